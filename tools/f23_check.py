"""Check of the opt-in Winograd F(2,3) conv path (run with ADX_HS_F23=1): relative rms error against fp64 of the HIP
result and of a torch-CPU fp32 conv on the same inputs, for shapes with full, partial and single tiles."""
import os, sys, torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops
DEV = "cuda:0"
torch.manual_seed(0)
for cin, cout, h, w, n in [(64, 64, 64, 225, 2), (128, 128, 32, 113, 2), (256, 256, 16, 57, 3), (64, 128, 13, 37, 1), (32, 192, 9, 70, 1), (64,64,5,64,1)]:
    x = torch.randn(n, cin, h, w); wt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    res = torch.randn(n, cout, h, w); sc = torch.rand(cout) + 0.5; sh = torch.randn(cout)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), wt.double(), padding=1) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None] + res.double())
    y, _ = ops.conv2d(x.to(DEV), wt.to(DEV), stride=1, pad=1, scale=sc.to(DEV), shift=sh.to(DEV), res=res.to(DEV), relu=True)
    f32 = torch.relu(torch.nn.functional.conv2d(x, wt, padding=1) * sc[None, :, None, None] + sh[None, :, None, None] + res)
    e = ((y.cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    e32 = ((f32.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(cin, cout, h, w, "err", e, "cpu fp32 err", e32, "max abs", (y.cpu().double() - ref).abs().max().item())
