"""Per-shape timing of the 3x3 stride-1 convs in the cell layout (as the executor launches them at B = 64) for the tile mode
pinned by ADX_HS_MODE (unset: the default rule)."""
import os, sys, torch
sys.path.insert(0, ".")
import bench
from autonomous_driving_with_diffusion_model_amd import ops
dev = "cuda:0"
B = int(os.environ.get("B", "64"))
shapes = [(64, 64, 64, 225), (128, 128, 32, 113), (256, 256, 16, 57), (512, 512, 8, 29)]
out = []
for cin, cout, h, w in shapes:
    x = torch.randn((B, cin, h, w), device=dev)
    wt = torch.randn((cout, cin, 3, 3), device=dev) * (1.0 / (cin * 9)) ** 0.5
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    res = torch.randn((B, cout, h, w), device=dev)
    _, packed = ops.conv2d(x, wt, stride=1, pad=1)
    xc, rc = ops.to_cells(x), ops.to_cells(res)
    f1 = lambda: ops.conv2d_cells(xc, packed, cin, cout, B, h, w, x_cells=True, scale=sc, shift=sh, relu=True)
    f2 = lambda: ops.conv2d_cells(xc, packed, cin, cout, B, h, w, x_cells=True, scale=sc, shift=sh, res=rc, res_cells=True, relu=True)
    f1(); f2()
    out.append((round(bench.time_events(f1, 20), 4), round(bench.time_events(f2, 20), 4)))
print("MODE", os.environ.get("ADX_HS_MODE"), out)
