"""Error of the conv2d kernels against fp64, per K (run with and without ADX_CONV_EXACT=1)."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops  # noqa: E402

for cin, h, w in ((16, 32, 64), (64, 32, 64), (256, 16, 57), (512, 8, 29)):
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(2, cin, h, w, generator=g)
    wt = torch.randn(64 if cin < 64 else cin, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    f32 = F.conv2d(x, wt, padding=1)
    gpu32 = F.conv2d(x.cuda(), wt.cuda(), padding=1).cpu()
    y, _ = ops.conv2d(x.cuda(), wt.cuda(), stride=1, pad=1)
    den = ref.abs().max()
    rms = lambda a: ((a.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()  # noqa: E731
    mx = lambda a: ((a.double() - ref).abs().max() / den).item()  # noqa: E731
    print(f"K={9 * cin:5d}  adx max {mx(y.cpu()):.2e} rms {rms(y.cpu()):.2e} | torch-cpu max {mx(f32):.2e} rms {rms(f32):.2e} | "
          f"torch-gpu(MIOpen) max {mx(gpu32):.2e} rms {rms(gpu32):.2e}")
