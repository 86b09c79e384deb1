"""3x3 stride-1 conv on tiny maps (the layer4 shapes of small images), with and without an in-place residual, vs fp64."""
import sys
import torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops  # noqa: E402
DEV = "cuda:0"
torch.manual_seed(0)
for (B, C, h, w) in ((1, 512, 4, 6), (2, 512, 4, 6), (2, 512, 2, 3), (3, 512, 2, 3), (2, 256, 8, 12), (2, 512, 8, 12), (4, 512, 4, 6)):
    x = torch.randn(B, C, h, w, device=DEV)
    wt = torch.randn(C, C, 3, 3, device=DEV) * (1.0 / (9 * C)) ** 0.5
    r = torch.randn(B, C, h, w, device=DEV)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), padding=1)
    y0, packed = ops.conv2d(x, wt, stride=1, pad=1)
    e0 = ((y0.double() - ref).norm() / ref.norm()).item()
    y1, _ = ops.conv2d(x, wt, stride=1, pad=1, res=r, packed=packed)
    e1 = ((y1.double() - (ref + r.double())).norm() / (ref + r.double()).norm()).item()
    buf = r.clone()
    ops.conv2d(x, wt, stride=1, pad=1, res=buf, packed=packed, out=buf)
    e2 = ((buf.double() - (ref + r.double())).norm() / (ref + r.double()).norm()).item()
    print(f"B{B} C{C} {h}x{w}: plain {e0:.2e}  +res {e1:.2e}  in-place res {e2:.2e}")
