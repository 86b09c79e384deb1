"""Per-shape timing of the 3x3 weight-gradient launches of one training step (B = 64, 3x256x900 geometry).
RANGE=0: without the range-estimate kernel in front (dy in fp16 range as given), i.e. memset + the weight-gradient kernel."""
import os
import sys
import torch
sys.path.insert(0, ".")
import bench
from autonomous_driving_with_diffusion_model_amd import ops
dev = "cuda:0"
est = os.environ.get("RANGE", "1") == "1"
out = []
for cin, cout, s, h, w in ((64, 64, 1, 64, 225), (128, 128, 1, 32, 113), (256, 256, 1, 16, 57), (512, 512, 1, 8, 29), (64, 128, 2, 64, 225)):
    x = torch.randn(bench.B, cin, h, w, device=dev)
    oh, ow = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    dy = torch.randn(bench.B, cout, oh, ow, device=dev) * (1e-4 if est else 1.0)
    if os.environ.get("ZERO") == "1":      # the same instruction stream without toggling operands: what the power limit costs
        x.zero_(); dy.zero_()
    f = lambda: ops.conv2d_weight_grad(x, dy, 3, stride=s, pad=1, estimate_range=est)
    f()
    out.append(round(bench.time_events(f, 10), 4))
print(out)
