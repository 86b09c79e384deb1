"""The stem's weight gradient at the training size (B = 64, 3 x 256 x 900): split-fp16 kernel (range estimate in front, as
ops.conv2d_weight_grad runs it standalone) and the exact-fp32 kernel.  Under rocprofv3 --kernel-trace --stats the kernels'
own durations are in the summary."""
import sys
import torch
sys.path.insert(0, ".")
import bench
from autonomous_driving_with_diffusion_model_amd import ops
dev = "cuda:0"
x = torch.randn(bench.B, 3, 256, 900, device=dev)
dy = torch.randn(bench.B, 64, 128, 450, device=dev) * 1e-4
for est in (True, False):
    f = lambda: ops.conv2d_weight_grad(x, dy, 7, stride=2, pad=3, estimate_range=est)
    f()
    print("estimate_range", est, round(bench.time_events(f, 10), 4), "ms per call")
