"""Per-shape timing of the perception convs (B = 64, 3x256x900 geometry) with HIP events on torch's stream."""
import sys
import torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [(64, 64, 3, 1, 1, 64, 225), (128, 128, 3, 1, 1, 32, 113), (256, 256, 3, 1, 1, 16, 57), (512, 512, 3, 1, 1, 8, 29),
          (64, 128, 3, 2, 1, 64, 225), (128, 256, 3, 2, 1, 32, 113), (256, 512, 3, 2, 1, 16, 57),
          (64, 128, 1, 2, 0, 64, 225), (3, 64, 7, 2, 3, 256, 900)]
import os
B = int(os.environ.get("B", "64"))
for cin, cout, k, s, p, h, w in SHAPES:
    x = torch.randn(B, cin, h, w, device=DEV)
    wt = torch.randn(cout, cin, k, k, device=DEV) * 0.05
    if os.environ.get("ZERO") == "1":     # all-zero operands: same instruction stream, no toggling in the matrix pipe
        x.zero_(); wt.zero_()
    y, packed = ops.conv2d(x, wt, stride=s, pad=p)
    res = None if k == 7 or os.environ.get("RES") == "0" else torch.randn_like(y)
    sc, sh = torch.rand(cout, device=DEV), torch.rand(cout, device=DEV)
    for _ in range(3):
        ops.conv2d(x, wt, stride=s, pad=p, packed=packed, out=y, scale=sc, shift=sh, res=res, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.conv2d(x, wt, stride=s, pad=p, packed=packed, out=y, scale=sc, shift=sh, res=res, relu=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    oh, ow = y.shape[2:]
    gf = 2.0 * B * cout * oh * ow * cin * k * k / 1e9
    mb = 4.0 * (x.numel() + 2 * y.numel()) / 1e6
    print(f"{cin:4d}->{cout:4d} k{k} s{s} @{h}x{w}: {ms:7.3f} ms  {gf / ms:8.1f} GFLOP/ms(=TF/s fp32-equiv)  {mb / ms:7.1f} GB/s algorithmic")
