"""Per-layer timing of the temporal stack's launches (every distinct conv of one UNet forward, H = 32) through the C ABI,
HIP events on torch's stream.  ROWS = UNet batch (128 = the CFG batch of 64 scenes), EXACT=1 -> exact-fp32 kernel."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import _lib as L  # noqa: E402

DEV = torch.device("cuda:0")
ROWS = int(os.environ.get("ROWS", "128"))
EXACT = int(os.environ.get("EXACT", "0"))
N = int(os.environ.get("N", "30"))
SCRATCH = int(os.environ.get("SCRATCH", "1"))  # hand the launches a split-reduction scratch, as the UNet executor does
HZ = int(os.environ.get("H", "32"))           # horizon: the layer lengths below are written for 32 and scaled
# (kind, taps, stride, pad, c0, c1, cout, lin, lout, groups, count per forward)
LAYERS = [(0, 5, 1, 2, 7, 0, 64, 32, 32, 8, 1), (0, 5, 1, 2, 64, 0, 64, 32, 32, 8, 4), (0, 1, 1, 0, 7, 0, 64, 32, 32, 0, 1),
          (0, 3, 2, 1, 64, 0, 64, 32, 16, 0, 1), (0, 5, 1, 2, 64, 0, 128, 16, 16, 8, 1), (0, 5, 1, 2, 128, 0, 128, 16, 16, 8, 3),
          (0, 1, 1, 0, 64, 0, 128, 16, 16, 0, 1), (0, 3, 2, 1, 128, 0, 128, 16, 8, 0, 1),
          (0, 5, 1, 2, 128, 0, 256, 8, 8, 8, 1), (0, 5, 1, 2, 256, 0, 256, 8, 8, 8, 3), (0, 1, 1, 0, 128, 0, 256, 8, 8, 0, 1),
          (0, 3, 2, 1, 256, 0, 256, 8, 4, 0, 1), (0, 5, 1, 2, 256, 0, 512, 4, 4, 8, 1), (0, 5, 1, 2, 512, 0, 512, 4, 4, 8, 7),
          (0, 1, 1, 0, 256, 0, 512, 4, 4, 0, 1), (0, 5, 1, 2, 512, 512, 256, 4, 4, 8, 1), (0, 5, 1, 2, 256, 0, 256, 4, 4, 8, 3),
          (0, 1, 1, 0, 512, 512, 256, 4, 4, 0, 1), (1, 4, 2, 1, 256, 0, 256, 4, 8, 0, 1),
          (0, 5, 1, 2, 256, 256, 128, 8, 8, 8, 1), (0, 5, 1, 2, 128, 0, 128, 8, 8, 8, 3), (0, 1, 1, 0, 256, 256, 128, 8, 8, 0, 1),
          (1, 4, 2, 1, 128, 0, 128, 8, 16, 0, 1), (0, 5, 1, 2, 128, 128, 64, 16, 16, 8, 1), (0, 5, 1, 2, 64, 0, 64, 16, 16, 8, 3),
          (0, 1, 1, 0, 128, 128, 64, 16, 16, 0, 1), (1, 4, 2, 1, 64, 0, 64, 16, 32, 0, 1), (0, 1, 1, 0, 64, 0, 7, 32, 32, 0, 1),
          (0, 1, 1, 0, 128, 0, 3840, 1, 1, 0, 1)]
lib = L.lib()
s = L.stream_ptr(DEV)
scratch = torch.empty(2 << 20, device=DEV) if SCRATCH else None
tot = 0.0
tot_w = 0.0
for kind, taps, stride, pad, c0, c1, cout, lin, lout, groups, cnt in LAYERS:
    if lin > 1 or lout > 1:
        lin, lout = max(1, lin * HZ // 32), max(1, lout * HZ // 32)
    d = L.TConvDesc(kind, taps, stride, pad, c0, c1, cout, lin, lout, groups, 1e-5, 0, 0, EXACT)
    cin = c0 + c1
    w = torch.randn((cout, cin, taps) if kind == 0 else (cin, cout, taps), device=DEV) * (1.0 / (taps * cin)) ** 0.5
    packed = torch.empty(lib.adx_tconv_packed_bytes(C.byref(d)) // 4, device=DEV)
    L.check(lib.adx_tconv_pack(C.byref(d), w.data_ptr(), packed.data_ptr(), s))
    x0 = torch.randn((ROWS, c0, lin), device=DEV)
    x1 = torch.randn((ROWS, c1, lin), device=DEV) if c1 else None
    y = torch.empty((ROWS, cout, lout), device=DEV)
    b, g, be = (torch.randn(cout, device=DEV) * 0.1 for _ in range(3))
    io = L.TConvIO()
    io.x0, io.x0_sb, io.x0_sc, io.x0_sl = x0.data_ptr(), c0 * lin, lin, 1
    if x1 is not None:
        io.x1, io.x1_sb, io.x1_sc, io.x1_sl = x1.data_ptr(), c1 * lin, lin, 1
    io.packed_w, io.bias = packed.data_ptr(), b.data_ptr()
    if groups:
        io.gamma, io.beta = g.data_ptr(), be.data_ptr()
    io.y, io.y_sb, io.y_sc, io.y_sl, io.batch = y.data_ptr(), cout * lout, lout, 1, ROWS
    if scratch is not None:
        io.scratch, io.scratch_floats = scratch.data_ptr(), scratch.numel()
    run = lambda: L.check(lib.adx_tconv_forward(C.byref(d), C.byref(io), s))  # noqa: E731
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / N
    gf = 2.0 * ROWS * lout * cout * cin * taps / (1 if kind == 0 else 2) / 1e9
    wkb = 4.0 * cout * cin * taps / 1e3
    tot += us * cnt
    tot_w += wkb * cnt
    print(f"k{kind} t{taps} s{stride} {cin:5d}->{cout:5d} L{lin:2d}->{lout:2d} gn{groups} x{cnt}: {us:7.2f} us  {gf / us * 1e3:8.1f} TF/s-alg  "
          f"weights {wkb:8.1f} KB  {wkb / us:7.1f} GB/s-weights-once")
print(f"sum over one forward (launched back to back, same layer repeated): {tot:.1f} us, weights {tot_w / 1e3:.1f} MB")
