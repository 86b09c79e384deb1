"""Sustained fp16 MFMA rate (adx_probe_mfma_fp16) as a function of the operand bit statistics: how much of the power-limited
rate is data-dependent, and which property of the operands it follows."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
import bench
from autonomous_driving_with_diffusion_model_amd import _lib as L
dev = torch.device("cuda:0")
N = 4096 * 8
g = torch.Generator(device=dev).manual_seed(0)
r = torch.rand(N, device=dev, generator=g)
sgn = (torch.randint(0, 2, (N,), device=dev, generator=g) * 2 - 1).float()


def trunc(x, bits):      # keep `bits` mantissa bits of an fp16 value
    i = x.half().view(torch.int16)
    return (i & ~((1 << (10 - bits)) - 1)).view(torch.float16)


cases = {
    "zero": torch.zeros(N, device=dev).half(),
    "ones": torch.ones(N, device=dev).half(),
    "random [0.125,1) +-": ((r * 0.875 + 0.125) * sgn).half(),
    "random [0.125,1) +": (r * 0.875 + 0.125).half(),
    "random [0.5,1) +- (one exponent)": ((r * 0.5 + 0.5) * sgn).half(),
    "random, 8 mantissa bits": trunc((r * 0.875 + 0.125) * sgn, 8),
    "random, 5 mantissa bits": trunc((r * 0.875 + 0.125) * sgn, 5),
    "random, 2 mantissa bits": trunc((r * 0.875 + 0.125) * sgn, 2),
    "random, 0 mantissa bits (powers of two) +-": trunc((r * 0.875 + 0.125) * sgn, 0),
    "normal(0,1)": torch.randn(N, device=dev, generator=g).half(),
    "relu(normal): half zeros": torch.randn(N, device=dev, generator=g).relu().half(),
}
out = torch.empty(512 * 256, device=dev)
fl = C.c_double(0.0)
for name, t in cases.items():
    fn = lambda: L.check(L.lib().adx_probe_mfma_fp16(t.data_ptr(), out.data_ptr(), 512, 4000, C.byref(fl), L.stream_ptr(dev)))
    ms = bench.time_events(fn, 3, warm=1)
    print(f"{name:48s} {fl.value / ms / 1e9:8.1f} TFLOP/s")
