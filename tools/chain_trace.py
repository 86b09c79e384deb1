"""Phase times inside the chained-level kernel (csrc/tconv_chain.hip) from a -DADX_CHAIN_TRACE build:
    ADX_OUT=../libadx_trace.so ADX_OBJDIR=build_trace bash autonomous_driving_with_diffusion_model_amd/csrc/build.sh -DADX_CHAIN_TRACE
    ADX_LIB=autonomous_driving_with_diffusion_model_amd/libadx_trace.so ADX_CHAIN_MASK=0x1 python tools/chain_trace.py
One chain at a time (ADX_CHAIN_MASK: bit i = down level i, bit 8 + i = up level i): thread 0 of every workgroup stamps
s_memtime (100 MHz ticks = 10 ns) at 0 start, 1 input staged, then per stage s: 2+4s K loops done, 3+4s epilogue done, 4+4s
barrier passed, 5+4s cells re-split."""
import ctypes as C, os, statistics, sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from autonomous_driving_with_diffusion_model_amd import _lib as L
from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
from test_gpu_model import make_model
DEV = "cuda:0"
ROWS, H = int(os.environ.get("ROWS", "128")), int(os.environ.get("H", "32"))
m, _ = make_model("FREE_GUIDANCE", H)
d = P.synthetic_batch(ROWS, H, image_hw=(32, 32), seed=12)
feat = P._uniform("feat", 12, (ROWS, 64), -3.0, 3.0).to(DEV)
m.perception.forward = lambda img: feat
x, img, t, c = d["trajs"].to(DEV), d["imgs"].to(DEV), d["t"].to(DEV), d["target"].to(DEV)
dbg = C.CDLL(L.LIB_PATH)
with torch.no_grad():
    for _ in range(3):
        m(x, img, t, cond=c)
    torch.cuda.synchronize()
    dbg.adx_debug_chain_trace_clear()
    m(x, img, t, cond=c)
    torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 256))()
assert dbg.adx_debug_chain_trace(buf, 64 * 256) == 0
rows = [[buf[i * 64 + k] for k in range(64)] for i in range(256)]
rows = [r for r in rows if r[0]]
n_st = max(k for r in rows for k in range(64) if r[k]) // 4
print(f"{len(rows)} workgroups, {n_st} stages; ticks of 10 ns")
def med(k0, k1):
    dl = [r[k1] - r[k0] for r in rows if r[k1] and r[k0] and r[k1] >= r[k0]]
    return (statistics.median(dl), max(dl)) if dl else (float("nan"), float("nan"))
print("input staged      median %6.0f max %6.0f" % med(0, 1))
prev = 1
for s in range(n_st):
    a = med(prev, 2 + 4 * s); b = med(2 + 4 * s, 3 + 4 * s); c2 = med(3 + 4 * s, 4 + 4 * s); e = med(4 + 4 * s, 5 + 4 * s)
    print(f"stage {s}: K loops {a[0]:6.0f} (max {a[1]:6.0f})  epilogue {b[0]:6.0f}  barrier {c2[0]:6.0f}  re-split {e[0]:6.0f}")
    prev = 5 + 4 * s
print("whole workgroup   median %6.0f max %6.0f" % med(0, prev))
t0 = min(r[0] for r in rows)
print("grid span %d ticks; last start %d" % (max(r[prev] for r in rows) - t0, max(r[0] for r in rows) - t0))
