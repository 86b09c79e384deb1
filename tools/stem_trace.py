"""Phase totals of the stem kernel from a -DADX_HS_TRACE build (ADX_LIB=tools/micro/libadx_trace.so): per workgroup the
shader clocks spent in the MFMA passes (+ BN / stores or the vertical maximum), the pooled epilogue and patch staging +
barriers, against its lifetime.  `python tools/stem_trace.py` = the plain stem (training), `... pool` = the fused
stem + MaxPool of the inference executor (one perception pass)."""
import ctypes
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops, _lib  # noqa: E402

DEV = "cuda:0"
x = torch.randn(64, 3, 256, 900, device=DEV)
if len(sys.argv) > 1 and sys.argv[1] == "pool":
    import contextlib
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    from autonomous_driving_with_diffusion_model_amd.modeling import build_model
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
    cfg = create_cfg()
    with contextlib.redirect_stdout(sys.stderr):
        m = build_model(cfg)
    P.load_procedural(m, 0)
    m = m.to(DEV).eval()
    with torch.no_grad():
        for _ in range(2):
            m.perception(x)
else:
    wt = torch.randn(64, 3, 7, 7, device=DEV) * 0.05
    sc, sh = torch.rand(64, device=DEV), torch.rand(64, device=DEV)
    y, packed = ops.conv2d(x, wt, stride=2, pad=3)
    for _ in range(3):
        ops.conv2d(x, wt, stride=2, pad=3, packed=packed, out=y, scale=sc, shift=sh, relu=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 16384
buf = np.zeros(n * 8, dtype=np.uint64)
assert lib.adx_hs_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n * 8)) == 0
t = buf.reshape(n, 8)[8192:]
t = t[t[:, 0] != 0].astype(np.int64)
life = t[:, 5] - t[:, 0]
print(f"{len(t)} workgroups; lifetime mean {life.mean():.0f} clocks; prologue {np.mean(t[:, 1] - t[:, 0]):.0f}")
for name, slot in (("MFMA passes + stores", 2), ("pooled epilogue", 3), ("staging + barriers", 4)):
    print(f"  {name:22s} {t[:, slot].mean():9.0f} clocks  ({100 * t[:, slot].mean() / life.mean():.0f} %)")
