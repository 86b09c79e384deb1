"""Per-wave phase stamps of one steady-state iteration of conv2d_wgrad_hs1_kernel (a trace build of the library, ADX_LIB points
at it): shader clocks relative to the workgroup's first wave at the top of the iteration.
Columns: top | addresses done | per k-step: MFMAs issued, (fetched data arrived, converted + next fetch issued) | next top."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops, _lib
cin, cout, h, w = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 64, 64, 225))]
x = torch.randn(64, cin, h, w, device="cuda:0")
dy = torch.randn(64, cout, h, w, device="cuda:0")
for _ in range(3):
    ops.conv2d_weight_grad(x, dy, 3, stride=1, pad=1, estimate_range=False)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 256 * 12 * 16
buf = np.zeros(n, dtype=np.uint64)
assert lib.adx_wg_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n)) == 0
t = buf.reshape(256, 12, 16).astype(np.int64)
t = t[t[:, 0, 0] != 0]
print(f"{len(t)} workgroups, shape {cin}->{cout} @{h}x{w}")
ks = 4 if w > 48 else 2
pit = 2 if w > 48 else 1
base = t[:, :, 0].min(1)
cols = [0, 1]
names = ["top", "addr"]
for k in range(ks):
    cols.append(2 + 3 * k); names.append(f"mm{k}")
    if k < pit:
        if k == 0:
            cols += [3, 12, 13, 4]; names += ["arr0", "math0", "stor0", "fetch0"]
        else:
            cols += [3 + 3 * k, 4 + 3 * k]; names += [f"arr{k}", f"cvt{k}"]
cols.append(15); names.append("next")
print("          " + " ".join(f"{n:>6s}" for n in names))
for wv in range(12):
    print(f"wave {wv:2d}:  " + " ".join(f"{(t[:, wv, c] - base).mean():6.0f}" for c in cols))
