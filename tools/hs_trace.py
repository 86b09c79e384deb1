"""Phase timeline of conv2d_hs3x3_kernel from a -DADX_HS_TRACE build (csrc/build.sh -DADX_HS_TRACE with
ADX_OUT=tools/micro/libadx_trace.so; run with ADX_LIB pointing at it).  Every workgroup stamps the shader clock at:
0 entry, 1 prologue loads issued + stage 0 stored, 2 first barrier passed, 3 main loop done, 4 epilogue stores issued,
5 stores drained; slot 7 = XCC_ID << 32 | HW_ID.  Prints per-phase durations and how the workgroups that shared a CU
overlapped."""
import ctypes
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops, _lib  # noqa: E402

DEV = "cuda:0"
cin, cout, h, w = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 64, 64, 225))]
B = 64
x = torch.randn(B, cin, h, w, device=DEV)
wt = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
y, packed = ops.conv2d(x, wt, stride=1, pad=1)
res = torch.randn_like(y)
sc, sh = torch.rand(cout, device=DEV), torch.rand(cout, device=DEV)
import os
if os.environ.get("CELLS", "1") == "1":      # as the executor launches it at this batch: cell tensors in and out, cell residual
    xc, rc_ = ops.to_cells(x), ops.to_cells(res)
    for _ in range(3):
        ops.conv2d_cells(xc, packed, cin, cout, B, h, w, x_cells=True, scale=sc, shift=sh, res=rc_, res_cells=True, relu=True)
else:
    for _ in range(3):
        ops.conv2d(x, wt, stride=1, pad=1, packed=packed, out=y, scale=sc, shift=sh, res=res, relu=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 16384
buf = np.zeros(n * 8, dtype=np.uint64)
rc = lib.adx_hs_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n * 8))
assert rc == 0, rc
t = buf.reshape(n, 8)
idx = np.nonzero(t[:, 0])[0]
t = t[idx]
nwg = len(t)
ts = t[:, :6].astype(np.int64)
hw = (t[:, 7] & 0xFFFFFFFF).astype(np.int64)
xcc = (t[:, 7] >> 32).astype(np.int64) & 0xF
cu = (hw >> 8) & 0xF
sh_ = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh_) * 16 + cu
print(f"{nwg} workgroups; distinct CUs seen {len(np.unique(cuid))}")
names = ["entry->prologue issued", "first barrier wait", "main loop", "epilogue issue", "store drain"]
for i, nm in enumerate(names):
    d = ts[:, i + 1] - ts[:, i]
    print(f"  {nm:24s} mean {d.mean():9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}")
t6 = t[:, 6].astype(np.int64)
if t6.any():
    d = t6 - ts[:, 3]
    print(f"  main end -> residual arrived mean {d.mean():9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}")
life = ts[:, 5] - ts[:, 0]
print(f"  workgroup lifetime        mean {life.mean():9.0f}")
busy = np.zeros(3)
spans = []
for c in np.unique(cuid):     # clocks are per XCD: everything relative to the CU's own first entry
    m = cuid == c
    tc = ts[m] - ts[m, 0].min()
    spans.append(tc[:, 5].max())
    ev = []
    for a_, b_ in zip(tc[:, 2], tc[:, 3]):
        ev.append((a_, 1)); ev.append((b_, -1))
    ev.sort()
    cur, last = 0, 0
    for tt, dlt in ev:
        busy[min(cur, 2)] += tt - last
        last, cur = tt, cur + dlt
    busy[0] += tc[:, 5].max() - last
print(f"  per-CU span mean {np.mean(spans):.0f} clocks (min {np.min(spans):.0f}, max {np.max(spans):.0f})")
print("  per-CU time with 0/1/2+ workgroups in the main loop:", np.round(busy / busy.sum(), 3))
for c in np.unique(cuid)[[5, 100]]:
    m = cuid == c
    tc = ts[m] - ts[m, 0].min()
    o = np.argsort(tc[:, 0])
    print(f"  timeline of CU {c} (clocks): blockIdx, entry, barrier, main end, stores issued, drained")
    for b_, r in list(zip(idx[m][o], tc[o]))[:8]:
        print("   ", b_, r[0], r[2], r[3], r[4], r[5])
first = [sorted(idx[cuid == c][np.argsort(ts[cuid == c, 0])][:2]) for c in np.unique(cuid)]
print("  first two blockIdx per CU (first 24 CUs):", first[:24])
