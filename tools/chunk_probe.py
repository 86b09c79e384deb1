"""Does running layer1 (six 64->64 convs at 64x225) in batch chunks keep the activations in the 256 MB Infinity Cache?
Chain of 6 convs (cells in/out, cell residual on every second) at B = 64 as one pass vs CH chunks run one after the other."""
import os, sys, torch
sys.path.insert(0, ".")
import bench
from autonomous_driving_with_diffusion_model_amd import ops
dev = "cuda:0"
B = 64
for (cin, h, w, nconv) in ((64, 64, 225, 6), (128, 32, 113, 7), (256, 16, 57, 11)):
    wt = torch.randn((cin, cin, 3, 3), device=dev) * (1.0 / (cin * 9)) ** 0.5
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    x = torch.randn((B, cin, h, w), device=dev)
    _, packed = ops.conv2d(x[:8].contiguous(), wt, stride=1, pad=1)
    bufs = [ops.to_cells(x).clone() for _ in range(3)]
    per_img = cin * h * w * 4
    d = ops.L.Conv2dDesc(cin, cin, 3, 1, 1)
    import ctypes as C
    lib = ops.L.lib()
    s = ops.L.stream_ptr(x.device)
    def conv(src, dst, res, n0, n):
        off = n0 * per_img
        fmt = 1 | 2 | (4 if res is not None else 0)
        ops.L.check(lib.adx_conv2d_forward_cells(C.byref(d), src.data_ptr() + off, packed.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                 (res.data_ptr() + off) if res is not None else None, dst.data_ptr() + off, n, h, w, 1, fmt, s))
    def chain(n0, n):
        cur = 0
        for i in range(nconv):
            if i % 2 == 0:
                conv(bufs[cur], bufs[(cur + 1) % 3], None, n0, n)
            else:
                conv(bufs[(cur + 1) % 3], bufs[(cur + 2) % 3], bufs[cur], n0, n)
                cur = (cur + 2) % 3
    for ch in (1, 2, 4, 8):
        def run():
            step = B // ch
            for c in range(ch):
                chain(c * step, step)
        run()
        print(f"{cin} ch @{h}x{w} x{nconv} convs, {ch} chunk(s): {bench.time_events(run, 5):.3f} ms")
