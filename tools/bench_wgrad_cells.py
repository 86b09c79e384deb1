"""Per-shape time of the 3x3 weight gradient with both operands as cell tensors (as the training executor launches it at B = 64):
memset + kernel, HIP events.  ADX_WGRAD_128=0|1 picks the kernel (read once per process)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, ".")
import bench
from autonomous_driving_with_diffusion_model_amd import _lib as L, ops
dev = "cuda:0"
B = int(os.environ.get("B", "64"))
out = []
for cin, cout, h, w in ((128, 128, 32, 113), (256, 256, 16, 57), (512, 512, 8, 29)):
    x = torch.randn((B, cin, h, w), device=dev).relu_()
    dy = torch.randn((B, cout, h, w), device=dev) * 1e-3
    s = 2.0 ** 23
    scale = torch.tensor([s, 1.0 / s], device=dev)
    xc, dc = ops.to_cells(x), ops.to_cells(dy * s)
    d = L.Conv2dDesc(cin, cout, 3, 1, 1)
    dw = torch.empty((cout, cin, 3, 3), device=dev)
    scratch = torch.empty(max(64, L.lib().adx_conv2d_wgrad_scratch_bytes()), dtype=torch.uint8, device=dev)
    f = lambda: L.check(L.lib().adx_conv2d_wgrad_cells(C.byref(d), xc.data_ptr(), dc.data_ptr(), scale.data_ptr(), dw.data_ptr(), B, h, w,
                                                       scratch.data_ptr(), L.stream_ptr(torch.device(dev))))
    f()
    out.append(round(bench.time_events(f, 20), 4))
print("ADX_WGRAD_128", os.environ.get("ADX_WGRAD_128"), out)
