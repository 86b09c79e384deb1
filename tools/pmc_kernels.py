"""Launch the two roofline kernels alone (for `rocprofv3 --pmc ...` passes): the four 3x3 stride-1 conv shapes of one
ResNet-34 pass at B=64, 3x256x900 (cell tensors in and out, like the executor's launches), and the 512->512 temporal conv at
the CFG batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from autonomous_driving_with_diffusion_model_amd import ops
dev = torch.device("cuda:0")
shapes = {}
for c in bench.resnet_conv_table(*bench.IMG):
    if c[3] == 1:
        shapes[c] = shapes.get(c, 0) + 1
for (cin, cout, k, s, p, h, w), cnt in shapes.items():
    x = torch.randn((bench.B, cin, h, w), device=dev)
    wt = torch.randn((cout, cin, k, k), device=dev) * (1.0 / (cin * k * k)) ** 0.5
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    res = torch.randn((bench.B, cout, h, w), device=dev)
    _, packed = ops.conv2d(x, wt, stride=s, pad=p)
    xc, rc = ops.to_cells(x), ops.to_cells(res)
    # as adx_resnet_forward launches them at this batch: cell tensors in and out, a cell residual on every second conv
    for i in range(4):
        ops.conv2d_cells(xc, packed, cin, cout, bench.B, h, w, x_cells=True, scale=sc, shift=sh, relu=True,
                         **(dict(res=rc, res_cells=True) if i % 2 else {}))
    torch.cuda.synchronize()
print(bench.tconv_roofline(None, dev, reps=3))
