"""One conv shape, a few launches (for rocprofv3 --pmc passes).  usage: pmc_one.py cin h w"""
import sys
import torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops
cin, h, w = (int(v) for v in sys.argv[1:4])
dev = "cuda:0"
x = torch.randn(64, cin, h, w, device=dev).relu_()
wt = torch.randn(cin, cin, 3, 3, device=dev) * 0.05
sc, sh = torch.rand(cin, device=dev), torch.rand(cin, device=dev)
_, packed = ops.conv2d(x, wt, stride=1, pad=1)
res = torch.randn_like(x)
xc, rc = ops.to_cells(x), ops.to_cells(res)
for _ in range(4):          # as the executor launches it at B = 64: cell tensors in and out, cell residual
    ops.conv2d_cells(xc, packed, cin, cin, 64, h, w, x_cells=True, scale=sc, shift=sh, res=rc, res_cells=True, relu=True)
torch.cuda.synchronize()
