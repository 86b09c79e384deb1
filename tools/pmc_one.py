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
y, packed = ops.conv2d(x, wt, stride=1, pad=1)
res = torch.randn_like(y)
for _ in range(4):
    ops.conv2d(x, wt, stride=1, pad=1, packed=packed, out=y, scale=sc, shift=sh, res=res, relu=True)
torch.cuda.synchronize()
