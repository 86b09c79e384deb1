"""Cell-layout 3x3 convolutions of a few launch geometries in a process of its own, so that a switch that is read once per process
(ADX_HS_PERSIST, ADX_HS_DMA, ADX_HS_MODE) can be set for it.  usage: python tests/conv_cells_worker.py OUT.pt; saves {case: cells}."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (cin, cout, batch, h, w, residual): grids of several tiles per CU at one, two and four cout tiles, a ragged last row tile, a
# range of spatial tiles that does not divide by eight, and a single tile
CASES = ((128, 128, 64, 32, 113, True), (128, 256, 40, 16, 57, False), (256, 512, 64, 8, 29, True), (64, 128, 33, 21, 50, True),
         (64, 128, 2, 8, 29, False), (256, 256, 48, 13, 31, True))


def outputs():
    from autonomous_driving_with_diffusion_model_amd import ops
    dev = "cuda:0"
    out = {}
    for cin, cout, b, h, w, with_res in CASES:
        g = torch.Generator(device="cpu").manual_seed(cin * 7 + cout + b)
        x = torch.randn((b, cin, h, w), generator=g).to(dev)
        wt = (torch.randn((cout, cin, 3, 3), generator=g) * (1.0 / (cin * 9)) ** 0.5).to(dev)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.1).to(dev)
        res = torch.randn((b, cout, h, w), generator=g).to(dev) if with_res else None
        _, packed = ops.conv2d(x[:1], wt, stride=1, pad=1)
        y = ops.conv2d_cells(ops.to_cells(x), packed, cin, cout, b, h, w, x_cells=True, scale=sc, shift=sh,
                             res=None if res is None else ops.to_cells(res), res_cells=with_res, relu=True)
        out[f"{cin}x{cout}b{b}@{h}x{w}"] = y.cpu()
    return out


if __name__ == "__main__":
    torch.save(outputs(), sys.argv[1])
