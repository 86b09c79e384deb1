"""ADX_WGRAD_DETERMINISTIC=1 in a process of its own (the switches are read once per process): the 3x3 weight gradients of one
training step's shapes at the full batch, each computed twice; writes {shape: (equal, dw)} for the parent to compare with the
atomic path.  usage: python tests/wgrad_det_worker.py <out.pt> <batch>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autonomous_driving_with_diffusion_model_amd import ops  # noqa: E402

SHAPES = ((64, 64, 1, 64, 225), (128, 128, 1, 32, 113), (256, 256, 1, 16, 57), (512, 512, 1, 8, 29), (64, 128, 2, 64, 225),
          (64, 64, 1, 13, 37))


def inputs(cin, cout, s, h, w, batch, dev):
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(batch, cin, h, w, generator=g).to(dev)
    oh, ow = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    dy = (torch.randn(batch, cout, oh, ow, generator=g) * 1e-4).to(dev)
    return x, dy


if __name__ == "__main__":
    out, batch = sys.argv[1], int(sys.argv[2])
    dev = "cuda:0"
    res = {}
    for cin, cout, s, h, w in SHAPES:
        x, dy = inputs(cin, cout, s, h, w, batch, dev)
        a = ops.conv2d_weight_grad(x, dy, 3, stride=s, pad=1)
        b = ops.conv2d_weight_grad(x, dy, 3, stride=s, pad=1)
        c = ops.conv2d_weight_grad(x, dy, 3, stride=s, pad=1)
        res[(cin, cout, s, h, w)] = (bool(torch.equal(a, b) and torch.equal(a, c)), a.cpu())
    torch.save(res, out)
