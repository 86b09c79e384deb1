"""One NO_GUIDANCE training step (train.py:221-261) at a given size in a process of its own; writes per-tensor CHECKSUMS of the
gradients (sum and a position-weighted sum of their bit patterns) instead of the tensors, so that runs under different once-per-process
switches (ADX_HS_PERSIST, ADX_WGRAD_DETERMINISTIC, ...) can be compared bit for bit without moving 150 MB per run.
Usage: python tests/train_checksum_worker.py OUT.json BATCH HORIZON IMG_H IMG_W SEED"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def checksum(t: torch.Tensor):
    b = t.detach().contiguous().view(torch.int32).to(torch.int64).reshape(-1)
    w = (torch.arange(b.numel(), device=b.device, dtype=torch.int64) % 65521) + 1
    return [int(b.sum().item()), int((b * w).sum().item())]


if __name__ == "__main__":
    from train_step_worker import train_step
    out, batch, horizon, ih, iw, seed = sys.argv[1], *map(int, sys.argv[2:7])
    loss, grads = train_step(batch, horizon, (ih, iw), seed)
    with open(out, "w") as f:
        json.dump({"loss": loss, "sums": {k: checksum(g) for k, g in grads.items()}}, f)
