"""Dump the perception train-forward tape (the workspace the backward reads) and compare two dumps (e.g. split-fp16 vs
ADX_CONV_EXACT=1): python tools/dbg_tape_cmp.py dump /tmp/a.pt ; ADX_CONV_EXACT=1 ... dump /tmp/b.pt ; ... cmp /tmp/a.pt /tmp/b.pt"""
import os
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
if sys.argv[1] == "dump":
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
    from test_gpu_model import make_model
    Bn = int(os.environ.get("B", "2"))
    hw = tuple(int(v) for v in os.environ.get("HW", "128,192").split(","))
    m, _ = make_model("NO_GUIDANCE", 16)
    m.train()
    img = P.synthetic_batch(Bn, 16, image_hw=hw, seed=41)["imgs"].to("cuda:0")
    feat = m.perception(img)
    torch.cuda.synchronize()
    ws = feat.grad_fn.ws.view(torch.float32).cpu()
    torch.save({"ws": ws, "feat": feat.detach().cpu()}, sys.argv[2])
    print("dumped", ws.numel(), "floats")
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    print("feature max diff", (a["feat"] - b["feat"]).abs().max().item())
    x, y = a["ws"], b["ws"]
    n = min(x.numel(), y.numel())
    x, y = x[:n], y[:n]
    d = (x - y).abs()
    d = torch.nan_to_num(d, nan=0.0, posinf=0.0)
    blk = 1 << 16
    nb = n // blk
    db = d[: nb * blk].view(nb, blk).max(dim=1).values
    mag = y[: nb * blk].view(nb, blk).abs().max(dim=1).values
    bad = [(i, db[i].item(), mag[i].item()) for i in range(nb) if db[i] > 1e-4 * max(mag[i].item(), 1e-6) and mag[i] < 1e30]
    print(len(bad), "of", nb, "blocks of 65536 floats differ by more than 1e-4 of the block's magnitude")
    for i, dv, mv in bad[:40]:
        print(f"  block {i} (float offset {i * blk}): max diff {dv:.3e}, magnitude {mv:.3e}")
    if len(sys.argv) > 4:
        lo_, hi_ = 80000, 7681664          # skip the fp64 statistics block in front and the backward scratch behind
        flips = (x > 0) != (y > 0)
        flips[:lo_] = False
        flips[hi_:] = False
        idx = torch.nonzero(flips).flatten()
        print("sign(>0) flips:", idx.numel())
        if idx.numel():
            mags = torch.maximum(x[idx].abs(), y[idx].abs())
            print("  magnitudes of flipped elements: max", mags.max().item(), "median", mags.median().item())
            big = idx[mags > 1e-5]
            print("  flips with |value| > 1e-5:", big.numel(), "first offsets", big[:20].tolist())
            for o in big[:10].tolist():
                print("    offset", o, "hs", x[o].item(), "exact", y[o].item())
