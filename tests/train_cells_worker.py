"""One train-mode perception forward + backward per case in a process of its own (ADX_TRAIN_CELLS and ADX_WGRAD_DETERMINISTIC are
read once per process): the feature, every parameter gradient and the updated running statistics, for the parent to compare between
the two activation layouts of the training executor.  usage: python tests/train_cells_worker.py <out.pt>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
CASES = (((96, 128), 8), ((70, 102), 3), ((128, 224), 1), ((256, 900), 4))


def run():
    from test_gpu_model import make_model
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
    out = {}
    for hw, b in CASES:
        m, _ = make_model("NO_GUIDANCE", 16)
        m.train()
        img = P.synthetic_batch(b, 16, image_hw=hw, seed=23)["imgs"].to("cuda:0")
        w = P._uniform("perc.w", 23, (b, 64), -1.0, 1.0).to("cuda:0")
        feat = m.perception(img)
        # ADX_TEST_GRAD_SCALE_LOG2=k: d(loss)/d(feature) times 2^k (the backward pass is linear in it: test_gpu_train.py)
        (feat * (w * 2.0 ** float(os.environ.get("ADX_TEST_GRAD_SCALE_LOG2", "0")))).sum().backward()
        res = {"feature": feat.detach().cpu()}
        for k, p in m.perception.named_parameters():
            res["grad." + k] = p.grad.cpu()
        for k, buf in m.perception.named_buffers():
            res["buffer." + k] = buf.cpu()
        out[f"{hw[0]}x{hw[1]}b{b}"] = res
    return out


if __name__ == "__main__":
    torch.save(run(), sys.argv[1])
