"""Diagnostic: stop the perception backward after N blocks (ADX_DBG_STOP_BLOCKS=N) and compare the gradient buffer it leaves
(d loss / d output of block 16 - N) with torch autograd through the oracle in fp64."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import resnet as R  # noqa: E402
from autonomous_driving_with_diffusion_model_amd.utils import procedural as P  # noqa: E402
from helpers import oracle_sd  # noqa: E402
from test_gpu_model import make_model  # noqa: E402

N = int(os.environ["ADX_DBG_STOP_BLOCKS"])
Bn = int(os.environ.get("B", "2"))
hw = tuple(int(v) for v in os.environ.get("HW", "128,192").split(","))
m, _ = make_model("NO_GUIDANCE", 16)
m.train()
img = P.synthetic_batch(Bn, 16, image_hw=hw, seed=41)["imgs"]
w = P._uniform("perc.w", 61, (Bn, 64), -1.0, 1.0)
feat = m.perception(img.to("cuda:0"))
ws = feat.grad_fn.ws
(feat * w.to("cuda:0")).sum().backward()
torch.cuda.synchronize()
wsf = ws.view(torch.float32).cpu().double()

# oracle with the block outputs kept
sd = {k: (v.double() if v.is_floating_point() else v) for k, v in oracle_sd("NO_GUIDANCE").items()}
p = "perception."
x = F.conv2d(img.double(), sd[p + "conv1.weight"], None, stride=2, padding=3)
x = F.relu(R.batch_norm(sd, p + "bn1.", x, True))
x = F.max_pool2d(x, 3, 2, 1)
outs = []
for li, n in enumerate(R.LAYERS, start=1):
    for bi in range(n):
        x = R.basic_block(sd, f"{p}layer{li}.{bi}.", x, 2 if (li > 1 and bi == 0) else 1, True)
        x.retain_grad() if x.requires_grad else None
        outs.append(x)
# gradient w.r.t. block outputs: make the input of the tail differentiable
k = len(outs) - 1 - N          # output of this block is what the stopped backward holds
xk = outs[k].detach().requires_grad_()
y = xk
idx = 0
blocks = [(li, bi) for li, n in enumerate(R.LAYERS, start=1) for bi in range(n)]
for (li, bi) in blocks[k + 1:]:
    y = R.basic_block(sd, f"{p}layer{li}.{bi}.", y, 2 if (li > 1 and bi == 0) else 1, True)
f = F.linear(torch.flatten(F.adaptive_avg_pool2d(y, (1, 1)), 1), sd[p + "fc.weight"], sd[p + "fc.bias"])
(f * w.double()).sum().backward()
g = xk.grad.reshape(-1)
print("block", blocks[k], "d(out) elements", g.numel(), "norm", g.norm().item())
best = (1e9, -1)
for off in range(0, wsf.numel() - g.numel() + 1, 64):
    cand = wsf[off:off + g.numel()]
    e = ((cand - g).norm() / g.norm()).item()
    if e < best[0]:
        best = (e, off)
print("best match in the workspace: relative error", best[0], "at float offset", best[1])
cand = wsf[best[1]:best[1] + g.numel()]
d = (cand - g).abs().reshape(xk.shape)
print("max abs diff", d.max().item(), "at", [int(v) for v in torch.nonzero(d == d.max())[0]], "value", g.reshape(xk.shape)[tuple(torch.nonzero(d == d.max())[0])].item())
per_img = [(d[i].norm() / g.reshape(xk.shape)[i].norm()).item() for i in range(Bn)]
print("relative error per image", per_img)
rows = d.amax(dim=(0, 1))
print("max abs diff by pixel:\n", rows)
