import os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_model import make_model
from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
m, _ = make_model("NO_GUIDANCE", 16)
img = P.synthetic_batch(33, 16, image_hw=(97, 131), seed=5)["imgs"].to("cuda:0")
with torch.no_grad():
    ref = m.perception(img).cpu()
    for lo, hi in ((0, 17), (17, 33), (0, 16), (16, 33), (5, 22)):
        f = m.perception(img[lo:hi].contiguous()).cpu()
        d = (f - ref[lo:hi]).abs().amax(dim=1)
        print((lo, hi), "max diff", d.max().item(), "rows", (d > 1e-3).nonzero().flatten().tolist())
