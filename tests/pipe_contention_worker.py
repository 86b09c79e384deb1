"""One of several processes that sample on the SAME GPU at the same time (tests/test_gpu_callers.py): one scene, H = 16,
classifier-free guidance -- the configuration whose deepest level runs as the pipeline launch (csrc/tconv_pipe.hip: 225 workgroups,
one per CU, later stages spinning on earlier ones).  With another process holding CUs the pipeline's workgroups cannot all be
resident at once; the launch must still complete (a stage waits only for workgroups dispatched before it) and give the same bits.
Usage: python tests/pipe_contention_worker.py OUT TICKS"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from autonomous_driving_with_diffusion_model_amd import scheduler as S  # noqa: E402
from autonomous_driving_with_diffusion_model_amd.sampling import GraphedSampler, generate_traj  # noqa: E402
from autonomous_driving_with_diffusion_model_amd.utils import procedural as P  # noqa: E402
from helpers import SCHED_KW  # noqa: E402
from test_gpu_model import make_model  # noqa: E402

DEV = "cuda:0"
out, ticks = sys.argv[1], int(sys.argv[2])
m, cfg = make_model("FREE_GUIDANCE", 16)
cfg.EVAL.SAMPLE_STEPS, cfg.GUIDANCE.FREE_SCALE = 20, 7.5
sch = S.GuidanceDDIMScheduler(cfg=cfg, thresholding=True, **SCHED_KW)
d = {k: v.to(DEV) for k, v in P.synthetic_batch(1, 16, image_hw=(64, 96), seed=5).items()}
res = []
with torch.no_grad():
    first = generate_traj(m, sch, cfg, d["imgs"], d["target"], d["init_trajs"])
    gs = GraphedSampler(m, sch, cfg)
    for _ in range(ticks):
        res.append(gs(d["imgs"], d["target"], d["init_trajs"]).clone())
torch.cuda.synchronize()
same = all(torch.equal(r, first) for r in res)
torch.save({"first": first.cpu(), "all_equal": bool(same), "finite": bool(torch.isfinite(first).all())}, out)
