"""Wall time of a 50-step tick written like the reference's own loop (interact.py:128-166): model() and scheduler.step() per step,
classifier-free combine in torch, nothing hoisted or fused by the caller.  B = 1, H = 16."""
import contextlib, sys, time, torch
sys.path.insert(0, ".")
import bench
from autonomous_driving_with_diffusion_model_amd import scheduler as S
from autonomous_driving_with_diffusion_model_amd.config import create_cfg
from autonomous_driving_with_diffusion_model_amd.modeling import build_model
from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
dev = torch.device("cuda:0")
cfg = create_cfg(); cfg.MODEL.HORIZON = 16
cfg.TRAIN.USE_COND = cfg.GUIDANCE.USE_COND = "FREE_GUIDANCE"
with contextlib.redirect_stdout(sys.stderr):
    m = build_model(cfg)
P.load_procedural(m, 0); m = m.to(dev).eval()
sch = S.GuidanceDDIMScheduler(cfg=cfg, thresholding=True, **bench.SCHED_KW)
d = {k: v.to(dev) for k, v in P.synthetic_batch(1, 16, image_hw=bench.IMG, seed=3).items()}
cond = torch.cat([d["target"], torch.zeros_like(d["target"])], 0)
def tick():      # the reference's own loop shape (interact.py:128-166): model() and scheduler.step() per step
    sch.set_timesteps(50, device=dev)
    trajs = d["init_trajs"].clone()
    img = d["imgs"].clone()
    for t in sch.timesteps:
        with torch.no_grad():
            out = m(torch.cat([trajs, trajs], 0), img, t.reshape(-1), cond=cond)
        c, u = out.chunk(2, dim=0)
        mo = u + 7.5 * (c - u)
        trajs = sch.step(mo, t, trajs).prev_sample
        trajs[:, 0, :3] = 0.0
    return trajs
for _ in range(3): tick()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): tick()
torch.cuda.synchronize(); print(f"reference-shaped eager loop, B=1 H=16: {1e3*(time.perf_counter()-t0)/5:.2f} ms per tick")
