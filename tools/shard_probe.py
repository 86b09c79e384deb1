"""Would scene shards on streams of their own pay in the hoisted (perception-memo) sampling loop?  The UNet step is a chain of ~28
latency-bound launches (0.33 ms at 128 rows, 0.30 ms at 2 rows), so two half-batch loops side by side should take little longer than one.
Probe: S model objects (same procedural weights, own workspaces), each driving its shard's loop on its own stream from its own
host thread; features from ONE perception pass of the whole batch, pre-seeded into each model's memo."""
import contextlib, sys, threading, time, weakref
import torch
sys.path.insert(0, ".")
import bench
from autonomous_driving_with_diffusion_model_amd import scheduler as S, _lib as L
from autonomous_driving_with_diffusion_model_amd.config import create_cfg
from autonomous_driving_with_diffusion_model_amd.modeling import build_model
from autonomous_driving_with_diffusion_model_amd.sampling import generate_traj
from autonomous_driving_with_diffusion_model_amd.utils import procedural as P

dev = torch.device("cuda:0")
cfg = create_cfg()
cfg.MODEL.HORIZON = bench.H
cfg.TRAIN.USE_COND = cfg.GUIDANCE.USE_COND = "FREE_GUIDANCE"
cfg.GUIDANCE.FREE_SCALE, cfg.EVAL.SAMPLE_STEPS = bench.FREE_SCALE, bench.N_INFER
d = {k: v.to(dev) for k, v in P.synthetic_batch(bench.B, bench.H, image_hw=bench.IMG, seed=0).items()}


def make():
    with contextlib.redirect_stdout(sys.stderr):
        m = build_model(cfg)
    P.load_procedural(m, 0)
    return m.to(dev).eval(), S.GuidanceDDIMScheduler(cfg=cfg, thresholding=True, **bench.SCHED_KW)


def tick(models, shards):
    m0 = models[0][0]
    with torch.no_grad():
        feat = m0.perception(d["imgs"])
    n = bench.B // shards
    outs = [None] * shards
    cur = torch.cuda.current_stream(dev)

    def work(k):
        m, sch = models[k]
        img = d["imgs"][k * n:(k + 1) * n]
        m._feat_cache = (weakref.ref(img), L.write_stamp(img), m.perception.weights_key(), feat[k * n:(k + 1) * n].contiguous())
        st = streams[k]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs[k] = generate_traj(m, sch, cfg, img, d["target"][k * n:(k + 1) * n], d["init_trajs"][k * n:(k + 1) * n])
        keep.append(img)
    keep = []
    th = [threading.Thread(target=work, args=(k,)) for k in range(shards)]
    for t in th: t.start()
    for t in th: t.join()
    for k in range(shards):
        cur.wait_stream(streams[k])
    return torch.cat(outs, 0)


for shards in (1, 2, 4):
    models = [make() for _ in range(shards)]
    for m, _ in models:
        m.cache_perception = True
    streams = [torch.cuda.Stream(device=dev) for _ in range(shards)]
    ref = tick(models, shards)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = tick(models, shards)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    if shards == 1:
        base = ref
    print(f"shards {shards}: {ms:.2f} ms per 50-step tick = {bench.N_INFER / ms * 1e3:.0f} steps/s; max |diff| vs one shard {(ref - base).abs().max().item():.2e}")
    del models
