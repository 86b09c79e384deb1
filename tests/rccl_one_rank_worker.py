"""One process, one GPU, a ONE-rank RCCL group: the data-parallel transport of parallel.GradientAverager with the world-1 early
returns kept out (`force=True`), so that the real RCCL kernels -- all_reduce, or reduce_scatter_tensor -> all_gather_into_tensor
-- run on the side stream behind the perception backward's per-group completion events while backward is still running
(train.py:174-178,249).  One-rank collectives are identities: every gradient must come out bit-equal to a plain
backward without the averager.  Run with ADX_WGRAD_DETERMINISTIC=1 (the default weight-gradient reduction uses float atomics).

Usage: python tests/rccl_one_rank_worker.py OUT.json PRIMITIVE BATCH HORIZON IMG_H IMG_W [TRACE_OUT.json]"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = "cuda:0"


def build(horizon):
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    from autonomous_driving_with_diffusion_model_amd.modeling import build_model
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
    cfg = create_cfg()
    cfg.MODEL.HORIZON = horizon
    cfg.TRAIN.USE_COND = cfg.GUIDANCE.USE_COND = "FREE_GUIDANCE"
    m = build_model(cfg)
    P.load_procedural(m, 0)
    return m.to(DEV).train()


def loss_of(model, d):
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from helpers import SCHED_KW
    noisy = S.DDPMScheduler(**SCHED_KW).add_noise(d["trajs"], d["noise"], d["t"], zero_first=True)
    return torch.nn.functional.mse_loss(model(noisy, d["imgs"], d["t"], cond=d["target"]), d["trajs"])


def same_bits(ma, mb, only=None):
    """Names of the gradient tensors that differ in some bit (restricted to `only`), and the largest relative L2 difference
    over ALL tensors (the temporal stack's and the stem's weight gradients add with float atomics: two plain runs differ there)."""
    bad, worst = [], 0.0
    for (k, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        if not torch.equal(p.grad, q.grad):
            if only is None or k in only:
                bad.append(k)
            worst = max(worst, ((p.grad.double() - q.grad.double()).norm() / (q.grad.double().norm() + 1e-300)).item())
    return bad, worst


def main():
    out, primitive = sys.argv[1], sys.argv[2]
    batch, horizon, ih, iw = map(int, sys.argv[3:7])
    trace_out = sys.argv[7] if len(sys.argv) > 7 else None
    from autonomous_driving_with_diffusion_model_amd.parallel import DataParallel
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    res = {"primitive": primitive, "backend": str(dist.get_backend())}
    d = {k: v.to(DEV) for k, v in P.synthetic_batch(batch, horizon, image_hw=(ih, iw), seed=7).items()}

    # (1) the step: plain backward vs the averager with forced one-rank collectives and per-bucket device events
    plain = build(horizon)
    loss_p = loss_of(plain, d)
    loss_p.backward()
    torch.cuda.synchronize()
    # which tensors are bit-reproducible at all: the same plain step twice (ADX_WGRAD_DETERMINISTIC=1 covers the 3x3 weight
    # gradients; kernels that reduce with float atomics -- the stem's and the temporal stack's weight gradients -- are not)
    plain2 = build(horizon)
    loss_of(plain2, d).backward()
    torch.cuda.synchronize()
    noisy, res["plain_vs_plain_worst_rel"] = same_bits(plain, plain2)
    stable = {k for k, _ in plain.named_parameters()} - set(noisy)
    res["n_tensors"], res["n_bit_reproducible"] = len(noisy) + len(stable), len(stable)
    del plain2
    forced = build(horizon)
    dp = DataParallel(forced, bucket_mb=64.0, primitive=primitive, force=True)
    av = dp.averager
    assert av.active and av.world == 1
    loss_of(dp, d).backward()          # warm-up of the communicator (its first collective allocates)
    dp.synchronize()
    forced.zero_grad(set_to_none=True)
    # (the warm-up moved the BatchNorm running statistics; train-mode gradients do not depend on them)
    av.trace = True
    loss_f = loss_of(dp, d)
    loss_f.backward()
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    dp.synchronize()
    torch.cuda.synchronize()
    res["overlap"] = av.overlap_report(end)
    av.trace = False
    res["n_buckets"] = len(av.buckets)
    res["copied_in"] = av.copied_in
    res["born_in_bucket"] = all(p.grad.data_ptr() == p._adx_grad_view.data_ptr() for p in forced.parameters())
    res["loss_equal"] = bool(loss_p.detach() == loss_f.detach())
    res["step_mismatch"], res["step_worst_rel"] = same_bits(plain, forced, stable)
    res["grad_bytes"] = sum(p.numel() * 4 for p in forced.parameters())

    # (2) accumulation: a second backward WITHOUT clearing .grad.  The perception node then gets buffers of its own and
    # AccumulateGrad's adds run on the compute stream after the whole native call -- no layer-group event covers them, so the
    # buckets must fall back to joining the compute stream (round 5 waited for the event only and reduced half-added buckets)
    d2 = {k: v.to(DEV) for k, v in P.synthetic_batch(batch, horizon, image_hw=(ih, iw), seed=8).items()}
    loss_of(plain, d2).backward()
    loss_of(dp, d2).backward()
    dp.synchronize()
    torch.cuda.synchronize()
    res["accum_copied_in"] = av.copied_in
    res["accum_mismatch"], res["accum_worst_rel"] = same_bits(plain, forced, stable)

    # (3) the module twice in one graph: the second node finds the bucket views lent and writes buffers of its own
    plain.zero_grad(set_to_none=True)
    forced.zero_grad(set_to_none=True)
    (loss_of(plain, d) + loss_of(plain, d2)).backward()
    (loss_of(dp, d) + loss_of(dp, d2)).backward()
    dp.synchronize()
    torch.cuda.synchronize()
    res["twice_copied_in"] = av.copied_in
    res["twice_mismatch"], res["twice_worst_rel"] = same_bits(plain, forced, stable)

    with open(out, "w") as f:
        json.dump(res, f)
    if trace_out is not None:
        with open(trace_out, "w") as f:
            json.dump({"what": "parallel.GradientAverager(force=True) on a one-rank RCCL group, one MI355X: per bucket, ms relative "
                               "to the END of backward on the device (negative = before); ready = the completion event of the "
                               "last layer group whose gradients the bucket holds, done = its RCCL collective finished",
                       "primitive": primitive, "backend": res["backend"], "batch": batch, "horizon": horizon, "image": [ih, iw],
                       "grad_mb": round(res["grad_bytes"] / 1e6, 1), "buckets": res["overlap"]}, f, indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
