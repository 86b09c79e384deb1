"""Data-parallel training on the real kernels, two ranks (reference: accelerate -> DistributedDataParallel,
train.py:115-117,176-178,251).  A gpurun box has ONE GPU and RCCL refuses two ranks on one device, so both ranks sit on
cuda:0 and the collectives go through gloo; everything else -- model, autograd nodes writing into the bucket views,
hook-launched reductions, fused optimizer with the deferred 1 / world -- is the product path."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("primitive", ["all_reduce", "reduce_scatter"])
def test_two_ranks_real_model_free_guidance_train_steps(tmp_path, primitive):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parallel_rank as W
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "parallel_rank.py"), str(tmp_path), primitive],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    res = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]

    # (1) every gradient was born inside its communication bucket: nothing was copied in or out
    assert all(r["born_in_bucket"] and r["copied_in"] == 0 and r["grad_scale"] == 0.5 for r in res)
    # (2) the weights the ranks ended with are identical (same start after the broadcast, same averaged gradients)
    for k, v in res[0]["weights"].items():
        assert torch.equal(v, res[1]["weights"][k]), k
    # (3) step-0 gradients = mean of two single-process runs on the two shards, from rank 0's initial weights
    singles = []
    for r in range(world):
        m = W.build(seed=0)
        W.loss_of(m, W.shard(r, 0), drop=W.drops_cond(r, 0)).backward()
        singles.append({k: p.grad.cpu() for k, p in m.named_parameters()})
        if r == 0:
            bufs0 = {k: b.clone().cpu() for k, b in m.named_buffers()}
        del m
    # rank 1 trained step 0 with cond=None: its d(cond_mlp.0.weight) is all zero, so the mean is half of rank 0's
    assert singles[1]["cond_mlp.0.weight"].abs().max().item() == 0.0 and singles[0]["cond_mlp.0.weight"].abs().max().item() > 0.0
    worst = 0.0
    for k, g in res[0]["grads"].items():
        assert torch.equal(g, res[1]["grads"][k]), k                  # both ranks hold the same reduced bucket
        want = 0.5 * (singles[0][k] + singles[1][k])
        err = ((g - want).norm() / (want.norm() + 1e-12)).item()
        worst = max(worst, err)
        assert err <= 1e-5, (k, err)      # the same kernels on the same data: only the order of the two-term sum differs
    # (4) BatchNorm running statistics: rank 0's after step 0 equal the single-process run on shard 0 (never averaged) ...
    for k, b in res[0]["buffers_after_step0"].items():
        assert torch.allclose(b.float(), bufs0[k].float(), atol=1e-6), k
    # ... the ranks' statistics differ between broadcasts, and a broadcast makes them rank 0's bit for bit
    rm = "perception.bn1.running_mean"
    assert not torch.equal(res[0]["buffers_before_sync"][rm], res[1]["buffers_before_sync"][rm])
    for k, b in res[0]["buffers_before_sync"].items():
        assert torch.equal(b, res[0]["buffers_after_sync"][k]) and torch.equal(b, res[1]["buffers_after_sync"][k]), k
    assert all(torch.isfinite(torch.tensor([r["loss0"], r["loss1"]])).all() for r in res)


def test_bench_two_ranks_same_device_fullsize_free_train_leg():
    """`bench.py --gpus 2` end to end on one GPU (ADX_BENCH_SAME_DEVICE=1: both ranks on cuda:0, gloo collectives): the
    launcher, the sharded sampling legs and -- the point -- BASELINE configs[4]'s per-GPU workload as the training leg:
    FREE_GUIDANCE at B = 64 per rank, H = 32, 3 x 256 x 900 under parallel.DataParallel with train.py:236-242's per-rank
    cond=None draw.  Every gradient must have been born in its bucket (nothing copied), every rank reports its own step
    time, the first step's loss is the oracle's, and the traced step reports when each bucket's reduction ran relative to
    the end of backward.  The times of such a run mean nothing."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["ADX_BENCH_SAME_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--train-steps", "3", "--short-sampling", "--no-roofline"], env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = lines[0]
    assert res["n_gpus"] == 2 and len(res["per_rank_ms_per_step"]) == 2 and res["value"] > 0
    assert "train" not in res                      # N > 1: the training leg IS configs[4]
    tf = res["train_free"]
    assert "error" not in tf and "parity_failed" not in tf, tf
    assert "FREE_GUIDANCE" in tf["workload"] and tf["value"] > 0 and len(tf["per_rank_ms_per_step"]) == 2
    assert abs(tf["first_step_loss"] - tf["first_step_loss_oracle"]) <= 2e-5 * max(1.0, abs(tf["first_step_loss_oracle"]))
    gb = tf["gradient_buckets"]
    assert gb["copied_in_last_step"] == 0 and gb["n"] >= 2
    ov = gb["overlap_rank0"]
    assert len(ov) == gb["n"] and all(o["done_ms"] >= o["ready_ms"] for o in ov)
    # the temporal stack's buckets are complete long before backward ends (the perception backward is two thirds of the step)
    assert min(o["ready_ms"] for o in ov) < -5.0, ov
    # round 5: the perception backward hands out a completion event per layer group (adx_resnet_backward_events), so every bucket
    # but the last (layer3's lower blocks ... the stem) is ready -- and its reduction launched -- while the layers below it
    # are still being differentiated; before, all 85 MB of perception gradients became "ready" together at the end of backward
    assert all(o["ready_ms"] < -3.0 for o in ov[:-1]), ov


@pytest.mark.parametrize("primitive", ["all_reduce", "reduce_scatter"])
def test_one_rank_rccl_behind_the_events(tmp_path, primitive):
    """RCCL behind the averager's events on the one GPU a box has (train.py:174-178,249): a ONE-rank nccl group with
    `GradientAverager(force=True)` keeps the world-1 early returns out, so the real RCCL kernels run on the side stream behind
    the perception backward's per-layer-group events while backward is still running.  Full-size FREE_GUIDANCE step (B = 64,
    H = 32, 3 x 256 x 900): one-rank collectives are identities, so every gradient is what a plain backward gives -- bit for bit wherever a
    plain step is bit-reproducible, like two plain runs elsewhere (float atomics); nothing is
    copied in; every bucket but the last is ready >= 3 ms before backward ends.  Then the two paths the per-group events do NOT
    cover: accumulation over two backwards without clearing .grad, and the module twice in one graph -- bit-equal as well.
    The all_reduce run's trace is the artifact profiles/r06_overlap_trace.json is a copy of."""
    import json
    out = os.path.join(tmp_path, "res.json")
    trace = os.path.join(ROOT, "gpurun_out", f"overlap_trace_{primitive}.json")
    os.makedirs(os.path.dirname(trace), exist_ok=True)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(ADX_WGRAD_DETERMINISTIC="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_worker.py"), out, primitive, "64", "32", "256", "900",
                        trace], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = json.load(open(out))
    assert "nccl" in res["backend"]
    assert res["n_buckets"] >= 3 and res["born_in_bucket"] and res["copied_in"] == 0 and res["loss_equal"]
    # bit-equal wherever a plain step is bit-reproducible at all (the 3x3 conv weights under ADX_WGRAD_DETERMINISTIC=1, every
    # BatchNorm affine ...: the worker runs the plain step twice to find out); the tensors reduced with float atomics agree like
    # two plain runs do
    assert res["n_bit_reproducible"] >= 100, res
    tol = max(10 * res["plain_vs_plain_worst_rel"], 2e-6)
    assert res["step_mismatch"] == [] and res["step_worst_rel"] <= tol, (res["step_mismatch"][:5], res["step_worst_rel"], tol)
    ov = res["overlap"]
    assert len(ov) == res["n_buckets"] and all(o["done_ms"] >= o["ready_ms"] for o in ov), ov
    assert all(o["ready_ms"] < -3.0 for o in ov[:-1]), ov          # reduced while the layers below are still differentiated
    assert res["accum_mismatch"] == [] and res["accum_worst_rel"] <= tol, (res["accum_mismatch"][:5], res["accum_worst_rel"], tol)
    assert res["twice_mismatch"] == [] and res["twice_worst_rel"] <= tol, (res["twice_mismatch"][:5], res["twice_worst_rel"], tol)
