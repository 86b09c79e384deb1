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
        W.loss_of(m, W.shard(r, 0)).backward()
        singles.append({k: p.grad.cpu() for k, p in m.named_parameters()})
        if r == 0:
            bufs0 = {k: b.clone().cpu() for k, b in m.named_buffers()}
        del m
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
