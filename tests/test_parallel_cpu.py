"""CPU, world_size 2 over gloo: the data-parallel host logic (scene sharding for sampling, bucketed
gradient averaging and parameter broadcast for training).  No kernels involved."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from autonomous_driving_with_diffusion_model_amd.parallel import (GradientAverager, broadcast_parameters, make_buckets,
                                                                  shard_range)


def test_shard_range_partitions_every_scene_once():
    for n, w in ((64, 8), (65, 8), (7, 8), (512, 3), (1, 1)):
        covered = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            assert 0 <= lo <= hi <= n and hi - lo in (n // w, n // w + 1)
            covered += list(range(lo, hi))
        assert covered == list(range(n))


def test_buckets_cover_all_in_reverse_order():
    sizes = [10, 200, 30, 4000, 5, 60]
    b = make_buckets(sizes, 250)
    flat = [i for bucket in b for i in bucket]
    assert flat == list(reversed(range(len(sizes))))
    assert all(sum(sizes[i] for i in bucket) <= 250 or len(bucket) == 1 for bucket in b)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(rank)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
    broadcast_parameters(model, src=0)
    w_after_bcast = [p.detach().clone() for p in model.parameters()]
    x = torch.full((3, 8), float(rank + 1))
    model(x).sum().backward()
    local = [p.grad.clone() for p in model.parameters()]
    avg = GradientAverager(model.parameters(), bucket_mb=1e-4)     # tiny buckets -> several collectives
    works = avg.average(async_op=True)
    avg.finish(works)
    gathered = [torch.zeros_like(local[0]) for _ in range(world)]
    dist.all_gather(gathered, local[0])
    ok_avg = torch.allclose(list(model.parameters())[0].grad, sum(gathered) / world, atol=1e-6)
    # overlap mode: the hooks launch each bucket during backward; synchronize() must give the same means
    model.zero_grad()
    avg2 = GradientAverager(model.parameters(), bucket_mb=1e-4).attach()
    model(x).sum().backward()
    avg2.synchronize()
    ok_avg = ok_avg and all(torch.allclose(p.grad, g) for p, g in zip(model.parameters(),
                                                                       [q.grad.clone() for q in model.parameters()]))
    ok_avg = ok_avg and torch.allclose(list(model.parameters())[0].grad, sum(gathered) / world, atol=1e-6)
    model.zero_grad()
    model(x).sum().backward()          # hooks stay attached: a second step works too
    avg2.synchronize()
    ok_avg = ok_avg and torch.allclose(list(model.parameters())[-1].grad * 0 + list(model.parameters())[0].grad[0, 0],
                                       list(model.parameters())[-1].grad * 0 + (sum(gathered) / world)[0, 0], atol=1e-6)
    avg2.detach()
    ws = [torch.zeros_like(w_after_bcast[0]) for _ in range(world)]
    dist.all_gather(ws, w_after_bcast[0])
    ok_bcast = all(torch.equal(ws[0], w) for w in ws)
    out.put((rank, bool(ok_avg), bool(ok_bcast), len(avg.buckets)))
    dist.destroy_process_group()


def test_gradient_averaging_and_broadcast_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] and r[2] for r in res), res
    assert res[0][3] > 1


class _BucketBornGrad(torch.autograd.Function):
    """A backward node written like this package's (modeling/temporal.py:_UnetTrainFn.backward): the parameter gradient
    is written into the buffer `_lib.grad_buffer` hands out and returned."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        from autonomous_driving_with_diffusion_model_amd import _lib as L
        x, w = ctx.saved_tensors
        gw = L.grad_buffer(w)
        torch.mm(g.t(), x, out=gw)
        return g @ w, gw


def _inplace_worker(rank, world, port, out, primitive):
    """In-place buckets: gradients born inside the flat bucket are reduced without copies (copied_in == 0); a foreign
    gradient (plain torch module) in the same averager is copied in and out; mean=False leaves the sum + grad_scale."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(5, 7))            # 35 elements: not a multiple of the world size (padding)
    lin = torch.nn.Linear(5, 3)
    x = torch.randn(4, 7) * (rank + 1)
    ok = True
    for mean in (True, False):
        avg = GradientAverager([w, *lin.parameters()], bucket_mb=1e-3, primitive=primitive, mean=mean).attach()
        for step in range(2):
            w.grad = None
            lin.zero_grad(set_to_none=True)
            lin(_BucketBornGrad.apply(x, w)).square().sum().backward()
            born_in_bucket = w.grad.data_ptr() == w._adx_grad_view.data_ptr()
            # reference values: the same loss with plain autograd, gathered over the ranks
            w2 = w.detach().clone().requires_grad_()
            l2 = torch.nn.Linear(5, 3)
            l2.load_state_dict(lin.state_dict())
            l2(x @ w2.t()).square().sum().backward()
            avg.synchronize()
            for got, loc in ((w.grad, w2.grad), (lin.weight.grad, l2.weight.grad), (lin.bias.grad, l2.bias.grad)):
                parts = [torch.zeros_like(loc) for _ in range(world)]
                dist.all_gather(parts, loc)
                want = sum(parts) / world
                ok = ok and torch.allclose(got * avg.grad_scale, want, atol=1e-5, rtol=1e-5)
            ok = ok and born_in_bucket and avg.copied_in == 2            # the Linear's two gradients only
            ok = ok and (avg.grad_scale == (1.0 if mean else 1.0 / world))
        # accumulation over two backwards without clearing: the node must NOT write into the live .grad's storage
        w.grad = None
        lin.zero_grad(set_to_none=True)
        avg.detach()
        for _ in range(2):
            lin(_BucketBornGrad.apply(x, w)).square().sum().backward()
        w2 = w.detach().clone().requires_grad_()
        (2 * lin(x @ w2.t()).square().sum()).backward()
        ok = ok and torch.allclose(w.grad, w2.grad, atol=1e-4, rtol=1e-5)
        # one parameter feeding TWO nodes of one graph (the model called twice before one backward; shared weights): both
        # nodes run before AccumulateGrad, so both see `.grad is None` -- the bucket view may be lent to one of them only
        avg = GradientAverager([w, *lin.parameters()], bucket_mb=1e-3, primitive=primitive, mean=mean).attach()
        for step in range(2):
            w.grad = None
            lin.zero_grad(set_to_none=True)
            x2 = torch.randn(4, 7, generator=torch.Generator().manual_seed(5 + rank))
            (lin(_BucketBornGrad.apply(x, w)).square().sum() + 3.0 * lin(_BucketBornGrad.apply(x2, w)).square().sum()).backward()
            w2 = w.detach().clone().requires_grad_()
            ref_lin = lambda v: torch.nn.functional.linear(v, lin.weight.detach(), lin.bias.detach())  # noqa: E731
            (ref_lin(x @ w2.t()).square().sum() + 3.0 * ref_lin(x2 @ w2.t()).square().sum()).backward()
            avg.synchronize()
            parts = [torch.zeros_like(w2.grad) for _ in range(world)]
            dist.all_gather(parts, w2.grad)
            ok = ok and torch.allclose(w.grad * avg.grad_scale, sum(parts) / world, atol=1e-4, rtol=1e-5)
        avg.detach()
    out.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_inplace_buckets_reduce_scatter_and_deferred_scale_world2():
    ctx = mp.get_context("spawn")
    for primitive in ("all_reduce", "reduce_scatter"):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_inplace_worker, args=(r, 2, port, q, primitive)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert all(r[1] for r in res), (primitive, res)


def _dp_worker(rank, world, port, out):
    """The reference's DDP semantics on a small conv + BatchNorm net: per-forward buffer broadcast from rank 0
    (statistics are NOT averaged), hook-driven gradient means, and the LR schedule that ticks `world` times per
    optimizer step under accelerate (FusedAdamWEMA.lr_ticks_per_step)."""
    from autonomous_driving_with_diffusion_model_amd.optim import FusedAdamWEMA
    from autonomous_driving_with_diffusion_model_amd.parallel import DataParallel
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different initial weights per rank: the wrapper must fix that
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                              torch.nn.Conv2d(8, 4, 1))
    dp = DataParallel(net, bucket_mb=1e-4)
    fused = FusedAdamWEMA(net.parameters(), lr=1e-2, warmup_steps=8, lr_ticks_per_step=world, use_ema=False)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2, betas=(0.95, 0.999), eps=1e-7)
    ok = True
    bn = net[1]
    for step in range(3):
        g = torch.Generator().manual_seed(1000 * step + rank)
        x = torch.randn(4, 3, 6, 6, generator=g) + rank          # rank-dependent statistics
        before = [torch.zeros_like(bn.running_mean) for _ in range(world)]
        dist.all_gather(before, bn.running_mean.clone())
        out_ = dp(x)
        # the forward started from rank 0's buffers on every rank: new = 0.9 * rank0_old + 0.1 * own batch mean
        pre = net[0](x)
        want = 0.9 * before[0] + 0.1 * pre.mean(dim=(0, 2, 3))
        ok = ok and torch.allclose(bn.running_mean, want, atol=1e-5)
        ok = ok and int(bn.num_batches_tracked) == step + 1
        out_.square().mean().backward()
        local = [p.grad.clone() for p in net.parameters()]
        dp.synchronize()
        for p, l in zip(net.parameters(), local):
            parts = [torch.zeros_like(l) for _ in range(world)]
            dist.all_gather(parts, l)
            ok = ok and torch.allclose(p.grad, sum(parts) / world, atol=1e-6)
        lr = fused.current_lr()
        ok = ok and abs(lr - 1e-2 * min(1.0, step * world / 8)) < 1e-12     # accelerate: `world` ticks per step
        for grp in opt.param_groups:
            grp["lr"] = lr
        opt.step()
        opt.zero_grad()
        fused.step_count += 1
    # identical averaged gradients + identical start => identical weights on every rank; statistics differ (not synced)
    w = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    ok = ok and all(torch.equal(ws[0], v) for v in ws)
    rm = [torch.zeros_like(bn.running_mean) for _ in range(world)]
    dist.all_gather(rm, bn.running_mean.clone())
    stats_differ = not torch.allclose(rm[0], rm[1])
    # an eval-mode forward issues no collective (DDP only syncs buffers when it will run a training forward)
    net.eval()
    n_before = dp.buffers_sync.sync.__self__ is dp.buffers_sync
    with torch.no_grad():
        dp(x)
    out.put((rank, bool(ok), bool(stats_differ), bool(n_before)))
    dist.destroy_process_group()


def test_data_parallel_wrapper_semantics_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert all(r[2] for r in res), "BatchNorm statistics must stay per rank between broadcasts"


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it must start N ranks itself (reference: `accelerate launch`,
    train.py:115-117), report the world size the process group saw, and fail as a whole when one rank fails."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines == [{"launch_check": True, "n_gpus": 3, "sum": 6.0, "env_world": 3}]
    # a launcher that started a different number of ranks than --gpus is an error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
    # one failing rank fails the run
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=dict(env, ADX_LAUNCH_CHECK_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 7, (r.returncode, r.stderr[-500:])


def _forced_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    res = {}
    for primitive in ("all_reduce", "reduce_scatter"):
        model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))
        plain = [p.detach().clone().requires_grad_() for p in model.parameters()]
        x = torch.randn(4, 5)
        idle = GradientAverager(model.parameters(), bucket_mb=1e-4, primitive=primitive)          # world 1, not forced: a no-op
        forced = GradientAverager(model.parameters(), bucket_mb=1e-4, primitive=primitive, force=True).attach()
        assert not idle.active and idle.average() == [] and forced.active and forced.world == 1 and len(forced.buckets) >= 2
        for step in range(2):                      # the second backward accumulates into .grad (no zero_grad in between)
            model(x * (step + 1)).square().sum().backward()
            forced.synchronize()
            h = torch.tanh(torch.nn.functional.linear(x * (step + 1), plain[0], plain[1]))
            torch.nn.functional.linear(h, plain[2], plain[3]).square().sum().backward()
        res[primitive] = {"equal": all(torch.equal(p.grad, q.grad) for p, q in zip(model.parameters(), plain)),
                          "copied_in_second": forced.copied_in}
        forced.detach()
    torch.save(res, out)
    dist.destroy_process_group()


def test_forced_one_rank_collectives_are_identities(tmp_path):
    """GradientAverager(force=True) / ADX_FORCE_COLLECTIVES=1 (the GPU form runs RCCL on a one-rank group behind the perception
    backward's events: tests/test_gpu_parallel.py): with one rank every collective is an identity -- the gradients are what a plain
    backward gives, also when a second backward accumulates into buckets that were already reduced once -- and without the switch a
    one-rank averager does nothing at all."""
    out = str(tmp_path / "forced.pt")
    mp.spawn(_forced_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    res = torch.load(out)
    for primitive, r in res.items():
        assert r["equal"], primitive
