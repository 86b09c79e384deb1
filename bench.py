#!/usr/bin/env python3
"""Headline benchmark: denoising-steps/sec of 50-step DDIM sampling with classifier-free guidance,
64 scenes per GPU, horizon 32, camera image 3x256x900 (BASELINE.json configs[2]; north_star target).

One "step" = one `model.forward` (ResNet-34 perception on the 64 images + TemporalMapUnet on the
2x64 CFG batch) + one `scheduler.step` (CFG combine + DDIM update + `[:,0,:3]=0`) advancing all 64
trajectories by one timestep, i.e. exactly the per-step work of the reference's sampling loop
(interact.py:131-164, which re-runs the perception every step: modeling/temporal.py:203).  `value`
is measured in that reference-faithful mode.  The `hoisted` object reports the product's default
mode, where the perception feature is memoised per image tensor (one ResNet pass per scene).

Contract: python bench.py --gpus N --steps K --warmup W  ->  one JSON line on rank 0.
Multi-GPU: scenes are independent, every rank samples its own 64 scenes, no collective on the data
path ("scaling": "weak"); the barrier only brackets the timed region.
"""
import argparse
import json
import os
import sys
import time

# Kernel arguments in device memory (what the package itself asks for at import, autonomous_driving_with_diffusion_model_amd/__init__.py;
# the ROCm runtime reads the variable when it initialises, and this script touches the GPU before it imports the package): same
# box, alternating: 158.8 -> 161.5 reference-faithful steps/s, hoisted eager loop 1742 -> 2007, train 31.1 -> 30.0 ms;
# graph replays unchanged (profiles/r06_ab_dev_kernarg_bench.txt).  Reported in the line as `runtime_env`.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, H, D, IMG = 64, 32, 7, (256, 900)
N_TRAIN, N_INFER, FREE_SCALE = 100, 50, 7.5
SCHED_KW = dict(num_train_timesteps=N_TRAIN, prediction_type="sample", beta_schedule="squaredcos_cap_v2",
                beta_start=1e-4, beta_end=0.02)
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak
PEAK_F16_TFLOPS = 2516.6  # dense fp16/bf16 MFMA: 16 x the fp32 rate (same guide; AMD's ~2.5 PF figure)
PEAK_HBM_GBS = 8000.0


def resnet_conv_table(h, w):
    """(cin, cout, k, stride, pad, in_h, in_w) for the 36 convs of ResNet-34 on an h x w image."""
    out = lambda n, k, s, p: (n + 2 * p - k) // s + 1  # noqa: E731
    t = [(3, 64, 7, 2, 3, h, w)]
    h, w = out(out(h, 7, 2, 3), 3, 2, 1), out(out(w, 7, 2, 3), 3, 2, 1)
    inpl = 64
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for bi in range(n):
            s = 2 if (li > 0 and bi == 0) else 1
            t.append((inpl, planes, 3, s, 1, h, w))
            oh, ow = out(h, 3, s, 1), out(w, 3, s, 1)
            t.append((planes, planes, 3, 1, 1, oh, ow))
            if s != 1 or inpl != planes:
                t.append((inpl, planes, 1, s, 0, h, w))
            inpl, h, w = planes, oh, ow
    return t


def conv_flops(n, cin, cout, k, s, p, h, w):
    oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    return 2.0 * n * cout * oh * ow * cin * k * k


def max_over_ranks(dt, world, dev):
    """(max over ranks, every rank's own time): the contract's time is the slowest rank's; the per-rank list makes a scaling
    run diagnosable (one slow GPU / one slow link shows up as one outlier instead of as a mystery in the maximum)."""
    if world == 1:
        return dt, [dt]
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    parts = [torch.zeros_like(t) for _ in range(world)]
    torch.distributed.all_gather(parts, t)
    per = [p.item() for p in parts]
    return max(per), per


def time_events(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps  # ms per call


def pmc_traffic_file():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    return files[-1] if files else None


def pmc_traffic(key):
    """HBM bytes per launch from the newest committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json); rocprof cannot
    run inside this process, so the number is read back with its provenance."""
    try:
        with open(pmc_traffic_file()) as f:
            for k, v in json.load(f)["kernels"].items():
                if k.startswith(key):
                    return round(v["traffic"])
    except Exception:
        pass
    return None


def pmc_traffic_per_shape(key):
    """{"64->64 @64x225": {"fetch": .., "write": ..}, ...} of the newest committed PMC passes for kernel `key` ({} if none)."""
    try:
        with open(pmc_traffic_file()) as f:
            for k, v in json.load(f)["kernels"].items():
                if k.startswith(key):
                    return {name.split(" (")[0]: rec for name, rec in v.get("per_shape", {}).items()}
    except Exception:
        pass
    return {}


def rocprof_avg_ms(prefix):
    """Launch-weighted average duration of the kernels whose name contains `prefix` (a string, several that must all
    occur, or a predicate on the name) in the newest committed rocprofv3 --kernel-trace --stats summary of this command (profiles/r*_bench_kernel_stats.csv)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")))
    if not files:
        return None
    calls = tot = 0
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if prefix(row["Name"]) if callable(prefix) else all(part in row["Name"] for part in ([prefix] if isinstance(prefix, str) else prefix)):
                calls += int(row["Calls"])
                tot += int(row["TotalDurationNs"])
    return round(tot / calls / 1e6, 4) if calls else None


def conv2d_roofline(dev, reps=10):
    """Dominant kernel of the timed region: conv2d_hs3x3_kernel (the 29 stride-1 3x3 convs of one
    perception pass).  Every distinct shape is launched alone through the C ABI, in each operand layout the executor uses for it
    (fp32 NCHW or pre-split cells, adx_conv2d_forward_cells), and timed with HIP events on the launch stream; the
    launch-mix average is what rocprofv3's per-kernel average shows.

    The kernel produces an fp32-grade result on the fp16 matrix cores: every operand is split into an fp16 hi and a
    scaled fp16 lo part and each algorithmic multiply-add is issued as three v_mfma_f32_32x32x16_f16 products
    (csrc/conv2d_hs.hip), so the roofline is the dense fp16 MFMA peak.  `achieved` / `frac` count the ALGORITHMIC work
    (the convolution's 2*M*N*K); `achieved_issued` / `frac_issued` the MFMA work actually issued = 3 x that."""
    from autonomous_driving_with_diffusion_model_amd import ops
    shapes = {}
    for c in resnet_conv_table(*IMG):
        if c[3] == 1:
            shapes[c] = shapes.get(c, 0) + 1
    tot_ms = tot_fl = tot_bytes = tot_alone = 0.0
    count = 0
    per_shape = []
    pmc = pmc_traffic_per_shape("conv2d_hs3x3_kernel")
    for (cin, cout, k, s, p, h, w), cnt in shapes.items():
        x = torch.randn((B, cin, h, w), device=dev)
        wt = torch.randn((cout, cin, k, k), device=dev) * (1.0 / (cin * k * k)) ** 0.5
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        res = torch.randn((B, cout, h, w), device=dev)
        y, packed = ops.conv2d(x, wt, stride=s, pad=p, scale=sc, shift=sh, relu=True)
        xc, rc = ops.to_cells(x), ops.to_cells(res)
        kw = dict(scale=sc, shift=sh, relu=True)
        run = lambda **f: (lambda: ops.conv2d_cells(xc if f["x_cells"] else x, packed, cin, cout, B, h, w, **f, **kw))  # noqa: E731
        # the launches of this shape as adx_resnet_forward issues them at this batch (csrc/conv2d.hip): every 3x3 stride-1 conv reads and
        # writes cells (the pooled stem map and both outputs of the stride-2 launches are cell tensors too); conv2 of every block
        # adds a cell residual.  Layer1 has as many conv1 as conv2 launches here, the other layers' first conv1 is the stride-2 one.
        n2 = (cnt + 1) // 2
        variants = [("cells in", run(x_cells=True), cnt - n2), ("cells in, cell residual", run(x_cells=True, res=rc, res_cells=True), n2)]
        assert sum(v[2] for v in variants) == cnt, (cnt, variants)
        ms_alone = 0.0
        detail = {}
        for name, fn, c in variants:
            fn()
            t = time_events(fn, reps)
            detail[name] = round(t, 4)
            ms_alone += t * c
        ms_alone /= cnt
        # ... and as they run inside a perception pass: the layer's launches in the executor's order on its three rotating
        # buffers (conv1: block input -> mid; conv2: mid + residual block input -> out; out is the next block's input), so every
        # launch reads what the launch before it wrote -- out of the Infinity Cache, as in the timed region -- instead of a
        # 236 MB tensor that fell out of it since the previous repetition
        bufs = [xc.clone(), xc.clone(), xc.clone()]
        first_is_conv2 = cnt % 2 == 1          # layers 2-4: the block's first conv is the stride-2 launch (another kernel)

        def chain():
            cur = 0
            for li in range(cnt):
                conv2 = (li % 2 == 0) if first_is_conv2 else (li % 2 == 1)
                if conv2:      # reads mid, adds the block input (or, first block of layers 2-4, the downsample output)
                    ops.conv2d_cells(bufs[(cur + 1) % 3], packed, cin, cout, B, h, w, x_cells=True, res=bufs[cur], res_cells=True,
                                     out=bufs[(cur + 2) % 3], **kw)
                    cur = (cur + 2) % 3
                else:
                    ops.conv2d_cells(bufs[cur], packed, cin, cout, B, h, w, x_cells=True, out=bufs[(cur + 1) % 3], **kw)
        chain()
        ms = 0.0
        nrep = max(3, reps // 2)
        for _ in range(nrep):        # every repetition starts from the same activations (a layer's output feeds its own input here,
            bufs[0].copy_(xc)        # and a residual chain grows): the copy stands for the previous layer's launch, outside the events
            bufs[1].copy_(xc)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            chain()
            e1.record()
            e1.synchronize()
            ms += e0.elapsed_time(e1)
        ms /= nrep * cnt
        detail["in the executor's launch order (per launch)"] = round(ms, 4)
        del bufs
        fl = conv_flops(B, cin, cout, k, s, p, h, w)
        byts = 4.0 * (x.numel() + y.numel() + wt.numel() + res.numel() * n2 / cnt)     # conv2 of every block also reads its residual
        tot_alone += ms_alone * cnt
        rec = {"shape": f"{cin}->{cout} k{k} @{h}x{w}", "count": cnt, "ms": round(ms, 4), "ms_launched_alone": round(ms_alone, 4),
               "ms_by_operand_layout": detail,
               "algorithmic_tflops": round(fl / ms / 1e9, 1), "issued_mfma_tflops": round(3 * fl / ms / 1e9, 1),
               "algorithmic_mb": round(byts / 1e6, 1)}
        t = pmc.get(f"{cin}->{cout} @{h}x{w}")
        if t is not None:       # HBM-side bytes of this shape from the committed PMC passes, against its algorithmic bytes
            rec["pmc_traffic_mb"] = round((t["fetch"] + t["write"]) / 1e6, 1)
            rec["pmc_traffic_over_algorithmic"] = round((t["fetch"] + t["write"]) / byts, 2)
        per_shape.append(rec)
        tot_ms += ms * cnt
        tot_fl += fl * cnt
        tot_bytes += byts * cnt
        count += cnt
        del x, y, wt, packed, xc, rc, res
    avg_ms = tot_ms / count
    equiv = tot_fl / count / avg_ms / 1e9      # algorithmic TFLOP/s
    achieved = 3.0 * equiv                     # fp16 MFMA TFLOP/s issued
    sus = sustained_mfma(dev)
    return {"kernel": "conv2d_hs3x3q_kernel (16x16x32 MFMA: 128 / 256 / 512-channel layers) + conv2d_hs3x3_kernel<0> (32x32x16: 64-channel "
                      "layers) -- the ResNet-34 3x3 stride-1 convs, B=64, 3x256x900; fp32-grade result from fp16 hi/lo split operands, "
                      "3 MFMA products per multiply-add",
            "bound": "mfma", "achieved": round(equiv, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(equiv / PEAK_F16_TFLOPS, 4),
            "achieved_issued": round(achieved, 1), "frac_issued": round(achieved / PEAK_F16_TFLOPS, 4),
            # measured in this run, same box, same clocks: the ceiling the power limit leaves under `peak`
            "method": "in_order",
            "sustained": {"mfma_fp16_tflops_random_operands": sus["random"], "mfma_fp16_tflops_zero_operands": sus["zero"],
                          "by_mfma_shape_random_operands": {"32x32x16": sus["random_32x32x16"], "16x16x32": sus["random_16x16x32"]},
                          "frac_of_peak": round(sus["random"] / PEAK_F16_TFLOPS, 4),
                          "achieved_issued_over_sustained": round(achieved / sus["random"], 4),
                          "note": "adx_probe_mfma_fp16 / adx_probe_mfma_fp16_16x16x32: the 3x3 kernels' inner loops alone (8 LDS operand reads per "
                                  "12 v_mfma_f32_32x32x16_f16, resp. 16 reads per 48 v_mfma_f32_16x16x32_f16: the same flops and reads per trip; "
                                  "two 4-wave workgroups per CU, no global memory) -- the fp16 MFMA rate this chip sustains at its power limit, "
                                  "the BETTER of the two shapes; `peak` is the datasheet figure at the boost clock.  A 3-product launch "
                                  "cannot exceed sustained / 3 algorithmic."},
            "achieved_note": "achieved / frac (the contract figures, SURVEY 8d) = ALGORITHMIC conv flops (2*M*N*K) per launch / "
                             "HIP-event launch time, against the dense fp16 MFMA peak; achieved_issued / frac_issued = the fp16 "
                             "MFMA flops the kernel actually issues (3 x algorithmic: hi*hi + hi*lo + lo*hi) / the same time",
            # the inference instantiations (cell tensors in and out); the same summary also holds the training leg's launches of
            # these kernels (statistics epilogues, fp32 outputs)
            "avg_launch_ms_rocprof": rocprof_avg_ms(lambda n: ("conv2d_hs3x3q_kernel<" in n and ", 0>(" in n)             # (<DMA, TRAIN = 0>)
                                                    or ("conv2d_hs3x3_kernel<" in n and ", 0, true, true" in n)),          # (<MODE, STATS = 0, XCELLS, YCELLS, ..>)
            "fp32_mfma_peak_tflops": PEAK_F32_TFLOPS, "frac_of_fp32_mfma_peak": round(equiv / PEAK_F32_TFLOPS, 3),
            "traffic": pmc_traffic("conv2d_hs3x3_kernel"),
            "traffic_note": "bytes/launch, 2 x FETCH_SIZE (gfx950 correction, calibrated: tools/micro/fetch_calib.hip) + WRITE_SIZE from profiles/" + os.path.basename(pmc_traffic_file() or "(none)") + " (separate rocprofv3 --pmc passes)",
            "avg_launch_ms": round(avg_ms, 4), "launches_per_step": count,
            "avg_launch_ms_rocprof_note": "launch-weighted average of the same kernels in the committed rocprofv3 summary of this command run with "
                                          "ADX_RESNET_STREAMS=1 ADX_PERCEPTION_AHEAD=0 (one chain of launches: a kernel that shares the chip with "
                                          "another sub-batch's has no duration of its own); that run's mix = faithful steps at B = 64 + this "
                                          "probe's chains and fixed-buffer repetitions",
            "avg_launch_ms_note": "HIP events around each layer's launches issued in the executor's order on its rotating buffers (every "
                                  "launch reads what the one before it wrote, as inside the timed region); avg_launch_ms_alone = the same "
                                  "launches repeated on fixed buffers (what rounds 1-3 reported: their 236 MB operands fall out of the 256 MB "
                                  "Infinity Cache between repetitions, which they do not inside a pass)",
            "avg_launch_ms_alone": round(tot_alone / count, 4),
            "frac_alone": round(tot_fl / count / (tot_alone / count) / 1e9 / PEAK_F16_TFLOPS, 4),
            "algorithmic_gflop_per_launch": round(tot_fl / count / 1e9, 2),
            "mfma_gflop_per_launch": round(3 * tot_fl / count / 1e9, 2),
            "algorithmic_mb_per_launch": round(tot_bytes / count / 1e6, 2),
            "hbm_gbs_at_algorithmic_bytes": round(tot_bytes / count / avg_ms / 1e6, 1), "per_shape": per_shape}


def sustained_mfma(dev):
    """What the matrix pipe of THIS chip sustains (adx_probe_mfma_fp16: the 3x3 kernel's inner loop -- 8 LDS operand reads per 12
    v_mfma_f32_32x32x16_f16 -- with no global traffic, staging or epilogue, two 4-wave workgroups per CU like the kernel), on random
    and on all-zero operands.  The datasheet peak assumes the boost clock; under matrix load the shader clock settles at the
    power limit, and how far depends on how much the operands toggle."""
    import ctypes as C
    from autonomous_driving_with_diffusion_model_amd import _lib as L
    out = torch.empty(512 * 256, dtype=torch.float32, device=dev)
    fl = C.c_double(0.0)
    res = {}
    probes = {"32x32x16": L.lib().adx_probe_mfma_fp16, "16x16x32": L.lib().adx_probe_mfma_fp16_16x16x32}
    for name in ("random", "zero"):
        if name == "random":      # fp16 values of both signs in [0.125, 1): every mantissa bit toggles
            ops_ = ((torch.rand(4096 * 8, device=dev) * 0.875 + 0.125) * (torch.randint(0, 2, (4096 * 8,), device=dev) * 2 - 1)).half()
        else:
            ops_ = torch.zeros(4096 * 8, dtype=torch.float16, device=dev)
        for shape, probe in probes.items():
            fn = lambda: L.check(probe(ops_.data_ptr(), out.data_ptr(), 512, 4000, C.byref(fl), L.stream_ptr(dev)), "adx_probe_mfma_fp16")  # noqa: E731
            ms = time_events(fn, 3, warm=1)
            res[f"{name}_{shape}"] = round(fl.value / ms / 1e9, 1)
        res[name] = max(res[f"{name}_32x32x16"], res[f"{name}_16x16x32"])
    return res


def tconv_roofline(model, dev, reps=20):
    """The temporal Conv1d kernel at the CFG batch (2x64): the seven 512->512 k5 convs at L=4."""
    from autonomous_driving_with_diffusion_model_amd import ops
    rows, c, L = 2 * B, 512, 4
    x = torch.randn((rows, c, L), device=dev)
    w = torch.randn((c, c, 5), device=dev) * (1.0 / (5 * c)) ** 0.5
    b, g, be = (torch.randn(c, device=dev) * 0.1 for _ in range(3))
    fn = lambda: ops.tconv(x, w, b, pad=2, gn_weight=g + 1, gn_bias=be, groups=8)  # noqa: E731  (includes the pack launch)
    # time the conv launch alone: pre-pack once through the model-level path instead
    import ctypes as C
    from autonomous_driving_with_diffusion_model_amd import _lib as Lb
    d = Lb.TConvDesc(0, 5, 1, 2, c, 0, c, L, L, 8, 1e-5)
    packed = torch.empty(Lb.lib().adx_tconv_packed_bytes(C.byref(d)) // 4, device=dev)
    Lb.check(Lb.lib().adx_tconv_pack(C.byref(d), w.data_ptr(), packed.data_ptr(), Lb.stream_ptr(dev)))
    y = torch.empty_like(x)
    io = Lb.TConvIO()
    io.x0, io.x0_sb, io.x0_sc, io.x0_sl = x.data_ptr(), c * L, L, 1
    io.packed_w, io.bias, io.gamma, io.beta = packed.data_ptr(), b.data_ptr(), g.data_ptr(), be.data_ptr()
    io.y, io.y_sb, io.y_sc, io.y_sl, io.batch = y.data_ptr(), c * L, L, 1, rows
    launch = lambda: Lb.check(Lb.lib().adx_tconv_forward(C.byref(d), C.byref(io), Lb.stream_ptr(dev)))  # noqa: E731
    ms = time_events(launch, reps, warm=3)
    fl = 2.0 * rows * L * c * c * 5
    byts = 4.0 * (2 * x.numel() + w.numel())
    del fn
    return {"kernel": "tconv_hs_kernel<2,8,4> (Conv1d 512->512 k5 + GroupNorm + Mish, 128x4 positions; fp32-grade result "
                      "from fp16 hi/lo split operands, 3 MFMA products per multiply-add)",
            "bound": "mfma", "achieved": round(fl / ms / 1e9, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(fl / ms / 1e9 / PEAK_F16_TFLOPS, 4),
            "achieved_issued": round(3 * fl / ms / 1e9, 2), "frac_issued": round(3 * fl / ms / 1e9 / PEAK_F16_TFLOPS, 4),
            "fp32_mfma_peak_tflops": PEAK_F32_TFLOPS,
            "traffic": pmc_traffic("tconv_hs_kernel<2"), "avg_launch_ms": round(ms, 4),
            "avg_launch_ms_rocprof": rocprof_avg_ms("tconv_hs_kernel<2"),
            "algorithmic_gflop_per_launch": round(fl / 1e9, 3), "algorithmic_mb_per_launch": round(byts / 1e6, 2),
            "hbm_gbs_at_algorithmic_bytes": round(byts / ms / 1e6, 1),
            "hbm_frac": round(byts / ms / 1e6 / PEAK_HBM_GBS, 4),
            "note": "neither roofline binds at 512 rows: one workgroup streams a 655 KB weight slab through one CU's "
                    "L2->L1 port (~64 B/clk) behind ~4 us of launch, staging and epilogue latency (DESIGN.md section 3)"}


def deployed_leg(dev):
    """The configuration the reference actually drives with (e2e_driving/diffusion_agent.py:179-232, interact.py:115-168):
    ONE scene per tick, horizon 16, classifier-free guidance (UNet batch 2), 50 DDIM steps, the camera frame's
    perception pass included, replayed as one HIP graph per tick (sampling.GraphedSampler)."""
    import contextlib
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    from autonomous_driving_with_diffusion_model_amd.modeling import build_model
    from autonomous_driving_with_diffusion_model_amd.sampling import GraphedSampler, generate_traj
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
    cfg = create_cfg()
    cfg.MODEL.HORIZON = 16
    cfg.TRAIN.USE_COND = cfg.GUIDANCE.USE_COND = "FREE_GUIDANCE"
    cfg.GUIDANCE.FREE_SCALE, cfg.EVAL.SAMPLE_STEPS = FREE_SCALE, N_INFER
    with contextlib.redirect_stdout(sys.stderr):
        model = build_model(cfg)
    P.load_procedural(model, 0)
    model = model.to(dev).eval()
    sch = S.GuidanceDDIMScheduler(cfg=cfg, thresholding=True, **SCHED_KW)
    d = {k: v.to(dev) for k, v in P.synthetic_batch(1, 16, image_hw=IMG, seed=3).items()}
    out = {}
    with torch.no_grad():
        for name, fn in (("eager", lambda: generate_traj(model, sch, cfg, d["imgs"].clone(), d["target"], d["init_trajs"])),
                         ("graph", None)):
            if fn is None:
                gs = GraphedSampler(model, sch, cfg)
                fn = lambda: gs(d["imgs"], d["target"], d["init_trajs"])  # noqa: E731
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            out[name] = (time.perf_counter() - t0) / 5
    # configs/guidance/classifier_guidance.yaml as deployed: one scene, 2 DDIM steps, TargetGuidance gradient through
    # state_pred at scale 15 (one fused launch per step), perception pass inside the tick
    cls = {}
    try:
        ccfg = create_cfg()
        ccfg.MODEL.HORIZON = 16
        ccfg.TRAIN.USE_COND = ccfg.GUIDANCE.USE_COND = "CLASSIFIER_GUIDANCE"
        ccfg.GUIDANCE.LOSS_LIST = [["TargetGuidance", []]]
        ccfg.GUIDANCE.CLASSIFIER_SCALE, ccfg.EVAL.SAMPLE_STEPS = 15.0, 2
        with contextlib.redirect_stdout(sys.stderr):
            cmodel = build_model(ccfg)
        P.load_procedural(cmodel, 0)
        cmodel = cmodel.to(dev).eval()
        csch = S.GuidanceDDIMScheduler(cfg=ccfg, thresholding=True, **SCHED_KW)
        cgs = GraphedSampler(cmodel, csch, ccfg)
        with torch.no_grad():
            for _ in range(3):
                cgs(d["imgs"], d["target"], d["init_trajs"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                cgs(d["imgs"], d["target"], d["init_trajs"])
            torch.cuda.synchronize()
            cls = {"classifier_guidance_tick_ms_graph": round(1e3 * (time.perf_counter() - t0) / 20, 3),
                   "classifier_guidance_note": "configs/guidance/classifier_guidance.yaml: B = 1, 2 DDIM steps, scale 15, "
                                               "perception pass inside the tick"}
        del cgs, cmodel
    except Exception as e:  # the FREE numbers above stand on their own
        cls = {"classifier_guidance_error": f"{type(e).__name__}: {e}"[:200]}
    wbytes = 64.5e6            # UNet weights read once per denoising step (B = 1, H = 16)
    per_step = out["graph"] / N_INFER
    return {"workload": "one scene per tick: B = 1 (UNet batch 2, classifier-free guidance 7.5), horizon 16, 50 DDIM steps, "
                        "image 3x256x900, perception pass inside the tick",
            "tick_ms_graph": round(1e3 * out["graph"], 3), "tick_ms_eager": round(1e3 * out["eager"], 3),
            "denoising_steps_per_sec": round(N_INFER / out["graph"], 1), "us_per_step": round(1e6 * per_step, 1),
            "hbm_frac_weights_once_per_step": round(wbytes / per_step / 1e9 / PEAK_HBM_GBS, 4), **cls,
            "note": "bound by the chain of ~24 dependent launches per step (3 chained levels, the pipeline launch of the deepest level's "
                    "seven convs, ~20 per-layer launches; a wave retires an instruction per ~8-12 clocks at this occupancy, a hand-off "
                    "between workgroups costs ~4 us), not by HBM: DESIGN.md sections 3 and 8"}


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(steps=5):
    """The oracle (CPU restatement of the reference, kind 'port') on the host cores of this box,
    same workload: one reference-faithful denoising step = ResNet-34 on 64 images + UNet on the
    2x64 CFG batch + DDIM step.  Bounded sample: 1 warm-up + `steps` timed steps; `value` is 1 / the MEDIAN step."""
    from autonomous_driving_with_diffusion_model_amd.modeling.spec import unet_entries
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
    from oracle import unet as U
    from oracle.schedulers import GuidanceDDIM
    sd = P.procedural_state_dict(((e.key, e.shape) for e in unet_entries("FREE_GUIDANCE")), 0)
    d = P.synthetic_batch(B, H, image_hw=IMG, seed=0)
    sch = GuidanceDDIM(thresholding=True, **SCHED_KW)
    sch.set_timesteps(N_INFER)
    trajs = d["init_trajs"].clone()
    trajs[:, 0, :3] = 0
    cond = torch.cat([d["target"], torch.zeros_like(d["target"])], 0)
    cores = torch.get_num_threads()
    times = []
    with torch.no_grad():
        for i, t in enumerate(sch.timesteps[: steps + 1]):
            t0 = time.perf_counter()
            out = U.unet_forward(sd, torch.cat([trajs, trajs], 0), d["imgs"], t.reshape(-1), cond,
                                 use_cond="FREE_GUIDANCE")
            c, u = out.chunk(2, 0)
            trajs = sch.step(u + FREE_SCALE * (c - u), t, trajs).prev_sample
            trajs[:, 0, :3] = 0
            times.append(time.perf_counter() - t0)
        # UNet-only (hoisted) CPU time for the second comparison
        feat = torch.randn(B, 64)
        t0 = time.perf_counter()
        for _ in range(3):
            U.unet_forward(sd, torch.cat([trajs, trajs], 0), None, sch.timesteps[0].reshape(-1), cond,
                           use_cond="FREE_GUIDANCE", img_feature=feat)
        unet_s = (time.perf_counter() - t0) / 3
    timed = times[1:]
    per = sorted(timed)[len(timed) // 2]          # median: single steps wander by +-10 % on a shared host
    return {"value": round(1.0 / per, 4), "unit": "denoising-steps/sec", "cores": cores, "cpu": cpu_model_name(),
            "kind": "port", "per_step_s": [round(x, 3) for x in timed],
            "sample": f"{len(timed)} reference-faithful steps (ResNet-34 on 64x3x256x900 + UNet 2x64xH32 + DDIM step) "
                      f"after 1 warm-up, torch-CPU fp32, {cores} threads; value = 1 / median step",
            "s_per_step": round(per, 3), "unet_only_steps_per_sec": round(1.0 / unet_s, 3)}


def train_leg(dev, world, use_cond="NO_GUIDANCE", steps=20, warm=3, trace_overlap=False, force_collectives=False):
    """One training leg at B = 64 per GPU, H = 32, 3x256x900: add_noise -> train-mode forward (batch-statistics BatchNorm)
    -> MSE -> backward -> fused nan_to_num + AdamW + EMA (train.py:221-261); one optimizer step = one denoising step.

    use_cond "NO_GUIDANCE"   BASELINE configs[1] (configs/default.yaml).
             "FREE_GUIDANCE" BASELINE configs[4]'s per-GPU workload (configs/guidance/free_guidance.yaml): the model with
                             cond_mlp, and train.py:236-242's draw -- with probability 1 - USE_FREE_COND_PROB = 0.3 a batch
                             trains with cond=None.  The draw is per batch AND per process (`random.random()`), so under
                             data parallelism the ranks of one step take different branches; here it comes from a
                             `random.Random(1000 + rank)` so that a run is reproducible."""
    import contextlib
    import random
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    from autonomous_driving_with_diffusion_model_amd.modeling import build_model
    from autonomous_driving_with_diffusion_model_amd.optim import FusedAdamWEMA
    from autonomous_driving_with_diffusion_model_amd.parallel import DataParallel
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
    torch.cuda.empty_cache()      # the sampling legs' workspaces go back to the driver before the 27 GB of tapes arrive
    rank = int(os.environ.get("RANK", "0"))
    free = use_cond == "FREE_GUIDANCE"
    cfg = create_cfg()
    cfg.MODEL.HORIZON = H
    cfg.TRAIN.USE_COND = cfg.GUIDANCE.USE_COND = use_cond
    with contextlib.redirect_stdout(sys.stderr):
        model = build_model(cfg)
    P.load_procedural(model, 0)
    model = model.to(dev).train()
    opt = FusedAdamWEMA(model.parameters(), lr=1e-4, warmup_steps=1000, lr_ticks_per_step=world)
    # rank 0's weights and buffers to everyone, per-forward BatchNorm-buffer broadcast, bucketed all-reduce from hooks
    # during backward (the reference's DistributedDataParallel semantics, train.py:176-178)
    # --force-collectives (N = 1): a one-rank RCCL group and GradientAverager(force=True) -- the collectives are identities, but
    # the RCCL kernels run on the side stream behind the per-group events exactly as on a rank of an 8-GPU job
    dp = (DataParallel(model, optimizer=opt, force=force_collectives or None)
          if (world > 1 or force_collectives) else None)       # buckets carry the sum, 1 / world folded into opt.step
    fwd = dp if dp is not None else model
    sch = S.DDPMScheduler(**SCHED_KW)
    d = {k: v.to(dev) for k, v in P.synthetic_batch(B, H, image_hw=IMG, seed=7 + rank).items()}
    draw = random.Random(1000 + rank)
    branches = {"cond": 0, "cond_none": 0}

    def step(count=True):
        cond = None
        if free:
            drop = draw.random() > cfg.TRAIN.USE_FREE_COND_PROB           # train.py:237-241
            cond = None if drop else d["target"]
            if count:
                branches["cond_none" if drop else "cond"] += 1
        noisy = sch.add_noise(d["trajs"], d["noise"], d["t"], zero_first=True)
        pred = fwd(noisy, d["imgs"], d["t"], cond=cond)
        loss = torch.nn.functional.mse_loss(pred, d["trajs"])
        loss.backward()
        end = None
        if trace_overlap and dp is not None and dp.averager.trace:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
        if dp is not None:
            dp.synchronize()
        opt.step()
        opt.zero_grad()
        return loss, cond is None and free, end

    # the first step's loss is a known number: the CPU oracle's train-mode forward on the same inputs (rank 0's batch, seed 7;
    # tests/golden/make_bench_loss.py wrote it, tests/test_gpu_fullsize.py recomputes it on the box) -- a training leg that
    # computes something else must not report a time.  Every rank runs on whatever the verdict (no rank leaves the others
    # inside a collective); rank 0 reports `parity_failed` instead of a value.
    loss0, dropped0, _ = step(count=False)
    first = float(loss0.detach())
    expected = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "bench_train_loss.json")) as f:
            expected = json.load(f)[("free_drop_loss_fp32" if dropped0 else "free_loss_fp32") if free else "loss_fp32"]
    except (OSError, KeyError, ValueError):
        pass
    parity_ok = expected is None or abs(first - expected) <= 2e-5 * max(1.0, abs(expected))
    for _ in range(warm - 1):
        step(count=False)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _, _ = step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt, per_rank = max_over_ranks(time.perf_counter() - t0, world, dev)
    overlap = None
    if trace_overlap and dp is not None:
        # one more step with device events around every bucket's collective (outside the timed region): when a bucket's last
        # gradient was ready and when its reduction had finished, relative to the end of backward on the device
        dp.averager.trace = True
        _, _, end = step(count=False)
        torch.cuda.synchronize()
        overlap = dp.averager.overlap_report(end)
        dp.averager.trace = False
    flops = 3 * (64 * 34.02e9) + 3 * 10.03e9      # ~3x forward
    name = ("configs/guidance/free_guidance.yaml train step, FREE_GUIDANCE (cond=None with probability 0.3 per batch and rank)"
            if free else "configs/default.yaml train step, NO_GUIDANCE")
    res = {"workload": name + ", batch 64 per GPU, horizon 32, image 3x256x900, fwd + bwd + fused AdamW/EMA"
                       + (" + RCCL gradient all-reduce" if world > 1 else "")
                       + (" + one-rank RCCL all-reduce behind the per-group events (--force-collectives)" if force_collectives and world == 1 else ""),
           "value": round(world * steps / dt, 3), "unit": "train-steps/sec", "ms_per_step": round(1e3 * dt / steps, 2),
           "steps": steps, "approx_tflops_per_gpu": round(flops * steps / dt / 1e12, 1), "first_step_loss": round(first, 7),
           "first_step_loss_oracle": expected, "final_loss": round(float(loss.detach()), 5)}
    if free:
        res["first_step_branch"] = "cond_none" if dropped0 else "cond"
        res["timed_steps_by_branch_rank0"] = branches
    if not parity_ok:
        res["parity_failed"] = {"first_step_loss": first, "expected": expected, "tolerance": "2e-5 relative"}
        res["value"] = res["ms_per_step"] = None
    if dp is not None:
        res["per_rank_ms_per_step"] = [round(1e3 * t / steps, 2) for t in per_rank]
        res["gradient_buckets"] = {"primitive": dp.averager.primitive, "n": len(dp.averager.buckets),
                                   "copied_in_last_step": dp.averager.copied_in}
        if overlap is not None:
            res["gradient_buckets"]["overlap_rank0"] = overlap
            res["gradient_buckets"]["overlap_note"] = ("per bucket, ms relative to the END of backward on the device (negative = "
                                                       "before): ready = its last gradient written, done = its collective finished; "
                                                       "done_ms < 0 means the reduction was hidden behind backward")
    del model, opt, dp, fwd
    torch.cuda.empty_cache()
    return res


def visible_gpu_count() -> int:
    """GPUs this process would see, counted WITHOUT touching the HIP runtime (a torch.cuda.device_count() that falls back to
    hipGetDeviceCount initialises it, and a process that has done so must not start other GPU programs on this pool): the
    visibility variables if set, else the KFD topology (nodes with SIMDs are GPUs; CPU nodes report simd_count 0)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                for line in f:
                    k, _, val = line.partition(" ")
                    if k == "simd_count" and int(val) > 0:
                        n += 1
        except (OSError, ValueError):
            pass
    return n


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher around it: start N copies of this script, one per GPU, with the
    rendezvous variables torchrun would set.  The parent never touches a GPU (no HIP call before or after the spawn);
    the first child that fails takes the others down and its exit code becomes the parent's."""
    import socket
    import subprocess
    have = visible_gpu_count()                # no torch.cuda / HIP call in the parent: it forks the ranks
    # (0 = could not tell -- no visibility variable, no readable KFD topology: start the ranks and let them find out)
    if 0 < have < n and "--launch-check" not in sys.argv and os.environ.get("ADX_BENCH_SAME_DEVICE") != "1":
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, alive = 0, set(range(n))
    while alive:
        for i in sorted(alive):
            r = procs[i].poll()
            if r is None:
                continue
            alive.discard(i)
            if r != 0 and rc == 0:
                rc = r
                print(f"bench.py: rank {i} exited with status {r}; stopping the other ranks", file=sys.stderr)
                for j in alive:
                    procs[j].terminate()
        time.sleep(0.1)
    return rc


def launch_check(world: int, rank: int) -> None:
    """`--launch-check`: the rendezvous plumbing alone (gloo, no GPU): every rank contributes its rank + 1 to an
    all-reduce; rank 0 prints what the process group itself reports.  Used by the CPU test of the launcher."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if os.environ.get("ADX_LAUNCH_CHECK_FAIL_RANK") == str(rank):
        sys.exit(7)
    if dist.get_rank() == 0:
        print(json.dumps({"launch_check": True, "n_gpus": dist.get_world_size(), "sum": t.item(),
                          "env_world": world}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--train-steps", type=int, default=20, help="timed optimizer steps per training leg")
    ap.add_argument("--no-deployed", action="store_true")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1: run the FREE_GUIDANCE training leg under parallel.DataParallel(force=True) on a one-rank RCCL group "
                         "(train_free.gradient_buckets.overlap_rank0 then reports the bucket reductions against backward)")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--short-sampling", action="store_true", help=argparse.SUPPRESS)   # tests: skip the hoisted / graph legs
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))       # no launcher around us: become one (the parent stays off the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    if args.launch_check:
        return launch_check(world, rank)
    # the JSON line must be the ONLY thing on stdout: native libraries (gloo, RCCL with NCCL_DEBUG, the model's channel table)
    # write to file descriptor 1 behind Python's back, so fd 1 is pointed at stderr for the run and the line goes to the saved fd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # Test hook for boxes with ONE GPU (RCCL refuses two ranks on one device): ADX_BENCH_SAME_DEVICE=1 puts every rank on
    # cuda:0 and uses gloo for the collectives, so the multi-rank flow (launcher, DataParallel, bucketed all-reduce from
    # autograd hooks, max-over-ranks timing) runs end to end on the real kernels.  The numbers of such a run mean nothing.
    same_device = os.environ.get("ADX_BENCH_SAME_DEVICE") == "1"
    if same_device:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        world = dist.get_world_size()           # what RCCL itself reports is what goes into n_gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    from autonomous_driving_with_diffusion_model_amd.modeling import build_model
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P

    cfg = create_cfg()
    cfg.MODEL.HORIZON = H
    cfg.TRAIN.USE_COND = cfg.GUIDANCE.USE_COND = "FREE_GUIDANCE"
    cfg.GUIDANCE.FREE_SCALE, cfg.EVAL.SAMPLE_STEPS = FREE_SCALE, N_INFER
    os.environ.setdefault("LOCAL_RANK", str(local))
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):   # the model announces its channel table on stdout like the reference
        model = build_model(cfg)
    P.load_procedural(model, 0)
    model = model.to(dev).eval()
    sch = S.GuidanceDDIMScheduler(cfg=cfg, thresholding=True, **SCHED_KW)
    sch.set_timesteps(N_INFER, device=dev)
    d = {k: v.to(dev) for k, v in P.synthetic_batch(B, H, image_hw=IMG, seed=rank).items()}
    cond = torch.cat([d["target"], torch.zeros_like(d["target"])], 0)
    ts = list(sch.timesteps)

    def run(n_steps, start=0):
        trajs = d["init_trajs"].clone()
        trajs[:, 0, :3] = 0
        # the tick's image tensor is written by nobody while the loop runs (the reference's loops pass the same untouched tensor to
        # every step, interact.py:133-155): declared, so that the per-step encoder pass may run ahead (modeling/perception.py)
        with torch.no_grad(), model.perception.frozen_image(d["imgs"]):
            # product default (perception memo on): like sampling.generate_traj, everything the UNet derives from (t, target,
            # image feature) alone is computed once per loop; reference-faithful mode recomputes it every step
            tc = model.time_conditioning(d["imgs"], sch.timesteps.tensor, cond=cond, rows=2 * B) if model.cache_perception else None
            for i in range(n_steps):
                k = (start + i) % len(ts)
                t = ts[k]
                x2 = torch.cat([trajs, trajs], 0)
                out = model(x2, d["imgs"], t.reshape(-1), cond=cond) if tc is None else model(x2, None, None, time_cond=(tc, k))
                trajs = sch.step(out, t, trajs, cfg_scale=FREE_SCALE, zero_first=True).prev_sample
        return trajs

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, warm):
        run(warm)
        model._feat_cache = None   # hoisted mode: the timed region pays for its own perception pass
        barrier()
        t0 = time.perf_counter()
        run(n_steps)
        barrier()
        return max_over_ranks(time.perf_counter() - t0, world, dev)

    # the perception pass alone (HIP events on the caller's stream around 8 back-to-back passes on the tick's image tensor), as the
    # product runs it: two sub-batches on streams of their own (csrc/conv2d.hip), on the pass stream (modeling/perception.py)
    pass_ms = None
    if rank == 0 and not args.no_roofline:
        with torch.no_grad():
            model.perception(d["imgs"])
            pass_ms = round(time_events(lambda: model.perception(d["imgs"]), 8, warm=1), 4)
    model.cache_perception = False          # reference-faithful: perception re-run every step
    dt, per_rank = timed(args.steps, args.warmup)
    model.cache_perception = True           # product default: one perception pass per scene
    steps_h = args.steps if args.short_sampling else max(args.steps, N_INFER)
    dt_h, per_rank_h = timed(steps_h, args.warmup)

    # the same hoisted loop as ONE HIP graph per 50-step tick (sampling.GraphedSampler: perception pass, 50 x (UNet +
    # scheduler step), clamp and scaling captured once, replayed bit-identically): no host work between launches
    from autonomous_driving_with_diffusion_model_amd.sampling import GraphedSampler
    gs = GraphedSampler(model, sch, cfg)
    ticks = max(10, (max(args.steps, N_INFER) + N_INFER - 1) // N_INFER)     # >= 10 replays of ~25 ms: one sample is noise
    if args.short_sampling:
        ticks = 1
    with torch.no_grad():
        for _ in range(2):                   # capture + one replay
            gs(d["imgs"], d["target"], d["init_trajs"])
        barrier()
        t0 = time.perf_counter()
        for _ in range(ticks):
            gs(d["imgs"], d["target"], d["init_trajs"])
        barrier()
        dt_g, per_rank_g = max_over_ranks(time.perf_counter() - t0, world, dev)
    steps_g = ticks * N_INFER
    del gs

    del model
    torch.cuda.empty_cache()
    # training legs: configs[1] (NO_GUIDANCE) on one GPU; configs[4]'s per-GPU workload (FREE_GUIDANCE incl. the cond=None
    # draw) at every world size -- with N > 1 ranks that IS configs[4] (8 x 64 = 512 global), so it is the only one run there
    train = train_free = None
    if not args.no_train:
        try:
            if world == 1:
                train = train_leg(dev, world, "NO_GUIDANCE", steps=args.train_steps)
            force = args.force_collectives and world == 1
            if force:
                import socket
                import torch.distributed as dist
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                        device_id=torch.device("cuda", local))
            train_free = train_leg(dev, world, "FREE_GUIDANCE", steps=args.train_steps, trace_overlap=world > 1 or force,
                                   force_collectives=force)
        except Exception as e:   # the headline sampling metric must survive a failure of the secondary leg
            if world > 1:
                raise            # ... but not at the price of a hang: the other ranks are inside this leg's collectives
            err = {"error": f"{type(e).__name__}: {e}"[:300]}
            train, train_free = (train or err), (train_free or err)

    if rank == 0:
        res = {
            "metric": "denoising-steps/sec", "value": round(world * args.steps / dt, 3), "unit": "denoising-steps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_note": "fp32 inputs, outputs and accumulation; the 2-D convolutions multiply fp16 hi/lo split operands (3 MFMAs per product, error below fp32 accumulation's own: DESIGN.md section 3)",
            "parity_note": "north_star: fp32 trajectory outputs within 1e-4.  Measured at this workload (tests/test_gpu_fullsize.py, B = 64, "
                           "50 steps) against the CPU oracle in fp32: 1.6e-5 on the normalised trajectory = 2.1e-5 on channels 2-6 as returned "
                           "and 3.8e-4 on the RETURNED x, y, which the callers multiply by magic_num = 23.315 after the clamp (interact.py:167); "
                           "against the same oracle run in fp64 (the exact result): 2.4e-4 / 9.5e-6, where the reference's own fp32 arithmetic "
                           "(the fp32 oracle) is 3.3e-4 / 1.7e-5 from it -- the HIP path is closer to the exact trajectory than the fp32 "
                           "reference is, and 1e-4 on the returned x, y is below the distance between any two fp32 evaluations of this "
                           "50-step recurrence.  The tests bound channels 0-1 by 23.315e-4 and the rest by 1e-4 against the fp32 oracle "
                           "(tests/helpers.py:close_traj) and by 2x the fp32 oracle's own error against fp64",
            "data": "synthetic",
            "runtime_env": {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG"),
                            "note": "kernel arguments in device memory (set by the package at import unless the user set it)"},
            "config": {"workload": "configs/guidance/free_guidance.yaml: 50-step DDIM sampling, classifier-free "
                                   "guidance scale 7.5, 64 scenes per GPU (UNet batch 128), horizon 32, image 3x256x900, "
                                   "reference-faithful (ResNet-34 perception re-run every step)",
                       "batch_per_gpu": B, "horizon": H, "ddim_steps": N_INFER, "weights": "procedural (seed 0)"},
            "hoisted": {"value": round(world * steps_h / dt_h, 3), "ms_per_step": round(1e3 * dt_h / steps_h, 4),
                        "steps": steps_h, "note": "perception memoised per image tensor: one ResNet-34 pass per "
                        "scene inside the timed region (and one pass of the time / condition embedding for all timesteps), "
                        "then UNet + scheduler per step, launched eagerly",
                        "trajectories_per_sec": round(world * B / (dt_h * N_INFER / steps_h), 2)},
            "hoisted_graph": {"value": round(world * steps_g / dt_g, 3), "ms_per_step": round(1e3 * dt_g / steps_g, 4),
                              "steps": steps_g, "note": "sampling.GraphedSampler: each 50-step tick (perception pass "
                              "included) replayed as one HIP graph; bit-identical to the eager loop",
                              "trajectories_per_sec": round(world * B * ticks / dt_g, 2)},
        }
        if world > 1:
            res["per_rank_ms_per_step"] = [round(1e3 * t / args.steps, 3) for t in per_rank]
            res["hoisted"]["per_rank_ms_per_step"] = [round(1e3 * t / steps_h, 4) for t in per_rank_h]
            res["hoisted_graph"]["per_rank_ms_per_step"] = [round(1e3 * t / steps_g, 4) for t in per_rank_g]
        if train is not None:
            res["train"] = train
        if train_free is not None:
            res["train_free"] = train_free
        if not args.no_roofline:
            res["roofline"] = conv2d_roofline(dev)
            if pass_ms is not None:
                conv_gf = sum(conv_flops(B, *c) for c in resnet_conv_table(*IMG)) / 1e9
                s1_ms = res["roofline"]["avg_launch_ms"] * res["roofline"]["launches_per_step"]
                res["roofline"]["perception_pass"] = {
                    "ms": pass_ms, "algorithmic_tflops_all_convs": round(conv_gf / pass_ms, 1),
                    "frac_all_convs": round(conv_gf / pass_ms / PEAK_F16_TFLOPS, 4),
                    "sum_of_stride1_launches_in_order_ms": round(s1_ms, 3),
                    "note": "the whole ResNet-34 pass at B = 64 as the timed region runs it -- two sub-batches of 32 on streams of their own, so "
                            "that one's launches fill the CUs the other's last round of workgroups leaves idle (ADX_RESNET_STREAMS=1: one "
                            "chain) -- against the sum of its 29 stride-1 launches timed one after the other at B = 64 (`avg_launch_ms` x 29; "
                            "the pass also holds the stem + pool, three stride-2 launches and the pooling + fc).  `frac` / `avg_launch_ms` stay "
                            "per-launch figures on ONE stream: a launch that shares the chip has no duration of its own"}
            res["roofline_tconv"] = tconv_roofline(None, dev)
        if world == 1 and not args.no_deployed:
            try:
                res["deployed_b1_h16"] = deployed_leg(dev)
            except Exception as e:
                res["deployed_b1_h16"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
            res["speedup_vs_cpu_baseline"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
