"""GPU parity, op level: every HIP kernel through the C ABI against the CPU oracle on identical
seeded inputs.  Tolerances (fp32, different summation order than torch's CPU kernels):
conv/GN/Mish ops 2e-5 abs on O(1) activations.  Scheduler steps: integer tables bit-exact; fp32
outputs within 1 ulp of the O(1) intermediates (2.4e-7) of the golden vectors (torch's own CPU elementwise results move by 1 ulp
between the Xeon that generated the fixtures and the EPYC host of the GPU box), and bit-identical
to the same ops issued one by one through torch on the GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import schedulers as SCH
from oracle import unet as U
from helpers import SCHED_KW, close, oracle_sd, uni

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from autonomous_driving_with_diffusion_model_amd import ops
    return ops


# (cin0, cin1, cout, L, batch): every Conv1dBlock shape of the H=32 and H=16 UNets + ragged batches
BLOCK_SHAPES = [(7, 0, 64, 32, 3), (64, 0, 64, 32, 5), (64, 0, 128, 16, 4), (128, 0, 128, 16, 2),
                (128, 0, 256, 8, 3), (256, 0, 256, 8, 5), (256, 0, 512, 4, 6), (512, 0, 512, 4, 7),
                (512, 512, 256, 4, 5), (256, 256, 128, 8, 3), (128, 128, 64, 16, 2),
                (512, 0, 512, 2, 9), (7, 0, 64, 16, 1), (64, 0, 64, 64, 2), (512, 0, 512, 1, 17)]


@pytest.mark.parametrize("c0,c1,cout,L,B", BLOCK_SHAPES)
def test_conv1d_block(c0, c1, cout, L, B):
    name = f"blk.{c0}.{c1}.{cout}.{L}"
    x0 = uni(name + ".x0", (B, c0, L))
    x1 = uni(name + ".x1", (B, c1, L)) if c1 else None
    cin = c0 + c1
    w = uni(name + ".w", (cout, cin, 5), lo=-(3.0 / (5 * cin)) ** 0.5, hi=(3.0 / (5 * cin)) ** 0.5)
    b, g, be = uni(name + ".b", (cout,), lo=-.1, hi=.1), uni(name + ".g", (cout,), lo=.9, hi=1.1), uni(name + ".be", (cout,), lo=-.1, hi=.1)
    tb = uni(name + ".tb", (B, cout + 5))[:, 3:3 + cout]           # strided view: row stride != cout
    res = uni(name + ".res", (B, cout, L))
    xin = x0 if x1 is None else torch.cat([x0, x1], 1)
    ref = F.mish(F.group_norm(F.conv1d(xin, w, b, padding=2), 8, g, be, 1e-5))
    y = _ops().tconv(x0.to(DEV), w.to(DEV), b.to(DEV), x1=None if x1 is None else x1.to(DEV), pad=2,
                     gn_weight=g.to(DEV), gn_bias=be.to(DEV), groups=8)
    close(y.cpu(), ref, 2e-5)
    tbd = tb.to(DEV)
    y = _ops().tconv(x0.to(DEV), w.to(DEV), b.to(DEV), x1=None if x1 is None else x1.to(DEV), pad=2,
                     gn_weight=g.to(DEV), gn_bias=be.to(DEV), groups=8, tbias=tbd, res=res.to(DEV))
    close(y.cpu(), ref + tb[:, :, None] + res, 2e-5)


@pytest.mark.parametrize("c0,c1,cout,L,B", [(512, 0, 512, 2, 1), (512, 0, 512, 2, 2), (512, 0, 512, 4, 1), (512, 512, 256, 2, 2),
                                            (256, 256, 128, 4, 1), (256, 0, 512, 2, 2), (512, 0, 512, 4, 3), (256, 0, 256, 4, 2),
                                            (512, 0, 512, 4, 128), (512, 512, 256, 4, 128), (256, 0, 512, 4, 97)])
def test_conv1d_block_split_reduction_tiny_batch(c0, c1, cout, L, B):
    """Deployed batch sizes (B = 1: 1 or 2 UNet rows, H = 16): with a scratch buffer the K-split kernel spreads the
    input channels over more workgroups and a reduce launch runs the epilogue (tconv_hs.hip, HsArgs::ksplit).  Same
    oracle, same bar as the unsplit launch; the two launches' results may differ by summation order only."""
    name = f"blks.{c0}.{c1}.{cout}.{L}.{B}"
    x0 = uni(name + ".x0", (B, c0, L))
    x1 = uni(name + ".x1", (B, c1, L)) if c1 else None
    cin = c0 + c1
    w = uni(name + ".w", (cout, cin, 5), lo=-(3.0 / (5 * cin)) ** 0.5, hi=(3.0 / (5 * cin)) ** 0.5)
    b, g, be = uni(name + ".b", (cout,), lo=-.1, hi=.1), uni(name + ".g", (cout,), lo=.9, hi=1.1), uni(name + ".be", (cout,), lo=-.1, hi=.1)
    tb, res = uni(name + ".tb", (B, cout)), uni(name + ".res", (B, cout, L))
    xin = x0 if x1 is None else torch.cat([x0, x1], 1)
    ref = F.mish(F.group_norm(F.conv1d(xin, w, b, padding=2), 8, g, be, 1e-5)) + tb[:, :, None] + res
    scratch = torch.full((2 << 20,), float("nan"), device=DEV)       # stale contents must never reach the output
    kw = dict(x1=None if x1 is None else x1.to(DEV), pad=2, gn_weight=g.to(DEV), gn_bias=be.to(DEV), groups=8,
              tbias=tb.to(DEV), res=res.to(DEV))
    y_split = _ops().tconv(x0.to(DEV), w.to(DEV), b.to(DEV), scratch=scratch, **kw)
    y_plain = _ops().tconv(x0.to(DEV), w.to(DEV), b.to(DEV), **kw)
    close(y_split.cpu(), ref, 2e-5)
    close(y_split.cpu(), y_plain.cpu(), 4e-6)
    small = torch.full((1024,), float("nan"), device=DEV)            # too small for any split: falls back to one launch
    assert torch.equal(_ops().tconv(x0.to(DEV), w.to(DEV), b.to(DEV), scratch=small, **kw), y_plain)
    # with ticket words the last workgroup to publish its partial tile adds them up in the same launch: same sums in the
    # same order as the reduce launch, and the words are zero again afterwards (three calls in a row on the same words)
    tickets = torch.zeros(256, dtype=torch.int32, device=DEV)
    for _ in range(3):
        scratch.fill_(float("nan"))
        y_ticket = _ops().tconv(x0.to(DEV), w.to(DEV), b.to(DEV), scratch=scratch, tickets=tickets, **kw)
        if B <= 3:
            assert torch.equal(y_ticket, y_split)
        else:        # UNet batch ~128: grids of 33..128 workgroups split in two, and only through the ticket path
            close(y_ticket.cpu(), ref, 2e-5)
            close(y_ticket.cpu(), y_plain.cpu(), 4e-6)
        assert int(tickets.abs().sum()) == 0


@pytest.mark.parametrize("c0,c1,cout,L,B", [(7, 0, 48, 32, 3), (48, 0, 48, 32, 2), (96, 96, 48, 16, 3), (192, 0, 192, 8, 2),
                                            (384, 0, 384, 4, 1), (512, 512, 256, 1, 3), (256, 0, 256, 1, 2), (24, 0, 24, 16, 2)])
def test_conv1d_block_general_shapes(c0, c1, cout, L, B):
    """Shapes neither MFMA kernel tiles -- GroupNorm(8, C) with groups of 6 / 12 / 24 / 48 / 3 channels (MODEL.DIM = 48, 96, 24:
    modeling/helpers.py:105-107 takes any C divisible by 8), groups of 32 elements (one position: the bottom of the up path
    at horizon 8) -- run on csrc/tconv_generic.hip: same operator, same bar, concat input / time bias / residual included."""
    name = f"gen.{c0}.{c1}.{cout}.{L}"
    x0 = uni(name + ".x0", (B, c0, L))
    x1 = uni(name + ".x1", (B, c1, L)) if c1 else None
    cin = c0 + c1
    w = uni(name + ".w", (cout, cin, 5), lo=-(3.0 / (5 * cin)) ** 0.5, hi=(3.0 / (5 * cin)) ** 0.5)
    b, g, be = uni(name + ".b", (cout,), lo=-.1, hi=.1), uni(name + ".g", (cout,), lo=.9, hi=1.1), uni(name + ".be", (cout,), lo=-.1, hi=.1)
    tb = uni(name + ".tb", (B, cout + 5))[:, 3:3 + cout]
    res = uni(name + ".res", (B, cout, L))
    xin = x0 if x1 is None else torch.cat([x0, x1], 1)
    ref = F.mish(F.group_norm(F.conv1d(xin, w, b, padding=2), 8, g, be, 1e-5)) + tb[:, :, None] + res
    y = _ops().tconv(x0.to(DEV), w.to(DEV), b.to(DEV), x1=None if x1 is None else x1.to(DEV), pad=2,
                     gn_weight=g.to(DEV), gn_bias=be.to(DEV), groups=8, tbias=tb.to(DEV), res=res.to(DEV))
    close(y.cpu(), ref, 2e-5)


@pytest.mark.parametrize("c,L,B", [(64, 32, 3), (128, 16, 5), (256, 8, 2), (64, 16, 4), (256, 4, 9)])
def test_downsample(c, L, B):
    x, w, b = uni(f"dn.x.{c}", (B, c, L)), uni(f"dn.w.{c}", (c, c, 3), lo=-.1, hi=.1), uni(f"dn.b.{c}", (c,))
    y = _ops().tconv(x.to(DEV), w.to(DEV), b.to(DEV), stride=2, pad=1)
    close(y.cpu(), F.conv1d(x, w, b, stride=2, padding=1), 2e-5)


@pytest.mark.parametrize("c,L,B", [(256, 4, 3), (128, 8, 5), (64, 16, 2), (256, 2, 9), (64, 32, 1)])
def test_upsample(c, L, B):
    x, w, b = uni(f"up.x.{c}", (B, c, L)), uni(f"up.w.{c}", (c, c, 4), lo=-.1, hi=.1), uni(f"up.b.{c}", (c,))
    y = _ops().tconv(x.to(DEV), w.to(DEV), b.to(DEV), kind=1, stride=2, pad=1)
    close(y.cpu(), F.conv_transpose1d(x, w, b, stride=2, padding=1), 2e-5)


@pytest.mark.parametrize("c0,c1,cout,L,B", [(7, 0, 64, 32, 3), (64, 0, 128, 16, 2), (512, 512, 256, 4, 5), (64, 0, 7, 32, 4),
                                            (64, 0, 3, 16, 2)])
def test_pointwise_conv(c0, c1, cout, L, B):
    x0 = uni(f"pw.x0.{c0}.{cout}", (B, c0, L))
    x1 = uni(f"pw.x1.{c1}.{cout}", (B, c1, L)) if c1 else None
    w, b = uni(f"pw.w.{c0}.{cout}", (cout, c0 + c1, 1), lo=-.1, hi=.1), uni(f"pw.b.{cout}", (cout,))
    xin = x0 if x1 is None else torch.cat([x0, x1], 1)
    y = _ops().tconv(x0.to(DEV), w.to(DEV), b.to(DEV), x1=None if x1 is None else x1.to(DEV))
    close(y.cpu(), F.conv1d(xin, w, b), 2e-5)


def test_strided_io_is_the_einops_rearrange():
    """[B,H,D] read as [B,D,H] and head output written as [B,H,D] (temporal.py:204,243)."""
    B, H, D = 3, 32, 7
    traj = uni("st.traj", (B, H, D))
    w, b = uni("st.w", (64, 7, 1), lo=-.3, hi=.3), uni("st.b", (64,))
    y = _ops().tconv(traj.to(DEV).transpose(1, 2), w.to(DEV), b.to(DEV))
    close(y.cpu(), F.conv1d(traj.transpose(1, 2), w, b), 2e-5)
    x = uni("st.x", (B, 64, H))
    w2, b2 = uni("st.w2", (7, 64, 1), lo=-.3, hi=.3), uni("st.b2", (7,))
    out = torch.empty((B, H, D), device=DEV)
    _ops().tconv(x.to(DEV), w2.to(DEV), b2.to(DEV), out=out.transpose(1, 2))
    close(out.cpu(), F.conv1d(x, w2, b2).transpose(1, 2), 2e-5)


def test_linear_as_length1_conv():
    """The fused block time_mlp Linear: [rows, 128] -> [rows, 3840]."""
    rows = 37
    x, w, b = uni("lin.x", (rows, 128)), uni("lin.w", (3840, 128), lo=-.1, hi=.1), uni("lin.b", (3840,))
    y = _ops().tconv(x.to(DEV)[:, :, None], w.to(DEV)[:, :, None], b.to(DEV))
    close(y.cpu()[:, :, 0], F.linear(x, w, b), 2e-5)


def test_tconv_rejects_bad_shapes():
    x = torch.zeros((2, 64, 24), device=DEV)   # L = 24 is not a power of two
    w = torch.zeros((64, 64, 5), device=DEV)
    with pytest.raises(ValueError):
        _ops().tconv(x, w, pad=2)


@pytest.mark.parametrize("free", [False, True])
def test_embed(free):
    sd = oracle_sd("FREE_GUIDANCE" if free else "NO_GUIDANCE")
    rows = 6
    t = torch.tensor([0, 37, 99], dtype=torch.int64)
    feat = uni("emb.feat", (2, 64), lo=-3, hi=3)
    cond = uni("emb.cond", (rows, 2))
    te = U.time_mlp(sd, t, 64).repeat(2, 1)
    if free:
        te = te + U.cond_mlp(sd, cond)
    ci = torch.cat([te, feat.repeat(3, 1)], -1)
    import math
    freqs = torch.exp(torch.arange(32) * -(math.log(10000) / 31))
    g = lambda k: sd[k].to(DEV)  # noqa: E731
    cm = (g("cond_mlp.0.weight"), g("cond_mlp.0.bias"), g("cond_mlp.2.weight"), g("cond_mlp.2.bias")) if free else None
    te_d, mc_d = _ops().embed(freqs.to(DEV), g("time_mlp.1.weight"), g("time_mlp.1.bias"), g("time_mlp.3.weight"),
                              g("time_mlp.3.bias"), t.to(DEV), feat.to(DEV), rows,
                              cond=cond.to(DEV) if free else None, cond_mlp=cm)
    close(te_d.cpu(), te, 5e-6)
    close(mc_d.cpu(), F.mish(ci), 5e-6)


def test_scheduler_steps(golden):
    """All four step() variants, three prediction types, against the golden vectors of the reference:
    integer tables bit-exact, fp32 outputs within 2 ulp."""
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
    g = golden("sched")
    cfg = create_cfg()
    for n in (100, 50, 10, 2):
        s = S.GuidanceDDIMScheduler(cfg=cfg, thresholding=True, **SCHED_KW)
        s.set_timesteps(n, device=DEV)
        assert np.array_equal(s.timesteps.numpy(), g[f"sched.timesteps.{n}"])
        assert np.array_equal(s.timesteps.tensor.cpu().numpy(), g[f"sched.timesteps.{n}"])
        assert all(int(t) == v for t, v in zip(s.timesteps, g[f"sched.timesteps.{n}"]))
    s = S.GuidanceDDIMScheduler(cfg=cfg, **SCHED_KW)
    assert np.array_equal(s.betas.numpy(), g["sched.betas"]) and np.array_equal(s.alphas_cumprod.numpy(), g["sched.alphas_cumprod"])
    u = lambda n, lo=-1.5, hi=1.5: P._uniform(n, 21, (3, 16, 7), lo, hi)  # noqa: E731
    mo, x = u("sched.mo").to(DEV), u("sched.x").to(DEV)
    z = P.step_noise(0, (3, 16, 7), seed=21).to(DEV)
    tt, tm = u("sched.tt", -1, 1).to(DEV), (P._uniform("sched.tm", 21, (3, 16, 7), 0, 1) > 0.5).float().to(DEV)
    bad = []

    def eq(got, key):
        ref = g[key]
        err = np.abs(got.cpu().numpy() - ref)
        if not (err <= 2.4e-7 + 2.4e-7 * np.abs(ref)).all():   # 1 ulp of the O(1) intermediate terms
            bad.append((key, float(err.max())))

    for pt in ("sample", "epsilon", "v_prediction"):
        kw = dict(SCHED_KW, prediction_type=pt)
        for n, ts in ((50, (98, 50, 0)), (10, (90, 0)), (100, (99, 1, 0))):
            for thr in (True, False):
                s = S.GuidanceDDIMScheduler(cfg=cfg, thresholding=thr, **kw)
                s.set_timesteps(n, device=DEV)
                for t in ts:
                    r = s.step(mo, torch.tensor(t), x)
                    eq(r.prev_sample, f"sched.ddim.{pt}.thr{int(thr)}.n{n}.t{t}.prev")
                    eq(r.pred_original_sample, f"sched.ddim.{pt}.thr{int(thr)}.n{n}.t{t}.x0")
            s = S.GuidanceDDIMScheduler(cfg=cfg, thresholding=True, **kw)
            s2 = S.GuidanceDDPMScheduler(cfg=cfg, thresholding=False, **kw)
            s3, s4 = S.InpaintingDDIMScheduler(**kw), S.InpaintingDDPMScheduler(**kw)
            for q in (s, s2, s3, s4):
                q.set_timesteps(n, device=DEV)
            for t in ts:
                T = torch.tensor(t)
                eq(s.step(mo, T, x, eta=0.5, variance_noise=z).prev_sample, f"sched.ddim.{pt}.eta.n{n}.t{t}.prev")
                eq(s2.step(mo, T, x, variance_noise=z).prev_sample, f"sched.ddpm.{pt}.n{n}.t{t}.prev")
                eq(s3.step(mo, T, x, variance_noise=z, target_traj=tt, target_mask=tm).prev_sample, f"sched.inp_ddim.{pt}.n{n}.t{t}.prev")
                eq(s3.step(mo, T, x, variance_noise=z).prev_sample, f"sched.inp_ddim.{pt}.n{n}.t{t}.plain")
                eq(s4.step(mo, T, x, variance_noise=z, target_traj=tt, target_mask=tm).prev_sample, f"sched.inp_ddpm.{pt}.n{n}.t{t}.prev")
                eq(s4.step(mo, T, x, variance_noise=z).prev_sample, f"sched.inp_ddpm.{pt}.n{n}.t{t}.plain")
    s = S.DDPMScheduler(**SCHED_KW)
    eq(s.add_noise(x, z, torch.tensor([0, 50, 99], device=DEV)), "sched.add_noise")
    assert not bad, f"{len(bad)} mismatches, first: {bad[:5]}"


def test_scheduler_fusions_and_errors():
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    cfg = create_cfg()
    s = S.GuidanceDDIMScheduler(cfg=cfg, thresholding=True, **SCHED_KW)
    x = uni("fz.x", (4, 32, 7), lo=-2, hi=2).to(DEV)
    out2 = uni("fz.out2", (8, 32, 7), lo=-2, hi=2).to(DEV)
    with pytest.raises(ValueError):
        s.step(out2[:4], torch.tensor(90), x)           # set_timesteps not called
    s.set_timesteps(50, device=DEV)
    c, u = out2.chunk(2, 0)
    ref = s.step(u + 7.5 * (c - u), s.timesteps[3], x).prev_sample
    ref[:, 0, :3] = 0
    fused = s.step(out2, s.timesteps[3], x, cfg_scale=7.5, zero_first=True).prev_sample
    assert torch.equal(ref, fused)
    with pytest.raises(ValueError):
        s.set_timesteps(101)
    bad = S.GuidanceDDIMScheduler(cfg=cfg, **dict(SCHED_KW, prediction_type="nope"))
    bad.set_timesteps(10, device=DEV)
    with pytest.raises(ValueError):
        bad.step(x, torch.tensor(90), x)
    # add_noise fused with the [...,0,:3] = 0 of train.py:235
    z = uni("fz.z", (4, 32, 7)).to(DEV)
    t = torch.tensor([0, 10, 50, 99], device=DEV)
    a = S.DDPMScheduler(**SCHED_KW).add_noise(x, z, t)
    a[..., 0, :3] = 0
    assert torch.equal(a, S.DDPMScheduler(**SCHED_KW).add_noise(x, z, t, zero_first=True))
    o = SCH.GuidanceDDIM(**SCHED_KW)
    assert torch.equal(o.add_noise(x.cpu(), z.cpu(), t.cpu()), S.DDPMScheduler(**SCHED_KW).add_noise(x, z, t).cpu())


def test_scheduler_step_equals_stepwise_torch_on_gpu():
    """The fused step kernel is bit-identical to the reference's operation sequence issued op by op
    (IEEE mul/add/sub and a true division) on the same device."""
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    cfg = create_cfg()
    x = uni("sw.x", (64, 32, 7), lo=-2, hi=2).to(DEV)
    mo = uni("sw.mo", (64, 32, 7), lo=-2, hi=2).to(DEV)
    for pt in ("sample", "epsilon", "v_prediction"):
        s = S.GuidanceDDIMScheduler(cfg=cfg, thresholding=True, **dict(SCHED_KW, prediction_type=pt))
        s.set_timesteps(50, device=DEV)
        for t in (98, 50, 2, 0):
            c = s._ddim_coef(t, 0.0, False)
            sa, sb = (torch.full((1,), v, device=DEV) for v in (c.sqrt_alpha_t, c.sqrt_beta_t))
            if pt == "sample":
                x0, eps = mo, (x - sa * mo) / sb
            elif pt == "epsilon":
                x0, eps = (x - sb * mo) / sa, mo
            else:
                x0, eps = sa * x - sb * mo, sa * mo + sb * x
            prev = torch.full((1,), c.c_x0, device=DEV) * x0.clamp(-1, 1) + torch.full((1,), c.c_dir, device=DEV) * eps
            r = s.step(mo, torch.tensor(t), x)
            assert torch.equal(r.prev_sample, prev), (pt, t)
            assert torch.equal(r.pred_original_sample, x0.clamp(-1, 1)), (pt, t)


def test_image_front_end():
    """uint8 HWC camera frame -> ToTensor + ImageNet Normalize (interact.py:73-78), fused."""
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (2, 37, 53, 3), generator=g, dtype=torch.uint8)
    ref = frames.permute(0, 3, 1, 2).float().div(255.0)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    ref = (ref - mean) / std
    out = _ops().image_transform(frames.to(DEV))
    close(out.cpu(), ref, 5e-7, rtol=1e-6)


def test_image_front_end_folded_into_the_stem_is_bit_identical():
    """uint8 HWC frames -> ToTensor + Normalize inside the ResNet stem's staging load
    (adx_resnet_forward_u8) == the separate normalise kernel followed by the fp32 perception pass, bit for bit
    (odd sizes: the zero padding of the NORMALISED image must stay zero, not normalise(0))."""
    from autonomous_driving_with_diffusion_model_amd import ops
    from autonomous_driving_with_diffusion_model_amd.modeling.perception import PerceptionResNet34
    from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
    m = PerceptionResNet34(64)
    P.load_procedural(m, 0)
    m = m.to(DEV).eval()
    g = torch.Generator().manual_seed(1)
    for shape in ((2, 64, 96, 3), (1, 71, 101, 3), (3, 256, 900, 3)):
        frames = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8).to(DEV)
        with torch.no_grad():
            want = m(ops.image_transform(frames))
            got = m.forward_frames(frames)
        assert torch.equal(got, want), (shape, (got - want).abs().max().item())
    with torch.no_grad():
        # a single [H, W, 3] frame; one frame and three take different reduction splits in the deep layers (small-grid
        # launches split their input channels over workgroups), so the match is to summation order, not to the bit
        close(m.forward_frames(frames[0]).cpu(), want[:1].cpu(), 2e-5, rtol=1e-6)
        assert torch.equal(m.forward_frames(frames[:1]), m(ops.image_transform(frames[:1])))
    from autonomous_driving_with_diffusion_model_amd._lib import AdxError
    with pytest.raises(AdxError):
        m.forward_frames(frames.float())


def test_gpu_augment_stand_in_vs_numpy_oracle():
    """adx_image_augment (blur / additive noise / coarse dropout / dropout / add / multiply / contrast in a
    host-drawn random order, dataset/augment.py:10-77) against oracle/augment.py on the same plan and seeds: bit-exact,
    except where the Gaussian noise operator ran (logf / cosf differ in the last bit between the device and numpy, which can
    move a value across a rounding boundary: those images may differ by one grey level on < 1 % of their pixels)."""
    from autonomous_driving_with_diffusion_model_amd.dataset import augment as A
    from oracle import augment as OA
    rng = np.random.default_rng(5)
    n, h, w = 24, 37, 53
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    plan = A.sample_plan(32 * 600000 + np.arange(n), h, w, rng)          # late in training: every operator is likely
    codes = plan[0][:, :, 0].astype(int)
    assert all((codes == c).any() for c in range(1, 8)), "the drawn plans do not cover every operator"
    aug = A.GpuAugmentor(seed=0)
    got = aug(torch.from_numpy(frames).to(DEV), 0, plan=plan).cpu().numpy()
    want = OA.augment(frames, *plan)
    assert (got != frames).mean() > 0.5
    for i in range(n):
        if (codes[i] == A.NOISE).any():
            d = np.abs(got[i].astype(int) - want[i].astype(int))
            # a one-grey-level difference before a Multiply / LinearContrast slot can be amplified by that slot's factor
            assert d.max() <= 4 and (d > 0).mean() < 0.01, (i, d.max(), (d > 0).mean())
        else:
            assert np.array_equal(got[i], want[i]), (i, codes[i].tolist())
    # same plan, same seeds -> same images; fresh plans from the augmentor's own generator differ
    again = aug(torch.from_numpy(frames).to(DEV), 0, plan=plan).cpu().numpy()
    assert np.array_equal(again, got)
    a, b = aug(torch.from_numpy(frames).to(DEV), 32 * 600000), aug(torch.from_numpy(frames).to(DEV), 32 * 600000)
    assert not torch.equal(a, b)
    with pytest.raises(Exception):
        aug(torch.from_numpy(frames), 0)                                 # CPU tensors are refused: no CPU path
