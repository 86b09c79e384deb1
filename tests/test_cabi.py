"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/adx.h declares
(no compute calls: there is no GPU in this container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from autonomous_driving_with_diffusion_model_amd import _lib
    return _lib


def test_header_symbols_exported(built):
    header = open(os.path.join(ROOT, "include", "adx.h")).read()
    declared = set(re.findall(r"\b(adx_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    handle = ctypes.CDLL(built.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(handle, s)]
    assert not missing, f"declared in adx.h but not exported: {missing}"
    assert set(built.EXPORTED_SYMBOLS) <= declared


def test_version_and_error_string(built):
    lib = built.lib()
    assert lib.adx_version() >= 1
    assert isinstance(lib.adx_last_error(), bytes)


def test_host_side_validation_without_gpu(built):
    """Shape validation happens on the host before any launch."""
    lib = built.lib()
    d = built.TConvDesc(0, 5, 1, 2, 64, 0, 64, 24, 24, 8, 1e-5)      # L = 24: not a power of two
    assert lib.adx_tconv_packed_bytes(ctypes.byref(d)) == 0
    assert b"power of two" in lib.adx_last_error()
    d = built.TConvDesc(0, 5, 1, 2, 64, 0, 64, 32, 32, 8, 1e-5)
    assert lib.adx_tconv_packed_bytes(ctypes.byref(d)) == 4 * (64 // 16) * 5 * (64 // 16) * 256
    cfg = built.UnetConfig()
    cfg.horizon, cfg.transition_dim, cfg.dim, cfg.n_mults, cfg.guidance = 32, 7, 64, 4, 0
    for i, m in enumerate((1, 2, 4, 8)):
        cfg.dim_mults[i] = m
    h = built.vp()
    assert lib.adx_unet_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    assert lib.adx_unet_num_params(h) == 196   # named_parameters() of the reference minus perception.*
    assert lib.adx_unet_packed_bytes(h) > 64_000_000         # 16 M UNet-side parameters
    assert lib.adx_unet_workspace_bytes(h, 64) > 0
    lib.adx_unet_destroy(h)
    cfg.horizon = 24          # not a power of two: runs on 32 with the real lengths as masks (24 -> 12 -> 6 -> 3)
    assert lib.adx_unet_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    lib.adx_unet_destroy(h)
    cfg.horizon = 8           # GroupNorm groups of 32 elements at the bottom of the up path: the general-shape kernel's
    assert lib.adx_unet_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    lib.adx_unet_destroy(h)
    cfg.horizon, cfg.dim = 32, 48     # GroupNorm(8, 48): groups of 6 channels (modeling/helpers.py:105-107 takes any C % 8 == 0)
    assert lib.adx_unet_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    lib.adx_unet_destroy(h)
    cfg.dim = 64
    for bad in (12, 72):      # not divisible by 8 (the reference's own down / up path breaks); longer than 64
        cfg.horizon = bad
        assert lib.adx_unet_create(ctypes.byref(cfg), ctypes.byref(h)) == -1, bad


def test_resnet_backward_groups_map_every_gradient_to_the_event_of_its_layer_group(built):
    """adx_resnet_backward_events (csrc/resnet_train.hip) records one completion event per layer group in the order the backward
    produces the gradients -- fc, the 16 BasicBlocks from layer4's last to layer1's first, the stem -- and
    adx_resnet_tensor_group tells which event covers which tensor slot of the state_dict order.  Host-side bookkeeping: checked
    against the parameter names without a GPU."""
    from autonomous_driving_with_diffusion_model_amd.modeling.spec import resnet34_entries
    lib = built.lib()
    h = built.vp()
    assert lib.adx_resnet_create(64, ctypes.byref(h)) == 0
    try:
        n = lib.adx_resnet_backward_groups(h)
        assert n == 18
        entries = [e for e in resnet34_entries("", 64) if e.dtype == "f32"]
        assert len(entries) == lib.adx_resnet_num_tensors(h)
        blocks = [f"layer{li}.{b}." for li, nb in zip((1, 2, 3, 4), (3, 4, 6, 3)) for b in range(nb)]
        seen = set()
        for i, e in enumerate(entries):
            g = lib.adx_resnet_tensor_group(h, i)
            if e.is_buffer:
                assert g == -1, e.key                       # running statistics: no gradient
                continue
            if e.key.startswith("fc."):
                want = 0
            elif e.key.startswith(("conv1.", "bn1.")):
                want = n - 1
            else:
                b = next(j for j, pre in enumerate(blocks) if e.key.startswith(pre))
                want = 1 + (len(blocks) - 1 - b)            # the backward visits the blocks in reverse
            assert g == want, (e.key, g, want)
            seen.add(g)
        assert seen == set(range(n))
        assert lib.adx_resnet_tensor_group(h, -1) == -1 and lib.adx_resnet_tensor_group(h, len(entries)) == -1
    finally:
        lib.adx_resnet_destroy(h)


def test_cell_layout_host_side(built):
    """The cell layout of csrc/conv2d_hs.hip on the host: ops.to_cells / from_cells (pure torch) place the 8 channels of a
    pixel as one 16-byte cell per plane, per image [C / 8][hi, lo][H][W], and lose at most 2^-22 of the value (values far below
    fp16's normal range: the lo half's subnormal step); which launches
    may use it is a host decision (adx_conv2d_cells_supported: plain launches of the pipelined 3x3 kernel only)."""
    import torch
    from autonomous_driving_with_diffusion_model_amd import ops
    n, c, h, w = 2, 16, 3, 5
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(5)) * torch.logspace(-3, 3, c).view(1, c, 1, 1)
    cells = ops.to_cells(x)
    assert cells.dtype == torch.uint8 and cells.numel() == x.numel() * 4
    halves = cells.view(torch.float16).view(n, c // 8, 2, h, w, 8)
    hi = x.to(torch.float16)
    assert torch.equal(halves[1, 1, 0, 2, 4], hi[1, 8:16, 2, 4])                      # image 1, cell group 1, hi plane, pixel (2, 4)
    assert torch.equal(halves[0, 0, 1, 1, 3], ((x - hi.float()) * 2048).to(torch.float16)[0, 0:8, 1, 3])
    back = ops.from_cells(cells, x.shape)
    # 2^-22 of the value, plus the lo half's fp16 subnormal step (2^-24 / 2^11) where the value is far below 1
    assert bool(((back - x).abs() <= 2.0 ** -21 * x.abs() + 6e-11).all())
    lib = built.lib()
    d3 = built.Conv2dDesc(64, 64, 3, 1, 1)
    assert lib.adx_conv2d_cells_supported(ctypes.byref(d3), 64, 64, 225) == 1        # 4096 tiles: a plain launch
    assert lib.adx_conv2d_cells_supported(ctypes.byref(built.Conv2dDesc(512, 512, 3, 1, 1)), 1, 8, 29) == 0   # one frame: split reduction
    assert lib.adx_conv2d_cells_supported(ctypes.byref(built.Conv2dDesc(64, 128, 3, 2, 1)), 64, 64, 225) == 0  # stride 2: another kernel
    assert lib.adx_conv2d_cells_supported(ctypes.byref(built.Conv2dDesc(64, 64, 1, 1, 0)), 64, 64, 225) == 0


def test_isa_has_no_wide_store_followed_by_a_valu_write_of_its_data():
    """tools/check_store_hazard.py over every kernel source: on gfx950 a VALU write to the data registers of a >= 8-byte
    store in the next issue slot can reach the store when the store carries an SGPR offset (found in the cell epilogue of
    csrc/conv2d_hs.hip, guarded there); the compiler only covers the stores without one.  Cross-compiles, needs no GPU."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_store_hazard as C
    wide, found = C.scan("""
k:
\tbuffer_store_dwordx4 v[98:101], v202, s[4:7], s11 offen
\tv_mul_f32_e32 v100, v200, v66
\tglobal_store_dwordx2 v1, v[2:3], s[0:1]
\ts_nop 0
\tv_mov_b32_e32 v2, 0
""")
    assert wide == 2 and len(found) == 1 and "v100" in found[0][1]
    assert C.main([]) == 0


def test_cpu_tensors_are_refused(built):
    import torch
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    s = S.DDPMScheduler(num_train_timesteps=100, beta_schedule="squaredcos_cap_v2", prediction_type="sample")
    s.set_timesteps(10)
    x = torch.zeros(1, 16, 7)
    with pytest.raises(built.AdxError):
        s.step(x, s.timesteps[0], x)
