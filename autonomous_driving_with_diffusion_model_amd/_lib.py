"""ctypes binding of libadx.so (C ABI declared in include/adx.h).

There is NO fallback: if the shared library is missing every op raises, so a GPU box can
never silently run an eager/PyTorch path instead of the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADX_LIB") or os.path.join(_HERE, "libadx.so")   # ADX_LIB: A/B builds in one run

c_f32p = C.POINTER(C.c_float)
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class TConvDesc(C.Structure):
    _fields_ = [("kind", i32), ("taps", i32), ("stride", i32), ("pad", i32), ("c0", i32), ("c1", i32),
                ("cout", i32), ("lin", i32), ("lout", i32), ("groups", i32), ("eps", f32), ("w_layout", i32),
                ("w_flip", i32), ("exact", i32), ("lin_valid", i32), ("lout_valid", i32)]


class TConvIO(C.Structure):
    _fields_ = [("x0", vp), ("x0_sb", i64), ("x0_sc", i64), ("x0_sl", i64),
                ("x1", vp), ("x1_sb", i64), ("x1_sc", i64), ("x1_sl", i64),
                ("packed_w", vp), ("bias", vp), ("gamma", vp), ("beta", vp),
                ("tbias", vp), ("tbias_stride", i64),
                ("res", vp), ("res_sb", i64), ("res_sc", i64), ("res_sl", i64),
                ("y", vp), ("y_sb", i64), ("y_sc", i64), ("y_sl", i64), ("batch", i32), ("pre", vp), ("stats", vp),
                ("scratch", vp), ("scratch_floats", i64), ("tickets", vp)]


class EmbedWeights(C.Structure):
    _fields_ = [(n, vp) for n in ("freqs", "w1", "b1", "w3", "b3", "cw0", "cb0", "cw2", "cb2")]


class UnetConfig(C.Structure):
    _fields_ = [("horizon", i32), ("transition_dim", i32), ("dim", i32), ("n_mults", i32),
                ("dim_mults", i32 * 8), ("guidance", i32)]


class UnetIO(C.Structure):
    _fields_ = [("x", vp), ("img_feature", vp), ("feat_rows", i32), ("t", vp), ("t_rows", i32), ("cond", vp),
                ("rows", i32), ("out", vp), ("time_embed", vp), ("x_rows", i32), ("time_bias", vp)]


class Conv2dDesc(C.Structure):
    _fields_ = [("cin", i32), ("cout", i32), ("k", i32), ("stride", i32), ("pad", i32)]


class StepCoef(C.Structure):
    _fields_ = [("prediction_type", i32), ("clip", i32), ("clip_range", f32), ("sqrt_alpha_t", f32),
                ("sqrt_beta_t", f32), ("c_x0", f32), ("c_dir", f32), ("c_x", f32), ("c_noise", f32),
                ("add_noise", i32), ("use_clipped_model_output", i32), ("inpaint", i32), ("c_const", f32),
                ("c_known", f32), ("c_known_noise", f32), ("known_noise", i32), ("cfg_combine", i32),
                ("free_scale", f32), ("zero_first", i32)]


_SIGS = {
    "adx_version": (i32, []),
    "adx_last_error": (C.c_char_p, []),
    "adx_source_hash": (C.c_char_p, []),
    "adx_tconv_packed_bytes": (C.c_size_t, [C.POINTER(TConvDesc)]),
    "adx_tconv_pack": (i32, [C.POINTER(TConvDesc), vp, vp, vp]),
    "adx_tconv_forward": (i32, [C.POINTER(TConvDesc), C.POINTER(TConvIO), vp]),
    "adx_embed_forward": (i32, [C.POINTER(EmbedWeights), i32, vp, i32, vp, vp, i32, i32, vp, vp, vp]),
    "adx_unet_create": (i32, [C.POINTER(UnetConfig), C.POINTER(vp)]),
    "adx_unet_destroy": (None, [vp]),
    "adx_unet_num_params": (i32, [vp]),
    "adx_unet_packed_bytes": (C.c_size_t, [vp]),
    "adx_unet_pack": (i32, [vp, C.POINTER(vp), i32, vp, vp, vp]),
    "adx_unet_workspace_bytes": (C.c_size_t, [vp, i32]),
    "adx_unet_forward": (i32, [vp, vp, vp, C.POINTER(UnetIO), vp]),
    "adx_unet_time_bias_width": (i32, [vp]),
    "adx_unet_time_conditioning_workspace_bytes": (C.c_size_t, [vp, i32]),
    "adx_unet_time_conditioning": (i32, [vp, vp, vp, C.POINTER(UnetIO), vp, vp, vp]),
    "adx_unet_tape_create": (i32, [C.POINTER(vp)]),
    "adx_unet_tape_destroy": (None, [vp]),
    "adx_unet_train_workspace_bytes": (C.c_size_t, [vp, i32]),
    "adx_unet_forward_train": (i32, [vp, vp, vp, C.c_size_t, C.POINTER(UnetIO), vp, vp]),
    "adx_unet_backward": (i32, [vp, vp, vp, C.c_size_t, vp, vp, vp, vp, C.POINTER(vp), C.POINTER(vp), i32, vp]),
    "adx_gn_mish_backward": (i32, [vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "adx_tconv_wgrad": (i32, [C.POINTER(TConvDesc), C.POINTER(TConvIO), vp, vp, vp]),
    "adx_bias_grad": (i32, [vp, vp, i32, i32, i32, vp]),
    "adx_resnet_create": (i32, [i32, C.POINTER(vp)]),
    "adx_resnet_destroy": (None, [vp]),
    "adx_resnet_num_tensors": (i32, [vp]),
    "adx_resnet_packed_bytes": (C.c_size_t, [vp]),
    "adx_resnet_pack": (i32, [vp, C.POINTER(vp), i32, vp, vp]),
    "adx_resnet_workspace_bytes": (C.c_size_t, [vp, i32, i32, i32]),
    "adx_resnet_forward": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "adx_resnet_forward_u8": (i32, [vp, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), i32, i32, i32, vp, vp]),
    "adx_resnet_tape_create": (i32, [C.POINTER(vp)]),
    "adx_resnet_tape_destroy": (None, [vp]),
    "adx_resnet_train_workspace_bytes": (C.c_size_t, [vp, i32, i32, i32]),
    "adx_resnet_forward_train": (i32, [vp, C.POINTER(vp), i32, vp, vp, C.c_size_t, vp, i32, i32, i32, vp, vp, i32, vp]),
    "adx_resnet_backward": (i32, [vp, C.POINTER(vp), C.POINTER(vp), i32, vp, C.c_size_t, vp, vp, vp]),
    "adx_resnet_backward_groups": (i32, [vp]),
    "adx_resnet_tensor_group": (i32, [vp, i32]),
    "adx_resnet_backward_events": (i32, [vp, C.POINTER(vp), C.POINTER(vp), i32, vp, C.c_size_t, vp, vp, C.POINTER(vp), i32, vp]),
    "adx_conv2d_packed_bytes": (C.c_size_t, [C.POINTER(Conv2dDesc)]),
    "adx_conv2d_pack": (i32, [C.POINTER(Conv2dDesc), vp, vp, vp]),
    "adx_conv2d_forward": (i32, [C.POINTER(Conv2dDesc), vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "adx_conv2d_cells_supported": (i32, [C.POINTER(Conv2dDesc), i32, i32, i32]),
    "adx_conv2d_forward_cells": (i32, [C.POINTER(Conv2dDesc), vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "adx_conv2d_wgrad_scratch_bytes": (C.c_size_t, []),
    "adx_conv2d_wgrad": (i32, [C.POINTER(Conv2dDesc), vp, vp, vp, i32, i32, i32, vp, vp]),
    "adx_conv2d_wgrad_ex": (i32, [C.POINTER(Conv2dDesc), vp, vp, vp, i32, i32, i32, vp, i32, vp]),
    "adx_conv2d_wgrad_cells": (i32, [C.POINTER(Conv2dDesc), vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "adx_trajpred_create": (i32, [i32, C.POINTER(vp)]),
    "adx_trajpred_destroy": (None, [vp]),
    "adx_trajpred_num_params": (i32, [vp]),
    "adx_trajpred_packed_bytes": (C.c_size_t, [vp]),
    "adx_trajpred_scratch_bytes": (C.c_size_t, [vp, i32, i32]),
    "adx_trajpred_set_scratch": (i32, [vp, vp, C.c_size_t]),
    "adx_trajpred_pack": (i32, [vp, C.POINTER(vp), i32, vp, vp, vp]),
    "adx_trajpred_forward": (i32, [vp, vp, vp, i64, i64, vp, vp, i32, i32, vp]),
    "adx_trajpred_backward": (i32, [vp, vp, vp, i64, i64, vp, vp, vp, i32, i32, vp]),
    "adx_trajpred_backward_params": (i32, [vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, i32, i32, C.c_float, C.c_uint64, vp]),
    "adx_trajpred_forward_train": (i32, [vp, vp, vp, i64, i64, vp, vp, i32, i32, C.c_float, C.c_uint64, vp]),
    "adx_trajpred_param_offsets": (i32, [vp, C.POINTER(i64), i32]),
    "adx_guided_output": (i32, [vp, vp, vp, vp, vp, f32, f32, vp, vp, i32, i32, vp]),
    "adx_optim_chunk": (i32, []),
    "adx_adamw_ema_step": (i32, [vp, vp, vp, i32, f32, f32, f32, f32, f32, i32, f32, i32, i32, vp]),
    "adx_adamw_ema_step_scaled": (i32, [vp, vp, vp, i32, f32, f32, f32, f32, f32, i32, f32, i32, i32, f32, vp]),
    "adx_image_normalize": (i32, [vp, vp, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), vp]),
    "adx_probe_mfma_fp16": (i32, [vp, vp, i32, i32, C.POINTER(C.c_double), vp]),
    "adx_probe_mfma_fp16_16x16x32": (i32, [vp, vp, i32, i32, C.POINTER(C.c_double), vp]),
    "adx_image_augment": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp]),
    "adx_ddim_step": (i32, [C.POINTER(StepCoef), vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "adx_ddpm_step": (i32, [C.POINTER(StepCoef), vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "adx_add_noise": (i32, [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib: Optional[C.CDLL] = None


class AdxError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libadx.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AdxError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or autonomous_driving_with_diffusion_model_amd/csrc/build.sh). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class NativeTape:
    """Owner of one native training tape (adx_*_tape_create / _destroy).  The handle is destroyed exactly once: by
    `release()` after the backward pass, or by the finalizer when the autograd node that holds it is dropped without a
    backward (a train-mode forward under no_grad, a loss that is never back-propagated)."""

    def __init__(self, create, destroy, what: str):
        import weakref
        h = vp()
        check(create(C.byref(h)), what)
        self.handle = h
        self._fin = weakref.finalize(self, destroy, h)

    @property
    def alive(self) -> bool:
        return self._fin.alive

    def release(self) -> None:
        self._fin()          # idempotent


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().adx_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}" if what else msg)
        raise AdxError(f"{what} failed (status {rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu_f32(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise AdxError(f"{name} lives on {t.device}; the adx kernels only run on an MI355X (no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def write_stamp(t: torch.Tensor) -> int:
    """A number that moves when `t` is written in place: autograd's version counter.  Inference tensors -- everything
    created under `torch.inference_mode()`, which is how the reference's `train.evaluate` runs (train.py:53) -- carry no
    counter (`._version` raises on them); they answer 0, so a memo keyed on (identity, stamp) then rests on identity alone:
    an in-place write to an inference tensor between two forwards is NOT seen.  The image-feature memo therefore skips
    inference tensors unless the caller opts in (`model.cache_perception = "identity"`, modeling/temporal.py:image_feature);
    the weight-image memos key on parameters, which optimizers and loaders replace or write outside inference mode."""
    return 0 if t.is_inference() else t._version


def grad_buffer(p: torch.Tensor) -> torch.Tensor:
    """Where a backward node writes d(loss)/d(p).  Under `parallel.GradientAverager` every parameter owns a view into a
    flat communication bucket (`p._adx_grad_view`); when the parameter has no gradient yet, a fresh alias of that view is
    handed out -- autograd adopts a gradient tensor nobody else holds as `.grad` without copying, so the gradient is born
    inside the bucket the collective runs on.  With a gradient already present (accumulation over several backwards)
    autograd ADDS what the node returns to `.grad`, which may be this very storage: then, and without an averager, the
    node gets a buffer of its own.

    The view is LENT at most once per backward (`p._adx_grad_leased`): a parameter that feeds two nodes of one graph (the
    model called twice before one `loss.backward()`, shared weights) reaches both nodes with `.grad is None` -- AccumulateGrad
    has not run yet -- and two nodes writing the same storage would leave autograd summing two aliases of the second
    gradient.  The averager clears the lease when the gradient has been accumulated (its post-accumulate hook) or reduced."""
    v = getattr(p, "_adx_grad_view", None)
    if (v is not None and p.grad is None and not getattr(p, "_adx_grad_leased", False)
            and v.device == p.device and v.shape == p.shape):
        p._adx_grad_leased = True
        return v.detach()
    return torch.empty_like(p)


def ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (vp * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
