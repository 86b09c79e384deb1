"""The callers' sampling loops, restated over the drop-in model/scheduler objects.

`generate_traj` mirrors `interact.Agent.generate_traj` (interact.py:115-168) ==
`DiffusionAgent.generate_traj` (e2e_driving/diffusion_agent.py:179-232); `evaluate_sample` mirrors
the loop of `train.evaluate` (train.py:62-90).  They exist so that parity tests and the benchmark can
drive the exact call sequence the reference's callers issue: model.forward + scheduler.step per
timestep, `trajs[:, 0, :3] = 0` before the loop and after every step, final clamp and xy scaling.

Differences from the callers, both optional and numerically identical:
  * `target` may be [B, 2] (one goal per scene) instead of one [2] goal repeated over the batch;
  * `fuse=True` folds the classifier-free combine and the `[:, 0, :3] = 0` write into the scheduler's
    step kernel instead of issuing them as separate torch ops.
"""
from __future__ import annotations

import contextlib
from typing import Callable, Optional

import torch

from .misc.constant import GuidanceType


def _targets(target: Optional[torch.Tensor], batch: int) -> Optional[torch.Tensor]:
    if target is None:
        return None
    t = target.reshape(-1, 2)
    if t.shape[0] == 1:
        t = t.repeat(batch, 1)   # interact.py:120
    if t.shape[0] != batch:
        raise ValueError(f"target must be [2] or [{batch}, 2], got {tuple(target.shape)}")
    return t.contiguous()


def generate_traj(model, scheduler, cfg, image: torch.Tensor, target: Optional[torch.Tensor] = None,
                  init_trajs: Optional[torch.Tensor] = None, *, fuse: bool = True, scale_xy: bool = True,
                  step_noise: Optional[Callable[[int, tuple], torch.Tensor]] = None,
                  set_timesteps: bool = True) -> torch.Tensor:
    use = GuidanceType[cfg.GUIDANCE.USE_COND]
    model.eval()
    device = image.device
    if init_trajs is None:
        init_trajs = torch.randn((image.shape[0], cfg.MODEL.HORIZON, cfg.MODEL.TRANSITION_DIM), device=device)
    trajs = init_trajs.clone().detach()
    B = trajs.shape[0]
    tgt = _targets(target, B)
    cond = None
    if tgt is not None and use == GuidanceType.FREE_GUIDANCE:
        cond = torch.cat([tgt, torch.zeros_like(tgt)], dim=0)   # interact.py:121-127
    trajs[:, 0, :3] = 0.0
    if set_timesteps:
        scheduler.set_timesteps(cfg.EVAL.SAMPLE_STEPS, device=device)
    is_ddpm = not getattr(scheduler, "_is_ddim", False)
    # What the UNet derives from (t, target, image feature) alone does not change inside the loop: with the perception
    # memo on (the product default) it is computed for all timesteps in one pass, and each step then starts at the first
    # convolution.  The reference-faithful mode (cache_perception = False) keeps the reference's per-step recomputation.
    tc = None
    if fuse and getattr(model, "cache_perception", False) and hasattr(model, "time_conditioning"):
        rows = 2 * B if use == GuidanceType.FREE_GUIDANCE else B
        ts = scheduler.timesteps
        ts = ts.tensor if hasattr(ts, "tensor") else torch.as_tensor(ts)
        with torch.no_grad():
            tc = model.time_conditioning(image, ts.to(device), cond=cond, rows=rows)
    pair = (lambda x: x) if B == 1 and tc is not None else (lambda x: torch.cat([x, x], dim=0))
    # nothing in this loop writes `image`: say so, so that the reference-faithful per-step encoder pass of a batched tick may run
    # beside the previous step's temporal stack (modeling/perception.py:frozen_image; a no-op for holders without the method)
    frozen = getattr(getattr(model, "perception", None), "frozen_image", None)
    with (frozen(image) if frozen is not None else contextlib.nullcontext()):
        trajs = _tick_loop(model, scheduler, cfg, use, image, trajs, B, tgt, cond, tc, pair, fuse, is_ddpm, step_noise, device)
    trajs = trajs.to(torch.float32).clamp(-1, 1)
    if scale_xy:
        trajs[..., :2] *= model.magic_num
    return trajs


def _tick_loop(model, scheduler, cfg, use, image, trajs, B, tgt, cond, tc, pair, fuse, is_ddpm, step_noise, device):
    action = None
    for i, t in enumerate(scheduler.timesteps):
        tck = None if tc is None else (tc, i)
        extra = {}
        if is_ddpm and step_noise is not None:
            extra["variance_noise"] = step_noise(i, tuple(trajs.shape)).to(device)
        if use == GuidanceType.FREE_GUIDANCE:
            with torch.no_grad():
                out = model(pair(trajs), image, t.reshape(-1), cond=cond, time_cond=tck)
            if fuse:
                trajs = scheduler.step(out, t, trajs, cfg_scale=cfg.GUIDANCE.FREE_SCALE, zero_first=True,
                                       **extra).prev_sample
                continue
            c, u = out.chunk(2, dim=0)
            model_output = u + cfg.GUIDANCE.FREE_SCALE * (c - u)
            trajs = scheduler.step(model_output, t, trajs, **extra).prev_sample
        elif use == GuidanceType.CLASSIFIER_GUIDANCE:
            with torch.no_grad():
                action, time_embed = model(trajs, image, t.reshape(-1).repeat(B), return_action_and_time_only=True,
                                           time_cond=tck)
            guided = getattr(scheduler, "use_classifier_guidance", False) and tgt is not None
            if fuse and guided and scheduler.guidance_loss.guidance_step == 1:
                # one launch: state_pred forward + TargetGuidance + its gradient through state_pred + update + clip
                model_output = model.state_pred.guided_output(action, time_embed, tgt, scheduler.guidance_std(t),
                                                              scheduler.guidance_loss.scale)
                trajs = scheduler.step(model_output, t, trajs, zero_first=True, **extra).prev_sample
                continue
            action = action.detach().requires_grad_()
            with torch.enable_grad():
                state = model.state_pred(action[:, :-1], time_embed)
                state = torch.cat([torch.zeros_like(state[:, :1]), state], dim=1)
                model_output = torch.cat([state, action], dim=-1)
            trajs = scheduler.step(model_output, t, trajs, target=tgt, action=action, **extra).prev_sample.detach()
        else:
            with torch.no_grad():
                model_output = model(trajs, image, t.reshape(-1).repeat(B), time_cond=tck)
            if fuse:
                trajs = scheduler.step(model_output, t, trajs, zero_first=True, **extra).prev_sample
                continue
            trajs = scheduler.step(model_output, t, trajs, **extra).prev_sample
        trajs[:, 0, :3] = 0.0
    return trajs


@torch.no_grad()
def evaluate_sample(model, noise_scheduler, image: torch.Tensor, init_trajs: torch.Tensor, n_steps: int,
                    step_noise: Optional[Callable[[int, tuple], torch.Tensor]] = None) -> torch.Tensor:
    """train.evaluate's loop: stock DDPM scheduler, B copies of one image, fresh/injected noise."""
    model.eval()
    B = init_trajs.shape[0]
    trajs = init_trajs.clone()
    trajs[:, 0, :3] = 0
    noise_scheduler.set_timesteps(n_steps, device=image.device)
    tc = None
    if getattr(model, "cache_perception", False) and hasattr(model, "time_conditioning"):     # as in generate_traj
        ts = noise_scheduler.timesteps
        tc = model.time_conditioning(image, (ts.tensor if hasattr(ts, "tensor") else torch.as_tensor(ts)).to(image.device), rows=B)
    for i, t in enumerate(noise_scheduler.timesteps):
        out = model(trajs, image, t.reshape(-1).repeat(B), time_cond=None if tc is None else (tc, i))
        kw = {}
        if step_noise is not None:
            kw["variance_noise"] = step_noise(i, tuple(trajs.shape)).to(image.device)
        trajs = noise_scheduler.step(out, t, trajs, **kw).prev_sample
        trajs[:, 0, :3] = 0
    return trajs


class GraphedSampler:
    """`generate_traj` captured once as a HIP graph and replayed per tick.

    The loop is a fixed sequence of ~50 launches per denoising step with no host decision inside it (the timestep
    values travel as kernel arguments, the guidance rule runs on the device), so for the small batches of real driving
    (one scene, B = 1 or 2 with classifier-free guidance) the host's launch work is a visible part of the tick: at
    B = 1 the 50-step DDIM loop takes 29.2 ms eagerly and 26.4 ms as one graph launch on an MI355X
    (tools/graph_probe.py; at B = 64 the GPU is the bound either way).  Results are bit-identical to the eager loop.

    Deterministic samplers only (DDIM with eta = 0; a DDPM loop would replay its captured noise), eval mode, fused
    step path.  Inputs are copied into static buffers; the camera frame's perception pass is part of the graph, so
    every replay sees the new frame.
    """

    def __init__(self, model, scheduler, cfg, *, scale_xy: bool = True):
        if not getattr(scheduler, "_is_ddim", False) or float(getattr(cfg.EVAL, "ETA", 0) or 0) != 0.0:
            raise ValueError("GraphedSampler needs a deterministic sampler (DDIM, eta = 0)")
        self.model, self.scheduler, self.cfg, self.scale_xy = model, scheduler, cfg, scale_xy
        self._key = None
        self._graph = None

    def _capture(self, image, target, init_trajs):
        dev = image.device
        self.model.eval()
        self.scheduler.set_timesteps(self.cfg.EVAL.SAMPLE_STEPS, device=dev)   # host tables + device timesteps, once
        # the graph reads these device tensors on every replay: keep them alive even if somebody calls
        # scheduler.set_timesteps() again (which replaces the scheduler's own references)
        self._timesteps = list(self.scheduler.timesteps)
        self._img, self._init = image.clone(), init_trajs.clone()
        self._tgt = None if target is None else target.clone()
        run = lambda: generate_traj(self.model, self.scheduler, self.cfg, self._img, self._tgt, self._init,  # noqa: E731
                                    fuse=True, scale_xy=self.scale_xy, set_timesteps=False)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # warm-up off the capture: lazy packs, workspaces, tile tables
            run()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.model._feat_cache = None          # the perception pass must be IN the graph (new frame every tick)
        self._graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: a process group's watchdog thread (multi-rank runs) may query events while this
        # thread captures; in the default global mode that would invalidate the capture
        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
            self._out = run()
        self.model._feat_cache = None          # the memo now points at the static frame buffer: drop it
        self._pointers = self._model_pointers()

    def _model_pointers(self):
        """Addresses of the model-owned buffers the captured launches read and write (workspaces and packed weight
        images).  The model re-allocates them lazily (a later eager forward at a larger batch or image size, a weight
        re-pack); a replay over stale addresses would touch freed memory, so `__call__` re-captures when any moved."""
        m = self.model
        owners = [m, getattr(m, "perception", None), getattr(m, "state_pred", None)]
        return tuple(None if t is None else t.data_ptr()
                     for o in owners if o is not None for t in (getattr(o, "_ws", None), getattr(o, "_packed", None)))

    def reset(self) -> None:
        """Forget the captured graph (call after the model's weights changed: the weight images are packed outside
        the graph, during the warm-up pass of the next capture)."""
        self._key, self._graph = None, None

    @torch.no_grad()
    def __call__(self, image: torch.Tensor, target: Optional[torch.Tensor] = None,
                 init_trajs: Optional[torch.Tensor] = None) -> torch.Tensor:
        if init_trajs is None:
            init_trajs = torch.randn((image.shape[0], self.cfg.MODEL.HORIZON, self.cfg.MODEL.TRANSITION_DIM),
                                     device=image.device)
        key = (tuple(image.shape), None if target is None else tuple(target.shape), tuple(init_trajs.shape), image.device,
               self.cfg.EVAL.SAMPLE_STEPS, self.cfg.GUIDANCE.USE_COND)
        if key != self._key or self._graph is None or self._pointers != self._model_pointers():
            self._capture(image, target, init_trajs)
            self._key = key
        else:
            self._img.copy_(image)
            self._init.copy_(init_trajs)
            if target is not None:
                self._tgt.copy_(target)
        self._graph.replay()
        return self._out.clone()
