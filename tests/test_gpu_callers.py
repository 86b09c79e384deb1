"""The reference's own callers, in the contexts they really run in (the boundary is their call surface).

(a) `train.evaluate` (train.py:53-90) is decorated with `@torch.inference_mode()`: every tensor it creates is an
    inference tensor (no version counter, no autograd metadata).  Its loop body is restated here under the same
    decorator, with the package's objects standing where `build_model(cfg)` / `DDPMScheduler(...)` stand in train.py.
(b) `interact.Agent.generate_traj` (interact.py:115-168) after INTEGRATION.md's `sys.modules` shim: the shim is the
    code block of INTEGRATION.md, extracted and exec'd as it is printed there, and the loop imports the model, the
    schedulers and the enums under the REFERENCE's module names (`modeling`, `scheduler`, `misc.constant`,
    `misc.load_param`) -- no helper of this package appears in it.  FREE guidance runs under `no_grad` only where the
    reference has it, the NO / CLASSIFIER branches run with grad enabled like the reference's.

Both against the golden vectors the real reference produced for the same inputs (tests/golden/make_golden.py)."""
import os
import re
import sys
import types

import pytest
import torch

from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
from helpers import IMG_SMALL, SCHED_KW, close, close_traj

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMMED = ("modeling", "scheduler", "control", "misc", "misc.constant", "misc.load_param")


# ---------------------------------------------------------------------------------------------------------------------
# (a) train.evaluate
@torch.inference_mode()
def evaluate_like_train_py(cfg, unet, noise_scheduler, device, front_image_chw, init_trajs):
    """train.py:62-90 minus the file I/O and the plotting: `front_image_chw` stands for
    `img_transform(Image.open(...))`, `init_trajs` overwrites the fresh `torch.randn` draw (the golden run's draw)."""
    unet.eval()
    num_traj = cfg.EVAL.BATCH_SIZE
    traj_shape = (num_traj, cfg.MODEL.HORIZON, cfg.MODEL.TRANSITION_DIM)
    trajs = torch.randn(traj_shape, device=device)
    trajs.copy_(init_trajs)
    trajs[:, 0, :3] = 0
    front_image = torch.stack([front_image_chw]).to(device).repeat(num_traj, 1, 1, 1)
    assert front_image.is_inference() and trajs.is_inference()
    noise_scheduler.set_timesteps(cfg.TRAIN.TIME_STEPS, device=device)
    for t in noise_scheduler.timesteps:
        model_output = unet(trajs, front_image, t.reshape(-1).repeat(num_traj))
        trajs = noise_scheduler.step(model_output, t, trajs).prev_sample
        trajs[:, 0, :3] = 0
    return trajs[..., :2].to(torch.float32).clamp(-1, 1)


@pytest.mark.parametrize("cache", [True, False])
def test_train_evaluate_body_under_inference_mode_vs_golden(golden, cache):
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    from autonomous_driving_with_diffusion_model_amd.modeling import build_model
    cfg = create_cfg()
    cfg.EVAL.BATCH_SIZE, cfg.TRAIN.TIME_STEPS = 8, 10          # BASELINE cfg-1
    unet = build_model(cfg)
    P.load_procedural(unet, 0)
    unet = unet.to(DEV)
    unet.cache_perception = cache
    d = P.synthetic_batch(8, 16, image_hw=IMG_SMALL, seed=34)

    class InjectedNoise(S.DDPMScheduler):       # the golden run injected its noise; evaluate() calls step(mo, t, x)
        def set_timesteps(self, *a, **k):
            super().set_timesteps(*a, **k)
            self._i = 0

        def step(self, model_output, timestep, sample):
            z = P.step_noise(self._i, tuple(sample.shape), seed=35).to(sample.device)
            self._i += 1
            return super().step(model_output, timestep, sample, variance_noise=z)

    sch = InjectedNoise(**SCHED_KW)
    got = evaluate_like_train_py(cfg, unet, sch, torch.device(DEV), d["imgs"][0], d["init_trajs"].to(DEV))
    want = torch.as_tensor(golden("loop")["loop.evaluate.cfg1"])[..., :2].clamp(-1, 1)
    close(got.cpu(), want, 1e-4)
    # ... and the train loop that follows evaluate() (train.py:319: unet.train()) still works on the same object
    unet.train()
    dd = {k: v.to(DEV) for k, v in P.synthetic_batch(2, 16, image_hw=IMG_SMALL, seed=41).items()}
    noisy = sch.add_noise(dd["trajs"], dd["noise"], dd["t"], zero_first=True)
    loss = torch.nn.functional.mse_loss(unet(noisy, dd["imgs"], dd["t"]), dd["trajs"])
    loss.backward()
    assert abs(loss.item() - float(golden("train")["train.NO_GUIDANCE.loss"])) <= 2e-5


def test_inference_tensor_memo_follows_the_image_object():
    """The perception memo under inference_mode.  Inference tensors carry no version counter, so by default they are NOT
    memoised (an in-place refill of a frame buffer could not be seen): the encoder runs at every forward, as in the
    reference.  With `cache_perception = "identity"` the caller vouches for immutability: same image object -> one encoder
    pass, another object -> a new one."""
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    from autonomous_driving_with_diffusion_model_amd.modeling import build_model
    m = build_model(create_cfg())
    P.load_procedural(m, 0)
    m = m.to(DEV).eval()
    calls = []
    real = m.perception.forward
    m.perception.forward = lambda img: (calls.append(1), real(img))[1]
    d = P.synthetic_batch(2, 16, image_hw=IMG_SMALL, seed=3)
    with torch.inference_mode():
        img = d["imgs"].to(DEV)
        x, t = d["trajs"].to(DEV), d["t"].to(DEV)
        y0 = m(x, img, t)
        img.mul_(0.5)                                   # the next frame written into the same buffer
        y_refilled = m(x, img, t)
        assert len(calls) == 2 and not torch.equal(y0, y_refilled)
        img.mul_(2.0)
        calls.clear()
        m.cache_perception = "identity"
        y0 = m(x, img, t)
        y1 = m(x, img, t)
        assert len(calls) == 1 and torch.equal(y0, y1)
        img2 = (d["imgs"] * 0.5).to(DEV)
        y2 = m(x, img2, t)
        assert len(calls) == 2 and not torch.equal(y0, y2)
        tc = m.time_conditioning(img2, torch.tensor([5, 3], device=DEV), rows=2)      # same memo
        assert len(calls) == 2
        assert torch.equal(m(x, None, None, time_cond=(tc, 0)), m(x, img2, torch.tensor([5, 5], device=DEV)))


# ---------------------------------------------------------------------------------------------------------------------
# (b) interact.Agent.generate_traj behind the INTEGRATION.md shim
@pytest.fixture
def reference_names(tmp_path):
    """Execute INTEGRATION.md's shim block verbatim.  The reference's own `misc/` package "stays" (INTEGRATION.md): an
    empty stand-in package of that name is put on sys.path, as the reference's checkout would be."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "sys.modules[" in b]
    assert len(blocks) == 1, "INTEGRATION.md must hold exactly one shim block"
    (tmp_path / "misc").mkdir()
    (tmp_path / "misc" / "__init__.py").write_text("")
    saved = {k: sys.modules.get(k) for k in SHIMMED}
    for k in SHIMMED:
        sys.modules.pop(k, None)
    sys.path.insert(0, str(tmp_path))
    try:
        exec(compile(blocks[0], "INTEGRATION.md", "exec"), {"__name__": "shim"})
        yield
    finally:
        sys.path.remove(str(tmp_path))
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


class AgentLikeInteractPy:
    """The members `interact.Agent.__init__` sets up for `generate_traj` (interact.py:70-106), built from the REFERENCE's
    import lines (interact.py:22-27) -- which resolve to this package only through the shim."""

    def __init__(self, cfg, init_trajs, device):
        from misc.constant import GuidanceType                                   # interact.py:24
        from modeling import build_model                                         # interact.py:26
        from scheduler import GuidanceDDIMScheduler, GuidanceDDPMScheduler       # interact.py:27
        self.cfg, self.device = cfg, device
        self.use_guidance_type = GuidanceType[cfg.GUIDANCE.USE_COND]
        kw = dict(num_train_timesteps=cfg.TRAIN.SAMPLE_STEPS, prediction_type=cfg.TRAIN.NOISE_SCHEDULER.PRED_TYPE,
                  beta_schedule=cfg.TRAIN.NOISE_SCHEDULER.TYPE, beta_start=cfg.TRAIN.NOISE_SCHEDULER.BETA_START,
                  beta_end=cfg.TRAIN.NOISE_SCHEDULER.BETA_END, thresholding=True, cfg=cfg)
        self.noise_scheduler = {"ddim": GuidanceDDIMScheduler, "ddpm": GuidanceDDPMScheduler}[cfg.EVAL.SCHEDULER](**kw)
        self.init_trajs = init_trajs.to(device)
        self.model = build_model(cfg).to(device)
        self.GuidanceType = GuidanceType

    def load(self, weight):
        from misc.load_param import copy_parameters                              # interact.py:25
        self.model.load_state_dict(weight["state_dict"])                         # interact.py:104-105
        copy_parameters(weight["ema_state_dict"]["shadow_params"], self.model.parameters())

    def generate_traj(self, image, target=None):
        """interact.py:115-168, statement for statement."""
        G = self.GuidanceType
        self.model.eval()
        trajs = self.init_trajs.clone().detach()
        image = image.to(self.device)
        if target is not None and self.use_guidance_type == G.FREE_GUIDANCE:
            target = target.repeat(trajs.size(0), 1)
            target = torch.cat([target, torch.zeros_like(target)], dim=0)
        trajs[:, 0, :3] = 0.0
        self.noise_scheduler.set_timesteps(self.cfg.EVAL.SAMPLE_STEPS, device=self.device)
        action = None
        for t in self.noise_scheduler.timesteps:
            if self.use_guidance_type == G.FREE_GUIDANCE:
                input_trajs = torch.cat([trajs, trajs], dim=0)
                with torch.no_grad():
                    with_cond, without_cond = self.model(input_trajs, image, t.reshape(-1), cond=target).chunk(2, dim=0)
                model_output = without_cond + self.cfg.GUIDANCE.FREE_SCALE * (with_cond - without_cond)
            else:
                model_output = self.model(trajs, image, t.reshape(-1),
                                          return_action_and_time_only=(self.use_guidance_type == G.CLASSIFIER_GUIDANCE))
            if self.use_guidance_type == G.CLASSIFIER_GUIDANCE:
                action, time_embed = model_output
                if not action.requires_grad:
                    action.requires_grad_()
                state = self.model.state_pred(action[:, :-1], time_embed)
                state = torch.cat([torch.zeros_like(state[:, :1]), state], dim=1)
                model_output = torch.cat([state, action], dim=-1)
            trajs = self.noise_scheduler.step(model_output, t, trajs, target=target, action=action).prev_sample
            trajs[:, 0, :3] = 0.0
        trajs = trajs.to(torch.float32).clamp(-1, 1)
        trajs[..., :2] *= self.model.magic_num
        return trajs


@pytest.mark.parametrize("name,n_steps", [("NO_GUIDANCE", 10), ("FREE_GUIDANCE", 10), ("CLASSIFIER_GUIDANCE", 5)])
def test_interact_generate_traj_through_the_shim_vs_golden(golden, reference_names, name, n_steps):
    import modeling                                            # the reference's module names, aliased by the shim
    import scheduler
    import autonomous_driving_with_diffusion_model_amd as adx
    assert modeling is adx.modeling and scheduler is adx.scheduler
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    cfg = create_cfg()
    cfg.TRAIN.USE_COND = cfg.GUIDANCE.USE_COND = name
    cfg.EVAL.SAMPLE_STEPS, cfg.GUIDANCE.FREE_SCALE, cfg.GUIDANCE.CLASSIFIER_SCALE = n_steps, 7.5, 15.0
    if name == "CLASSIFIER_GUIDANCE":
        cfg.GUIDANCE.LOSS_LIST = [["TargetGuidance", []]]
    d = P.synthetic_batch(1, 16, image_hw=IMG_SMALL, seed=31)
    agent = AgentLikeInteractPy(cfg, d["init_trajs"], torch.device(DEV))
    # a checkpoint in the reference's layout, read the way interact.py:102-106 reads it
    sd = P.procedural_state_dict(((k, tuple(v.shape)) for k, v in agent.model.state_dict().items()), 0)
    shadow = [sd[k].clone() for k, _ in agent.model.named_parameters()]
    agent.load({"state_dict": {k: torch.zeros_like(v) if v.is_floating_point() and k in dict(agent.model.named_parameters())
                               else v for k, v in sd.items()},
                "ema_state_dict": {"shadow_params": shadow}})
    target = None if name == "NO_GUIDANCE" else d["target"][0].to(DEV)
    got = agent.generate_traj(d["imgs"], target)
    close_traj(got.cpu(), golden("loop")[f"loop.ddim.{name}"], 1e-4)
    # a second tick with a new camera frame object must not see the first one's feature
    got2 = agent.generate_traj(d["imgs"] * 0.25, target)
    assert not torch.equal(got, got2)
    assert torch.equal(agent.generate_traj(d["imgs"].clone(), target), got)


def test_pipeline_launch_completes_with_the_gpu_shared_between_processes(tmp_path):
    """csrc/tconv_pipe.hip is a launch of 225 workgroups in which later stages spin on earlier ones -- without a cooperative launch.
    Three processes sample on this one GPU at once (one scene, H = 16: every step of every process issues that launch), so the
    pipeline's workgroups are not all resident together; every tick of every process must still complete and reproduce the
    bits of that process's first, eagerly launched tick, and the processes must agree with each other."""
    import subprocess
    procs, outs = [], []
    for i in range(3):
        outs.append(str(tmp_path / f"w{i}.pt"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "pipe_contention_worker.py"), outs[-1], "40"],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=900)[0] for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    res = [torch.load(o) for o in outs]
    assert all(r["all_equal"] and r["finite"] for r in res), [(r["all_equal"], r["finite"]) for r in res]
    assert torch.equal(res[0]["first"], res[1]["first"]) and torch.equal(res[0]["first"], res[2]["first"])
