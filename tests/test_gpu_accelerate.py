"""The reference's training loop driven through `accelerate` ITSELF (train.py:115-125, 176-178, 221-261, 283-299, 181-194).

north_star: "drops into ... accelerate-launched training unchanged".  `train.py` is not importable on the box (yacs, aim,
loguru, diffusers ...), so its loop body is restated here statement for statement around the REAL `accelerate.Accelerator`:
`prepare(model, optimizer, lr_scheduler, dataloader)`, `accumulate`, `accelerator.backward`, `sync_gradients`, the
nan_to_num loop, `torch.optim.AdamW`, the `LambdaLR` that `get_constant_schedule_with_warmup` returns, and an EMA with
diffusers 0.28's update rule (restated from the published API, as in oracle/diffusers_base.py).  Two iterations on the
fixture's batch must reproduce what the REFERENCE's model and the same torch objects produced (tests/golden/ckpt.npz:
losses, weights, both moments, EMA shadow); then train.py:283-299's save and :181-194's resume run on the file.
World size 1 (a gpurun box has one GPU): `Accelerator.prepare` then leaves the model unwrapped -- the DDP wrap of the
multi-process case is covered by test_model_under_torch_ddp_world1_nccl and tests/test_gpu_parallel.py."""
import json
import os

import pytest
import torch

from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
from helpers import IMG_SMALL, SCHED_KW, close

pytestmark = pytest.mark.gpu
accelerate = pytest.importorskip("accelerate")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class EMAModel:
    """diffusers==0.28.0 `EMAModel` as train.py:146-153,258-259,294 uses it (step / state_dict / load_state_dict / to),
    restated from the published API; the decay rule is oracle.diffusers_base.ema_decay's."""

    def __init__(self, parameters, decay=0.9999, min_decay=0.0, update_after_step=0, use_ema_warmup=False, inv_gamma=1.0,
                 power=2 / 3):
        self.shadow_params = [p.clone().detach() for p in parameters]
        self.decay, self.min_decay, self.update_after_step = decay, min_decay, update_after_step
        self.use_ema_warmup, self.inv_gamma, self.power = use_ema_warmup, inv_gamma, power
        self.optimization_step = 0

    def get_decay(self, optimization_step):
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        cur = 1 - (1 + step / self.inv_gamma) ** -self.power if self.use_ema_warmup else (1 + step) / (10 + step)
        return max(min(cur, self.decay), self.min_decay)

    @torch.no_grad()
    def step(self, parameters):
        self.optimization_step += 1
        one_minus_decay = 1 - self.get_decay(self.optimization_step)
        for s_param, param in zip(self.shadow_params, list(parameters)):
            s_param.sub_(one_minus_decay * (s_param - param))

    def to(self, device):
        self.shadow_params = [p.to(device) for p in self.shadow_params]

    def state_dict(self):
        return {"decay": self.decay, "min_decay": self.min_decay, "optimization_step": self.optimization_step,
                "update_after_step": self.update_after_step, "use_ema_warmup": self.use_ema_warmup,
                "inv_gamma": self.inv_gamma, "power": self.power, "shadow_params": self.shadow_params}

    def load_state_dict(self, sd):
        self.decay, self.min_decay, self.optimization_step = sd["decay"], sd["min_decay"], sd["optimization_step"]
        self.update_after_step, self.use_ema_warmup = sd["update_after_step"], sd["use_ema_warmup"]
        self.inv_gamma, self.power = sd["inv_gamma"], sd["power"]
        self.shadow_params = [p.clone() for p in sd["shadow_params"]]


def test_reference_train_loop_through_accelerate_vs_reference_written_checkpoint(golden, tmp_path):
    import random
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from autonomous_driving_with_diffusion_model_amd.config import create_cfg
    from autonomous_driving_with_diffusion_model_amd.misc.constant import GuidanceType
    from autonomous_driving_with_diffusion_model_amd.modeling import build_model
    g = golden("ckpt")
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "ckpt_spec.json")))
    cfg = create_cfg()
    cfg.TRAIN.LR_WARMUP, cfg.TRAIN.MAX_ITER = spec["warmup"], spec["iter"]

    # ---- train.py:115-178 -------------------------------------------------------------------------------------------
    accelerator = accelerate.Accelerator(gradient_accumulation_steps=cfg.TRAIN.GRADIENT_ACCUMULATION_STEPS)
    assert accelerator.num_processes == 1 and accelerator.device.type == "cuda"
    device = accelerator.device
    model = build_model(cfg)
    P.load_procedural(model, 0)                           # (the reference starts from ImageNet + default init)
    noise_scheduler = S.DDPMScheduler(num_train_timesteps=cfg.TRAIN.SAMPLE_STEPS,
                                      prediction_type=cfg.TRAIN.NOISE_SCHEDULER.PRED_TYPE,
                                      beta_schedule=cfg.TRAIN.NOISE_SCHEDULER.TYPE,
                                      beta_start=cfg.TRAIN.NOISE_SCHEDULER.BETA_START,
                                      beta_end=cfg.TRAIN.NOISE_SCHEDULER.BETA_END)
    kw = spec["ema_kw"]
    ema_model = EMAModel(model.parameters(), update_after_step=kw["update_after_step"], decay=kw["max_decay"],
                         use_ema_warmup=True, inv_gamma=kw["inv_gamma"], power=kw["power"])
    d = P.synthetic_batch(2, 16, image_hw=IMG_SMALL, seed=51)
    dataloader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(d["imgs"], d["trajs"], d["target"]), batch_size=2)
    optimizer = torch.optim.AdamW(model.parameters(), lr=cfg.TRAIN.LR, betas=(0.95, 0.999), eps=1e-7)
    lr_scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda k: min(1.0, k / max(1, cfg.TRAIN.LR_WARMUP)))
    model, optimizer, lr_scheduler, dataloader = accelerator.prepare(model, optimizer, lr_scheduler, dataloader)
    ema_model.to(accelerator.device)
    assert next(model.parameters()).is_cuda

    # ---- train.py:205-325 -------------------------------------------------------------------------------------------
    weight_dtype = torch.float32
    max_iter = cfg.TRAIN.MAX_ITER
    loader = iter(dataloader)
    use_cond = GuidanceType[cfg.TRAIN.USE_COND]
    cur_iter, losses = 0, []
    save_name = None
    while True:
        model.train()
        try:
            imgs, trajs, target_point = next(loader)
        except StopIteration:
            loader = iter(dataloader)
            imgs, trajs, target_point = next(loader)
        assert imgs.is_cuda                                # accelerate's DataLoaderShard put the batch on the device
        imgs = imgs.to(weight_dtype)
        trajs = trajs.to(weight_dtype)
        target_point = target_point.to(weight_dtype)
        # the fixture fixed (t, noise) instead of drawing them (train.py:232-233)
        t = d["t"].to(device).long()
        noise = d["noise"].to(device)
        noise_data = noise_scheduler.add_noise(trajs, noise, t)
        noise_data[..., 0, :3] = 0
        with accelerator.accumulate(model):
            if use_cond == GuidanceType.FREE_GUIDANCE and random.random() > cfg.TRAIN.USE_FREE_COND_PROB:
                target_point = None
            pred = model(noise_data, imgs, t, cond=target_point)
            if cfg.TRAIN.NOISE_SCHEDULER.PRED_TYPE == "epsilon":
                loss = torch.nn.functional.mse_loss(pred.float(), noise.float())
            elif cfg.TRAIN.NOISE_SCHEDULER.PRED_TYPE == "sample":
                loss = torch.nn.functional.mse_loss(pred.float(), trajs.float())
            else:
                raise ValueError("Not supported prediction type.")
            accelerator.backward(loss)
            if accelerator.sync_gradients:
                for param in model.parameters():
                    if param.grad is not None:
                        torch.nan_to_num(param.grad, nan=0, posinf=1e5, neginf=-1e5, out=param.grad)
            optimizer.step()
            lr_scheduler.step()
            optimizer.zero_grad()
        if accelerator.sync_gradients:
            ema_model.step(model.parameters())
        losses.append(loss.item())
        if ((cur_iter + 1) % 1000 == 0 or cur_iter + 1 == max_iter) and accelerator.is_main_process and accelerator.sync_gradients:
            state_dict = {"state_dict": accelerator.unwrap_model(model).state_dict(),
                          "optimizer": optimizer.optimizer.state_dict(),
                          "lr_scheduler": lr_scheduler.scheduler.state_dict(),
                          "iter": cur_iter + 1,
                          "ema_state_dict": ema_model.state_dict()}
            save_name = str(tmp_path / ("final.pth" if cur_iter + 1 == max_iter else f"checkpoint_{cur_iter + 1}.pth"))
            torch.save(state_dict, save_name)
        if accelerator.sync_gradients:
            cur_iter += 1
        if cur_iter == max_iter:
            break
        accelerator.wait_for_everyone()

    # ---- against the checkpoint the REFERENCE's objects wrote ---------------------------------------------------------
    for it in range(spec["iter"]):
        assert abs(losses[it] - float(g[f"ckpt.loss.{it}"])) < 2e-5, (it, losses[it])
    ck = torch.load(save_name, map_location="cpu", weights_only=False)
    assert list(ck) == spec["keys"] and ck["iter"] == spec["iter"] and list(ck["state_dict"]) == spec["state_dict_keys"]
    names = spec["parameter_names"]
    for k in g.files:
        kind, _, name = k.partition(".")[2].partition(".")
        if kind not in ("param", "exp_avg", "exp_avg_sq", "shadow"):
            continue
        ref = torch.from_numpy(g[k])
        i = names.index(name)
        got = {"param": lambda: ck["state_dict"][name], "exp_avg": lambda: ck["optimizer"]["state"][i]["exp_avg"],
               "exp_avg_sq": lambda: ck["optimizer"]["state"][i]["exp_avg_sq"],
               "shadow": lambda: ck["ema_state_dict"]["shadow_params"][i]}[kind]().cpu()
        if kind in ("param", "shadow"):
            close(got, ref, 2e-5, rtol=1e-6)       # same bars as test_two_optimizer_steps_and_checkpoint_vs_reference_written_fixture
        else:
            e = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
            assert e <= (2e-3 if kind == "exp_avg_sq" else 1e-3), (k, e)
    close(ck["state_dict"]["perception.bn1.running_mean"], g["ckpt.bn_running_mean"], 1e-5)
    assert ck["lr_scheduler"]["last_epoch"] == spec["iter"]

    # ---- train.py:181-194: resume from that file into fresh objects, through accelerate again --------------------------
    model2 = build_model(cfg)
    optimizer2 = torch.optim.AdamW(model2.parameters(), lr=cfg.TRAIN.LR, betas=(0.95, 0.999), eps=1e-7)
    lr_scheduler2 = torch.optim.lr_scheduler.LambdaLR(optimizer2, lambda k: min(1.0, k / max(1, cfg.TRAIN.LR_WARMUP)))
    ema2 = EMAModel(model2.parameters(), update_after_step=kw["update_after_step"], decay=kw["max_decay"], use_ema_warmup=True,
                    inv_gamma=kw["inv_gamma"], power=kw["power"])
    model2, optimizer2, lr_scheduler2 = accelerator.prepare(model2, optimizer2, lr_scheduler2)
    ema2.to(accelerator.device)
    with accelerator.main_process_first():
        state_dict = torch.load(save_name, map_location=device, weights_only=False)
    ema2.load_state_dict(state_dict["ema_state_dict"])
    accelerator.unwrap_model(model2).load_state_dict(state_dict["state_dict"])
    optimizer2.optimizer.load_state_dict(state_dict["optimizer"])
    lr_scheduler2.scheduler.load_state_dict(state_dict["lr_scheduler"])
    start_iter = state_dict["iter"] + 1
    assert start_iter == spec["iter"] + 1
    # the resumed objects continue exactly like the originals: one more iteration on both, same loss, same weights
    def one_more(mdl, opt, lrs):
        mdl.train()
        imgs, trajs = d["imgs"].to(device), d["trajs"].to(device)
        t, noise = d["t"].to(device).long(), d["noise"].to(device)
        noise_data = noise_scheduler.add_noise(trajs, noise, t)
        noise_data[..., 0, :3] = 0
        with accelerator.accumulate(mdl):
            loss = torch.nn.functional.mse_loss(mdl(noise_data, imgs, t, cond=None).float(), trajs.float())
            accelerator.backward(loss)
            opt.step()
            lrs.step()
            opt.zero_grad()
        return loss.item()
    la, lb = one_more(model, optimizer, lr_scheduler), one_more(model2, optimizer2, lr_scheduler2)
    assert abs(la - lb) <= 1e-6 * max(1.0, abs(la)), (la, lb)
    for (k, p), q in zip(model.named_parameters(), model2.parameters()):
        # weight gradients are reduced with float atomics: two runs of the same step differ in the last bits
        assert (p - q).abs().max().item() <= 1e-6 + 1e-5 * p.abs().max().item(), k

    # ---- the package's own checkpoint reader accepts the file train.py's loop wrote (interact.py:102-106) ---------------
    from autonomous_driving_with_diffusion_model_amd.checkpoint import load_checkpoint
    m3 = build_model(cfg).to(device)
    load_checkpoint(save_name, m3, use_ema=True)
    for p, s in zip(m3.parameters(), ck["ema_state_dict"]["shadow_params"]):
        assert torch.equal(p.detach().cpu(), s.cpu())
