"""SURVEY 8(f) rows 3 and 4 on the host: the post-sampling controller against golden outputs of the real reference
(tests/golden/control.npz, made by make_golden.py control), and the TrajDataset reader on a dataset written in the
reference's on-disk format."""
import os

import numpy as np
import pytest
import torch

from autonomous_driving_with_diffusion_model_amd.config import create_cfg
from autonomous_driving_with_diffusion_model_amd.control import Controller, PIDController, post_process_control
from autonomous_driving_with_diffusion_model_amd.dataset import TrajDataset, get_loader, read_waypoint_file
from autonomous_driving_with_diffusion_model_amd.utils import procedural as P


def test_controller_matches_reference_over_80_ticks(golden):
    ref = golden("control")["control.pid80"]
    ctl = Controller(create_cfg())
    n_brake = 0
    for tick in range(80):
        wp, vel, tgt = P.control_inputs(tick)
        th, st, br = ctl.control_pid(wp, vel, tgt)
        assert bool(br) == bool(ref[tick, 2]), tick
        assert float(th) == ref[tick, 0], (tick, th, ref[tick, 0])          # bit-exact: host arithmetic, same dtypes
        assert float(st) == ref[tick, 1], (tick, st, ref[tick, 1])
        n_brake += bool(br)
    assert 5 < n_brake < 75          # the run exercises both branches


def test_pid_window_semantics():
    pid = PIDController(K_P=2.0, K_I=0.5, K_D=1.0, n=4)
    outs = [pid.step(e) for e in (1.0, 3.0, -2.0, 0.5, 4.0)]
    hist = [0, 0, 0, 0]
    want = []
    for e in (1.0, 3.0, -2.0, 0.5, 4.0):
        hist = hist[1:] + [e]
        want.append(2.0 * e + 0.5 * float(np.mean(hist)) + 1.0 * (hist[-1] - hist[-2]))
    assert np.allclose(outs, want, rtol=0, atol=1e-12)


def test_post_process_control():
    assert post_process_control(0.6, 0.1, 0.04) == (0.6, 0.1, 0.0)
    assert post_process_control(0.2, 0.1, 0.3) == (0.2, 0.1, 0.3)
    assert post_process_control(0.4, -0.2, 0.3) == (0.4, -0.2, 0.0)
    assert post_process_control(0.1, 0.0, 0.9) == (0.0, 0.0, 0.9)


def _write_dataset(root, n, h=12, w=20, horizon=16, seed=0):
    from PIL import Image
    os.makedirs(os.path.join(root, "front"))
    os.makedirs(os.path.join(root, "waypoints"))
    rng = np.random.default_rng(seed)
    imgs, wps, tgts = [], [], []
    for i in range(n):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        Image.fromarray(img).save(os.path.join(root, "front", f"{i:06d}.png"))
        wp = rng.uniform(-1.4, 1.4, size=(horizon, 7))
        tgt = rng.uniform(-1, 1, size=2)
        with open(os.path.join(root, "waypoints", f"{i:06d}.txt"), "w") as f:   # misc/data_collect.py:200-208
            f.write(f"{tgt[0]} {tgt[1]}\n")
            for row in wp:
                f.write(" ".join(map(str, row)) + "\n")
            f.write("\n")
        imgs.append(img); wps.append(wp); tgts.append(tgt)
    return imgs, wps, tgts


def test_traj_dataset_reads_the_reference_format(tmp_path):
    root = str(tmp_path / "ds")
    imgs, wps, tgts = _write_dataset(root, 5)
    ds = TrajDataset(root)
    assert len(ds) == 5
    for i in range(5):
        img, wp, tgt = ds[i]
        assert img.dtype == torch.uint8 and tuple(img.shape) == (12, 20, 3)
        assert np.array_equal(img.numpy(), imgs[i])
        assert torch.allclose(wp, torch.tensor(wps[i], dtype=torch.float32).clip(-1, 1))
        assert wp.abs().max() <= 1.0 and tuple(wp.shape) == (16, 7)
        assert torch.allclose(tgt, torch.tensor(tgts[i], dtype=torch.float32))
    # a transform sees the HWC uint8 frame, like torchvision's ToTensor pipeline in the reference
    ds2 = TrajDataset(root, img_transforms=lambda a: torch.from_numpy(a.astype(np.float32) / 255.0).permute(2, 0, 1))
    assert tuple(ds2[0][0].shape) == (3, 12, 20)
    with pytest.raises(NotImplementedError):
        TrajDataset(root, use_img_augmentor=True)
    flipped = TrajDataset(root, use_img_augmentor=True, augment=lambda a, k: a[:, ::-1].copy())
    assert np.array_equal(flipped[1][0].numpy(), imgs[1][:, ::-1])
    w2, t2 = read_waypoint_file(os.path.join(root, "waypoints", "000003.txt"))
    assert tuple(w2.shape) == (16, 7) and tuple(t2.shape) == (2,)


def test_loader_batches(tmp_path):
    root = str(tmp_path / "ds")
    _write_dataset(root, 7)
    cfg = create_cfg()
    cfg.TRAIN.ROOT, cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_WORKERS = root, 3, 0
    batches = list(get_loader(cfg, train=False))
    assert len(batches) == 2                                 # drop_last
    img, wp, tgt = batches[0]
    assert tuple(img.shape) == (3, 12, 20, 3) and img.dtype == torch.uint8
    assert tuple(wp.shape) == (3, 16, 7) and tuple(tgt.shape) == (3, 2)
