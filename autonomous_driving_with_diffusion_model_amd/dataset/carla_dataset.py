"""Reader for the reference's on-disk training format (dataset/carla_dataset.py:11-58; writer at
misc/data_collect.py:176-208):

    <root>/front/*.png             RGB camera frames, sorted by name; sample idx = position in that order
    <root>/waypoints/%06d.txt      line 1: "tx ty" (target point, ego frame, / magic_num)
                                   then 16 lines "x y speed sin cos ... " (7 numbers: the trajectory state rows)

`TrajDataset[idx]` -> (image, waypoints [16, 7] clipped to [-1, 1], target_point [2]) like the reference.  Frames are
decoded with Pillow (the reference's cv2.imread + BGR->RGB gives the same bytes for PNG).  With `img_transforms=None`
the frame is returned as a uint8 HWC tensor so that a whole batch can go through the package's GPU front-end
(`ops.image_transform`, ToTensor + ImageNet Normalize in one kernel) instead of per-sample CPU transforms.
The reference's imgaug pipeline (dataset/augment.py) runs per sample on the CPU; its stand-in here is batch-level and on
the GPU (`dataset.augment.GpuAugmentor`: same seven operators and iteration schedule, applied to the uint8 batch where
it lives, before `ops.image_transform`).  A per-sample CPU hook remains: `augment` takes any callable
(HWC uint8 ndarray, access counter) -> ndarray.
"""
from __future__ import annotations

import glob
import os
from typing import Callable, Optional

import numpy as np
import torch


def read_waypoint_file(path: str):
    """-> (waypoints float32 [n, 7] clipped to [-1, 1], target float32 [2])."""
    with open(path, "r") as f:
        rows = [ln.split() for ln in f.read().splitlines()]
    rows = [r for r in rows if r]
    if not rows:
        raise ValueError(f"{path}: empty waypoint file")
    target = torch.tensor([float(v) for v in rows[0]], dtype=torch.float32)
    wps = torch.tensor([[float(v) for v in r] for r in rows[1:]], dtype=torch.float32)
    return wps.clip(-1, 1), target


class TrajDataset(torch.utils.data.Dataset):
    def __init__(self, root_path: str, img_transforms: Optional[Callable] = None, use_img_augmentor: bool = False,
                 augment: Optional[Callable] = None, horizon: int = 16):
        if use_img_augmentor and augment is None:
            raise NotImplementedError("imgaug is not part of this package: augment the uint8 batch on the GPU with "
                                      "dataset.augment.GpuAugmentor (the stand-in for dataset/augment.py), or pass a "
                                      "per-sample augment=callable(image_uint8_hwc, access_count)")
        self.root_path, self.img_transforms, self.augment = root_path, img_transforms, augment
        self.horizon = horizon
        self.count_access = 0
        self.front_image = sorted(glob.glob(os.path.join(root_path, "front", "*.png")))

    def __len__(self) -> int:
        return len(self.front_image)

    def __getitem__(self, idx: int):
        from PIL import Image
        with Image.open(self.front_image[idx]) as im:
            img = np.array(im.convert("RGB"))          # a writable copy
        if self.augment is not None:
            self.count_access += 1
            img = self.augment(img, self.count_access)
        img = self.img_transforms(img) if self.img_transforms is not None else torch.from_numpy(np.ascontiguousarray(img))
        wps, target = read_waypoint_file(os.path.join(self.root_path, "waypoints", f"{idx:06d}.txt"))
        if len(wps) != self.horizon:
            raise ValueError(f"sample {idx}: {len(wps)} waypoint rows, expected {self.horizon}")
        return img, wps, target


def get_loader(cfg, train: bool, img_transforms: Optional[Callable] = None, augment: Optional[Callable] = None):
    """dataset/carla_dataset.py:44-58: shuffled when training, drop_last, pinned."""
    ds = TrajDataset(cfg.TRAIN.ROOT, img_transforms=img_transforms,
                     use_img_augmentor=bool(getattr(cfg.TRAIN, "USE_IMG_AUGMENTOR", False)), augment=augment,
                     horizon=cfg.MODEL.HORIZON)
    return torch.utils.data.DataLoader(ds, shuffle=train, batch_size=cfg.TRAIN.BATCH_SIZE,
                                       num_workers=getattr(cfg.TRAIN, "NUM_WORKERS", 0), pin_memory=torch.cuda.is_available(),
                                       drop_last=True)
