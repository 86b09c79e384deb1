"""Minimal stand-in for the reference's yacs config (config.py:9-118): an attribute namespace with
the same defaults for the keys the hot path reads, `_BASE_` yaml inheritance and `--opts K V ...`
overrides.  yacs/colorama are not dependencies of the hot path (out of scope)."""
from __future__ import annotations

import ast
import os
from typing import Any, Dict, Sequence

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def merge_from_dict(self, d: Dict[str, Any]):
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_dict(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def merge_from_list(self, opts: Sequence[str]):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            try:
                v = ast.literal_eval(v) if isinstance(v, str) else v
            except (ValueError, SyntaxError):
                pass
            node[parts[-1]] = v


def create_cfg() -> CfgNode:
    """Defaults of reference config.py:9-103 for the keys read on the hot path."""
    c = CfgNode()
    c.MODEL = CfgNode(HORIZON=16, TRANSITION_DIM=7, USE_ATTN=False, DIM=64, DIM_MULTS=(1, 2, 4, 8),
                      DIFFUSER_BUILDING_BLOCK="concat")
    c.TRAIN = CfgNode(RESUME=None, ROOT=None, USE_IMG_AUGMENTOR=False, NUM_WORKERS=4, USE_COND="NO_GUIDANCE", USE_FREE_COND_PROB=0.7, BATCH_SIZE=32, MAX_ITER=100000,
                      IMAGE_HEIGHT=256, IMAGE_WIDTH=900, GRAD_NORM=1.0, EMA_MAX_DECAY=0.9999, EMA_INV_GAMMA=1.0,
                      EMA_POWER=0.75, LR=1e-4, LR_WARMUP=1000, TIME_STEPS=100, SAMPLE_STEPS=100,
                      GRADIENT_ACCUMULATION_STEPS=1,
                      NOISE_SCHEDULER=CfgNode(BETA_START=1e-4, BETA_END=0.02, TYPE="squaredcos_cap_v2",
                                              PRED_TYPE="sample"))
    c.GUIDANCE = CfgNode(USE_COND="NO_GUIDANCE", LOSS_LIST=None, STEP=1, CLASSIFIER_SCALE=0.1, FREE_SCALE=1.0)
    c.EVAL = CfgNode(BATCH_SIZE=4, ETA=0, CHECKPOINT=None, SCHEDULER="ddim", SAMPLE_STEPS=100)
    # post-sampling control (reference config.py:67-86)
    c.PID = CfgNode(TURN_KP=1, TURN_KI=0.5, TURN_KD=1.0, TURN_N=40, SPEED_KP=5, SPEED_KI=0.5, SPEED_KD=1.0, SPEED_N=40)
    c.CONTROL = CfgNode(AIM_DIST=4.0, ANGLE_THRESH=0.3, DIST_THRESH=10, BRAKE_SPEED=0.4, BRAKE_RATIO=1.1, CLIP_DELTA=0.25,
                        MAX_THROTTLE=9)
    return c


def merge_possible_with_base(cfg: CfgNode, config_path: str) -> CfgNode:
    """config.py:106-111: merge `_BASE_` first, then the file itself."""
    with open(config_path) as f:
        new = yaml.safe_load(f) or {}
    base = new.pop("_BASE_", None)
    if base:
        merge_possible_with_base(cfg, os.path.join(os.path.dirname(config_path), base))
    cfg.merge_from_dict(new)
    return cfg
