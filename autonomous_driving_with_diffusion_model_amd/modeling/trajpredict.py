"""`TrajPredict` state head used by classifier guidance (reference: modeling/helpers.py:22-59).

Parameter holder with the reference's keys (input_proj, encoder_traj.layers.N.*, encoder_traj.norm,
output_proj).  The HIP forward + input-gradient kernels are the next hot-path row to land
(SURVEY.md §8a M7/G1); until then calling it raises instead of falling back to torch ops.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .holders import populate
from .spec import traj_predict_entries


class TrajPredict(nn.Module):
    def __init__(self, in_dim: int = 3, out_dim: int = 3, pred_len: int = 16, hidden_dim: int = 256,
                 num_heads: int = 4, num_layers: int = 3):
        super().__init__()
        self.in_dim, self.out_dim, self.pred_len = in_dim, out_dim, pred_len
        self.hidden_dim, self.num_heads, self.num_layers = hidden_dim, num_heads, num_layers
        populate(self, traj_predict_entries("", in_dim, out_dim, hidden_dim, num_layers))

    def forward(self, x: torch.Tensor, time_embed: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("TrajPredict: HIP forward/backward kernels not implemented yet")
