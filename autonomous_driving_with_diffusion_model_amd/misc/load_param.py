"""Positional EMA -> model copy used by the agents (reference: misc/load_param.py:4-8,
called at interact.py:104 and e2e_driving/diffusion_agent.py:70)."""
import torch


def copy_parameters(from_parameters, to_parameters):
    to_parameters = list(to_parameters)
    assert len(from_parameters) == len(to_parameters)
    with torch.no_grad():
        for src, dst in zip(from_parameters, to_parameters):
            # copy through the Parameter itself (not .data) so that its version counter moves and
            # the model re-packs its HIP weight image on the next forward
            dst.copy_(src.to(dst.device))
