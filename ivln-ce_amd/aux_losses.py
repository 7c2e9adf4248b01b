"""Auxiliary-loss registry of the update step.

Interface contract (what policy.py / latent_policy.py / trainers.py call, same method names as the
reference's module-level `AuxLosses` object, ivlnce_baselines/common/aux_losses.py:4-44):
`activate()`, `deactivate()`, `is_active()`, `clear()`, `register_loss(name, loss, alpha)`,
`get_loss(name)`, `reduce(mask)`.  The only registered term on this hot path is the progress monitor's
(TN,TN) squared-error matrix (quirk Q7, `train.progress_monitor_loss`), whose masked mean times `alpha`
joins the action loss in `trainers.update_agent`.
"""
from typing import Dict, NamedTuple

import torch


class _Term(NamedTuple):
    # per-element loss, any shape `mask` broadcasts against under masked_select - or an object that offers
    # `.values` (the same tensor, built on demand) and `.masked_mean(mask, alpha)` (the reduction done directly,
    # e.g. train.PMLossTerm: no (TN, TN) matrix, no masked_select, no host synchronisation in the middle of an update)
    values: object
    alpha: float


class AuxLossRegistry:
    """Process-wide registry: terms are only accepted between `activate()` and `deactivate()` (the trainers
    switch it on around the updates, never during rollouts) and live until the next `clear()`."""

    def __init__(self):
        self._terms: Dict[str, _Term] = {}
        self._enabled = False

    # -- switch ---------------------------------------------------------------------------------
    def activate(self):
        self._enabled = True

    def deactivate(self):
        self._enabled = False

    def is_active(self) -> bool:
        return self._enabled

    # -- terms ----------------------------------------------------------------------------------
    def clear(self):
        self._terms = {}

    def __len__(self):
        return len(self._terms)

    def register_loss(self, name: str, loss: torch.Tensor, alpha: float = 1.0):
        if not self._enabled:
            raise AssertionError("AuxLosses.register_loss while inactive")
        if name in self._terms:
            raise AssertionError(f"aux loss `{name}` registered twice before clear()")
        self._terms[name] = _Term(loss, float(alpha))

    def get_loss(self, name: str) -> torch.Tensor:
        v = self._terms[name].values
        return v if torch.is_tensor(v) else v.values

    def reduce(self, mask: torch.Tensor):
        """sum_k alpha_k * mean(values_k[mask]); 0.0 when nothing is registered."""
        if not self._enabled:
            raise AssertionError("AuxLosses.reduce while inactive")
        total = None
        for term in self._terms.values():
            if torch.is_tensor(term.values):
                t = term.alpha * torch.masked_select(term.values, mask).mean()
            else:
                t = term.values.masked_mean(mask, term.alpha)
            total = t if total is None else total + t
        return 0.0 if total is None else total


AuxLosses = AuxLossRegistry()
