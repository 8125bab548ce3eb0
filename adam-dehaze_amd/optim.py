"""Adam on the fused HIP kernel, with the reference's joint-training quirk available.

/root/reference training/train_joint.py:81-89 builds the optimiser from `router.parameters()` (which
already contains every branch through nn.ModuleDict) and then appends every branch parameter again:
torch applies TWO consecutive Adam updates per step to those tensors (shared state, same gradient).
`Adam(params)` reproduces that when a parameter appears more than once in `params` (repeats = its
multiplicity), so `Adam(list(router.parameters()) + [p for m in models.values() for p in m.parameters()])`
matches the reference step for step.  weight decay is torch's L2 form (added to the gradient).
"""
from __future__ import annotations

from typing import Dict, Iterable, List

import torch

from . import _hip as H


class Adam:
    def __init__(self, params: Iterable[torch.Tensor], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.params: List[torch.Tensor] = []
        self.repeats: Dict[int, int] = {}
        for p in params:
            if id(p) in self.repeats:
                self.repeats[id(p)] += 1
            else:
                self.repeats[id(p)] = 1
                self.params.append(p)
        self.state: Dict[int, dict] = {}
        self.param_groups = [{"lr": lr, "params": self.params}]   # ReduceLROnPlateau-style access

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        lr = self.param_groups[0]["lr"]
        for p in self.params:
            if p.grad is None:
                continue
            H.require_cuda(p, "parameter")
            st = self.state.get(id(p))
            if st is None:
                st = {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)}
                self.state[id(p)] = st
            g = p.grad.contiguous()
            rep = self.repeats[id(p)]
            H.call("adh_adam_step", p.data_ptr(), g.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(), p.numel(),
                   st["step"], lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, rep)
            st["step"] += rep
