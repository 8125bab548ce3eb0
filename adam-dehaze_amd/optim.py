"""Adam on the fused multi-tensor HIP kernel, with the reference's joint-training quirk available.

/root/reference training/train_joint.py:81-89 builds the optimiser from `router.parameters()` (which
already contains every branch through nn.ModuleDict) and then appends every branch parameter again, so
those tensors appear TWICE in torch's parameter list.  What torch does with that depends on the code
path:

* `duplicates="sequential"` (default): the single-tensor loop (`foreach=False`: torch's default on CPU --
  the "reference PyTorch CPU path" the north-star names -- and on every device before torch 2.0, i.e. the
  README's torch 1.12 pin): two consecutive full updates per step with shared state.
* `duplicates="foreach"`: torch >= 2.0 picks the `_foreach_` implementation on CUDA; there the duplicate
  list entries alias each other inside each foreach op: step += 2, weight decay taken from the
  un-updated p, m lerped twice, v = beta2^2 v + 2 (1-beta2) g'^2, ONE bias correction at step+2 and two
  identical parameter updates (pinned by tests/golden/adam_dup_foreach.npz, generated with
  torch.optim.Adam(foreach=True)).  The two differ by ~8 % of an update.

weight decay is torch's L2 form (added to the gradient).  One kernel launch updates every tensor
(`adh_adam_multi`); `state_dict()` / `load_state_dict()` use torch.optim.Adam's layout (duplicates share
the index of their last occurrence, as torch packs them) so checkpoints written by either side load in
the other (train_joint.py:280, train_dehazing.py:199).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from . import _hip as H


class Adam:
    def __init__(self, params: Iterable[torch.Tensor], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 duplicates: str = "sequential"):
        if duplicates not in ("sequential", "foreach"):
            raise ValueError(f"duplicates must be 'sequential' or 'foreach', got {duplicates!r}")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.duplicates = duplicates
        self.params: List[torch.Tensor] = []
        self.repeats: Dict[int, int] = {}
        self._listed: List[torch.Tensor] = []          # the list as given (with duplicates), for state_dict()
        for p in params:
            self._listed.append(p)
            if id(p) in self.repeats:
                self.repeats[id(p)] += 1
            else:
                self.repeats[id(p)] = 1
                self.params.append(p)
        if any(r > 4 for r in self.repeats.values()):
            raise ValueError("a parameter may be listed at most 4 times")
        self.state: Dict[int, dict] = {}
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay,
                              "params": self.params}]   # ReduceLROnPlateau-style access
        self.grad_scale = 1.0     # multiplies every gradient as it is read (1/world for summed DP gradients)
        self._table_key = None
        self._table_dev = None
        self._chunks_dev = None
        self._nchunks = 0
        self._calls_since_upload = 0
        self._table_host = None

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _state(self, p: torch.Tensor) -> dict:
        st = self.state.get(id(p))
        if st is None:
            st = {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)}
            self.state[id(p)] = st
        return st

    @torch.no_grad()
    def step(self):
        lr = self.param_groups[0]["lr"]
        live = []
        for p in self.params:
            if p.grad is None:
                continue
            H.require_cuda(p, "parameter")
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            live.append((p, g, self._state(p)))
        if not live:
            return
        dev = live[0][0].device
        # The pointer table (48 B per tensor, ~15 KB for the joint model) and the (tensor, chunk) list live on the device
        # and are uploaded only when a pointer or the live set changes (a per-step pageable upload made every
        # optimiser step a host-device sync).  The step counters advance on the device side of the ABI: the kernel adds
        # `calls_since_upload * repeats` to the uploaded value.  An upload goes through a pinned staging buffer and is
        # asynchronous on the current stream.
        since = self._calls_since_upload
        key = tuple((p.data_ptr(), g.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(), p.numel(), self.repeats[id(p)],
                     st["step"] - since * self.repeats[id(p)])        # the step the resident table was uploaded with
                    for p, g, st in live)
        if key != self._table_key:
            chunk = H.value("adh_adam_chunk_elems")
            table = (H.AdamTensor * len(live))()
            for i, (p, g, st) in enumerate(live):
                t = table[i]
                t.p, t.g, t.m, t.v = p.data_ptr(), g.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr()
                t.n, t.step, t.repeats = p.numel(), st["step"], self.repeats[id(p)]
            chunks = np.array([(i, c) for i, (p, _, _) in enumerate(live) for c in range((p.numel() + chunk - 1) // chunk)],
                              dtype=np.int32).reshape(-1)
            raw = np.frombuffer(bytes(table), dtype=np.uint8)
            host = torch.empty(raw.size + chunks.size * 4, dtype=torch.uint8).pin_memory()
            host[:raw.size].copy_(torch.from_numpy(raw.copy()))
            host[raw.size:].copy_(torch.from_numpy(chunks.view(np.uint8).copy()))
            both = host.to(dev, non_blocking=True)
            self._table_host = host                      # keep the staging buffer alive until the copy has run
            self._table_dev, self._chunks_dev = both[:raw.size], both[raw.size:]
            self._nchunks = chunks.size // 2
            self._calls_since_upload = 0
            self._table_key = tuple(k[:6] + (st["step"],) for k, (_, _, st) in zip(key, live))
        H.call("adh_adam_multi", self._table_dev.data_ptr(), self._chunks_dev.data_ptr(), self._nchunks, lr, self.betas[0],
               self.betas[1], self.eps, self.weight_decay, self.grad_scale, 1 if self.duplicates == "foreach" else 0,
               max(self.repeats[id(p)] for p, _, _ in live), self._calls_since_upload)
        self._calls_since_upload += 1
        for p, _, st in live:
            st["step"] += self.repeats[id(p)]
        from .engine import invalidate_weight_cache   # the kernel wrote the parameters behind torch's version counter
        invalidate_weight_cache()

    # ------------------------------------------------------------------ torch.optim.Adam-compatible (de)serialisation
    def state_dict(self) -> dict:
        index: Dict[int, int] = {}
        for i, p in enumerate(self._listed):
            index[id(p)] = i          # torch packs a duplicated parameter under the index of its LAST occurrence
        state = {}
        for p in self.params:
            st = self.state.get(id(p))
            if st is not None:
                state[index[id(p)]] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["m"].clone(),
                                       "exp_avg_sq": st["v"].clone()}
        group = {"lr": self.param_groups[0]["lr"], "betas": tuple(self.betas), "eps": self.eps,
                 "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None,
                 "params": [index[id(p)] for p in self._listed]}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self._listed):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        g = groups[0]
        by_index: Dict[int, torch.Tensor] = {}
        for idx, p in zip(g["params"], self._listed):
            by_index.setdefault(idx, p)
        self.param_groups[0]["lr"] = g["lr"]
        self.betas = tuple(g.get("betas", self.betas))
        self.eps = g.get("eps", self.eps)
        self.weight_decay = g.get("weight_decay", self.weight_decay)
        self.state = {}
        for idx, st in sd["state"].items():
            p = by_index[int(idx)]
            step = st["step"]
            self.state[id(p)] = {"step": int(round(float(step))),
                                 "m": st["exp_avg"].to(device=p.device, dtype=torch.float32).clone().contiguous(),
                                 "v": st["exp_avg_sq"].to(device=p.device, dtype=torch.float32).clone().contiguous()}
        self._table_key = None
