"""Shared helpers for the tests: golden-fixture loading and comparison utilities."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub_sd(rec, prefix="sd."):
    """Extract a state_dict (torch tensors, fresh copies) stored under `prefix`."""
    out = {}
    for k, v in rec.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(np.array(v, copy=True))
    return out


def t(a):
    return torch.from_numpy(np.array(a, copy=True))


def max_abs(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max())


def rel_err(a, b):
    """max|a-b| / max(|b|max, tiny): scale-aware error for gradients."""
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))
