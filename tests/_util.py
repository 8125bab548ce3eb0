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


class kink_matched:
    """Context manager around a HIP forward (+ backward): records the post-ReLU output of every fused conv+ReLU op of
    `model` (engine.RELU_CAPTURE) so that `.masks()` can hand the ReLU masks the kernels actually used to the oracle
    (oracle.ref_cpu.KINK_MASKS).  With matched masks both sides differentiate the same piece of the piecewise-smooth
    network, and the gradient comparison no longer has to tolerate flipped kinks."""

    def __init__(self, model):
        self.model = model
        self.cap = {}

    def __enter__(self):
        import adam_dehaze_amd.engine as E
        self._E = E
        self._old = E.RELU_CAPTURE
        self._old_cbam = E.CBAM_CAPTURE
        E.RELU_CAPTURE = self.cap
        self.cbam = {}
        E.CBAM_CAPTURE = self.cbam
        return self

    def __exit__(self, *exc):
        self._E.RELU_CAPTURE = self._old
        self._E.CBAM_CAPTURE = self._old_cbam
        return False

    def cbam_indices(self):
        """{AttentionBlock 'fc.0.weight' parameter name: (global max-pool arg-max [N, C], per-pixel channel arg-max [N, H*W])} on
        the CPU: the arg-max positions the HIP attention kernels used (the network's other kinks; round 4)."""
        out = {}
        for name, p in self.model.named_parameters():
            o = self.cbam.get(id(p))
            if o is not None:
                out[name] = (o[0].cpu(), o[1].cpu())
        return out

    def masks(self):
        """{conv weight parameter name: bool mask [N, >=C, H, W] on the CPU} (channel padding of the NHWC buffers kept;
        oracle_with_masks crops it)."""
        out = {}
        for name, p in self.model.named_parameters():
            o = self.cap.get(id(p))
            if o is not None:
                out[name] = (o > 0).permute(0, 3, 1, 2).cpu()
        return out


def oracle_with_masks(fn, masks, cbam=None):
    """Run `fn()` (an oracle forward + backward) with the ReLUs that follow the listed convolutions forced to `masks` and, given
    `cbam` (kink_matched.cbam_indices()), the attention blocks' arg-max positions forced as well."""
    from oracle import ref_cpu as R
    old = R.KINK_MASKS
    R.KINK_MASKS = {k: m for k, m in masks.items()}
    old_relu = R._relu
    old_cbam = R.CBAM_INDICES
    R.CBAM_INDICES = cbam

    def _relu(y, key):
        m = masks.get(key)
        if m is None:
            return old_relu(y, key)
        return y * m[:, :y.shape[1]].to(y.dtype)
    R._relu = _relu
    try:
        return fn()
    finally:
        R._relu = old_relu
        R.KINK_MASKS = old
        R.CBAM_INDICES = old_cbam
