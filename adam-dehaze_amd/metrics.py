"""Evaluation harness on device: per-image PSNR / SSIM / LPIPS for NCHW batches without per-image D2H copies.

Mirrors /root/reference evaluation/metrics.py:13-124 (`calculate_image_metrics`, `ImageQualityMetrics` with the same
method names and results-JSON schema) and the validation loops of training/train_joint.py:187-229 and
training/train_dehazing.py:110-166.  The reference copies every image to the host and calls skimage
(`peak_signal_noise_ratio(target, pred, data_range=1.0)`; `structural_similarity` on the channel-mean grayscale with
its defaults) -- here both are HIP reduction kernels over the whole batch (csrc/train_io.hip), LPIPS goes through the
LPIPS-alex kernels of the perceptual loss, and values stay on the device until `compute_averages()` reads them once.
"""
from __future__ import annotations

import json
import os
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import torch

from . import _hip as H
from .detection_metrics import DetectionMetrics   # noqa: F401  (evaluation/metrics.py:126-270)

CATEGORY_BY_LABEL = {0: "low_intensity", 1: "medium_intensity", 2: "high_intensity"}   # evaluate.py:160-166


def _check_pair(pred: torch.Tensor, target: torch.Tensor):
    H.require_cuda(pred, "predicted images")
    H.require_cuda(target, "target images")
    if pred.dim() != 4 or pred.shape[1] != 3 or pred.shape != target.shape:
        raise RuntimeError(f"expected two [N,3,H,W] batches, got {tuple(pred.shape)} and {tuple(target.shape)}")
    return pred.contiguous(), target.contiguous()


def psnr_batch(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Per-image PSNR [N] (device tensor): 10 log10(R^2 / mean((t-p)^2)), mean in float64 (metrics.py:27)."""
    pred, target = _check_pair(pred, target)
    N = pred.shape[0]
    per = pred[0].numel()
    nblk = H.value("adh_psnr_num_blocks", per)
    partial = torch.empty(N * nblk, device=pred.device, dtype=torch.float64)
    out = torch.empty(N, device=pred.device, dtype=torch.float32)
    H.call("adh_psnr", pred.data_ptr(), target.data_ptr(), N, per, float(data_range), partial.data_ptr(), nblk, None,
           out.data_ptr())
    return out


def ssim_batch(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Per-image SSIM [N] of the channel-mean grayscale images with skimage's defaults (metrics.py:29-32)."""
    pred, target = _check_pair(pred, target)
    N, _, Hh, Ww = pred.shape
    if Hh < 7 or Ww < 7:
        raise ValueError("win_size exceeds image extent (images must be at least 7x7)")   # skimage's message
    nblk = H.value("adh_ssim_num_blocks", Hh, Ww)
    partial = torch.empty(N * nblk, device=pred.device, dtype=torch.float64)
    out = torch.empty(N, device=pred.device, dtype=torch.float32)
    H.call("adh_ssim_gray", pred.data_ptr(), target.data_ptr(), N, Hh, Ww, float(data_range), partial.data_ptr(), nblk,
           out.data_ptr())
    return out


def calculate_image_metrics(pred: torch.Tensor, target: torch.Tensor) -> Dict[str, torch.Tensor]:
    """metrics.py:13-36 for a batch (or one [3,H,W] image): {'psnr': [N], 'ssim': [N]} device tensors."""
    if pred.dim() == 3:
        pred, target = pred.unsqueeze(0), target.unsqueeze(0)
    return {"psnr": psnr_batch(pred, target), "ssim": ssim_batch(pred, target)}


class ImageQualityMetrics:
    """metrics.py:38-124: accumulate per-image metrics per category, average, print, save as JSON
    ({category: {psnr, ssim, lpips, samples}})."""

    def __init__(self, device="cuda", lpips_fn=None, use_lpips: bool = True):
        self.device = torch.device(device)
        self.lpips_fn = lpips_fn
        if lpips_fn is None and use_lpips:
            from .loss import PerceptualLoss
            self.lpips_fn = PerceptualLoss().to(self.device)
        self._chunks: Dict[str, List[Dict[str, torch.Tensor]]] = defaultdict(list)

    @property
    def results(self) -> Dict[str, List[Dict[str, float]]]:
        """The reference's `results` attribute: {category: [per-sample metric dicts]} (one host read-back)."""
        out: Dict[str, List[Dict[str, float]]] = defaultdict(list)
        for cat, chunks in self._chunks.items():
            for ch in chunks:
                host = {k: v.detach().cpu().tolist() for k, v in ch.items()}
                n = len(next(iter(host.values())))
                for i in range(n):
                    out[cat].append({k: host[k][i] for k in host})
        return out

    def add_batch(self, pred: torch.Tensor, target: torch.Tensor, categories: Optional[Sequence[str]] = None):
        """Whole [N,3,H,W] batches at once; `categories`: one name per image (or None -> 'all').  No host sync."""
        with torch.no_grad():
            m = calculate_image_metrics(pred, target)
            if self.lpips_fn is not None:
                # loss.PerceptualLoss maps [0,1] -> [-1,1] itself, i.e. it IS lpips.LPIPS(net='alex')(2p-1, 2t-1) of
                # metrics.py:69-75
                m["lpips"] = self.lpips_fn(pred.contiguous(), target.contiguous()).reshape(-1)
        if categories is None:
            self._chunks["all"].append(m)
            return
        cats = list(categories)
        for cat in sorted(set(cats)):
            sel = torch.tensor([i for i, c in enumerate(cats) if c == cat], device=pred.device)
            self._chunks[cat].append({k: v.index_select(0, sel) for k, v in m.items()})

    def add_sample(self, pred: torch.Tensor, target: torch.Tensor, category=None):
        """metrics.py:47-84 signature: one [3,H,W] pair."""
        self.add_batch(pred.unsqueeze(0), target.unsqueeze(0), None if not category else [category])

    def compute_averages(self):
        avg = {}
        for cat, chunks in self._chunks.items():
            if not chunks:
                continue
            avg[cat] = {}
            for name in chunks[0]:
                allv = torch.cat([c[name].double() for c in chunks])
                avg[cat][name] = float(allv.mean())
            avg[cat]["samples"] = int(sum(len(next(iter(c.values()))) for c in chunks))
        return avg

    def print_results(self):
        avg = self.compute_averages()
        print("Image Quality Evaluation Results:")
        for cat, ms in sorted(avg.items()):
            print(f"\n{cat.upper()} ({ms['samples']} samples):")
            for k, v in ms.items():
                if k != "samples":
                    print(f"  {k.upper()}: {v:.4f}")
        return avg

    def save_results(self, output_path):
        d = os.path.dirname(output_path)
        if d:
            os.makedirs(d, exist_ok=True)
        with open(output_path, "w") as f:
            json.dump(self.compute_averages(), f, indent=2)
        print(f"Results saved to {output_path}")
