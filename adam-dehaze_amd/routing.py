"""Routing between the dehazing branches (/root/reference models/routing.py) on the HIP kernels.

Class names, constructor arguments, forward signatures and returned dict keys are the reference's:
  HardRouter.forward(x, intensity=None)          -> (outputs, {'intensity','low_mask','medium_mask','high_mask'})
  SoftRouter.forward(x, classifier_logits=None)  -> (blended, {'weights','individual_outputs'})
  GatedRouter.forward(x)                         -> (final,   {'gate_weights','individual_outputs'})
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _hip as H
from .engine import Act, Engine
from .layers import Seq

_NAMES = ("low", "medium", "high")


class _SoftBlendFn(torch.autograd.Function):
    """out = sum_i softmax(logits/T)[:, i] * branch_i   (routing.py:110-127), blend and both backward
    products (g_branch_i = w_i * g, g_logits through the softmax) in HIP kernels."""

    @staticmethod
    def forward(ctx, logits, temperature, present, o0, o1, o2):
        ctx.set_materialize_grads(False)
        H.require_cuda(logits, "classifier logits")
        logits = logits.contiguous()
        N = logits.shape[0]
        w = torch.empty_like(logits)
        H.call("adh_softmax3", logits.data_ptr(), float(temperature), N, w.data_ptr())
        w_eff = w
        if not all(present):   # a missing branch contributes nothing (routing.py:116,124 `if name in ...`)
            w_eff = w * torch.tensor([1.0 if p else 0.0 for p in present], device=w.device)
        outs = [o.contiguous() for o in (o0, o1, o2)]
        per = outs[0].numel() // N
        out = torch.empty_like(outs[0])
        H.call("adh_soft_blend", w_eff.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), N, per,
               out.data_ptr())
        ctx.save_for_backward(w, w_eff, *outs)
        ctx.temperature, ctx.present, ctx.per = float(temperature), present, per
        return out, w

    @staticmethod
    def backward(ctx, g, g_w_ext):
        w, w_eff, o0, o1, o2 = ctx.saved_tensors
        N, per = w.shape[0], ctx.per
        if g is None:
            g = torch.zeros_like(o0)
        g = g.contiguous()
        need = ctx.needs_input_grad
        gs = [torch.empty_like(o0) if (need[3 + i] and ctx.present[i]) else None for i in range(3)]
        nblk = max(1, min((per // 4 + 255) // 256, 256))
        gw_partial = torch.empty((N, nblk, 3), device=w.device, dtype=torch.float32)
        H.call("adh_soft_blend_bwd", w_eff.data_ptr(), g.data_ptr(), o0.data_ptr(), o1.data_ptr(), o2.data_ptr(), N, per,
               H.ptr(gs[0]), H.ptr(gs[1]), H.ptr(gs[2]), gw_partial.data_ptr(), nblk)
        g_logits = None
        if need[0]:
            if not all(ctx.present):
                gw_partial = gw_partial * torch.tensor([1.0 if p else 0.0 for p in ctx.present], device=w.device)
            if g_w_ext is not None:   # gradient arriving through the returned weights tensor
                gw_partial = gw_partial.clone()
                gw_partial[:, 0, :] += g_w_ext
            g_logits = torch.empty_like(w)
            H.call("adh_softmax3_bwd", w.data_ptr(), gw_partial.contiguous().data_ptr(), nblk, ctx.temperature, N,
                   g_logits.data_ptr())
        return g_logits, None, None, gs[0], gs[1], gs[2]


def _blend(weights_logits, temperature, outputs, x):
    present = tuple(n in outputs for n in _NAMES)
    any_out = next(iter(outputs.values()))
    o = [outputs.get(n, any_out) for n in _NAMES]
    return _SoftBlendFn.apply(weights_logits, temperature, present, o[0], o[1], o[2])


class _HardAssembleFn(torch.autograd.Function):
    """outputs[mask_c] = branch_c(x[mask_c]) for every class (routing.py:53-61) with the scatter and
    its adjoint (a gather of the incoming gradient rows) on the HIP copy kernels."""

    @staticmethod
    def forward(ctx, base, sel, meta, per, *subs):
        out = base   # freshly zero-filled by the caller
        for (cls, k), sub in zip(meta, subs):
            H.call("adh_scatter_images", sub.contiguous().data_ptr(), sel[cls].data_ptr(), k, per, out.data_ptr())
        ctx.sel, ctx.meta, ctx.per = sel, meta, per
        ctx.shapes = [tuple(s.shape) for s in subs]
        ctx.mark_dirty(base)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        grads = []
        for (cls, k), shp in zip(ctx.meta, ctx.shapes):
            gs = torch.empty(shp, device=g.device, dtype=g.dtype)
            H.call("adh_gather_images", g.data_ptr(), ctx.sel[cls].data_ptr(), k, ctx.per, gs.data_ptr())
            grads.append(gs)
        return (None, None, None, None, *grads)


class HardRouter(nn.Module):
    """Hard routing: every image goes through the one branch its class selects (routing.py:5-68)."""

    def __init__(self, models, classifier=None, device="cuda"):
        super().__init__()
        self.models = nn.ModuleDict(models)
        self.classifier = classifier
        self.device = device

    def forward(self, x, intensity=None):
        H.require_cuda(x, "input image batch")
        x = x.contiguous()
        N = x.shape[0]
        outputs = torch.zeros_like(x)
        if intensity is None and self.classifier is not None:
            with torch.no_grad():
                logits, _ = self.classifier(x)
                logits = logits.contiguous()
                intensity = torch.empty(N, device=x.device, dtype=torch.int64)
                H.call("adh_argmax3", logits.data_ptr(), N, intensity.data_ptr())   # first index on ties
        masks = {"low": intensity == 0, "medium": intensity == 1, "high": intensity == 2}
        if intensity.dim() != 1:
            # The reference compares whatever it is given with 0/1/2 (routing.py:46-50); logits passed by
            # mistake (as its drivers do) select nothing and the result is all zeros.
            if any(bool(m.any()) for m in masks.values()):
                raise RuntimeError("HardRouter: `intensity` must be a 1-D tensor of class indices")
            return outputs, {"intensity": intensity, "low_mask": masks["low"], "medium_mask": masks["medium"],
                             "high_mask": masks["high"]}
        # device-side compaction; ONE host read-back of the three counts (the sub-batch sizes are needed
        # on the host to size the branch launches), instead of the reference's torch.any per class
        idx64 = intensity.to(torch.int64).contiguous()
        sel = torch.empty((3, N), device=x.device, dtype=torch.int32)
        counts = torch.empty(3, device=x.device, dtype=torch.int32)
        H.call("adh_route_compact", idx64.data_ptr(), N, sel.data_ptr(), counts.data_ptr())
        counts_h = counts.cpu().tolist()
        per = x.numel() // N
        subs, meta = [], []
        for cls, name in enumerate(_NAMES):
            if name not in self.models or counts_h[cls] == 0:
                continue
            k = counts_h[cls]
            sub = torch.empty((k,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
            H.call("adh_gather_images", x.data_ptr(), sel[cls].data_ptr(), k, per, sub.data_ptr())
            subs.append(self.models[name](sub))
            meta.append((cls, k))
        if subs:
            outputs = _HardAssembleFn.apply(outputs, sel, tuple(meta), per, *subs)
        return outputs, {"intensity": intensity, "low_mask": masks["low"], "medium_mask": masks["medium"],
                         "high_mask": masks["high"]}


class SoftRouter(nn.Module):
    """Soft routing: every branch runs on the whole batch, outputs blended by softmax(logits/T)
    (routing.py:70-132); gradients reach the classifier through the weights."""

    def __init__(self, models, classifier=None, temperature=1.0, device="cuda"):
        super().__init__()
        self.models = nn.ModuleDict(models)
        self.classifier = classifier
        self.temperature = temperature
        self.device = device

    def forward(self, x, classifier_logits=None):
        H.require_cuda(x, "input image batch")
        if classifier_logits is None and self.classifier is not None:
            logits, _ = self.classifier(x)
        else:
            logits = classifier_logits
        outputs = {}
        for name in _NAMES:
            if name in self.models:
                outputs[name] = self.models[name](x)
        blended, weights = _blend(logits, self.temperature, outputs, x)
        return blended, {"weights": weights, "individual_outputs": outputs}


class _LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) on the MFMA conv kernel (a Linear is a 1x1 conv over a 1x1 image)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        N, fin = x.shape
        eng = Engine(x.device, record=any(ctx.needs_input_grad[:3]))
        fin8 = (fin + 7) // 8 * 8
        xt = x.contiguous() if fin8 == fin else torch.nn.functional.pad(x, (0, fin8 - fin))
        xa = Act(xt.view(N, 1, 1, fin8), fin, needs_grad=ctx.needs_input_grad[0])
        w4 = w.view(w.shape[0], w.shape[1], 1, 1)
        eng.alias[id(w4)] = id(w4)   # a view made under no_grad: mark it as a trainable weight for the engine
        o = eng.conv(xa, w4, b, None, kind="conv", k=1, stride=1, pad=0, relu=relu)
        ctx.eng, ctx.xa, ctx.o, ctx.w4, ctx.b = eng, xa, o, w4, b
        ctx.fin, ctx.fout = fin, w.shape[0]
        return o.t.view(N, -1)[:, :w.shape[0]]

    @staticmethod
    def backward(ctx, g):
        eng, o = ctx.eng, ctx.o
        gp = torch.zeros(o.t.shape, device=g.device, dtype=torch.float32)
        gp.view(g.shape[0], -1)[:, :ctx.fout] = g
        o.grad = gp
        eng.backward()
        gx = ctx.xa.grad.view(g.shape[0], -1)[:, :ctx.fin] if ctx.xa.grad is not None else None
        gw = eng.param_grads.get(id(ctx.w4))
        gb = eng.param_grads.get(id(ctx.b)) if ctx.b is not None else None
        return gx, (gw.view(ctx.fout, ctx.fin) if gw is not None else None), gb, None


def linear(x, w, b, relu=False):
    return _LinearFn.apply(x, w, b, relu)


class _LinearParams(nn.Module):
    """nn.Linear parameter container (same init: kaiming_uniform(a=sqrt(5)) + bias bound)."""

    def __init__(self, fin, fout):
        super().__init__()
        import math
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(fin)
        nn.init.uniform_(self.bias, -bound, bound)


def dropout(x, p, training):
    """nn.Dropout: identity in eval; in training the mask comes from torch's generator (the
    reference's dropout is RNG-dependent too) and is applied by the HIP multiply kernel."""
    if not training or p == 0.0:
        return x
    mask = (torch.rand_like(x) >= p).to(x.dtype) / (1.0 - p)
    return _MulFn.apply(x, mask)


class _MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, m):
        x, m = x.contiguous(), m.contiguous()
        out = torch.empty_like(x)
        H.call("adh_mul", out.data_ptr(), x.data_ptr(), m.data_ptr(), x.numel())
        ctx.save_for_backward(m)
        return out

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(g)
        H.call("adh_mul", out.data_ptr(), g.data_ptr(), m.data_ptr(), g.numel())
        return out, None


class GatedRouter(nn.Module):
    """Gated routing: a small MLP on the classifier features produces the blending weights
    (routing.py:134-226).  feature_dim is 512 as in the reference (ResNet18/34 backbones)."""

    def __init__(self, models, classifier=None, feature_dim=512, device="cuda"):
        super().__init__()
        self.models = nn.ModuleDict(models)
        self.classifier = classifier
        self.device = device
        # indices follow the reference's nn.Sequential: Linear, ReLU, Dropout, Linear, ReLU, Linear, Softmax
        self.gate_network = Seq([(0, _LinearParams(feature_dim, 256)), (3, _LinearParams(256, 128)),
                                 (5, _LinearParams(128, len(models)))])
        self.use_feature_fusion = False

    def forward(self, x):
        H.require_cuda(x, "input image batch")
        N = x.shape[0]
        if self.classifier is not None:
            logits, features = self.classifier(x)
            g = self.gate_network
            h = linear(features, g.at(0).weight, g.at(0).bias, relu=True)
            h = dropout(h, 0.3, self.training)
            h = linear(h, g.at(3).weight, g.at(3).bias, relu=True)
            gate_logits = linear(h, g.at(5).weight, g.at(5).bias)
        else:
            gate_logits = torch.zeros(N, len(self.models), device=x.device)
        outputs = {}
        for name in _NAMES:
            if name in self.models:
                outputs[name] = self.models[name](x)
        if gate_logits.shape[1] != 3:
            raise RuntimeError("GatedRouter kernels expect the reference's three branches")
        final, gate_weights = _blend(gate_logits, 1.0, outputs, x)   # Softmax(dim=1) == softmax(logits/1)
        return final, {"gate_weights": gate_weights, "individual_outputs": outputs}


def create_router(models, classifier, config):
    """routing.py:228-252."""
    routing_type = config["routing"]["type"]
    if routing_type == "hard":
        return HardRouter(models=models, classifier=classifier, device=config["device"])
    elif routing_type == "soft":
        return SoftRouter(models=models, classifier=classifier, temperature=config["routing"]["temperature"],
                          device=config["device"])
    elif routing_type == "gated":
        return GatedRouter(models=models, classifier=classifier, device=config["device"])
    else:
        raise ValueError(f"Unsupported routing type: {routing_type}")
