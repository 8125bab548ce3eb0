"""Losses of the hot path (/root/reference training/loss.py) on the HIP kernels.

`DehazingLoss` = 1.0*L1 + 0.1*content(VGG16 taps) + 0.1*LPIPS(alex); `JointLoss` adds 0.2*CE
(+ 0.5 * detection, always 0 in the drivers).  Forward signatures and returned dict keys follow
loss.py:125-162 and :179-224.  The third-party feature networks (VGG16 / LPIPS-AlexNet) are built here
with torchvision / lpips parameter names; their pretrained weights cannot be downloaded offline, so they
start randomly initialised (load real ones with load_state_dict).
"""
from __future__ import annotations

import warnings
from typing import Optional

import torch
import torch.nn as nn

from . import _hip as H

_WARNED = set()


def _warn_random_extractor(what: str, how: str):
    """The reference builds these feature extractors with pretrained weights (loss.py:20-22,91); they cannot be
    downloaded here, so the networks start randomly initialised -- say so once, loudly, instead of silently optimising
    0.1 * content + 0.1 * perceptual against random features."""
    if what not in _WARNED:
        _WARNED.add(what)
        warnings.warn(f"{what}: pretrained weights are not available offline -- the feature extractor is RANDOMLY "
                      f"initialised until real weights are loaded ({how}).", stacklevel=3)


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


VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class _ContentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, pred, target, *params):
        from .engine import Engine
        rec = ctx.needs_input_grad[1]
        taps = module.tap_indices
        scale = 1.0 / len(taps)
        # target features: no gradient
        eng_t = Engine(pred.device, False)
        eng_t.wino43 = "dgrad"
        feats_t = module._features(eng_t, target.contiguous(), taps, {})
        eng = Engine(pred.device, rec)
        eng.wino43 = "dgrad"
        holder = {}
        feats_p = module._features(eng, pred.contiguous(), taps, holder)
        vals = []
        for fp, ft in zip(feats_p, feats_t):
            eng.mse(fp, ft, scale, vals)
        total = vals[0].clone()
        for v in vals[1:]:
            H.call("adh_add_inplace", total.data_ptr(), v.data_ptr(), 1)
        ctx.eng, ctx.holder = eng, holder
        return total.reshape(())

    @staticmethod
    def backward(ctx, g):
        eng = ctx.eng
        eng.upstream["g"] = g.contiguous().reshape(1)
        eng.backward()
        gx = ctx.holder.get("gx")
        ctx.eng = None
        return (None, gx, None) + tuple(None for _ in range(len(ctx.needs_input_grad) - 3))


class ContentLoss(nn.Module):
    """Content loss on VGG features (loss.py:7-84): ImageNet-normalise, then mean over the taps
    ['relu2_2','relu3_3','relu4_3'] of mse(features[:idx+1](x), features[:idx+1](target)) with the reference's
    layer_mapping (idx 9/16/23 = the MaxPool layers after those relus in torchvision's vgg16.features).
    One pass per image with three taps replaces the reference's six prefix passes (identical values).
    Weights: `model.{idx}.weight/bias` (torchvision naming); pretrained weights cannot be downloaded here,
    load them with load_state_dict.  The extractor is frozen (loss.py:27-28)."""

    LAYER_MAPPING = {"relu1_1": 2, "relu1_2": 4, "relu2_1": 7, "relu2_2": 9, "relu3_1": 12, "relu3_2": 14, "relu3_3": 16,
                     "relu4_1": 19, "relu4_2": 21, "relu4_3": 23, "relu5_1": 26, "relu5_2": 28, "relu5_3": 30}

    def __init__(self, pretrained_model="vgg16", content_layers=None):
        super().__init__()
        from .layers import ConvParams, Seq
        if content_layers is None:
            content_layers = ["relu2_2", "relu3_3", "relu4_3"]
        self.content_layers = content_layers
        if pretrained_model == "vgg16":
            cfg = VGG16_CFG
        elif pretrained_model == "vgg19":
            cfg = VGG19_CFG
        else:
            raise ValueError(f"Unsupported model: {pretrained_model}")
        items, idx, cin = [], 0, 3
        self.plan = []   # ('conv', idx) | ('relu', idx) | ('pool', idx)
        for v in cfg:
            if v == "M":
                self.plan.append(("pool", idx))
                idx += 1
            else:
                items.append((idx, ConvParams(cin, v, 3, bias=True)))
                self.plan.append(("conv", idx))
                self.plan.append(("relu", idx + 1))
                cin = v
                idx += 2
        self.model = Seq(items)
        for p in self.model.parameters():
            p.requires_grad = False
        self.tap_indices = [self.LAYER_MAPPING[n] for n in content_layers]
        self.weights_loaded = False
        _warn_random_extractor(f"ContentLoss({pretrained_model})", "ContentLoss.load_torchvision_vgg(path_or_state_dict)")

    def load_torchvision_vgg(self, src):
        """Load a stock torchvision VGG checkpoint (`features.{idx}.weight/bias`, classifier keys ignored) or a state_dict
        in this module's own naming (`model.{idx}.*`).  `src`: path or dict."""
        sd = torch.load(src, map_location="cpu") if isinstance(src, (str, bytes)) or hasattr(src, "__fspath__") else src
        if "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
        out = {}
        for k, v in sd.items():
            if k.startswith("features."):
                out["model." + k[len("features."):]] = v
            elif k.startswith("model."):
                out[k] = v
        mine = self.state_dict()
        missing = [k for k in mine if k not in out]
        if missing:
            raise KeyError(f"VGG checkpoint lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
        self.load_state_dict({k: out[k] for k in mine}, strict=True)
        self.weights_loaded = True
        from .engine import invalidate_weight_cache
        invalidate_weight_cache()
        return self

    def _features(self, eng, img, taps, holder):
        h = eng.image_normalize_to_nhwc8(img, IMAGENET_MEAN, IMAGENET_STD, holder)
        feats, last = [], max(taps)
        i = 0
        plan = self.plan
        while i < len(plan):
            kind, idx = plan[i]
            if idx > last:
                break
            if kind == "conv":
                p = self.model.at(idx)
                # features[:idx+1] ending ON a conv index (never the case for the reference's mapping) would
                # exclude the relu; the mapping's indices are relu / pool outputs, so conv+relu fuse safely
                fuse_relu = (idx + 1) <= last and idx not in taps
                h = eng.conv(h, p.weight, p.bias, None, k=3, stride=1, pad=1, relu=fuse_relu)
                if idx in taps:
                    feats.append(h)
                i += 2 if fuse_relu else 1
                if fuse_relu and (idx + 1) in taps:
                    feats.append(h)
                continue
            if kind == "pool":
                h = eng.maxpool(h, 2)
                if idx in taps:
                    feats.append(h)
            i += 1
        return feats

    def forward(self, x, target):
        H.require_cuda(x, "prediction")
        H.require_cuda(target, "target")
        params = list(self.model.parameters())
        return _ContentFn.apply(self, x, target, *params)


class _ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-0.030, -0.088, -0.188]).view(1, 3, 1, 1))
        self.register_buffer("scale", torch.tensor([0.458, 0.448, 0.450]).view(1, 3, 1, 1))


class _NetLin(nn.Module):
    """lpips NetLinLayer: Dropout (identity in eval) + 1x1 conv without bias at index 1."""

    def __init__(self, cin):
        super().__init__()
        from .layers import ConvParams, Seq
        self.model = Seq([(1, ConvParams(cin, 1, 1, bias=False))])


class _AlexSlices(nn.Module):
    def __init__(self):
        super().__init__()
        from .layers import ConvParams, Seq
        self.slice1 = Seq([(0, ConvParams(3, 64, 11, bias=True))])
        self.slice2 = Seq([(3, ConvParams(64, 192, 5, bias=True))])
        self.slice3 = Seq([(6, ConvParams(192, 384, 3, bias=True))])
        self.slice4 = Seq([(8, ConvParams(384, 256, 3, bias=True))])
        self.slice5 = Seq([(10, ConvParams(256, 256, 3, bias=True))])


class _LPIPSAlex(nn.Module):
    """lpips.LPIPS(net='alex') parameter layout: net.slice{1..5}.{idx}.*, scaling_layer.{shift,scale},
    lin{k}.model.1.weight and lins.{k}.model.1.weight (the same tensors under both names, as lpips has)."""

    CHNS = (64, 192, 384, 256, 256)

    def __init__(self):
        super().__init__()
        self.scaling_layer = _ScalingLayer()
        self.net = _AlexSlices()
        lins = []
        for k, c in enumerate(self.CHNS):
            lin = _NetLin(c)
            setattr(self, f"lin{k}", lin)
            lins.append(lin)
        self.lins = nn.ModuleList(lins)
        for p in self.parameters():
            p.requires_grad = False


class _LPIPSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, record, pred, target):
        from .engine import Engine
        eng_t = Engine(pred.device, False)
        eng_t.wino43 = "dgrad"
        feats_t = module._features(eng_t, target.contiguous(), {})
        eng = Engine(pred.device, record)
        eng.wino43 = "dgrad"
        holder = {}
        feats_p = module._features(eng, pred.contiguous(), holder)
        N = pred.shape[0]
        val = torch.empty(N, device=pred.device, dtype=torch.float32)
        lin = module.loss_fn
        taps = []
        for k, (fp, ft) in enumerate(zip(feats_p, feats_t)):
            w = getattr(lin, f"lin{k}").model.at(1).weight
            HW, Cc = fp.Hh * fp.Ww, fp.C
            nblk = H.value("adh_lpips_layer_num_blocks", HW)
            partial = torch.empty((N, nblk), device=pred.device, dtype=torch.float32)
            H.call("adh_lpips_layer", fp.t.data_ptr(), ft.t.data_ptr(), w.data_ptr(), N, HW, Cc, partial.data_ptr(), nblk)
            H.call("adh_rows_sum", partial.data_ptr(), N, nblk, 1.0 / HW, val.data_ptr(), 1 if k > 0 else 0)
            taps.append((fp, ft, w))
        ctx.eng, ctx.holder, ctx.taps, ctx.N = eng, holder, taps, N
        return val.view(N, 1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        eng = ctx.eng
        gv = g.contiguous().view(-1)
        for fp, ft, w in ctx.taps:
            if not fp.needs_grad:
                continue
            ga = torch.empty_like(fp.t)
            H.call("adh_lpips_layer_bwd", fp.t.data_ptr(), ft.t.data_ptr(), w.data_ptr(), gv.data_ptr(), ctx.N,
                   fp.Hh * fp.Ww, fp.C, ga.data_ptr())
            eng.accum(fp, ga)
        eng.backward()
        gx = ctx.holder.get("gx")
        ctx.eng = None
        return None, None, gx, None


class PerceptualLoss(nn.Module):
    """Perceptual loss based on LPIPS (loss.py:86-108): lpips.LPIPS(net='alex') on 2x-1 inputs -> [N,1,1,1].
    AlexNet's 11x11 stride-4 stem runs as a 3x3 stride-1 MFMA conv on a space-to-depth(4) image.  Weights use
    lpips' own key names (load a real checkpoint with load_state_dict); extractor and lin layers are frozen."""

    def __init__(self, net="alex"):
        super().__init__()
        if net != "alex":
            raise ValueError(f"Unsupported LPIPS net: {net}")
        self.loss_fn = _LPIPSAlex()
        self.weights_loaded = False
        _warn_random_extractor("PerceptualLoss(LPIPS-alex)", "PerceptualLoss.load_lpips(alexnet_sd, lpips_linear_sd)")

    def load_lpips(self, alexnet, linear=None):
        """Load real LPIPS weights: `alexnet` = torchvision alexnet checkpoint (`features.{idx}.*`) or lpips' own
        `net.slice{k}.{idx}.*` keys; `linear` = lpips' v0.1 `alex.pth` (`lin{k}.model.1.weight`), optional if `alexnet`
        already is a full lpips state_dict.  Paths or dicts."""
        def _sd(x):
            return torch.load(x, map_location="cpu") if isinstance(x, (str, bytes)) or hasattr(x, "__fspath__") else x
        a = _sd(alexnet)
        slice_of = {0: 1, 3: 2, 6: 3, 8: 4, 10: 5}
        out = {}
        for k, v in a.items():
            if k.startswith("features."):
                idx, leaf = k[len("features."):].split(".", 1)
                if int(idx) in slice_of:
                    out[f"loss_fn.net.slice{slice_of[int(idx)]}.{idx}.{leaf}"] = v
            elif k.startswith(("net.", "lin", "scaling_layer.")):
                out["loss_fn." + k] = v
            elif k.startswith("loss_fn."):
                out[k] = v
        if linear is not None:
            for k, v in _sd(linear).items():
                if k.startswith("lin"):
                    out["loss_fn." + k] = v
                    out["loss_fn.lins." + k[3] + k[4:]] = v
        mine = self.state_dict()
        for k in mine:
            if k not in out and k.startswith("loss_fn.lins."):
                alt = "loss_fn.lin" + k[len("loss_fn.lins."):]
                if alt in out:
                    out[k] = out[alt]
            if k not in out and "scaling_layer" in k:
                out[k] = mine[k]
        missing = [k for k in mine if k not in out]
        if missing:
            raise KeyError(f"LPIPS checkpoint lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
        self.load_state_dict({k: out[k] for k in mine}, strict=True)
        self.weights_loaded = True
        from .engine import invalidate_weight_cache
        invalidate_weight_cache()
        return self

    def _features(self, eng, img, holder):
        import ctypes as C
        from .engine import Act
        lp = self.loss_fn
        N, _, Hh, Ww = img.shape
        OH, OW = (Hh + 4 - 11) // 4 + 1, (Ww + 4 - 11) // 4 + 1
        shift = lp.scaling_layer.shift.view(-1).tolist()
        scale = lp.scaling_layer.scale.view(-1).tolist()
        a = [2.0 / s_ for s_ in scale]                       # ((2x-1) - shift)/scale = x*a + b
        b = [(-1.0 - sh) / s_ for sh, s_ in zip(shift, scale)]
        a3, b3 = (C.c_float * 3)(*a), (C.c_float * 3)(*b)
        s2d = eng._f(N, OH + 2, OW + 2, 48)
        H.call("adh_lpips_s2d", img.data_ptr(), N, Hh, Ww, a3, b3, OH + 2, OW + 2, s2d.data_ptr())
        x = Act(s2d, 48, needs_grad=eng.record)
        if eng.record:
            def bwd():
                g = x.grad
                x.grad = None
                if g is None:
                    return
                gx = torch.empty_like(img)
                H.call("adh_lpips_s2d_bwd", g.contiguous().data_ptr(), N, Hh, Ww, a3, OH + 2, OW + 2, gx.data_ptr())
                holder["gx"] = gx
            eng.tape.append(bwd)
        # conv1 11x11 s4 p2 as a 3x3 s1 conv over 48 = (by,bx,c) channels: W'[co][(by*4+bx)*3+c][ty][tx] = W[co][c][4ty+by][4tx+bx]
        c1 = lp.net.slice1.at(0)
        w = torch.zeros(64, 3, 12, 12, device=img.device)
        w[:, :, :11, :11] = c1.weight
        w1 = w.view(64, 3, 3, 4, 3, 4).permute(0, 3, 5, 1, 2, 4).reshape(64, 48, 3, 3).contiguous()
        feats = []
        h = eng.conv(x, w1, c1.bias, None, k=3, stride=1, pad=0, relu=True)
        feats.append(h)
        h = eng.maxpool(h, 3, 2, 0)
        c2 = lp.net.slice2.at(3)
        h = eng.conv(h, c2.weight, c2.bias, None, k=5, stride=1, pad=2, relu=True)
        feats.append(h)
        h = eng.maxpool(h, 3, 2, 0)
        for sl, idx in ((lp.net.slice3, 6), (lp.net.slice4, 8), (lp.net.slice5, 10)):
            c = sl.at(idx)
            h = eng.conv(h, c.weight, c.bias, None, k=3, stride=1, pad=1, relu=True)
            feats.append(h)
        return feats

    def forward(self, x, target):
        H.require_cuda(x, "prediction")
        H.require_cuda(target, "target")
        record = torch.is_grad_enabled() and x.requires_grad
        return _LPIPSFn.apply(self, record, x, target)


class DehazingLoss(nn.Module):
    """Combined loss for image dehazing (loss.py:110-162)."""

    def __init__(self, lambda_l1=1.0, lambda_content=0.1, lambda_perceptual=0.1, content=True, perceptual=True):
        """`content` / `perceptual` = False drop the VGG16 / LPIPS terms (their pretrained weights cannot be
        downloaded here: the extractors are randomly initialised until a checkpoint is loaded)."""
        super().__init__()
        self.lambda_l1, self.lambda_content, self.lambda_perceptual = lambda_l1, lambda_content, lambda_perceptual
        self.content_loss = ContentLoss() if content else None
        self.perceptual_loss = PerceptualLoss() if perceptual else None

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
