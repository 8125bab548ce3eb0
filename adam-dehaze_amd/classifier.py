"""Fog-intensity classifier "HDEN" (/root/reference models/classifier.py) on the HIP engine.

`FogIntensityClassifier(model_name, num_classes=3, pretrained)` keeps the reference's constructor,
`forward(x) -> (logits, features)`, `extract_features`, `feature_dim`, and torchvision-compatible
state_dict keys (`backbone.*`, `classifier.{1,4}.*`).  Backbones built here:
  resnet18 / resnet34 / resnet50  (reference default: resnet18, classifier.py:24-36; resnet50: Bottleneck
                        [3, 4, 6, 3], feature_dim 2048, classifier.py:31-33)  -- forward and backward
  densenet121          (the north-star's HDEN backbone; the reference itself raises for this name,
                        classifier.py:69: build-side extension)             -- forward (eval) only
Every other name raises ValueError like the reference does for unknown backbones (efficientnet /
mobilenet need timm / torchvision, which are not part of this build).  `pretrained=True` cannot
download weights offline: a warning is printed and torchvision's random initialisation is used;
real checkpoints load through `load_state_dict` (key names match torchvision).
PARITY UNPINNED for the backbones (no torchvision here): checked against the CPU oracle only.
"""
from __future__ import annotations

import math
import warnings

import torch
import torch.nn as nn

from . import _hip as H
from .engine import Act, Engine
from .layers import BNParams, ConvParams, Seq

DENSENET121_BLOCKS = (6, 12, 24, 16)


def _tv_conv(cin, cout, k, mode="fan_out"):
    """Conv2d(bias=False) container with torchvision's init (kaiming_normal)."""
    p = ConvParams(cin, cout, k, bias=False)
    if mode == "fan_out":
        nn.init.kaiming_normal_(p.weight, mode="fan_out", nonlinearity="relu")   # torchvision resnet.py
    else:
        nn.init.kaiming_normal_(p.weight)                                        # torchvision densenet.py
    return p


class _Linear(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.uniform_(self.bias, -1.0 / math.sqrt(fin), 1.0 / math.sqrt(fin))


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.stride = stride
        self.conv1 = _tv_conv(cin, cout, 3)
        self.bn1 = BNParams(cout)
        self.conv2 = _tv_conv(cout, cout, 3)
        self.bn2 = BNParams(cout)
        if stride != 1 or cin != cout:
            self.downsample = Seq([(0, _tv_conv(cin, cout, 1)), (1, BNParams(cout))])
        else:
            self.downsample = None

    def run(self, eng: Engine, x: Act, tr: bool) -> Act:
        h = eng.conv(x, self.conv1.weight, None, self.bn1.state(), k=3, stride=self.stride, pad=1, relu=True, training=tr)
        idt = x
        if self.downsample is not None:
            idt = eng.conv(x, self.downsample.at(0).weight, None, self.downsample.at(1).state(), k=1, stride=self.stride,
                           pad=0, relu=False, training=tr)
        return eng.conv(h, self.conv2.weight, None, self.bn2.state(), k=3, stride=1, pad=1, relu=True, residual=idt,
                        training=tr)


class _Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5: the stride sits on the 3x3 convolution), expansion 4."""

    def __init__(self, cin, width, stride):
        super().__init__()
        self.stride = stride
        self.conv1 = _tv_conv(cin, width, 1)
        self.bn1 = BNParams(width)
        self.conv2 = _tv_conv(width, width, 3)
        self.bn2 = BNParams(width)
        self.conv3 = _tv_conv(width, 4 * width, 1)
        self.bn3 = BNParams(4 * width)
        if stride != 1 or cin != 4 * width:
            self.downsample = Seq([(0, _tv_conv(cin, 4 * width, 1)), (1, BNParams(4 * width))])
        else:
            self.downsample = None

    def run(self, eng: Engine, x: Act, tr: bool) -> Act:
        h = eng.conv(x, self.conv1.weight, None, self.bn1.state(), k=1, stride=1, pad=0, relu=True, training=tr)
        h = eng.conv(h, self.conv2.weight, None, self.bn2.state(), k=3, stride=self.stride, pad=1, relu=True, training=tr)
        idt = x
        if self.downsample is not None:
            idt = eng.conv(x, self.downsample.at(0).weight, None, self.downsample.at(1).state(), k=1, stride=self.stride,
                           pad=0, relu=False, training=tr)
        return eng.conv(h, self.conv3.weight, None, self.bn3.state(), k=1, stride=1, pad=0, relu=True, residual=idt, training=tr)


class _ResNet(nn.Module):
    """torchvision.models.resnet18 / 34 (BasicBlock) / 50 (Bottleneck) with fc = Identity (classifier.py:24-36)."""

    def __init__(self, layers, bottleneck: bool = False):
        super().__init__()
        self.conv1 = _tv_conv(3, 64, 7)
        self.bn1 = BNParams(64)
        cin = 64
        for li, (c, n) in enumerate(zip((64, 128, 256, 512), layers), start=1):
            blocks = []
            for bi in range(n):
                stride = 2 if (bi == 0 and li > 1) else 1
                blocks.append((bi, _Bottleneck(cin, c, stride) if bottleneck else _BasicBlock(cin, c, stride)))
                cin = 4 * c if bottleneck else c
            setattr(self, f"layer{li}", Seq(blocks))
        self.nlayers = layers
        self.out_channels = cin

    def feature_map(self, eng: Engine, x8: Act, tr: bool) -> Act:
        h = eng.conv(x8, self.conv1.weight, None, self.bn1.state(), k=7, stride=2, pad=3, relu=True, training=tr)
        h = eng.maxpool(h, 3, 2, 1)
        for li, n in enumerate(self.nlayers, start=1):
            layer = getattr(self, f"layer{li}")
            for bi in range(n):
                h = layer.at(bi).run(eng, h, tr)
        return h

    def run(self, eng: Engine, x8: Act, tr: bool) -> Act:
        return eng.global_avgpool(self.feature_map(eng, x8, tr))


class _DenseLayer(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.norm1 = BNParams(cin)
        self.conv1 = _tv_conv(cin, 128, 1, mode="fan_in")
        self.norm2 = BNParams(128)
        self.conv2 = _tv_conv(128, 32, 3, mode="fan_in")


class _Transition(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.norm = BNParams(cin)
        self.conv = _tv_conv(cin, cin // 2, 1, mode="fan_in")


class _DenseNetFeatures(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = _tv_conv(3, 64, 7, mode="fan_in")
        self.norm0 = BNParams(64)
        c = 64
        for bi, nl in enumerate(DENSENET121_BLOCKS, start=1):
            block = nn.Module()
            for li in range(1, nl + 1):
                block.add_module(f"denselayer{li}", _DenseLayer(c))
                c += 32
            setattr(self, f"denseblock{bi}", block)
            if bi < 4:
                setattr(self, f"transition{bi}", _Transition(c))
                c //= 2
        self.norm5 = BNParams(c)
        self.out_channels = c


class _DenseNet121(nn.Module):
    """torchvision.models.densenet121 with classifier = Identity: features -> relu -> global avg pool."""

    def __init__(self):
        super().__init__()
        self.features = _DenseNetFeatures()

    def run(self, eng: Engine, x8: Act, tr: bool) -> Act:
        if tr or eng.record:
            raise RuntimeError("densenet121 backbone is forward-only (eval mode, no gradients) in this build")
        f = self.features
        h = eng.conv(x8, f.conv0.weight, None, f.norm0.state(), k=7, stride=2, pad=3, relu=True, training=False)
        h = eng.maxpool(h, 3, 2, 1)
        c = 64
        for bi, nl in enumerate(DENSENET121_BLOCKS, start=1):
            block = getattr(f, f"denseblock{bi}")
            N, Hh, Ww = h.N, h.Hh, h.Ww
            total = c + 32 * nl
            buf = eng._f(N, Hh, Ww, total)     # all concatenated features of the block, written in place
            H.call("adh_axpby_strided", buf.data_ptr(), total, h.t.data_ptr(), h.cs, N * Hh * Ww, c, 0.0, 1.0)
            for li in range(1, nl + 1):
                layer = getattr(block, f"denselayer{li}")
                cur = Act(buf[..., :c], c)
                a = eng.bn_relu_eval(cur, layer.norm1.state())
                # conv1x1 -> norm2 -> relu folded into the conv epilogue
                b = eng.conv(a, layer.conv1.weight, None, layer.norm2.state(), k=1, stride=1, pad=0, relu=True)
                eng.conv(b, layer.conv2.weight, None, None, k=3, stride=1, pad=1, relu=False, out=buf[..., c:c + 32])
                c += 32
            h = Act(buf, c)
            if bi < 4:
                tr_ = getattr(f, f"transition{bi}")
                a = eng.bn_relu_eval(h, tr_.norm.state())
                a = eng.conv(a, tr_.conv.weight, None, None, k=1, stride=1, pad=0, relu=False)
                h = eng.avgpool(a, 2)
                c //= 2
        h = eng.bn_relu_eval(h, f.norm5.state())
        return eng.global_avgpool(h)


class _ClassifierFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, record, x, masks, *params):
        eng = Engine(x.device, record)
        logits_act, feat_act = module._run(eng, x, masks)
        ctx.eng, ctx.logits_act, ctx.feat_act, ctx.params = eng, logits_act, feat_act, params
        N = x.shape[0]
        logits = logits_act.t.view(N, -1)[:, :module.num_classes].clone()
        feats = feat_act.t.view(N, -1)[:, :module.feature_dim].clone()
        ctx.set_materialize_grads(False)
        return logits, feats

    @staticmethod
    def backward(ctx, g_logits, g_feats):
        eng = ctx.eng
        la, fa = ctx.logits_act, ctx.feat_act
        if g_logits is not None:
            gp = torch.zeros(la.t.shape, device=g_logits.device, dtype=torch.float32)
            gp.view(g_logits.shape[0], -1)[:, :g_logits.shape[1]] = g_logits
            la.grad = gp
        if g_feats is not None:
            gf = torch.zeros(fa.t.shape, device=g_feats.device, dtype=torch.float32)
            gf.view(g_feats.shape[0], -1)[:, :g_feats.shape[1]] = g_feats
            eng.accum(fa, gf)
        eng.backward()
        grads = []
        for p in ctx.params:
            gp_ = eng.param_grads.get(id(p))
            grads.append(gp_.reshape(p.shape) if gp_ is not None else None)
        ctx.eng = None
        return (None, None, None, None, *grads)


class FogIntensityClassifier(nn.Module):
    """Classifier for fog intensity (low, medium, high)  (classifier.py:6-103)."""

    def __init__(self, model_name="resnet18", num_classes=3, pretrained=True):
        super().__init__()
        self.model_name = model_name
        self.num_classes = num_classes
        if model_name.startswith("resnet"):
            if model_name == "resnet18":
                self.backbone, self.feature_dim = _ResNet((2, 2, 2, 2)), 512
            elif model_name == "resnet34":
                self.backbone, self.feature_dim = _ResNet((3, 4, 6, 3)), 512
            elif model_name == "resnet50":
                self.backbone, self.feature_dim = _ResNet((3, 4, 6, 3), bottleneck=True), 2048
            else:
                raise ValueError(f"Unsupported ResNet variant: {model_name}")
        elif model_name == "densenet121":
            self.backbone, self.feature_dim = _DenseNet121(), 1024
        else:
            raise ValueError(f"Unsupported model: {model_name}")
        if pretrained:
            warnings.warn("pretrained backbone weights cannot be downloaded in this environment; using torchvision's "
                          "random initialisation (load a checkpoint with load_state_dict)")
        # nn.Sequential(Dropout(.3), Linear(fd,256), ReLU, Dropout(.2), Linear(256,nc))  (classifier.py:71-78)
        self.classifier = Seq([(1, _Linear(self.feature_dim, 256)), (4, _Linear(256, num_classes))])

    def _run(self, eng: Engine, x: torch.Tensor, masks):
        tr = self.training
        feats = self.backbone.run(eng, eng.image_to_nhwc8(x), tr)       # [N,1,1,fd]
        h = feats
        if masks is not None:
            h = eng.mul_mask(h, masks[0])
        l1, l4 = self.classifier.at(1), self.classifier.at(4)
        w1 = l1.weight.view(256, self.feature_dim, 1, 1)      # a Linear is a 1x1 conv over the 1x1 pooled map
        w4 = l4.weight.view(self.num_classes, 256, 1, 1)
        eng.alias[id(w1)], eng.alias[id(w4)] = id(l1.weight), id(l4.weight)
        h = eng.conv(h, w1, l1.bias, None, k=1, stride=1, pad=0, relu=True)
        if masks is not None:
            h = eng.mul_mask(h, masks[1])
        logits = eng.conv(h, w4, l4.bias, None, k=1, stride=1, pad=0, relu=False)
        return logits, feats

    def forward(self, x):
        H.require_cuda(x, "input image batch")
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"expected [N,3,H,W] input, got {tuple(x.shape)}")
        x = x.contiguous()
        N = x.shape[0]
        masks = None
        if self.training:   # nn.Dropout(0.3) / nn.Dropout(0.2): masks from torch's generator
            m0 = (torch.rand(N, 1, 1, self.feature_dim, device=x.device) >= 0.3).float() / 0.7
            m1 = (torch.rand(N, 1, 1, 256, device=x.device) >= 0.2).float() / 0.8
            masks = (m0, m1)
        params = list(self.parameters())
        record = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _ClassifierFunction.apply(self, record, x, masks, *params)

    def extract_features(self, x):
        with torch.no_grad():
            _, feats = self.forward(x)
        return feats


class DenseFeatureExtractor(nn.Module):
    """Extract dense feature maps from the backbone (classifier.py:105-137): torchvision's resnet18 / resnet34 without the
    average pool and fc -- `nn.Sequential(*children[:-2])`, so the state_dict keys are `backbone.0.weight` (conv1),
    `backbone.1.*` (bn1), `backbone.{4,5,6,7}.<block>.*` (layer1 .. layer4).  forward(x [N,3,H,W]) -> [N, 512, H/32, W/32].  The
    reference's own drivers never use it; here it is the inference path (BatchNorm in the module's train / eval mode, no
    gradients -- a training use would go through FogIntensityClassifier's autograd function).  mobilenet_v2 / efficientnet* need
    torchvision / timm and raise ValueError like any unknown name."""

    def __init__(self, model_name="resnet18", pretrained=True):
        super().__init__()
        if model_name == "resnet18":
            net = _ResNet((2, 2, 2, 2))
        elif model_name == "resnet34":
            net = _ResNet((3, 4, 6, 3))
        else:
            raise ValueError(f"Unsupported model for feature extraction: {model_name}")
        self.model_name = model_name
        self._nlayers = net.nlayers
        self.backbone = Seq([(0, net.conv1), (1, net.bn1), (4, net.layer1), (5, net.layer2), (6, net.layer3), (7, net.layer4)])
        if pretrained:
            warnings.warn("pretrained backbone weights cannot be downloaded in this environment; using torchvision's "
                          "random initialisation (load a checkpoint with load_state_dict)")

    @torch.no_grad()
    def forward(self, x):
        """Extract dense feature maps"""
        H.require_cuda(x, "input image batch")
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"expected [N,3,H,W] input, got {tuple(x.shape)}")
        eng = Engine(x.device, record=False)
        tr = self.training
        b = self.backbone
        h = eng.conv(eng.image_to_nhwc8(x.contiguous()), b.at(0).weight, None, b.at(1).state(), k=7, stride=2, pad=3, relu=True,
                     training=tr)
        h = eng.maxpool(h, 3, 2, 1)
        for idx, n in zip((4, 5, 6, 7), self._nlayers):
            for bi in range(n):
                h = b.at(idx).at(bi).run(eng, h, tr)
        return h.t[..., :h.C].permute(0, 3, 1, 2).contiguous()


def create_classifier(config):
    """classifier.py:139-145."""
    return FogIntensityClassifier(model_name=config["classifier"]["model"], num_classes=config["classifier"]["num_classes"],
                                  pretrained=config["classifier"]["pretrained"])
