"""Seeded random state_dicts with torchvision / lpips parameter names for the third-party networks
(no pretrained weights are available offline).  Test helper only."""
import math

import torch

from oracle.ref_cpu import VGG16_CFG, ALEX_CFG, ALEX_SLICE_OF, DENSENET121_BLOCKS


def _conv(sd, name, cout, cin, k, g, bias=False):
    fan_in = cin * k * k
    sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / fan_in)
    if bias:
        sd[name + ".bias"] = 0.05 * torch.randn(cout, generator=g)


def _bn(sd, name, c, g):
    sd[name + ".weight"] = 0.8 + 0.4 * torch.rand(c, generator=g)
    sd[name + ".bias"] = 0.05 * torch.randn(c, generator=g)
    sd[name + ".running_mean"] = 0.05 * torch.randn(c, generator=g)
    sd[name + ".running_var"] = 0.8 + 0.4 * torch.rand(c, generator=g)
    sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)


def _head(sd, fd, g):
    sd["classifier.1.weight"] = torch.randn(256, fd, generator=g) / math.sqrt(fd)
    sd["classifier.1.bias"] = 0.05 * torch.randn(256, generator=g)
    sd["classifier.4.weight"] = torch.randn(3, 256, generator=g) / 16.0
    sd["classifier.4.bias"] = 0.05 * torch.randn(3, generator=g)


def resnet18_sd(seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    _conv(sd, "backbone.conv1", 64, 3, 7, g)
    _bn(sd, "backbone.bn1", 64, g)
    cin = 64
    for li, c in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            q = f"backbone.layer{li}.{bi}"
            _conv(sd, q + ".conv1", c, cin if bi == 0 else c, 3, g)
            _bn(sd, q + ".bn1", c, g)
            _conv(sd, q + ".conv2", c, c, 3, g)
            _bn(sd, q + ".bn2", c, g)
            if bi == 0 and li > 1:
                _conv(sd, q + ".downsample.0", c, cin, 1, g)
                _bn(sd, q + ".downsample.1", c, g)
        cin = c
    _head(sd, 512, g)
    return sd


def resnet50_sd(seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    _conv(sd, "backbone.conv1", 64, 3, 7, g)
    _bn(sd, "backbone.bn1", 64, g)
    cin = 64
    for li, (c, nb) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
        for bi in range(nb):
            q = f"backbone.layer{li}.{bi}"
            _conv(sd, q + ".conv1", c, cin, 1, g)
            _bn(sd, q + ".bn1", c, g)
            _conv(sd, q + ".conv2", c, c, 3, g)
            _bn(sd, q + ".bn2", c, g)
            _conv(sd, q + ".conv3", 4 * c, c, 1, g)
            _bn(sd, q + ".bn3", 4 * c, g)
            sd[q + ".bn3.weight"] *= 0.5       # keeps the residual sums of 16 random-weight blocks from growing
            if bi == 0:
                _conv(sd, q + ".downsample.0", 4 * c, cin, 1, g)
                _bn(sd, q + ".downsample.1", 4 * c, g)
            cin = 4 * c
    _head(sd, 2048, g)
    return sd


def resnet18_feature_extractor_sd(seed=0):
    """DenseFeatureExtractor('resnet18') keys: the classifier's resnet18 backbone renumbered as nn.Sequential children."""
    src = resnet18_sd(seed)
    ren = {"conv1": "0", "bn1": "1", "layer1": "4", "layer2": "5", "layer3": "6", "layer4": "7"}
    out = {}
    for k, v in src.items():
        if not k.startswith("backbone."):
            continue
        parts = k.split(".")
        parts[1] = ren[parts[1]]
        out[".".join(parts)] = v
    return out


def densenet121_sd(seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    p = "backbone.features."
    _conv(sd, p + "conv0", 64, 3, 7, g)
    _bn(sd, p + "norm0", 64, g)
    c = 64
    for bi, nl in enumerate(DENSENET121_BLOCKS, start=1):
        for li in range(1, nl + 1):
            q = f"{p}denseblock{bi}.denselayer{li}"
            _bn(sd, q + ".norm1", c, g)
            _conv(sd, q + ".conv1", 128, c, 1, g)
            _bn(sd, q + ".norm2", 128, g)
            _conv(sd, q + ".conv2", 32, 128, 3, g)
            c += 32
        if bi < 4:
            q = f"{p}transition{bi}"
            _bn(sd, q + ".norm", c, g)
            _conv(sd, q + ".conv", c // 2, c, 1, g)
            c //= 2
    _bn(sd, p + "norm5", c, g)
    assert c == 1024
    _head(sd, 1024, g)
    return sd


def vgg16_sd(seed=0, prefix="model."):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    idx, cin = 0, 3
    for v in VGG16_CFG:
        if v == "M":
            idx += 1
        else:
            _conv(sd, f"{prefix}{idx}", v, cin, 3, g, bias=True)
            cin = v
            idx += 2
    return sd


def lpips_alex_sd(seed=0, prefix="loss_fn."):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    idx, cin = 0, 3
    chans = []
    for v in ALEX_CFG:
        if v == "M":
            idx += 1
        else:
            c, k, _, _ = v
            _conv(sd, f"{prefix}net.{ALEX_SLICE_OF[idx]}", c, cin, k, g, bias=True)
            cin = c
            chans.append(c)
            idx += 2
    for k, c in enumerate(chans):
        sd[f"{prefix}lin{k}.model.1.weight"] = torch.rand(1, c, 1, 1, generator=g) / c
        sd[f"{prefix}lins.{k}.model.1.weight"] = sd[f"{prefix}lin{k}.model.1.weight"]   # lpips registers both names
    sd[f"{prefix}scaling_layer.shift"] = torch.tensor([-0.030, -0.088, -0.188]).view(1, 3, 1, 1)
    sd[f"{prefix}scaling_layer.scale"] = torch.tensor([0.458, 0.448, 0.450]).view(1, 3, 1, 1)
    return sd
