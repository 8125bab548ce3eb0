"""Parameter containers with the reference's state_dict key layout.

These nn.Modules only OWN parameters/buffers (so reference checkpoints load with
`load_state_dict`); none of them calls ATen arithmetic -- compute happens in
`engine.Engine` through the HIP library.  Initialisation consumes the torch RNG exactly like
nn.Conv2d / nn.ConvTranspose2d (kaiming_uniform(a=sqrt(5)) then the bias bound) so that a model
built under `torch.manual_seed(s)` has the same weights as the reference built under the same seed
(/root/reference models/dehazing/base_model.py:4-78 uses PyTorch's default init everywhere).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .engine import Act, BNState, Engine


class ConvParams(nn.Module):
    """weight [Cout,Cin,k,k] (+ bias) of a Conv2d, or [Cin,Cout,k,k] of a ConvTranspose2d."""

    def __init__(self, cin: int, cout: int, k: int, bias: bool, transposed: bool = False):
        super().__init__()
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            fan_in = shape[1] * k * k   # torch: fan_in = weight.size(1) * receptive field
            bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, *a, **k):
        raise RuntimeError("parameter container: compute runs through the HIP engine")


class BNParams(nn.Module):
    """BatchNorm2d parameters and running statistics (eps 1e-5, momentum 0.1)."""

    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def state(self) -> BNState:
        return BNState(self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked)

    def forward(self, *a, **k):
        raise RuntimeError("parameter container: compute runs through the HIP engine")


class Seq(nn.ModuleDict):
    """Children addressed by their index in the reference's nn.Sequential ('0', '1', '3', ...)."""

    def __init__(self, items):
        super().__init__({str(i): m for i, m in items})

    def at(self, i: int):
        return self[str(i)]


class ConvBlock(nn.Module):
    """Conv2d(bias = not use_bn) -> BatchNorm2d -> ReLU|None   (base_model.py:4-24)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, use_bn=True, activation="relu",
                 relu=None):
        """`activation`: the reference passes an nn.ReLU instance or None (base_model.py:7-8); 'relu' / None /
        an nn.ReLU are accepted, anything else is not on the path."""
        super().__init__()
        cin, cout = in_channels, out_channels
        if relu is None:
            if activation is None:
                relu = False
            elif activation == "relu" or isinstance(activation, nn.ReLU):
                relu = True
            else:
                raise ValueError(f"unsupported activation for the HIP ConvBlock: {activation!r}")
        self.k, self.stride, self.padding, self.relu = kernel_size, stride, padding, relu
        items = [(0, ConvParams(cin, cout, kernel_size, bias=not use_bn))]
        if use_bn:
            items.append((1, BNParams(cout)))
        self.block = Seq(items)
        self.use_bn = use_bn

    def run(self, eng: Engine, x: Act, training: bool, residual: Optional[Act] = None, out=None) -> Act:
        conv = self.block.at(0)
        bn = self.block.at(1).state() if self.use_bn else None
        return eng.conv(x, conv.weight, conv.bias, bn, kind="conv", k=self.k, stride=self.stride, pad=self.padding,
                        relu=self.relu, residual=residual, training=training, out=out)


class ResidualBlock(nn.Module):
    """relu(BN(conv(relu(BN(conv(x))))) + x)   (base_model.py:26-41)."""

    def __init__(self, channels, kernel_size=3):
        super().__init__()
        self.conv1 = ConvBlock(channels, channels, kernel_size, padding=kernel_size // 2)
        self.conv2 = ConvBlock(channels, channels, kernel_size, padding=kernel_size // 2, relu=False)

    def run(self, eng: Engine, x: Act, training: bool, out=None) -> Act:
        h = self.conv1.run(eng, x, training)
        c2 = self.conv2
        conv, bn = c2.block.at(0), c2.block.at(1).state()
        # the block's tail: BN -> (+x) -> ReLU fused into conv2's epilogue / normalise pass
        return eng.conv(h, conv.weight, conv.bias, bn, kind="conv", k=c2.k, stride=1, pad=c2.padding, relu=True,
                        residual=x, training=training, out=out)


class AttentionBlock(nn.Module):
    """CBAM-style channel + spatial attention   (base_model.py:43-78)."""

    def __init__(self, channels, reduction=16):
        super().__init__()
        hidden = channels // reduction
        self.fc = Seq([(0, ConvParams(channels, hidden, 1, bias=False)), (2, ConvParams(hidden, channels, 1, bias=False))])
        self.conv_spatial = ConvParams(2, 1, 7, bias=False)

    def run(self, eng: Engine, x: Act, out=None) -> Act:
        return eng.attention(x, self.fc.at(0).weight, self.fc.at(2).weight, self.conv_spatial.weight, out=out)


def run_up(eng: Engine, convT: ConvParams, bn: BNParams, x: Act, training: bool, out=None) -> Act:
    """ConvTranspose2d(k4,s2,p1) -> BatchNorm2d -> ReLU: entries 0-2 of a decoder stage
    (medium_intensity.py:52-56; high_intensity.py:56-60)."""
    return eng.conv(x, convT.weight, convT.bias, bn.state(), kind="convT", k=4, stride=2, pad=1, relu=True,
                    training=training, out=out)
