#!/usr/bin/env python3
"""Dev tool: rounding error of the convolution kernels against the fp64 definition at the headline model's channel counts:
the 3x3 stride-1 forward kernels (direct, Winograd F(2x2,3x3), F(4x4,3x3) on fp32 MFMA, F(4x4,3x3) with the opt-in bf16 x 3
contraction) and the k4 s2 / transposed forms (direct, F(3x3,2x2) on fp32 MFMA, F(3x3,2x2) bf16 x 3).  Prints max and rms error
relative to the output's max, and -- for the bf16 x 3 rows -- the ratio of its rms error to the fp32 MFMA path's."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import adam_dehaze_amd.engine as E  # noqa: E402
from adam_dehaze_amd.engine import Act, Engine  # noqa: E402

dev = torch.device("cuda:0")


def run(kind, k, stride, x, wd, Co, OH, OW):
    eng = Engine(dev, record=False)
    y = torch.zeros(x.shape[0], OH, OW, Co, device=dev)
    eng._run_gather(eng._launch_plan(kind, k, stride, 1, wd, "fwd"), Act(x), y, Co, wd)
    torch.cuda.synchronize()
    return y.cpu().double()


print("== 3x3 stride 1, C -> C, 2 x 64 x 96 pixels (headline shapes' channel counts), error / max |y|")
for C in (96, 192, 384):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(2, 64, 96, C, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    rms = {}
    for name, wino, w43, contract in (("direct", False, False, "fp32"), ("f23", True, False, "fp32"), ("f43 fp32 MFMA", True, True, "fp32"),
                                      ("f43 bf16x3", True, True, "bf16x3")):
        E.USE_WINOGRAD, E.USE_WINO43, E.CONTRACT = wino, w43, contract
        e = run("conv", 3, 1, x.to(dev), w.to(dev), C, 64, 96) - ref
        rms[name] = float(e.pow(2).mean().sqrt()) / scale
        extra = f"   rms ratio to the fp32 MFMA path {rms[name] / rms['f43 fp32 MFMA']:.2f}" if contract == "bf16x3" else ""
        print(f"C={C:4d} {name:14s} max {float(e.abs().max()) / scale:.2e}  rms {rms[name]:.2e}{extra}")

print("== Conv2d k4 s2 p1 / ConvTranspose2d k4 s2 p1 (the down / up-sampling layers), error / max |y|")
for kind, Ci, Co, Hh, Ww in (("conv", 96, 192, 64, 96), ("conv", 192, 384, 64, 96), ("convT", 384, 192, 32, 48), ("convT", 384, 96, 32, 48)):
    g = torch.Generator().manual_seed(Ci + Co)
    x = torch.randn(2, Hh, Ww, Ci, generator=g)
    if kind == "conv":
        w = torch.randn(Co, Ci, 4, 4, generator=g) / (16 * Ci) ** 0.5
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=2, padding=1).permute(0, 2, 3, 1)
        OH, OW = Hh // 2, Ww // 2
    else:
        w = torch.randn(Ci, Co, 4, 4, generator=g) / (4 * Ci) ** 0.5
        ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=2, padding=1).permute(0, 2, 3, 1)
        OH, OW = Hh * 2, Ww * 2
    scale = float(ref.abs().max())
    rms = {}
    for name, wino, contract in (("direct", False, "fp32"), ("f32 fp32 MFMA", True, "fp32"), ("f32 bf16x3", True, "bf16x3")):
        E.USE_WINOGRAD, E.USE_WINO43, E.CONTRACT = wino, True, contract
        e = run(kind, 4, 2, x.to(dev), w.to(dev), Co, OH, OW) - ref
        rms[name] = float(e.pow(2).mean().sqrt()) / scale
        extra = f"   rms ratio to the fp32 MFMA path {rms[name] / rms['f32 fp32 MFMA']:.2f}" if contract == "bf16x3" else ""
        print(f"{kind:5s} {Ci:3d}->{Co:3d} {name:14s} max {float(e.abs().max()) / scale:.2e}  rms {rms[name]:.2e}{extra}")
