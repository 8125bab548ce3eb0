#!/usr/bin/env python3
"""Dev tool: rounding error of the three 3x3 s1 forward kernels (direct, Winograd F(2x2,3x3), F(4x4,3x3)) against the
fp64 definition, at the headline model's channel counts.  Prints max and rms error relative to the output's max."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import adam_dehaze_amd.engine as E  # noqa: E402
from adam_dehaze_amd.engine import Act, Engine  # noqa: E402

dev = torch.device("cuda:0")
for C in (96, 192, 384):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(1, 64, 96, C, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    for name, wino, w43 in (("direct", False, False), ("f23", True, False), ("f43", True, True)):
        E.USE_WINOGRAD, E.USE_WINO43 = wino, w43
        eng = Engine(dev, record=False)
        wd = w.to(dev)
        y = torch.zeros(1, 64, 96, C, device=dev)
        eng._run_gather(eng._launch_plan("conv", 3, 1, 1, wd, "fwd"), Act(x.to(dev)), y, C, wd)
        torch.cuda.synchronize()
        e = (y.cpu().double() - ref)
        print(f"C={C:4d} {name:7s} max {float(e.abs().max()) / scale:.2e}  rms {float(e.pow(2).mean().sqrt()) / scale:.2e}")
