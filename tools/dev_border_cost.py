#!/usr/bin/env python3
"""Dev probe: what do border strips / regions cost?  The same number of pixels and channels at image sizes whose share of border
strips (conv_wgrad_wino43_kernel: 4 x 16-pixel strips; conv_wgrad32v2_kernel) or border regions (forward kernels) differs.

    python tools/dev_border_cost.py [--cin 96]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adam_dehaze_amd.engine import Act, Engine  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=96)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    C = args.cin
    base_px = 8 * 512 * 1024 * 96 // C
    for kind, k, stride in (("conv", 3, 1), ("conv", 4, 2), ("convT", 4, 2)):
        for Hh, Ww in ((512, 1024), (256, 512), (128, 256), (64, 128), (32, 64)):
            N = base_px // (Hh * Ww)
            if N < 1 or N > 4096:
                continue
            eng = Engine(dev, record=False)
            x = Act(torch.randn(N, Hh, Ww, C, device=dev))
            w = (torch.randn(C, C, k, k, device=dev) * 0.05).requires_grad_(True)
            o = eng.conv(x, w, None, None, kind=kind, k=k, stride=stride, pad=1, relu=False)
            out_t = o.t
            g = torch.randn_like(out_t)
            gsrc = Act(g, C)
            gx = torch.empty(N, Hh, Ww, C, device=dev)

            def fwd():
                eng._run_gather(eng._launch_plan(kind, k, stride, 1, w, "fwd"), x, out_t, C, w)

            def dgrad():
                eng._run_gather(eng._launch_plan(kind, k, stride, 1, w, "dgrad"), gsrc, gx, C, w)

            def wgrad():
                eng._wgrad(eng._launch_plan(kind, k, stride, 1, w, "fwd"), x, g, C, w)
            print(f"{kind} k{k} s{stride} C={C} N={N:4d} {Hh:4d}x{Ww:<4d}  fwd {timeit(fwd, args.iters):7.3f}  dgrad {timeit(dgrad, args.iters):7.3f}"
                  f"  wgrad {timeit(wgrad, args.iters):7.3f} ms", flush=True)
            del x, o, g, gx, gsrc, eng
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
