#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels at the shapes the headline config uses
(CORUN-Complex, bs 8, 512x1024).  Dev tool: times each entry point with HIP events on the launch
stream, `--only` filters by substring, `--iters` repeats.  Used under rocprofv3 for PMC runs.

    python tools/bench_kernels.py --only conv96 --iters 5
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from adam_dehaze_amd import _hip as H          # noqa: E402
from adam_dehaze_amd.engine import Act, Engine  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--pass", dest="passes", default="fwd,dgrad,wgrad", help="comma list of fwd,dgrad,wgrad")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N = args.batch
    cases = [
        # name, Cin, Cout, H, W, kind, k, stride, pad
        ("conv96_full", 96, 96, 512, 1024, "conv", 3, 1, 1),
        ("conv192_half", 192, 192, 256, 512, "conv", 3, 1, 1),
        ("conv384_quarter", 384, 384, 128, 256, "conv", 3, 1, 1),
        ("head192to96_full", 192, 96, 512, 1024, "conv", 3, 1, 1),
        ("down96to192", 96, 192, 512, 1024, "conv", 4, 2, 1),
        ("up384to96", 384, 96, 256, 512, "convT", 4, 2, 1),
        ("down192to384", 192, 384, 256, 512, "conv", 4, 2, 1),
        ("probe_down128to192", 128, 192, 512, 1024, "conv", 4, 2, 1),     # (not a model layer: the 64 x 96 split at the full-res geometry)
        ("probe_down96to128", 96, 128, 512, 1024, "conv", 4, 2, 1),       # (the 96 x 64 split with two output groups)
        ("up384to192", 384, 192, 128, 256, "convT", 4, 2, 1),
        ("stem7x7", 8, 96, 512, 1024, "conv", 7, 1, 3),
    ]
    for name, Cin, Cout, Hh, Ww, kind, k, stride, pad in cases:
        if args.only and args.only not in name:
            continue
        eng = Engine(dev, record=False)
        x = Act(torch.randn(N, Hh, Ww, Cin, device=dev))
        if kind == "conv":
            w = torch.randn(Cout, 3 if name == "stem7x7" else Cin, k, k, device=dev) * 0.05
        else:
            w = torch.randn(Cin, Cout, k, k, device=dev) * 0.05
        w.requires_grad_(True)
        o = eng.conv(x, w, None, None, kind=kind, k=k, stride=stride, pad=pad, relu=False)
        OH, OW = o.Hh, o.Ww
        cin_real = w.shape[1] if kind == "conv" else Cin
        flops = 2.0 * N * OH * OW * k * k * cin_real * Cout if kind == "conv" else 2.0 * N * Hh * Ww * 16 * Cin * Cout
        out_t = o.t

        def fwd():
            eng._run_gather(eng._launch_plan(kind, k, stride, pad, w, "fwd"), x, out_t, Cout, w)

        g = torch.randn_like(out_t)
        gsrc = Act(g, Cout)
        gx = torch.empty(N, Hh, Ww, Cin, device=dev)

        def dgrad():
            eng._run_gather(eng._launch_plan(kind, k, stride, pad, w, "dgrad"), gsrc, gx, Cin, w)

        def wgrad():
            eng._wgrad(eng._launch_plan(kind, k, stride, pad, w, "fwd"), x, g, Cout, w)

        if "bnred" in args.passes.split(",") and kind == "conv" and k == 3:
            # DESIGN 4.13a (BatchNorm pass removal): the data-gradient launch with the producer's BN-backward sums in its
            # epilogue (adh_conv_wino43_dgrad_bnred) against the plain launch + the bn_bwd_reduce pass it replaces
            yprev = torch.randn(N, Hh, Ww, Cin, device=dev)
            sc = torch.rand(Cin, device=dev) + 0.5
            sh = torch.randn(Cin, device=dev) * 0.1
            ss = torch.stack([sc, sh]).contiguous()
            mean0 = torch.zeros(Cin, device=dev)

            def dgrad_fused():
                rows, _ = eng._run_gather(eng._launch_plan(kind, k, stride, pad, w, "dgrad"), gsrc, gx, Cin, w,
                                          bnred=(yprev, ss, mean0))
                assert rows is not None
            P = N * Hh * Ww
            mean = torch.zeros(Cin, device=dev)
            inv = torch.ones(Cin, device=dev)
            part = torch.empty(H.value("adh_bn_bwd_num_blocks", P, Cin), 2, Cin, device=dev)
            mss = torch.cat([sc, sh]).contiguous()

            def bn_bwd_reduce():
                H.call("adh_bn_bwd_reduce", gx.data_ptr(), Cin, None, 0, 1, yprev.data_ptr(), Cin, mean.data_ptr(),
                       inv.data_ptr(), part.data_ptr(), P, Cin, mss.data_ptr(), None)
            for tag, fn in (("dgrad", dgrad), ("dgrad+bnred", dgrad_fused), ("bn_bwd_reduce", bn_bwd_reduce)):
                print(f"{name:18s} {tag:14s} {timeit(fn, args.iters):8.3f} ms", flush=True)
            continue
        for tag, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
            if (name == "stem7x7" and tag == "dgrad") or tag not in args.passes.split(","):
                continue
            ms = timeit(fn, args.iters)
            print(f"{name:18s} {tag:6s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s ({flops / 1e9:.0f} GFLOP)", flush=True)

    if not args.only or "bn" in args.only:
        P, Cc = N * 512 * 1024, 96
        y = torch.randn(N, 512, 1024, Cc, device=dev)
        out = torch.empty_like(y)
        sc = torch.ones(Cc, device=dev)
        sh = torch.zeros(Cc, device=dev)

        def bn_apply():
            H.call("adh_bn_apply", y.data_ptr(), Cc, sc.data_ptr(), sh.data_ptr(), None, 0, 1, out.data_ptr(), Cc, P, Cc, None)
        ms = timeit(bn_apply, args.iters)
        print(f"bn_apply 96ch full  {ms:8.3f} ms  {2 * y.numel() * 4 / ms / 1e9:7.2f} TB/s")
        g = torch.randn_like(y)
        mean = torch.zeros(Cc, device=dev)
        inv = torch.ones(Cc, device=dev)
        nblk = H.value("adh_bn_bwd_num_blocks", P, Cc)
        part = torch.empty(nblk, 2, Cc, device=dev)
        coef = torch.ones(3, Cc, device=dev)
        gy = torch.empty_like(y)

        def bn_bwd_reduce():
            H.call("adh_bn_bwd_reduce", g.data_ptr(), Cc, out.data_ptr(), Cc, 1, y.data_ptr(), Cc, mean.data_ptr(),
                   inv.data_ptr(), part.data_ptr(), P, Cc, None, None)
        ms = timeit(bn_bwd_reduce, args.iters)
        print(f"bn_bwd_reduce 96ch  {ms:8.3f} ms  {3 * y.numel() * 4 / ms / 1e9:7.2f} TB/s")

        def bn_bwd_apply():
            H.call("adh_bn_bwd_apply", g.data_ptr(), Cc, out.data_ptr(), Cc, 1, y.data_ptr(), Cc, mean.data_ptr(),
                   inv.data_ptr(), coef.data_ptr(), 1, gy.data_ptr(), Cc, None, 0, P, Cc, None, None)
        ms = timeit(bn_bwd_apply, args.iters)
        print(f"bn_bwd_apply 96ch   {ms:8.3f} ms  {4 * y.numel() * 4 / ms / 1e9:7.2f} TB/s")


if __name__ == "__main__":
    main()
