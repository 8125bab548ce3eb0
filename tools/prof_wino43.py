#!/usr/bin/env python3
"""Reads the per-workgroup stamps of a -DW4_PROF build of conv_wino43_kernel (tools/prof_wino43.sh) and prints where a
workgroup's time goes and how long a CU idles between two workgroups.  Dev tool."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adam_dehaze_amd import _hip as H          # noqa: E402
from adam_dehaze_amd.engine import Act, Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=96)
    ap.add_argument("--h", type=int, default=512)
    ap.add_argument("--w", type=int, default=1024)
    ap.add_argument("--kernel", default="w43", choices=["w43", "w32", "wgw"], help="wgw: Winograd-domain weight gradient of the 3x3 layers")
    ap.add_argument("--case", default="down", choices=["down", "up"], help="w32: Conv2d k4 s2 96->192 / ConvTranspose2d k4 s2 384->96")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = H.load()
    Cc = args.cin
    eng = Engine(dev, record=False)
    if args.kernel in ("w43", "wgw"):
        kind, k, stride, Cout = "conv", 3, 1, Cc
        x = Act(torch.randn(8, args.h, args.w, Cc, device=dev))
        w = (torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05).requires_grad_(True)
    elif args.case == "down":
        kind, k, stride, Cc, Cout = "conv", 4, 2, 96, 192
        x = Act(torch.randn(8, 512, 1024, Cc, device=dev))
        w = (torch.randn(Cout, Cc, 4, 4, device=dev) * 0.05).requires_grad_(True)
    else:
        kind, k, stride, Cc, Cout = "convT", 4, 2, 384, 96
        x = Act(torch.randn(8, 256, 512, Cc, device=dev))
        w = (torch.randn(Cc, Cout, 4, 4, device=dev) * 0.05).requires_grad_(True)
    o = eng.conv(x, w, None, None, kind=kind, k=k, stride=stride, pad=1, relu=False)
    out_t = o.t
    gy = torch.randn_like(out_t)

    def run():
        if args.kernel == "wgw":
            eng._wgrad(eng._launch_plan(kind, k, stride, 1, w, "fwd"), x, gy, Cout, w)
        else:
            eng._run_gather(eng._launch_plan(kind, k, stride, 1, w, "fwd"), x, out_t, Cout, w)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    ev_us = e0.elapsed_time(e1) * 1000.0
    buf = np.zeros(16384 * 32, dtype=np.uint64)
    rc = {"w43": lib.adh_w4_prof_read, "w32": lib.adh_w3_prof_read, "wgw": lib.adh_wr_prof_read}[args.kernel](C.c_void_p(buf.ctypes.data))
    assert rc == 0
    b = buf.reshape(16384, 32)
    b = b[b[:, 0] != 0]
    t = b[:, :4].astype(np.float64)
    tc = b[:, 8:14].astype(np.float64)
    te = b[:, 16:22].astype(np.float64)
    # s_memtime counts shader clocks and the counters of different CUs are not synchronised: calibrate on the per-CU
    # span (first start to last end of the 8..32 workgroups a CU ran) against the HIP-event time of the launch
    hw0 = b[:, 6].astype(np.int64)
    key0 = (b[:, 7].astype(np.int64) & 0xf) * 65536 + ((hw0 >> 8) & 0xff)
    spans = [t[key0 == k, 3].max() - t[key0 == k, 0].min() for k in np.unique(key0)]
    span_ticks = float(np.median(spans))
    tick_us = ev_us / span_ticks
    print(f"event time {ev_us:.1f} us, per-CU span p10/p50/p90 {np.percentile(spans, 10):.0f}/{span_ticks:.0f}/{np.percentile(spans, 90):.0f} ticks -> {1.0 / tick_us:.1f} ticks/us")
    print(f"workgroups {len(b)}")
    for name, a, c in (("prologue", 0, 1), ("main loop", 1, 2), ("epilogue", 2, 3), ("total", 0, 3)):
        dt = (t[:, c] - t[:, a]) * tick_us
        print(f"  {name:10s} mean {dt.mean():7.2f} us  p10 {np.percentile(dt, 10):7.2f}  p50 {np.percentile(dt, 50):7.2f}  p90 {np.percentile(dt, 90):7.2f}")
    if args.kernel == "w43" and (b[:, 4] != 0).all() and (b[:, 5] != 0).all():   # persistent form: the top of a region
        for name, a, c in (("region top: next table", 0, 4), ("first weights", 4, 5), ("zero + barrier", 5, 1)):
            dt = (b[:, c].astype(np.float64) - b[:, a].astype(np.float64)) * tick_us
            print(f"  {name:22s} mean {dt.mean():7.2f} us  p10 {np.percentile(dt, 10):7.2f}  p50 {np.percentile(dt, 50):7.2f}  p90 {np.percentile(dt, 90):7.2f}")
    chunk_rows = (("chunk 1: stage issue", 0, 1), ("transform", 1, 2), ("barrier", 2, 3), ("contraction", 3, 4), ("fix + barrier", 4, 5), ("whole chunk", 0, 5))
    if args.kernel == "w32":   # (a convT launch runs one class per launch: the event time covers the four launches, the stamps the last)
        chunk_rows = (("slab 1: contraction", 0, 1), ("fix + barrier", 1, 2), ("transform", 2, 3), ("barrier", 3, 4), ("whole slab", 0, 4))
    if args.kernel == "wgw":   # (the event time also covers the split-sum and reduce kernels)
        chunk_rows = (("tile 1: stage issue", 0, 1), ("16 k-steps", 1, 2), ("wait for the next tile's data", 2, 3), ("barrier", 3, 4), ("whole tile", 0, 4))
    for name, a, c in chunk_rows:
        dt = (tc[:, c] - tc[:, a]) * tick_us
        print(f"  {name:22s} mean {dt.mean():7.2f} us  p10 {np.percentile(dt, 10):7.2f}  p50 {np.percentile(dt, 50):7.2f}  p90 {np.percentile(dt, 90):7.2f}")
    for name, a, c in (("epilogue round 1: barrier", 0, 1), ("M write", 1, 2), ("barrier", 2, 3), ("half 0", 3, 4), ("half 1", 4, 5), ("whole round (no stats)", 0, 5)):
        if (te[:, c] == 0).any() or (te[:, a] == 0).any():
            continue   # (this kernel does not stamp that point)
        dt = (te[:, c] - te[:, a]) * tick_us
        print(f"  {name:26s} mean {dt.mean():7.2f} us  p10 {np.percentile(dt, 10):7.2f}  p50 {np.percentile(dt, 50):7.2f}  p90 {np.percentile(dt, 90):7.2f}")
    hw = b[:, 6].astype(np.int64)
    xcc = b[:, 7].astype(np.int64) & 0xf
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    gaps = []
    per_cu = []
    for k in np.unique(key):
        sel = np.where(key == k)[0]
        order = sel[np.argsort(t[sel, 0])]
        per_cu.append(len(order))
        for i in range(1, len(order)):
            gaps.append((t[order[i], 0] - t[order[i - 1], 3]) * tick_us)
    gaps = np.array(gaps)
    print(f"  CUs seen {len(per_cu)}; workgroups per CU min {min(per_cu)} max {max(per_cu)}")
    print(f"  gap between consecutive workgroups on a CU: mean {gaps.mean():6.2f} us  p10 {np.percentile(gaps, 10):6.2f}  p50 {np.percentile(gaps, 50):6.2f}  p90 {np.percentile(gaps, 90):6.2f}  (negative = overlap)")
    busy = (t[:, 3] - t[:, 0]).sum() * tick_us / len(per_cu)
    print(f"  busy time per CU {busy:8.1f} us of {ev_us:.1f}")


if __name__ == "__main__":
    main()
