#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): images/sec, CORUN-Complex fwd+bwd, 512x1024, bs = 8 per GPU (weak scaling),
plus PSNR of the HIP output against the CPU oracle.  One "step" = train-mode forward of
HighIntensityDehazeModel(96) on a synthetic foggy batch + L1 loss + backward + Adam update (and, for
N > 1, the RCCL gradient all-reduce).  fp32 arithmetic (fp32 MFMA) end to end.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (dominant kernel =
the MFMA gather convolution, timed live with HIP events on the launch stream) and `cpu_baseline`
(the CPU oracle timed on a bounded sample on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA peak (= vector peak)


def synthetic_batch(n, h, w, seed=42):
    """Synthetic (hazy, clear) frames: low-pass random clear images + the reference's fog model
    I = J*t + A*(1-t), t = exp(-beta*depth) (/root/reference utils/helpers.py:241-258), level = i % 3."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    clear = torch.rand(n, 3, h, w, generator=g)
    clear = F.avg_pool2d(F.pad(clear, (2, 2, 2, 2), mode="reflect"), 5, 1)
    xs = torch.linspace(0, 1, w).view(1, w)
    ys = torch.linspace(0, 1, h).view(h, 1)
    depth = 0.3 + 0.7 * torch.sqrt((xs - 0.5) ** 2 + (ys - 0.2) ** 2)
    u = torch.rand(n, 2, generator=g)
    ranges = {0: ((0.1, 0.4), (0.5, 0.7)), 1: ((0.4, 0.7), (0.7, 0.9)), 2: ((0.7, 1.0), (0.8, 1.0))}
    hazy = torch.empty_like(clear)
    for i in range(n):
        (b0, b1), (a0, a1) = ranges[i % 3]
        beta = b0 + (b1 - b0) * float(u[i, 0])
        A = a0 + (a1 - a0) * float(u[i, 1])
        t = torch.exp(-beta * depth)
        hazy[i] = (clear[i] * t + A * (1 - t)).clamp(0, 1)
    return hazy, clear


def host_threads():
    """Threads for the CPU baseline: the cores this process may actually run on.  A GPU box gives a one-GPU job a share
    of the host (16 cores on this pool) although os.cpu_count() reports the whole machine; asking torch for hundreds of
    threads inside that share oversubscribes it 10x and measures the scheduler, not the CPU.  ADH_CPU_THREADS overrides."""
    env = os.environ.get("ADH_CPU_THREADS")
    if env:
        return max(1, int(env))
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:    # cgroup v2 CPU quota, if any
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p_ = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p_))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(h, w, threads):
    """The CPU oracle (oracle/ref_cpu.py, a pinned restatement of the reference) timed on the host cores: train-mode
    Complex forward + L1 + backward on ONE 512x1024 image, 1 warm-up + 3 timed iterations, median (SURVEY 8d /
    BASELINE.md 3); all host threads."""
    import statistics
    import torch.nn.functional as F
    import adam_dehaze_amd as A
    from oracle import ref_cpu as R
    torch.set_num_threads(threads)
    torch.manual_seed(42)
    m = A.HighIntensityDehazeModel()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    hazy, clear = synthetic_batch(1, h, w, seed=7)
    times = []
    for it in range(4):
        print(f"[bench] cpu_baseline iteration {it} ({threads} threads) ...", file=sys.stderr, flush=True)
        sd = {k: v.clone() for k, v in sd0.items()}
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
        t0 = time.perf_counter()
        out = R.high_forward(hazy, sd, training=True)
        loss = F.l1_loss(out, clear)
        loss.backward()
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    med = statistics.median(times)
    # what the oracle computed on that frame is kept: the HIP path is compared with it on the SAME weights and input (headline_parity)
    sample = {"sd0": sd0, "hazy": hazy, "clear": clear, "out": out.detach(), "loss": float(loss.detach())}
    return {"value": 1.0 / med, "unit": "images/sec", "cores": threads, "kind": "port", "cpu_model": cpu_model_string(),
            "host_logical_cpus": os.cpu_count(),
            "sample": f"1 image {h}x{w}, CORUN-Complex train fwd + L1 + bwd, 1 warm-up + 3 timed iterations, median "
                      f"{med:.2f} s (all: {', '.join(f'{t_:.2f}' for t_ in times)}), torch CPU oracle, {threads} threads"}, sample


def headline_parity(sample, device):
    """The HIP path against the CPU oracle on the frame the cpu_baseline leg computed: TRAIN-mode CORUN-Complex forward + L1 on
    one full-resolution 512x1024 frame, same initial weights, same input (in eval mode with barely moved
    running statistics the residual is ~0.008 of the output and a PSNR mostly measures the pass-through of x; in train mode
    it is ~0.1).  Returns PSNR (data_range 1), max |out - out_oracle| and |loss - loss_oracle|."""
    import adam_dehaze_amd as A
    from adam_dehaze_amd.loss import l1_loss
    from oracle import ref_cpu as R
    m = A.HighIntensityDehazeModel()
    m.load_state_dict(sample["sd0"], strict=True)
    m = m.to(device).train()
    with torch.no_grad():
        out = m(sample["hazy"].to(device))
        loss = float(l1_loss(out, sample["clear"].to(device)))
    out = out.cpu()
    ref = sample["out"]
    ps = R.psnr(out, ref)
    return {"psnr_db": ps if ps != float("inf") else 999.0, "max_abs": float((out - ref).abs().max()),
            "abs_loss_diff": abs(loss - sample["loss"]), "loss_oracle": sample["loss"],
            "mean_abs_residual_oracle": float((ref - sample["hazy"]).abs().mean()),
            "frame": f"1x3x{ref.shape[2]}x{ref.shape[3]} train-mode forward + L1, weights and input of the cpu_baseline leg"}


_PMC_PREFIX = {"adh_conv_wino43_forward": "void conv_wino43_kernel<", "adh_conv_wino_forward": "void conv_wino_kernel<", "adh_conv_wino32_forward": "void conv_wino32_kernel<",
               "adh_conv_wgrad_wino": "void conv_wgrad_rows_kernel<3, 3, false", "adh_conv_wgrad_wino32": "void conv_wgrad32_kernel<", "adh_conv_wgrad": "void conv_wgrad_",
               "adh_conv_forward": "void conv_rows_kernel<"}


def pmc_traffic(family):
    """Average HBM bytes per launch of the kernels behind `family`, from the committed PMC passes of this command
    (tools/pmc_bench.sh -> profiles/r0N_pmc_bench.json: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled for
    gfx950).  None when the file is absent."""
    ks, src = None, None
    for name in ("r04b_pmc_bench.json", "r04_pmc_bench.json", "r03_pmc_bench.json", "r02_pmc_bench.json", "r01_pmc_bench.json"):     # the newest committed pass wins
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            with open(path) as f:
                ks = json.load(f)["kernels"]
            src = "profiles/" + name
            break
        except (OSError, ValueError, KeyError):
            continue
    if ks is None:
        return None, None
    pre = _PMC_PREFIX.get(family, "")
    sel = [v for k, v in ks.items() if pre and k.startswith(pre)]
    n = sum(v["launches"] for v in sel)
    if not n:
        return None, None
    return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / n, src


def psnr_check(model, device):
    """PSNR / max-abs of the HIP eval output against the CPU oracle on a 1x3x128x256 frame."""
    from oracle import ref_cpu as R
    hazy, _ = synthetic_batch(1, 128, 256, seed=11)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    was_training = model.training
    model.eval()
    with torch.no_grad():
        out = model(hazy.to(device)).cpu()
        ref = R.high_forward(hazy, sd, training=False)
    model.train(was_training)
    return R.psnr(out, ref), float((out - ref).abs().max())


def other_workloads(args, rank, world, device):
    """The other BASELINE.json configs (secondary measurements; same timing protocol, one JSON line)."""
    import warnings
    import torch.distributed as dist
    import adam_dehaze_amd as A
    from adam_dehaze_amd import train as T
    from adam_dehaze_amd.classifier import FogIntensityClassifier
    from adam_dehaze_amd.loss import DehazingLoss, l1_loss
    from adam_dehaze_amd.optim import Adam
    from adam_dehaze_amd.parallel import GradientSynchronizer
    from adam_dehaze_amd.routing import HardRouter
    warnings.simplefilter("ignore")
    wl = args.workload
    bs = args.batch
    if wl == "config2":      # HDEN classify + hard route -> CORUN-Light forward, bs 8, 512x1024 (eval)
        clf = FogIntensityClassifier("densenet121", 3, pretrained=False).to(device).eval()
        models = {"low": A.LightweightDehazeModel(), "medium": A.MediumIntensityDehazeModel(),
                  "high": A.HighIntensityDehazeModel()}
        router = HardRouter(models, clf, device=str(device)).to(device).eval()
        hazy, _ = synthetic_batch(bs, args.height, args.width, seed=42 + rank)
        hazy = hazy.to(device)
        labels = torch.zeros(bs, dtype=torch.int64, device=device)   # all-"low" labels

        def step():
            with torch.no_grad():
                logits, _ = clf(hazy)            # HDEN forward (its argmax is what an unlabeled run would route by)
                out, _ = router(hazy, labels)
            return out.sum()
        desc = "HDEN DenseNet121 classify + hard route (all-low labels) -> CORUN-Light forward (eval)"
    elif wl in ("complex_eval", "config5", "config5_dehaze"):   # eval-mode forward (config5: 1024x2048 frames, 4 per GPU)
        model = A.HighIntensityDehazeModel().to(device).train()
        c5 = wl.startswith("config5")
        hh, ww = (1024, 2048) if c5 else (args.height, args.width)
        bs = 4 if c5 else bs
        hazy, _ = synthetic_batch(bs, hh, ww, seed=42 + rank)
        hazy = hazy.to(device)
        with torch.no_grad():
            model(hazy[:1])          # one train-mode pass so the BN running statistics are not the identity
        model.eval()
        args.height, args.width = hh, ww

        detector = None
        if wl == "config5":      # BASELINE config 5 end to end: dehaze -> detect (models/detection.py:73-125; random-init Faster R-CNN R50-FPN)
            from adam_dehaze_amd.detection import DetectionModel, IMAGE_MEAN, IMAGE_STD
            detector = DetectionModel(num_classes=91, pretrained=False).to(device).eval()

        def step():
            with torch.no_grad():
                out = model(hazy)
                if detector is None:
                    return out.sum()
                dets = detector(out, None, pre_affine=(IMAGE_MEAN, IMAGE_STD))
                return out.sum() + sum(d["scores"].sum() for d in dets)
        desc = "CORUN-Complex eval-mode forward (BN folded into the conv epilogues)" + \
            (" -> Faster R-CNN ResNet-50 FPN detect (800 x 1333 transform, random-init weights: no mAP)" if detector is not None else "")
    elif wl == "config3" or wl == "complex_fullloss":
        model = (A.MediumIntensityDehazeModel() if wl == "config3" else A.HighIntensityDehazeModel()).to(device).train()
        crit = DehazingLoss().to(device)
        opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
        sync = None
        if world > 1:
            sync = GradientSynchronizer(list(model.parameters()), world)
            sync.broadcast_parameters(model)
            sync.install()
        hazy, clear = synthetic_batch(bs, args.height, args.width, seed=42 + rank)
        hazy, clear = hazy.to(device), clear.to(device)

        def step():
            opt.zero_grad()
            if sync is not None:
                sync.begin_step()
            loss, _ = crit(model(hazy), clear)
            loss.backward()
            if sync is not None:
                sync.finish()
            opt.step()
            return loss
        desc = ("CORUN-Medium" if wl == "config3" else "CORUN-Complex") + \
            " train fwd + DehazingLoss (L1 + 0.1 VGG16 content + 0.1 LPIPS, random-init extractors) + bwd + Adam"
    else:                    # config4: the joint step (classifier + SoftRouter over 3 branches + JointLoss + Adam)
        import yaml
        cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "config.yaml")))
        cfg["device"] = str(device)
        cfg["classifier"]["pretrained"] = False
        for k in ("classifier", "dehazing"):
            cfg[k]["checkpoint_dir"] = "/nonexistent"
        system = T.build_joint_system(cfg, world)
        system["classifier"].train()
        system["router"].train()
        hazy, clear = synthetic_batch(bs, args.height, args.width, seed=42 + rank)
        batch = {"hazy": hazy.to(device), "clear": clear.to(device),
                 "intensity": (torch.arange(bs) % 3).to(device)}

        def step():
            return T.joint_train_step(system, batch)["loss"]
        desc = "joint step: ResNet18 HDEN (train) + SoftRouter(Light, Medium, Complex) + JointLoss + duplicate-param Adam"
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        print(json.dumps({
            "metric": f"images/sec {wl}", "value": world * bs * args.steps / dt, "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "per_gpu_batch": bs, "global_batch": world * bs, "height": args.height,
                       "width": args.width, "parallelism": f"dp{world}"},
            "last_value": float(last.detach()) if torch.is_tensor(last) else None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


CONV_FAMILIES = {
    "adh_conv_wino43_forward": "conv_wino43_kernel (Winograd F(4x4,3x3) fwd + dgrad of the 3x3 s1 layers)",
    "adh_conv_wino_forward": "conv_wino_kernel (Winograd F(2x2,3x3) fwd + dgrad launches)",
    "adh_conv_wino32_forward": "conv_wino32_kernel (Winograd F(3x3,2x2) fwd + dgrad of the k4 s2 / transposed layers)",
    "adh_conv_forward": "conv_rows_kernel / conv_igemm_kernel (direct fwd + dgrad launches: stems, heads, ragged shapes)",
    "adh_conv_wgrad_wino": "conv_wgrad_rows_kernel<WINO> (Winograd-domain weight gradients of 3x3 s1)",
    "adh_conv_wgrad_wino32": "conv_wgrad32_kernel (F(3x3,2x2)-domain weight gradients of the k4 s2 / transposed layers)",
    "adh_conv_wgrad": "conv_wgrad_rows_kernel / conv_wgrad_kernel (direct weight gradients)",
    "adh_conv_wgrad_wino43": "conv_wgrad_wino43_kernel (F(4x4,3x3)-domain weight gradients)",
    "adh_conv_stem_forward": "conv_stem_fwd_kernel (7x7 stem, 16x16x4 tiles)",
    "adh_conv_wgrad_stem": "conv_wgrad_stem_kernel (7x7 stem weight gradient, 16x16x4 tiles)",
    "adh_conv_wgrad_small": "conv_wgrad_small_kernel / conv_wgrad_fewout_kernel (few-channel 3x3 weight gradients)",
}
# HBM-bound families: `work` of these launches is algorithmic BYTES (engine.py), judged against 8 TB/s
HBM_FAMILIES = {
    "adh_bn_apply": "bn_apply (train-mode normalise + residual + ReLU)",
    "adh_bn_bwd_reduce": "bn_bwd_reduce (sum g, sum g*xhat)",
    "adh_bn_bwd_apply": "bn_bwd_apply (gradient wrt the conv output, residual gradient)",
    "adh_cbam_pool": "cbam_pool (global avg/max pool)",
    "adh_cbam_spatial_stats": "cbam_spatial_stats (x*ca -> mean/max over C)",
    "adh_cbam_apply": "cbam_apply (7x7 conv + sigmoid + scale)",
    "adh_cbam_bwd_a": "cbam_bwd_a", "adh_cbam_bwd_c": "cbam_bwd_c", "adh_cbam_bwd_e": "cbam_bwd_e",
}
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def conv_family_table(ks):
    zero = {"launches": 0, "seconds": 0.0, "work": 0.0, "work_exec": 0.0}
    per = {}
    for key, label in CONV_FAMILIES.items():
        k = ks.get(key, zero)
        per[key] = {"kernel": label, "launches": k["launches"], "seconds": k["seconds"],
                    "avg_launch_ms": 1e3 * k["seconds"] / max(1, k["launches"]),
                    # executed MFMA FLOP/s (F(2x2,3x3) / F(3x3,2x2) execute 4/9 of the direct algorithm's FLOPs, F(4x4,3x3) 1/4)
                    "achieved": k["work_exec"] / k["seconds"] / 1e12 if k["seconds"] > 0 else 0.0,
                    # FLOP/s of the direct-convolution algorithm this launch replaces (2*MAC of conv/convT)
                    "algorithmic": k["work"] / k["seconds"] / 1e12 if k["seconds"] > 0 else 0.0,
                    "flops_exec_per_launch_avg": k["work_exec"] / max(1, k["launches"])}
    return per


def hbm_family_table(ks, steps):
    out = []
    for key, label in HBM_FAMILIES.items():
        k = ks.get(key)
        if not k or k["seconds"] <= 0:
            continue
        gbps = k["work"] / k["seconds"] / 1e9
        out.append({"kernel": label, "launches": k["launches"], "ms_per_step": 1e3 * k["seconds"] / max(1, steps),
                    "algorithmic_gbytes_per_launch": k["work"] / k["launches"] / 1e9, "achieved_gbps": gbps,
                    "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS})
    return out


# algorithmic work of CORUN-Complex eval forward per 512x1024 image (scales with H*W)
COMPLEX_FWD_GFLOP_PER_IMAGE = 2014.8
COMPLEX_FWD_GBYTES_PER_IMAGE = 8.14


def forward_eval(model, hazy, args, H):
    """The north-star's own target line: CORUN-Complex eval-mode forward at the headline batch, with its roofline
    fractions -- algorithmic FLOP/s against the fp32 MFMA peak (may exceed 1: Winograd executes 1/4 .. 4/9 of the
    direct MACs), algorithmic bytes/s against 8 TB/s (SURVEY 8d figures), and the per-family executed rates."""
    was_training = model.training
    model.eval()
    with torch.no_grad():
        model(hazy)                      # warm-up (packs the eval-mode weights)
        timer = H.KernelTimer(set(CONV_FAMILIES) | set(HBM_FAMILIES))
        H.TIMER = timer
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            model(hazy)
        e1.record()
        torch.cuda.synchronize()
        H.TIMER = None
    model.train(was_training)
    ms = e0.elapsed_time(e1) / reps
    n = hazy.shape[0]
    area = (args.height * args.width) / (512.0 * 1024.0)
    flop = COMPLEX_FWD_GFLOP_PER_IMAGE * 1e9 * n * area
    byts = COMPLEX_FWD_GBYTES_PER_IMAGE * 1e9 * n * area
    ks = timer.summary()
    per = conv_family_table(ks)
    return {"workload": "CORUN-Complex eval-mode forward (BN folded), same batch", "ms": ms, "images_per_sec": n / (ms * 1e-3),
            "algorithmic_tflops": flop / (ms * 1e-3) / 1e12, "frac_of_f32_mfma_peak_algorithmic": flop / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
            "algorithmic_gbps": byts / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "bound": "mfma (fp32): 248 FLOP/B against a 20 FLOP/B ridge, SURVEY 8d",
            "kernels": {k: {"ms_per_pass": 1e3 * v["seconds"] / reps, "launches_per_pass": v["launches"] // reps,
                            "achieved_exec_tflops": v["achieved"], "frac_of_f32_mfma_peak": v["achieved"] / F32_MFMA_PEAK_TFLOPS,
                            "algorithmic_tflops": v["algorithmic"]} for k, v in per.items() if v["launches"]},
            "hbm_kernels": hbm_family_table(ks, reps)}


BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak


def split_bf16x3(step, model, args, H, sample, device):
    """The OPT-IN contraction (ADH_CONTRACT=bf16x3): the same step with the F(4x4,3x3) forward / data-gradient
    launches on v_mfma_f32_32x32x16_bf16 over exact three-plane bf16 splits of both operands.  Second object next to the fp32
    headline, never the headline: its own ms/step, per-family times, the family's fp32-equivalent and executed bf16 MFMA
    rates, and its own parity numbers on the headline frame."""
    import adam_dehaze_amd.engine as E
    old = E.CONTRACT
    E.CONTRACT = "bf16x3"
    try:
        for _ in range(max(1, min(args.warmup, 2))):
            step()
        timer = H.KernelTimer(set(CONV_FAMILIES) | set(HBM_FAMILIES))
        H.TIMER = timer
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        H.TIMER = None
        per = conv_family_table(timer.summary())
        fam = per["adh_conv_wino43_forward"]
        parity = headline_parity(sample, device) if sample is not None else None
        return {"switch": "ADH_CONTRACT=bf16x3 (engine.CONTRACT)", "ms_per_step": 1e3 * dt / args.steps,
                "images_per_sec": args.batch * args.steps / dt,
                "families_ms_per_step": {k: 1e3 * v["seconds"] / args.steps for k, v in per.items() if v["launches"]},
                "conv_wino43_kernel": {"launches": fam["launches"], "avg_launch_ms": fam["avg_launch_ms"],
                                       "fp32_equivalent_executed_tflops": fam["achieved"],
                                       "algorithmic_tflops": fam["algorithmic"],
                                       # six bf16 MFMA terms per fp32-equivalent product
                                       "executed_bf16_mfma_tflops": 6.0 * fam["achieved"],
                                       "frac_of_bf16_mfma_peak": 6.0 * fam["achieved"] / BF16_MFMA_PEAK_TFLOPS},
                "psnr_db_vs_cpu_oracle": parity["psnr_db"] if parity else None,
                "max_abs_vs_cpu_oracle": parity["max_abs"] if parity else None,
                "abs_loss_diff_vs_cpu_oracle": parity["abs_loss_diff"] if parity else None}
    finally:
        H.TIMER = None
        E.CONTRACT = old


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forward-eval", action="store_true")
    ap.add_argument("--workload", default="complex",
                    choices=["complex", "complex_fullloss", "complex_eval", "config2", "config3", "config4", "config5", "config5_dehaze"],
                    help="complex = headline (BASELINE.json metric); config2/3/4 = the other BASELINE.json configs")
    ap.add_argument("--no-adam", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="skip the opt-in bf16 x 3 contraction leg (split_bf16x3)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    import torch.distributed as dist
    if local_rank >= torch.cuda.device_count():   # rehearsal of the N>1 path on a single-GPU box
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # ADH_DIST_FORCE=1: a world of ONE rank initialises the process group and issues every collective of the N > 1 path (bucket
    # all-reduces from inside backward, the self-check's gathers): the rehearsal of the RCCL branches on a one-GPU box
    forced = world == 1 and os.environ.get("ADH_DIST_FORCE", "0") == "1"
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        backend = os.environ.get("ADH_DIST_BACKEND", "nccl")   # nccl = RCCL over xGMI; gloo only for single-GPU rehearsal
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import adam_dehaze_amd as A
    from adam_dehaze_amd import _hip as H
    from adam_dehaze_amd.loss import l1_loss
    from adam_dehaze_amd.optim import Adam
    from adam_dehaze_amd.parallel import GradientSynchronizer

    torch.manual_seed(42)   # identical replicas on every rank
    if args.workload != "complex":
        return other_workloads(args, rank, world, device)
    model = A.HighIntensityDehazeModel().to(device).train()
    opt = None if args.no_adam else Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
    sync = None
    if world > 1 or forced:
        # flat gradient buckets, all-reduced from inside Engine.backward() as they fill (parallel.py)
        sync = GradientSynchronizer(list(model.parameters()), world)
        sync.broadcast_parameters(model)
        sync.install()

    hazy, clear = synthetic_batch(args.batch, args.height, args.width, seed=42 + rank)
    hazy, clear = hazy.to(device), clear.to(device)

    def step():
        for p in model.parameters():
            p.grad = None
        if sync is not None:
            sync.begin_step()
        out = model(hazy)
        loss = l1_loss(out, clear)
        loss.backward()
        if sync is not None:
            sync.finish()
        if opt is not None:
            opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    # world > 1: ONE checked step behind the warm-up, outside the timed region -- every rank verifies that the synchronized
    # gradient buckets agree across ranks, equal the mean of the ranks' local buckets, and that the parameters after the Adam
    # step are bit-equal (parallel.GradientSynchronizer.selfcheck_result).  The first RCCL execution of this code is the
    # driver's scaling run: it must not be able to print a throughput over diverged replicas.
    selfcheck = None
    if sync is not None:
        if args.warmup == 0:
            step()      # (the bucket layout is rebuilt in the very first step: check the one after)
        sync.begin_selfcheck()
        step()
        selfcheck = sync.selfcheck_result(list(model.parameters()))
    # timed region: exactly K steps between barrier + synchronize on both sides
    timer = H.KernelTimer(set(CONV_FAMILIES) | set(HBM_FAMILIES))
    H.TIMER = timer
    if world > 1 or forced:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1 or forced:
        dist.barrier()
    dt = time.perf_counter() - t0
    H.TIMER = None
    if world > 1 or forced:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        per = conv_family_table(timer.summary())
        dom = max(per, key=lambda k_: per[k_]["seconds"])   # dominant kernel = most time inside the timed region
        d0 = per[dom]
        traffic, traffic_src = pmc_traffic(dom)
        roofline = {"bound": "mfma", "kernel": d0["kernel"], "achieved": d0["achieved"], "peak": F32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": d0["achieved"] / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE)", "traffic_source": traffic_src,
                    "launches": d0["launches"], "avg_launch_ms": d0["avg_launch_ms"],
                    "flops_per_launch_avg": d0["flops_exec_per_launch_avg"],
                    "algorithmic_tflops": d0["algorithmic"],
                    "note": "achieved = executed MFMA FLOPs / HIP-event time over the timed region; traffic comes from the "
                            "committed rocprofv3 --pmc passes over this same command (rocprofv3 cannot run inside bench.py)",
                    "other_kernels": {k_: {kk: v for kk, v in per[k_].items() if kk != "flops_exec_per_launch_avg"}
                                      for k_ in per if k_ != dom},
                    "conv_seconds_per_step": sum(v["seconds"] for v in per.values()) / max(1, args.steps)}
        hbm_kernels = hbm_family_table(timer.summary(), args.steps)
        psnr_small, max_abs_small = psnr_check(model, device)
        cpu_line, sample, parity = None, None, None
        if not args.no_cpu_baseline and world == 1:
            cpu_line, sample = cpu_baseline(args.height, args.width, host_threads())
            parity = headline_parity(sample, device)
        backend = dist.get_backend() if (world > 1 or forced) else None
        coll = "" if (world == 1 and not forced) else (" + RCCL grad all-reduce (flat buckets, overlapped with backward)" if backend == "nccl"
                                      else f" + {backend} grad all-reduce (single-GPU rehearsal backend, NOT RCCL)")
        result = {
            "metric": "images/sec CORUN-Complex fwd+bwd 512x1024 bs=8; PSNR vs CPU ref",
            "value": world * args.batch * args.steps / dt,
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CORUN-Complex (HighIntensityDehazeModel base 96) train-mode fwd + L1 + bwd"
                                   + ("" if args.no_adam else " + Adam") + coll,
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch,
                       "height": args.height, "width": args.width, "parallelism": f"dp{world}",
                       "dist_backend": backend},
            # PSNR vs the CPU reference on the HEADLINE frame (train mode, 512x1024); the small eval frame of rounds 1-3 stays
            # under its own key (and is what these keys fall back to when the CPU leg is skipped)
            "psnr_db_vs_cpu_oracle": parity["psnr_db"] if parity else (psnr_small if psnr_small != float("inf") else 999.0),
            "max_abs_vs_cpu_oracle": parity["max_abs"] if parity else max_abs_small,
            "abs_loss_diff_vs_cpu_oracle": parity["abs_loss_diff"] if parity else None,
            "parity_frame": parity["frame"] if parity else "1x3x128x256 eval-mode forward (CPU leg skipped)",
            "parity_detail": parity,
            "psnr_db_small_eval_frame": psnr_small if psnr_small != float("inf") else 999.0,
            "max_abs_small_eval_frame": max_abs_small,
            "loss": float(loss.detach()),
            "roofline": roofline,
            "hbm_kernels": hbm_kernels,
        }
        if not args.no_forward_eval:
            result["forward_eval"] = forward_eval(model, hazy, args, H)
        if cpu_line is not None:
            result["cpu_baseline"] = cpu_line
        if selfcheck is not None:
            result["ddp_selfcheck"] = selfcheck
        if world == 1 and not args.no_split:
            result["split_bf16x3"] = split_bf16x3(step, model, args, H, sample, device)
        print(json.dumps(result))
    if selfcheck is not None and not selfcheck["ok"]:
        if world > 1 or forced:
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit(f"ddp_selfcheck FAILED on rank {rank}: {selfcheck}")
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
