"""Pin the CPU oracle (oracle/ref_cpu.py) against fixtures produced by the REFERENCE itself
(tools/gen_golden.py imported /root/reference's torch-only modules in the build container).
CPU-only; runs everywhere."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import adam_dehaze_amd as A
from oracle import ref_cpu as R
from tests._util import load_golden, sub_sd, t, max_abs, rel_err

TOL = 2e-6       # same ATen kernels, same order => essentially exact
GTOL = 2e-5      # gradients (relative to the tensor's max)


def _grads(out, gout, x, sd, names):
    (out * gout).sum().backward()
    return x.grad, {k: sd[k].grad for k in names}


def _prep(rec, train):
    sd = sub_sd(rec, "sd.")
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    x = t(rec["x"]).requires_grad_(True)
    return sd, x


BLOCK_CASES = [
    ("convblock_k3_c16", dict(stride=1, padding=1)),
    ("convblock_k3_c32", dict(stride=1, padding=1)),
    ("convblock_k4s2_c16", dict(stride=2, padding=1)),
    ("convblock_k4s2_c32", dict(stride=2, padding=1)),
    ("convblock_k1_c16", dict(stride=1, padding=0)),
    ("convblock_k1_c32", dict(stride=1, padding=0)),
    ("convblock_nobn_noact_c16", dict(stride=1, padding=1, use_bn=False, act=None)),
    ("convblock_nobn_noact_c32", dict(stride=1, padding=1, use_bn=False, act=None)),
    ("convblock_k7_stem", dict(stride=1, padding=3)),
    ("convblock_k3_stem", dict(stride=1, padding=1)),
]


@pytest.mark.parametrize("name,kw", BLOCK_CASES)
@pytest.mark.parametrize("train", [False, True])
def test_conv_block(name, kw, train):
    rec = load_golden(name)
    sd, x = _prep(rec, train)
    out = R.conv_block(x, sd, "", training=train, **kw)
    tag = "train" if train else "eval"
    assert max_abs(out, rec["out_" + tag]) < TOL
    names = [k[len("gp_%s." % tag):] for k in rec if k.startswith("gp_%s." % tag)]
    gx, gp = _grads(out, t(rec["gout"]), x, sd, names)
    assert rel_err(gx, rec["gx_" + tag]) < GTOL
    for k in names:
        assert rel_err(gp[k], rec["gp_%s.%s" % (tag, k)]) < GTOL, k
    if train and kw.get("use_bn", True):
        for k in ("block.1.running_mean", "block.1.running_var", "block.1.num_batches_tracked"):
            assert max_abs(sd[k], rec["sd_after_train." + k]) < TOL, k


@pytest.mark.parametrize("C", [16, 32])
@pytest.mark.parametrize("train", [False, True])
def test_residual_block(C, train):
    rec = load_golden(f"resblock_c{C}")
    sd, x = _prep(rec, train)
    out = R.residual_block(x, sd, "", training=train)
    tag = "train" if train else "eval"
    assert max_abs(out, rec["out_" + tag]) < TOL
    names = [k[len("gp_%s." % tag):] for k in rec if k.startswith("gp_%s." % tag)]
    gx, gp = _grads(out, t(rec["gout"]), x, sd, names)
    assert rel_err(gx, rec["gx_" + tag]) < GTOL
    for k in names:
        assert rel_err(gp[k], rec["gp_%s.%s" % (tag, k)]) < GTOL, k


@pytest.mark.parametrize("C", [16, 32])
def test_attention_block(C):
    rec = load_golden(f"attention_c{C}")
    sd, x = _prep(rec, False)
    out = R.attention_block(x, sd, "")
    assert max_abs(out, rec["out_eval"]) < TOL
    assert max_abs(out, rec["out_train"]) < TOL
    names = [k[len("gp_eval."):] for k in rec if k.startswith("gp_eval.")]
    gx, gp = _grads(out, t(rec["gout"]), x, sd, names)
    assert rel_err(gx, rec["gx_eval"]) < GTOL
    for k in names:
        assert rel_err(gp[k], rec["gp_eval." + k]) < GTOL, k


BRANCHES = ["light_b8", "lowint_b8", "medium_b8", "medium_b8_odd", "corun_b8", "high_b16", "high_b16_odd",
            "dual_b16"]


@pytest.mark.parametrize("name", BRANCHES)
def test_branch_eval_and_train(name):
    rec = load_golden(name)
    fwd = R.BRANCH_FORWARD[str(rec["class_name"])]
    sd = sub_sd(rec, "sd.")
    x = t(rec["x"])
    with torch.no_grad():
        out = fwd(x, sd, training=False)
    assert max_abs(out, rec["out_eval"]) < TOL
    # train mode: output, L1 loss, parameter gradients, BN buffer updates
    sd = sub_sd(rec, "sd.")
    pnames = [k[len("gp_train."):] for k in rec if k.startswith("gp_train.")]
    for k in pnames:
        sd[k].requires_grad_(True)
    out = fwd(x, sd, training=True)
    assert max_abs(out, rec["out_train"]) < TOL
    loss = F.l1_loss(out, t(rec["target"]))
    assert abs(float(loss.detach()) - float(rec["l1"])) < 1e-6
    loss.backward()
    for k in pnames:
        g = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        ref = t(rec["gp_train." + k])
        scale = max(float(ref.abs().max()), 1e-8)
        # (biases feeding a train-mode BN have an exactly-zero true gradient: only noise ~1e-9)
        assert float((g - ref).abs().max()) < 1e-3 * scale + 1e-7, k
    for k, v in sub_sd(rec, "sd_after_train.").items():
        if "running" in k or "num_batches" in k:
            assert max_abs(sd[k], v) < TOL, k


def test_routers():
    rec = load_golden("routers")
    x = t(rec["x"])
    logits = t(rec["logits"])
    sds = {n: sub_sd(rec, f"sd.models.{n}.") for n in ("low", "medium", "high")}
    fns = {"low": R.lightweight_forward, "medium": R.medium_forward, "high": R.high_forward}
    with torch.no_grad():
        outs = {n: fns[n](x, sds[n], training=False) for n in fns}
    for n in outs:
        assert max_abs(outs[n], rec["soft_ind." + n]) < TOL
    blended, w = R.soft_route(outs, logits, 0.5)
    assert max_abs(w, rec["soft_weights"]) < 1e-7
    assert max_abs(blended, rec["soft_out"]) < TOL
    # gradient wrt logits through the blend
    lg = logits.clone().requires_grad_(True)
    b2, _ = R.soft_route(outs, lg, 0.5)
    (b2 * t(rec["soft_gout"])).sum().backward()
    assert rel_err(lg.grad, rec["soft_glogits"]) < GTOL
    # hard routing: indices bit-exact (ties -> first index), outputs equal
    idx = R.hard_route_indices(logits)
    assert idx.dtype == torch.int64
    assert np.array_equal(idx.numpy(), rec["hard_idx_from_logits"])
    with torch.no_grad():
        hard = R.hard_route(x, idx, {n: (lambda v, n=n: fns[n](v, sds[n], training=False)) for n in fns})
    assert max_abs(hard, rec["hard_out"]) < TOL
    # classifier-driven paths with the stub classifier's stored logits/features
    clf_logits = t(rec["clf_logits"])
    assert np.array_equal(R.hard_route_indices(clf_logits).numpy(), rec["hard_idx_clf"])
    b3, w3 = R.soft_route(outs, clf_logits, 0.5)
    assert max_abs(b3, rec["soft_out_clf"]) < TOL
    gate_sd = sub_sd(rec, "sd.gate.")
    gw = R.gate_network(t(rec["clf_feats"]), gate_sd, prefix="")
    assert max_abs(gw, rec["gated_weights"]) < 1e-6
    gated = sum(gw[:, i].view(-1, 1, 1, 1) * outs[n] for i, n in enumerate(("low", "medium", "high")))
    assert max_abs(gated, rec["gated_out"]) < TOL


def test_adam_duplicate_params():
    """train_joint.py:81-89 lists every branch parameter twice => two updates per step."""
    rec = load_golden("adam_dup")
    wd, ws = t(rec["w_dup0"]), t(rec["w_single0"])
    md, vd = torch.zeros_like(wd), torch.zeros_like(wd)
    ms, vs = torch.zeros_like(ws), torch.zeros_like(ws)
    for step in range(3):
        R.adam_step(wd, t(rec[f"g_dup{step}"]), md, vd, step=2 * step, lr=5e-5, weight_decay=1e-4, repeats=2)
        R.adam_step(ws, t(rec[f"g_single{step}"]), ms, vs, step=step, lr=5e-5, weight_decay=1e-4, repeats=1)
        assert max_abs(wd, rec[f"w_dup{step + 1}"]) < 1e-7
        assert max_abs(ws, rec[f"w_single{step + 1}"]) < 1e-7


def test_third_party_restatements_shapes():
    """UNPINNED parts: only structural checks (shapes, finite values) with seeded random weights."""
    g = torch.Generator().manual_seed(0)
    from tests._thirdparty_init import resnet18_sd, densenet121_sd, vgg16_sd, lpips_alex_sd
    x = torch.rand(2, 3, 64, 96, generator=g)
    logits, feats = R.classifier_forward(x, resnet18_sd(0), "resnet18")
    assert logits.shape == (2, 3) and feats.shape == (2, 512) and torch.isfinite(logits).all()
    logits, feats = R.classifier_forward(x, densenet121_sd(0), "densenet121")
    assert logits.shape == (2, 3) and feats.shape == (2, 1024) and torch.isfinite(logits).all()
    y = torch.rand(2, 3, 64, 96, generator=g)
    c = R.content_loss(x, y, vgg16_sd(0))
    assert c.dim() == 0 and torch.isfinite(c) and c > 0
    p = R.perceptual_loss(x, y, lpips_alex_sd(0))
    assert p.shape == (2, 1, 1, 1) and torch.isfinite(p).all()
    assert float(R.perceptual_loss(x, x, lpips_alex_sd(0)).abs().max()) == 0.0


def test_psnr_and_synthetic_batch():
    hazy, clear, labels = R.synthetic_batch(3, 32, 64, seed=42)
    assert hazy.shape == clear.shape == (3, 3, 32, 64) and labels.tolist() == [0, 1, 2]
    assert 0.0 <= float(hazy.min()) and float(hazy.max()) <= 1.0
    h2, c2, _ = R.synthetic_batch(3, 32, 64, seed=42)
    assert torch.equal(hazy, h2) and torch.equal(clear, c2)
    assert R.psnr(clear, clear) == float("inf")
    a = torch.zeros(1, 3, 4, 4)
    b = torch.full((1, 3, 4, 4), 0.1)
    assert abs(R.psnr(a, b) - 20.0) < 1e-4


def test_adam_duplicate_params_foreach_form():
    """torch >= 2.0's CUDA default (_multi_tensor_adam) with duplicated list entries: pinned by
    adam_dup_foreach.npz (tools/gen_golden_adam.py, torch.optim.Adam(foreach=True)); differs from the sequential form."""
    rec = load_golden("adam_dup_foreach")
    seq = load_golden("adam_dup")
    assert np.array_equal(rec["w_dup0"], seq["w_dup0"]) and np.array_equal(rec["g_dup1"], seq["g_dup1"])   # same scenario
    st = {k: (t(rec[f"w_{k}0"]), torch.zeros_like(t(rec[f"w_{k}0"])), torch.zeros_like(t(rec[f"w_{k}0"])), r)
          for k, r in (("dup", 2), ("single", 1), ("tri", 3))}
    for step in range(3):
        for k, (w, m, v, r) in st.items():
            R.adam_step_foreach(w, t(rec[f"g_{k}{step}"]), m, v, step=r * step, lr=5e-5, weight_decay=1e-4, repeats=r)
            assert max_abs(w, rec[f"w_{k}{step + 1}"]) < 1e-7, (k, step)
    # the two semantics are really different (~8 % of an update) -- and identical without duplicates
    assert max_abs(t(rec["w_dup3"]), seq["w_dup3"]) > 1e-6
    assert max_abs(t(rec["w_single3"]), seq["w_single3"]) < 1e-7


def test_image_metrics_restatements():
    """PSNR / SSIM restatements (UNPINNED: skimage absent) against first-principles evaluations: the float64 window
    definition of SSIM on every interior pixel, and PSNR of a constant offset."""
    g = torch.Generator().manual_seed(1)
    a = torch.rand(2, 3, 18, 25, generator=g)
    b = (a + 0.05 * torch.randn(a.shape, generator=g)).clamp(0, 1)
    got = R.ssim_per_image(b, a)
    for i in range(2):
        g1, g2 = a[i].mean(0).double().numpy(), b[i].mean(0).double().numpy()
        vals = []
        for y in range(3, 15):
            for x in range(3, 22):
                w1, w2 = g1[y - 3:y + 4, x - 3:x + 4], g2[y - 3:y + 4, x - 3:x + 4]
                ux, uy = w1.mean(), w2.mean()
                vx, vy = w1.var(ddof=1), w2.var(ddof=1)
                vxy = ((w1 - ux) * (w2 - uy)).sum() / 48
                vals.append((2 * ux * uy + 1e-4) * (2 * vxy + 9e-4) / ((ux * ux + uy * uy + 1e-4) * (vx + vy + 9e-4)))
        assert abs(float(got[i]) - float(np.mean(vals))) < 1e-6      # float32 channel mean vs float64: ~1e-8
    assert float(R.ssim_per_image(a, a).min()) > 1 - 1e-12
    c = torch.full((1, 3, 8, 8), 0.25)
    assert abs(float(R.psnr_per_image(c + 0.1, c)) - 20.0) < 1e-5


def test_fog_restatement_properties():
    """apply_fog (UNPINNED restatement of helpers.py:241-258): beta = 0 leaves the image, A = 1 with huge beta
    saturates to 1, and the centre of the depth map (x=.5, y=.2) is the least foggy pixel."""
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 21, 41, generator=g) * 0.5
    assert torch.equal(R.apply_fog(x, [0.0, 0.0], [0.7, 0.9]), x)
    assert float(R.apply_fog(x, [1e3, 1e3], [1.0, 1.0]).min()) == 1.0
    f = R.apply_fog(torch.zeros(1, 3, 21, 41), [1.0], [1.0])[0, 0]
    yy, xx = np.unravel_index(int(f.argmin()), f.shape)
    assert (yy, xx) == (4, 20)     # y = 0.2 * 20, x = 0.5 * 40


FULLWIDTH = [("light", R.lightweight_forward, lambda: A.LightweightDehazeModel(base_channels=32, n_blocks=3)),
             ("medium", R.medium_forward, lambda: A.MediumIntensityDehazeModel(base_channels=64)),
             ("high", R.high_forward, lambda: A.HighIntensityDehazeModel(base_channels=96))]


@pytest.mark.parametrize("name,fwd,ctor", FULLWIDTH, ids=[c[0] for c in FULLWIDTH])
def test_fullwidth_default_models_vs_reference_held_outputs(name, fwd, ctor):
    """BASELINE config 1: the DEFAULT full-width Light / Medium / Complex (low_intensity.py:33-45,
    medium_intensity.py:78-117, high_intensity.py:92-138) at 1x3x256x256, seed 42, eval mode.  The reference computed
    `out_patch`, `out_mean`, `out_absmax` (tools/gen_golden.py section (v)); the oracle must reproduce them from the
    seeded constructor (whose parameters are SHA-256-identical to the reference's, tests/test_host_cpu.py)."""
    rec = load_golden("fullwidth_summaries")
    torch.manual_seed(42)
    sd = {k: v.detach().clone() for k, v in ctor().state_dict().items()}
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(42))
    with torch.no_grad():
        out = fwd(x, sd, training=False)
    assert max_abs(out[0, :, 100:108, 100:108], rec[name + ".out_patch"]) < TOL
    assert abs(float(out.double().mean()) - float(rec[name + ".out_mean"])) < 1e-7
    assert abs(float(out.abs().max()) - float(rec[name + ".out_absmax"])) < TOL


def test_detector_restatement_properties():
    """The detector oracle is UNPINNED (torchvision absent): first-principles checks of the two pieces whose definitions are
    easy to get subtly wrong.  RoIAlign of a constant map is the constant, and of the ramp f(y, x) = x the bin centres'
    x (bilinear interpolation is exact on linear functions; 2 x 2 samples are symmetric about the bin centre).  Grouped
    NMS: kept boxes of one group overlap by at most the threshold, and every dropped box overlaps a kept box of its group
    with a higher score by more than the threshold."""
    feat = torch.zeros(1, 2, 20, 30)
    feat[0, 0] = 3.5
    feat[0, 1] = torch.arange(30.0).view(1, 30).expand(20, 30)
    rois = torch.tensor([[0.0, 4.0, 3.0, 18.0, 15.0], [0.0, 8.0, 8.0, 29.0, 19.0]])       # feature-map units at scale 1
    out = R.det_roi_align(feat, rois, 1.0)
    assert float((out[:, 0] - 3.5).abs().max()) < 1e-6
    for r in range(2):
        x1, x2 = float(rois[r, 1]), float(rois[r, 3])
        centres = x1 + (torch.arange(7.0) + 0.5) * (x2 - x1) / 7
        assert float((out[r, 1] - centres.view(1, 7)).abs().max()) < 1e-5
    g = torch.Generator().manual_seed(4)
    c = torch.rand(200, 2, generator=g) * 100
    boxes = torch.cat([c, c + torch.rand(200, 2, generator=g) * 40 + 1], dim=1)
    scores, groups = torch.rand(200, generator=g), torch.randint(0, 3, (200,), generator=g)
    keep = R.det_nms(boxes, scores, groups, 0.4)
    kept = set(keep.tolist())

    def iou(i, j):
        a, b = boxes[i], boxes[j]
        iw = max(0.0, float(min(a[2], b[2]) - max(a[0], b[0])))
        ih = max(0.0, float(min(a[3], b[3]) - max(a[1], b[1])))
        inter = iw * ih
        return inter / (float((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1])) - inter)
    assert all(scores[keep[i]] >= scores[keep[i + 1]] for i in range(len(keep) - 1))
    for i in kept:
        for j in kept:
            if i < j and groups[i] == groups[j]:
                assert iou(i, j) <= 0.4
    for i in set(range(200)) - kept:
        assert any(groups[i] == groups[j] and scores[j] >= scores[i] and iou(i, j) > 0.4 for j in kept)
    # anchors, decode: a zero delta returns the anchor; the size clamp holds
    a = torch.tensor([[10.0, 20.0, 50.0, 80.0]])
    assert torch.allclose(R.det_decode(torch.zeros(1, 4), a, (1.0, 1.0, 1.0, 1.0)), a)
    big = R.det_decode(torch.tensor([[0.0, 0.0, 100.0, 100.0]]), a, (1.0, 1.0, 1.0, 1.0))
    assert abs(float(big[0, 2] - big[0, 0]) - 40.0 * 1000.0 / 16) < 1e-2
