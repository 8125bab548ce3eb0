"""GPU parity of the pieces around the hot path: device PSNR / SSIM / LPIPS metrics and the
results-JSON schema, the synthetic fog kernel, the multi-tensor Adam in both duplicate-parameter semantics, the packed
weight cache.  All through the C ABI; oracle = oracle/ref_cpu.py (skimage / cv2-dependent reference code is restated
there: unpinned, see its docstrings) and the torch-generated Adam fixtures."""
import json
import warnings

import numpy as np
import pytest
import torch

import adam_dehaze_amd as A
from adam_dehaze_amd import _hip as H
from adam_dehaze_amd import data as D
from adam_dehaze_amd import metrics as M
from adam_dehaze_amd.optim import Adam
from oracle import ref_cpu as R
from tests._util import load_golden, max_abs, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ---------------------------------------------------------------------------------------------------------------------
# metrics (evaluation/metrics.py:13-124, train_joint.py:214-227)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (3, 3, 7, 7), (1, 3, 45, 131), (2, 3, 33, 32)])
def test_psnr_ssim_vs_oracle(shape):
    g = torch.Generator().manual_seed(shape[2] * 7 + shape[3])
    target = torch.rand(shape, generator=g)
    pred = (target + 0.08 * torch.randn(shape, generator=g)).clamp(0, 1)
    got = M.calculate_image_metrics(pred.to(DEV), target.to(DEV))
    assert set(got) == {"psnr", "ssim"} and got["psnr"].shape == (shape[0],)
    assert max_abs(got["psnr"], R.psnr_per_image(pred, target).float()) < 1e-4      # dB
    assert max_abs(got["ssim"], R.ssim_per_image(pred, target).float()) < 1e-5
    same = M.calculate_image_metrics(target.to(DEV), target.to(DEV))
    assert bool(torch.isinf(same["psnr"]).all()) and float((same["ssim"] - 1).abs().max()) < 1e-6


def test_ssim_rejects_images_smaller_than_the_window():
    x = torch.rand(1, 3, 6, 20, device=DEV)
    with pytest.raises(ValueError, match="win_size exceeds image extent"):
        M.ssim_batch(x, x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.psnr_batch(x.cpu(), x.cpu())


def test_psnr_ssim_full_size_properties():
    """8 x 3 x 512 x 1024: per-image values equal the single-image launches (different grid), PSNR of a constant offset
    is exact, SSIM is symmetric and 1 on identical images."""
    g = torch.Generator(device=DEV).manual_seed(3)
    target = torch.rand(8, 3, 512, 1024, generator=g, device=DEV)
    pred = (target + 0.05 * torch.randn(target.shape, generator=g, device=DEV)).clamp(0, 1)
    p, s = M.psnr_batch(pred, target), M.ssim_batch(pred, target)
    for n in (0, 7):
        assert float((M.psnr_batch(pred[n:n + 1], target[n:n + 1]) - p[n]).abs()) < 1e-5
        assert float((M.ssim_batch(pred[n:n + 1], target[n:n + 1]) - s[n]).abs()) < 1e-6
    assert float((M.ssim_batch(target, pred) - s).abs().max()) < 1e-6
    off = M.psnr_batch(torch.full_like(target, 0.5), torch.full_like(target, 0.4))
    assert float((off - 20.0).abs().max()) < 1e-3
    # a 40 x 64 window of image 5 against the oracle's definition evaluated on the crop (interior windows only)
    crop_p, crop_t = pred[5:6, :, 100:140, 200:264].cpu(), target[5:6, :, 100:140, 200:264].cpu()
    assert abs(float(M.ssim_batch(crop_p.to(DEV), crop_t.to(DEV))) - float(R.ssim_per_image(crop_p, crop_t))) < 1e-5


def test_image_quality_metrics_class_and_json_schema(tmp_path):
    from tests._thirdparty_init import lpips_alex_sd
    from adam_dehaze_amd.loss import PerceptualLoss
    g = torch.Generator().manual_seed(5)
    target = torch.rand(5, 3, 64, 64, generator=g)
    pred = (target + 0.1 * torch.randn(target.shape, generator=g)).clamp(0, 1)
    lp = PerceptualLoss()
    lsd = lpips_alex_sd(0)
    lp.load_state_dict(lsd, strict=True)
    iq = M.ImageQualityMetrics(device=DEV, lpips_fn=lp.to(DEV))
    cats = ["low_intensity", "high_intensity", "low_intensity", "medium_intensity", "low_intensity"]
    iq.add_batch(pred[:3].to(DEV), target[:3].to(DEV), cats[:3])
    iq.add_sample(pred[3].to(DEV), target[3].to(DEV), cats[3])         # metrics.py:47 signature
    iq.add_sample(pred[4].to(DEV), target[4].to(DEV), cats[4])
    avg = iq.compute_averages()
    ps, ss = R.psnr_per_image(pred, target), R.ssim_per_image(pred, target)
    lpv = R.perceptual_loss(pred, target, lsd).reshape(-1)
    for cat in set(cats):
        idx = [i for i, c in enumerate(cats) if c == cat]
        assert avg[cat]["samples"] == len(idx)
        assert abs(avg[cat]["psnr"] - float(ps[idx].mean())) < 1e-4
        assert abs(avg[cat]["ssim"] - float(ss[idx].mean())) < 1e-5
        assert abs(avg[cat]["lpips"] - float(lpv[idx].mean())) < 1e-3 * max(1.0, float(lpv.abs().max()))
    out = tmp_path / "res" / "joint_model_results.json"
    iq.save_results(str(out))
    saved = json.load(open(out))
    assert set(saved) == set(cats) and set(saved["low_intensity"]) == {"psnr", "ssim", "lpips", "samples"}   # metrics.py:117-124
    assert len(iq.results["low_intensity"]) == 3 and set(iq.results["low_intensity"][0]) == {"psnr", "ssim", "lpips"}
    assert "LOW_INTENSITY" in "".join(_capture(iq.print_results))


def _capture(fn):
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn()
    return buf.getvalue()


# ---------------------------------------------------------------------------------------------------------------------
# synthetic fog (utils/helpers.py:201-265)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 3, 32, 48), (1, 3, 1, 1), (2, 3, 37, 5), (2, 3, 256, 256)])
def test_fog_kernel_vs_oracle(shape):
    g = torch.Generator().manual_seed(shape[2])
    clear = torch.rand(shape, generator=g)
    beta = torch.tensor([0.15, 0.55, 0.95][:shape[0]], dtype=torch.float64)
    A_ = torch.tensor([0.6, 0.8, 1.0][:shape[0]], dtype=torch.float64)
    got = D.apply_fog(clear.to(DEV), beta, A_)
    # float32 beta / A on the device, transmission in float64 like numpy: the last float32 bit may differ
    ref = R.apply_fog(clear, beta.float().double(), A_.float().double())
    assert max_abs(got, ref) <= 1.2e-7
    assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0


def test_apply_random_fog_draws_like_the_reference():
    """np.random.uniform(*beta_range) then np.random.uniform(*A_range), image after image (helpers.py:237-238), so a
    seeded numpy state reproduces the reference's parameters; single images and 0..255 inputs are handled as there."""
    clear = torch.rand(3, 3, 24, 40, generator=torch.Generator().manual_seed(1))
    np.random.seed(123)
    got = D.apply_random_fog(clear.to(DEV), ["low", "medium", "high"])
    np.random.seed(123)
    betas, As = [], []
    for name in ("low", "medium", "high"):
        (b0, b1), (a0, a1) = D.FOG_RANGES[name]
        betas.append(np.random.uniform(b0, b1))
        As.append(np.random.uniform(a0, a1))
    ref = R.apply_fog(clear, np.float32(betas).astype(np.float64), np.float32(As).astype(np.float64))
    assert max_abs(got, ref) <= 1.2e-7
    one = D.apply_random_fog((clear[0] * 255).to(DEV), "high", rng=np.random.RandomState(7))
    rs = np.random.RandomState(7)
    b, a = rs.uniform(0.7, 1.0), rs.uniform(0.8, 1.0)
    ref1 = R.apply_fog((clear[:1] * 255) / 255.0, [float(np.float32(b))], [float(np.float32(a))])
    assert one.shape == (3, 24, 40) and max_abs(one, ref1[0]) <= 2e-7


@pytest.mark.parametrize("shape", [(5, 3, 24, 40), (2, 3, 17, 33), (3, 3, 256, 512)])
def test_paired_augmentation_vs_oracle(shape):
    """SURVEY 8f-4 / VERDICT r2 item 10: the reference's train transform (data/dataset.py:59-64,100-116: horizontal flip,
    vertical flip, ColorJitter(brightness 0.1, contrast 0.1), ONE seed per sample for hazy and clear) on device against the
    torch restatement (oracle UNPINNED: torchvision absent): every combination of flips and both jitter orders, the same
    choices for both images of a pair, values within one fp32 rounding of the grayscale mean's effect."""
    from adam_dehaze_amd.data import draw_augment_params, paired_augment
    g = torch.Generator().manual_seed(shape[2])
    hazy = torch.rand(*shape, generator=g)
    clear = (hazy * 1.3 - 0.1).clamp(0, 1)
    n = shape[0]
    params = draw_augment_params(n, np.random.RandomState(7))
    for i in range(n):                      # make sure every branch is exercised whatever the draws were
        params[i, 0], params[i, 1], params[i, 2] = float(i & 1), float((i >> 1) & 1), float((i >> 2) & 1 ^ 1)
    (ah, ac), back = paired_augment([hazy.to(DEV), clear.to(DEV)], params)
    assert torch.equal(back, params)
    assert max_abs(ah, R.paired_augment(hazy, params)) < 2e-6
    assert max_abs(ac, R.paired_augment(clear, params)) < 2e-6
    assert float(ah.min()) >= 0.0 and float(ah.max()) <= 1.0
    # the draws: one row per sample, reproducible from the numpy stream, factors in torchvision's ranges
    p1, p2 = draw_augment_params(64, np.random.RandomState(3)), draw_augment_params(64, np.random.RandomState(3))
    assert torch.equal(p1, p2) and p1.shape == (64, 5)
    assert set(p1[:, :3].unique().tolist()) <= {0.0, 1.0} and 5 < int(p1[:, 0].sum()) < 59 and 5 < int(p1[:, 2].sum()) < 59
    assert float(p1[:, 3:].min()) >= 0.9 and float(p1[:, 3:].max()) <= 1.1
    # the loader applies it pairwise: with augmentation the pair stays a fog pair up to the jitter (flips agree)
    from adam_dehaze_amd.data import synthetic_loader
    b0 = next(synthetic_loader(4, (16, 24), 1, seed=5, device=DEV))
    b1 = next(synthetic_loader(4, (16, 24), 1, seed=5, device=DEV, augment=True))
    assert b1["hazy"].shape == b0["hazy"].shape and not torch.equal(b1["hazy"], b0["hazy"])


def test_synthetic_loader_is_on_device_and_seeded():
    a = list(D.synthetic_loader(3, (16, 24), 2, seed=4, device=DEV))
    b = list(D.synthetic_loader(3, (16, 24), 2, seed=4, device=DEV))
    c = list(D.synthetic_loader(3, (16, 24), 1, seed=4, rank=1, device=DEV))
    assert a[0]["hazy"].is_cuda and a[0]["intensity"].dtype == torch.int64 and set(a[0]) >= {"hazy", "clear", "intensity", "name"}
    assert torch.equal(a[1]["hazy"], b[1]["hazy"]) and not torch.equal(a[0]["hazy"], c[0]["hazy"])
    assert 0.0 <= float(a[0]["hazy"].min()) and float(a[0]["hazy"].max()) <= 1.0


# ---------------------------------------------------------------------------------------------------------------------
# optimiser (train_joint.py:81-90)
# ---------------------------------------------------------------------------------------------------------------------
def test_adam_foreach_duplicate_semantics_vs_fixture():
    rec = load_golden("adam_dup_foreach")
    ws = {k: t(rec[f"w_{k}0"]).to(DEV).requires_grad_(True) for k in ("dup", "single", "tri")}
    opt = Adam([ws["dup"], ws["single"], ws["dup"], ws["tri"], ws["tri"], ws["tri"]], lr=5e-5, weight_decay=1e-4,
               duplicates="foreach")
    for step in range(3):
        for k, w in ws.items():
            w.grad = t(rec[f"g_{k}{step}"]).to(DEV)
        opt.step()
        for k, w in ws.items():
            assert max_abs(w, rec[f"w_{k}{step + 1}"]) < 2e-7, (k, step)
    sd = opt.state_dict()
    assert sd["param_groups"][0]["params"] == rec["state_params"].tolist()
    assert [float(sd["state"][k]["step"]) for k in sorted(sd["state"])] == rec["state_steps"].tolist()


def test_adam_multi_tensor_matches_per_tensor_oracle_and_handles_tails():
    """One launch over tensors of awkward sizes (1, 3, chunk +- 1, unaligned views) == the oracle's per-tensor update;
    parameters without a gradient are skipped (their step count does not advance); grad_scale folds 1/world."""
    chunk = H.value("adh_adam_chunk_elems")
    g = torch.Generator().manual_seed(2)
    base = torch.randn(4 * chunk + 64, generator=g)
    shapes = [(), (3,), (chunk - 1,), (chunk + 1,), (2, chunk), (7, 5)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    dev_ps = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    # an unaligned view (offset 1 float): the scalar path
    store = base.clone().to(DEV)
    view = store[1:1 + 1000].requires_grad_(True)
    ps.append(base[1:1001].clone())
    dev_ps.append(view)
    opt = Adam(dev_ps, lr=1e-3, weight_decay=1e-2)
    opt.grad_scale = 0.5
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    steps = [0] * len(ps)
    for it in range(3):
        for i, (p, dp) in enumerate(zip(ps, dev_ps)):
            if it == 1 and i == 2:
                dp.grad = None
                continue
            gr = torch.randn(p.shape, generator=g)
            dp.grad = gr.to(DEV)
            R.adam_step(p, gr * 0.5, ms[i], vs[i], step=steps[i], lr=1e-3, weight_decay=1e-2)
            steps[i] += 1
        opt.step()
    for i, (p, dp) in enumerate(zip(ps, dev_ps)):
        assert max_abs(dp.detach(), p) < 5e-7, i
        assert opt.state[id(dp)]["step"] == steps[i]
    assert torch.equal(store[0].cpu(), base[0]) and torch.equal(store[1001:].cpu(), base[1001:])   # nothing outside the view moved


def test_packed_weight_cache_follows_parameter_updates(monkeypatch):
    """The Winograd / packed weights are cached per parameter version: an in-place torch update (version counter) and the
    HIP Adam step (explicit invalidation) must both be seen by the next forward."""
    import adam_dehaze_amd.engine as E
    torch.manual_seed(0)
    m = A.LightweightDehazeModel(base_channels=16, n_blocks=1).to(DEV).eval()
    x = torch.rand(1, 3, 32, 48, device=DEV)
    with torch.no_grad():
        y0 = m(x)
        n_cached = len(E._PACK_CACHE)
        assert n_cached > 0
        y0b = m(x)
        assert len(E._PACK_CACHE) == n_cached and torch.equal(y0, y0b)       # second pass: all hits
        w = m.residual_blocks.at(0).conv1.block.at(0).weight
        w.mul_(1.5)                                                           # torch in-place op: version bump
        y1 = m(x)
    assert not torch.equal(y0, y1)
    monkeypatch.setattr(E, "USE_PACK_CACHE", False)
    with torch.no_grad():
        assert torch.equal(m(x), y1)                                          # same result without the cache
    monkeypatch.setattr(E, "USE_PACK_CACHE", True)
    m.train()
    opt = Adam(m.parameters(), lr=1e-2)
    from adam_dehaze_amd.loss import l1_loss
    l1_loss(m(x), torch.zeros_like(x)).backward()
    m.eval()
    with torch.no_grad():
        before = m(x)
    opt.step()                                                                # writes behind torch's version counter
    with torch.no_grad():
        after = m(x)
        E.invalidate_weight_cache()
        fresh = m(x)
    assert not torch.equal(before, after) and torch.equal(after, fresh)


def test_relu_bitmask_path_matches_default(monkeypatch):
    """engine.USE_RELU_BITS (opt-in: measured slower, see engine.py): the ResidualBlock tail's backward with the
    bit-packed ReLU mask gives bit-identical gradients to the default path that reads `out`."""
    import adam_dehaze_amd.engine as E
    from adam_dehaze_amd.engine import Act, Engine
    from adam_dehaze_amd.layers import ResidualBlock
    torch.manual_seed(0)
    blk = ResidualBlock(32).to(DEV).train()
    x = torch.randn(2, 19, 37, 32, device=DEV)
    gout = torch.randn(2, 19, 37, 32, device=DEV)
    res = {}
    for bits in (False, True):
        monkeypatch.setattr(E, "USE_RELU_BITS", bits)
        eng = Engine(torch.device(DEV), record=True)
        xa = Act(x.clone())
        o = blk.run(eng, xa, True)
        o.grad = gout.clone()
        eng.backward()
        torch.cuda.synchronize()
        names = {id(p): n for n, p in blk.named_parameters()}
        res[bits] = (o.t.clone(), xa.grad.clone(), {names[k]: g.clone() for k, g in eng.param_grads.items()})
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for k, g in res[False][2].items():
        assert torch.equal(res[True][2][k], g), k
