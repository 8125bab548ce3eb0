"""Whole-model GPU checks of the BASELINE.json configurations that had only piecewise coverage:

  * config 1 -- the DEFAULT full-width Light / Medium / Complex at 1x3x256x256 against the output summaries the REFERENCE
    itself produced (tests/golden/fullwidth_summaries.npz, tools/gen_golden.py section (v)); this pins the default
    F(4x4,3x3) path of the full-width models to reference-held data, not only to the oracle;
  * the headline workload as a whole -- CORUN-Complex train forward + L1 + backward + Adam at 8 x 512 x 1024
    (high_intensity.py:92-138, train_dehazing.py:71-106 with the L1 term): finite, bit-reproducible, every parameter
    moves, BatchNorm buffers change;
  * config 2 as a whole -- DenseNet121 HDEN -> arg-max -> HardRouter -> full-width branches at 8 x 512 x 1024
    (routing.py:23-68): indices equal the host arg-max of the same logits, image i of the batch equals the single-image run.
"""
import warnings

import pytest
import torch

import adam_dehaze_amd as A
from adam_dehaze_amd import _hip as H
from oracle import ref_cpu as R
from tests._util import load_golden, max_abs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

FULLWIDTH = [("light", lambda: A.LightweightDehazeModel(base_channels=32, n_blocks=3)),
             ("medium", lambda: A.MediumIntensityDehazeModel(base_channels=64)),
             ("high", lambda: A.HighIntensityDehazeModel(base_channels=96))]


@pytest.mark.parametrize("name,ctor", FULLWIDTH, ids=[c[0] for c in FULLWIDTH])
def test_config1_fullwidth_defaults_vs_reference_held_outputs(name, ctor):
    """Same seeded constructor (SHA-256-identical parameters, tests/test_host_cpu.py), same
    `torch.rand(1,3,256,256, generator=42)` input, eval mode: the HIP path's output against the REFERENCE's stored
    8x8 patch (<= 2e-4; north-star 1e-3), mean (<= 1e-5) and abs-max."""
    rec = load_golden("fullwidth_summaries")
    torch.manual_seed(42)
    m = ctor().to(DEV).eval()
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(42))
    with torch.no_grad():
        out = m(x.to(DEV))
    torch.cuda.synchronize()
    out = out.cpu()
    assert max_abs(out[0, :, 100:108, 100:108], rec[name + ".out_patch"]) < 2e-4
    assert abs(float(out.double().mean()) - float(rec[name + ".out_mean"])) < 1e-5
    assert abs(float(out.abs().max()) - float(rec[name + ".out_absmax"])) < 2e-4


def test_headline_complex_train_step_8x512x1024():
    """The BASELINE.json metric's workload, exactly as bench.py times it: CORUN-Complex (base 96) train-mode forward +
    L1 + backward + Adam(lr 1e-4, wd 1e-4), bs 8, 512 x 1024.  Two runs from the same seed are bit-equal (deterministic
    kernels, fixed reduction orders); the loss is finite and equals the fp64 L1 of the returned output; every parameter
    receives a finite gradient and moves (ConvTranspose biases in front of train-mode BN have an exactly-zero gradient);
    every BatchNorm buffer changes; outputs stay in [0, 1]."""
    from adam_dehaze_amd.loss import l1_loss
    from adam_dehaze_amd.optim import Adam
    hazy, clear, _ = R.synthetic_batch(8, 512, 1024, seed=42)
    hazy, clear = hazy.to(DEV), clear.to(DEV)
    runs = []
    for run in range(2):
        torch.manual_seed(42)
        m = A.HighIntensityDehazeModel().to(DEV).train()
        opt = Adam(m.parameters(), lr=1e-4, weight_decay=1e-4)
        before = {k: v.detach().clone() for k, v in m.state_dict().items()}
        out = m(hazy)
        assert out.shape == hazy.shape and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
        loss = l1_loss(out, clear)
        want = float((out.detach().double() - clear.double()).abs().mean())
        assert abs(float(loss) - want) < 1e-6
        loss.backward()
        zero_grad = []
        for k, p in m.named_parameters():
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
            if float(p.grad.abs().max()) == 0.0:
                zero_grad.append(k)
        assert all(k.startswith("decoder.") and k.endswith(".0.bias") for k in zero_grad), zero_grad
        opt.step()
        torch.cuda.synchronize()
        after = m.state_dict()
        params = dict(m.named_parameters())
        still = [k for k in params if torch.equal(before[k], after[k])]
        assert len(still) <= 2 and all(k in zero_grad for k in still), still
        for k in after:
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                assert not torch.equal(before[k], after[k]), k
        runs.append((float(loss), {k: v.detach().clone() for k, v in params.items()}, out.detach().clone()))
        del m, opt, out, loss, before, after, params
        torch.cuda.empty_cache()
    assert runs[0][0] == runs[0][0] and 0.0 < runs[0][0] < 1.0
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][2], runs[1][2])
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k


def test_config2_pipeline_8x512x1024():
    """BASELINE.json config 2 end to end at its real size: FogIntensityClassifier('densenet121') eval forward ->
    `adh_argmax3` inside HardRouter.forward(x) (routing.py:36-43) -> per-class gather -> full-width branch -> scatter.
    (a) the routed int64 class indices are bit-equal to the host arg-max (first index on ties) of the SAME logits;
    (b) with all-"low" labels and with the classifier's own indices, image i of the batch output
    equals the single-image run of the branch its index selects (eval mode: images are independent)."""
    from adam_dehaze_amd.classifier import FogIntensityClassifier
    from adam_dehaze_amd.routing import HardRouter
    torch.manual_seed(42)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf = FogIntensityClassifier("densenet121", 3, pretrained=False).to(DEV).eval()
        models = {"low": A.LightweightDehazeModel(), "medium": A.MediumIntensityDehazeModel(),
                  "high": A.HighIntensityDehazeModel()}
    router = HardRouter(models, clf, device=DEV).to(DEV)
    hazy, _, _ = R.synthetic_batch(8, 512, 1024, seed=42)
    x = hazy.to(DEV)
    router.train()
    with torch.no_grad():                      # one train-mode pass of every branch: BN statistics are not the identity
        for mdl in models.values():
            mdl(x[:1, :, :128, :256].contiguous())
    router.eval()
    clf.eval()
    with torch.no_grad():
        logits, feats = clf(x)
        out_auto, info = router(x)
        out_low, info_low = router(x, torch.zeros(8, dtype=torch.int64, device=DEV))
    torch.cuda.synchronize()
    assert logits.shape == (8, 3) and feats.shape == (8, 1024) and bool(torch.isfinite(logits).all())
    idx = info["intensity"]
    assert idx.dtype == torch.int64
    host = logits.cpu()
    want = torch.tensor([max(range(3), key=lambda c: (float(host[i, c]), -c)) for i in range(8)], dtype=torch.int64)
    assert torch.equal(idx.cpu(), want)
    assert torch.equal(info["low_mask"].cpu(), want == 0) and torch.equal(info["high_mask"].cpu(), want == 2)
    names = ("low", "medium", "high")
    with torch.no_grad():
        for i in (0, 3, 7):
            xi = x[i:i + 1].contiguous()
            assert float((out_low[i:i + 1] - models["low"](xi)).abs().max()) < 1e-6
            assert float((out_auto[i:i + 1] - models[names[int(want[i])]](xi)).abs().max()) < 1e-6
    assert bool(info_low["low_mask"].all()) and float((out_low - x).abs().max()) > 1e-3


def test_complex_fullwidth_512x1024_vs_oracle(monkeypatch):
    """The whole full-width CORUN-Complex at the HEADLINE resolution against the CPU oracle (VERDICT r3 weak 1 / next 1a;
    /root/reference models/dehazing/high_intensity.py:92-138): one 512 x 1024 frame, default kernels.  Eval output and train-mode
    output within 1e-3 (north-star) and PSNR > 60 dB, L1 loss within 1e-4, every BatchNorm buffer within 1e-4, and a spread of
    parameter gradients -- stem, both encoder levels, bottleneck, both decoder levels (ConvTranspose and ResidualBlock), both
    head convolutions, the guidance branch, a BatchNorm affine pair and a CBAM MLP -- on the kink-matched float64 gate at 5e-4
    of each tensor's scale (tests/test_gpu_parity.py _fullwidth_vs_oracle; the float64 oracle with the kernels' own ReLU
    masks replayed is the exact gradient of the piece of the network the HIP path differentiated)."""
    from tests.test_gpu_parity import _fullwidth_vs_oracle
    names = ["init_conv.block.0.weight", "encoder.0.0.block.0.weight", "encoder.0.1.conv1.block.0.weight", "encoder.0.3.fc.0.weight",
             "encoder.1.0.block.0.weight", "encoder.1.2.conv2.block.0.weight", "bottleneck.0.conv1.block.0.weight",
             "bottleneck.2.conv2.block.1.weight", "bottleneck.2.conv2.block.1.bias", "decoder.0.0.weight", "decoder.0.3.conv1.block.0.weight",
             "decoder.1.0.weight", "decoder.1.3.conv2.block.0.weight", "output_conv.0.block.0.weight", "output_conv.1.block.0.weight",
             "output_conv.2.weight", "detail_branch.0.block.0.weight", "detail_branch.2.weight"]
    _fullwidth_vs_oracle("complex96", "f43", 1, 512, 1024, monkeypatch, grad_names=names, grad_tol=5e-4,
                         report="grad_gate_fullwidth_512x1024.txt")
