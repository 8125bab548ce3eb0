"""GPU checks of the composed training steps (T1/T2 in SURVEY.md 8a)."""
import copy
import warnings

import pytest
import torch
import torch.nn.functional as F

import adam_dehaze_amd as A
from adam_dehaze_amd import train as T
from oracle import ref_cpu as R
from tests._util import kink_matched, max_abs, oracle_with_masks, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg():
    return {"dataset": {"batch_size": 4, "img_size": 32},
            "classifier": {"model": "resnet18", "pretrained": False, "num_classes": 3, "checkpoint_dir": "/nonexistent"},
            "dehazing": {"checkpoint_dir": "/nonexistent",
                         "low": {"model_type": "lightweight", "channels": 8, "blocks": 2, "learning_rate": 1e-4},
                         "medium": {"model_type": "standard", "channels": 8, "blocks": 6, "learning_rate": 1e-4},
                         "high": {"model_type": "complex", "channels": 16, "blocks": 9, "learning_rate": 1e-4}},
            "routing": {"type": "soft", "temperature": 0.5},
            "joint_training": {"learning_rate": 5e-5, "epochs": 1, "lambda_dehazing": 1.0, "lambda_classification": 0.2,
                               "lambda_detection": 0.5, "checkpoint_dir": "/tmp/adh_test_ckpt"},
            "device": DEV, "seed": 42}


def test_joint_step_composition_vs_oracle():
    """classifier (eval) -> SoftRouter over three train-mode branches -> JointLoss -> backward, against the
    same composition built from the oracle's functions."""
    torch.manual_seed(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        system = T.build_joint_system(_cfg())
    clf, models, router, crit = system["classifier"], system["models"], system["router"], system["criterion"]
    sd_clf = {k: v.detach().cpu().clone() for k, v in clf.state_dict().items()}
    sds = {n: {k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for n, m in models.items()}
    batch = next(T.synthetic_loader(4, 32, 1, seed=3, device=DEV))
    hazy, clear, labels = batch["hazy"].cpu(), batch["clear"].cpu(), batch["intensity"].cpu()

    # oracle
    for sd in list(sds.values()) + [sd_clf]:
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
    logits_r, _ = R.classifier_forward(hazy, sd_clf, "resnet18")
    fns = {"low": R.lightweight_forward, "medium": R.medium_forward, "high": R.high_forward}
    outs = {n: fns[n](hazy, sds[n], training=True) for n in fns}
    blended, w = R.soft_route(outs, logits_r, 0.5)
    vgg_sd = {k: v.detach().cpu() for k, v in crit.dehazing_loss.content_loss.state_dict().items()}
    lp_sd = {k: v.detach().cpu() for k, v in crit.dehazing_loss.perceptual_loss.state_dict().items()}
    total_r, comps_r = R.joint_loss(blended, clear, logits_r, labels, vgg_sd, lp_sd)
    total_r.backward()

    clf.eval()
    for m in models.values():
        m.train()
    logits, _ = clf(hazy.to(DEV))
    dehazed, aux = router(hazy.to(DEV), logits)
    total, comps = crit(dehazed, clear.to(DEV), logits, labels.to(DEV))
    total.backward()
    assert max_abs(logits, logits_r.detach()) < 1e-3
    assert max_abs(aux["weights"], w.detach()) < 1e-3
    assert max_abs(dehazed, blended.detach()) < 1e-3
    assert abs(float(total) - float(total_r)) < 1e-3 * max(1.0, float(total_r))
    assert abs(float(comps["classification"]) - float(comps_r["classification"])) < 1e-4
    # gradients reach the classifier through the blend weights and the CE term, and every branch
    g = dict(clf.named_parameters())["classifier.4.weight"].grad
    assert rel_err(g, sd_clf["classifier.4.weight"].grad) < 2e-2
    for n in ("low", "medium", "high"):
        name = "output_conv.0.block.0.weight"
        assert rel_err(dict(models[n].named_parameters())[name].grad, sds[n][name].grad) < 2e-2, n


def test_joint_step_gradients_kink_matched():
    """T1 (train_joint.py:129-166) on the kink-matched gate: classifier (eval: its dropout is
    RNG-dependent in the reference's train mode) -> SoftRouter(T = 0.5) over the three train-mode branches -> the FULL
    JointLoss (L1 + 0.1 VGG16-content + 0.1 LPIPS + 0.2 CE) -> backward.  EVERY parameter of the three branches and of
    the classifier head against the float64 oracle that replays the ReLU masks the branch kernels used (tests/_util.py).
    Gate per tensor (the fixture gate of tests/test_gpu_parity.py): err_gpu <= 3 * err_ref + 3e-4 of the tensor's scale,
    err_ref = the reference arithmetic itself (fp32 CPU oracle) with the SAME masks against that float64 run -- on this
    composition the reference's own fp32 rounding is ~1e-3 on a few tensors, and kinks that cannot be replayed (CBAM
    arg-max, max-pools of the loss networks, the backbone's ReLUs) are resolved by fp32 arithmetic on both sides.
    The table goes to gpurun_out/grad_gate_joint.txt."""
    import os
    torch.manual_seed(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        system = T.build_joint_system(_cfg())
    clf, models, router, crit = system["classifier"], system["models"], system["router"], system["criterion"]
    sd_clf0 = {k: v.detach().cpu().clone() for k, v in clf.state_dict().items()}
    sds0 = {n: {k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for n, m in models.items()}
    vgg0 = {k: v.detach().cpu() for k, v in crit.dehazing_loss.content_loss.state_dict().items()}
    lp0 = {k: v.detach().cpu() for k, v in crit.dehazing_loss.perceptual_loss.state_dict().items()}
    hazy, clear, labels = R.synthetic_batch(4, 32, 32, seed=3)      # CPU-generated: the same batch on every box
    fns = {"low": R.lightweight_forward, "medium": R.medium_forward, "high": R.high_forward}

    def oracle(dtype, masks=None):
        def cast(sd, grad=True):
            out = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
            for k, v in out.items():
                if grad and v.is_floating_point() and "running" not in k:
                    v.requires_grad_(True)
            return out
        sd_clf, sds = cast(sd_clf0), {n: cast(sd) for n, sd in sds0.items()}
        logits, _ = R.classifier_forward(hazy.to(dtype), sd_clf, "resnet18")
        outs = {}
        for n in fns:
            def run(n=n):
                return fns[n](hazy.to(dtype), sds[n], training=True)
            outs[n] = oracle_with_masks(run, masks[n]) if masks is not None else run()
        blended, _ = R.soft_route(outs, logits, 0.5)
        total, _ = R.joint_loss(blended, clear.to(dtype), logits, labels, cast(vgg0, False), cast(lp0, False))
        total.backward()
        return float(total), sd_clf, sds

    clf.eval()
    for m in models.values():
        m.train()
    with kink_matched(router) as km:
        logits, _ = clf(hazy.to(DEV))
        dehazed, _ = router(hazy.to(DEV), logits)
        total, _ = crit(dehazed, clear.to(DEV), logits, labels.to(DEV))
        total.backward()
    torch.cuda.synchronize()
    allm = km.masks()
    masks = {n: {k[len("models.%s." % n):]: v for k, v in allm.items() if k.startswith("models.%s." % n)} for n in fns}
    assert all(len(masks[n]) >= 5 for n in fns)
    tot32, clf32, sds32 = oracle(torch.float32, masks)
    tot64, clf64, sds64 = oracle(torch.float64, masks)
    assert abs(float(total) - tot64) < 1e-4 * max(1.0, abs(tot64))
    bad, lines = [], []

    def check(tag, name, g, g64, g32):
        scale = max(float(g64.abs().max()), 1e-6)
        err_gpu = float((g.cpu().double() - g64).abs().max()) / scale
        err_cpu = float((g32.double() - g64).abs().max()) / scale
        lines.append(f"joint {tag:7s} {name:44s} scale {scale:.2e}  err_gpu {err_gpu:.2e}  err_ref {err_cpu:.2e}")
        if not err_gpu <= 3 * err_cpu + 3e-4:
            bad.append((tag, name, err_gpu, err_cpu))
    for n, m in models.items():
        for name, p in m.named_parameters():
            if name.startswith("decoder") and name.endswith(".0.bias"):
                continue   # ConvTranspose bias feeding train-mode BN: the true gradient is exactly 0
            check(n, name, p.grad, sds64[n][name].grad, sds32[n][name].grad)
    for name, p in clf.named_parameters():
        if name.startswith("classifier."):
            check("clfhead", name, p.grad, clf64[name].grad, clf32[name].grad)
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "grad_gate_joint.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
    assert not bad, bad[:8]


def test_joint_training_runs_and_uses_duplicate_param_adam():
    torch.manual_seed(2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        system = T.build_joint_system(_cfg())
    opt = system["optimizer"]
    high_p = next(system["models"]["high"].parameters())
    clf_p = next(system["classifier"].parameters())
    assert opt.repeats[id(high_p)] == 2 and opt.repeats[id(clf_p)] == 1   # train_joint.py:81-84
    before = high_p.detach().clone()
    system["classifier"].train()
    system["router"].train()
    losses = []
    for batch in T.synthetic_loader(4, 32, 3, seed=5, device=DEV):
        losses.append(float(T.joint_train_step(system, batch)["loss"]))
    assert all(l == l and l < 1e3 for l in losses)
    assert not torch.equal(before, high_p.detach())
    assert opt.state[id(high_p)]["step"] == 6 and opt.state[id(clf_p)]["step"] == 3


def test_branch_training_step_vs_oracle(monkeypatch):
    """T2 (train_dehazing.py:71-106) on a mixed-label batch against the oracle: keep the images of the branch's level ->
    Medium train-mode forward -> L1 -> backward -> one Adam(lr 1e-4, wd 1e-4) step.  Checked: the filtered loss, every
    parameter gradient, the optimiser applied to those gradients (exactly), the BN buffers, and the parameters after
    the step against the all-oracle step wherever the first Adam step is sign-stable (it moves every element by
    lr * g / (|g| + 1e-8): elements whose gradient is ~1e-8 amplify rounding into the update)."""
    import adam_dehaze_amd.engine as E
    from adam_dehaze_amd.loss import DehazingLoss
    from adam_dehaze_amd.optim import Adam
    monkeypatch.setattr(E, "USE_WINOGRAD", False)     # the direct kernels follow the reference's summation order closely
    torch.manual_seed(11)
    model = A.MediumIntensityDehazeModel(base_channels=8).to(DEV).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    hazy, clear, _ = R.synthetic_batch(6, 32, 48, seed=21)
    labels = torch.tensor([1, 0, 1, 2, 1, 1])
    crit = DehazingLoss(content=False, perceptual=False).to(DEV)
    opt = Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
    with kink_matched(model) as km:
        st = T.dehazing_train_step(model, crit, opt, {"hazy": hazy, "clear": clear, "intensity": labels}, 1, DEV)
    torch.cuda.synchronize()
    # oracle: the same step on the CPU, differentiating the same piece of the piecewise-smooth network (the ReLU masks
    # the kernels used are replayed: an activation within fp32 rounding of zero may sit on either side)
    keep = labels == 1
    sd = {k: v.clone() for k, v in sd0.items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)

    def run():
        out = R.medium_forward(hazy[keep], sd, training=True)
        loss = F.l1_loss(out, clear[keep])
        loss.backward()
        return loss
    loss_r = oracle_with_masks(run, km.masks())
    assert abs(float(st["loss"]) - float(loss_r)) < 1e-5 and abs(float(st["l1"]) - float(loss_r)) < 1e-5
    names = dict(model.named_parameters())
    stable = total = 0
    for k, p in names.items():
        g_ref = sd[k].grad
        g = p.grad.cpu()
        scale = max(float(g_ref.abs().max()), 1e-8)
        noise = k.startswith("decoder") and k.endswith(".0.bias")       # ConvTranspose bias feeding train-mode BN: true grad 0
        if not noise:
            assert float((g - g_ref).abs().max()) < 5e-3 * scale + 2e-7, k
        # the optimiser on the GPU's own gradient: exact
        p_exp, m, v = sd0[k].clone(), torch.zeros_like(g), torch.zeros_like(g)
        R.adam_step(p_exp, g, m, v, step=0, lr=1e-4, weight_decay=1e-4)
        assert max_abs(p.detach(), p_exp) < 2e-7, k
        assert max_abs(opt.state[id(p)]["m"], m) < 1e-7 * max(1.0, float(m.abs().max())), k
        # the whole step against the all-oracle step where the update is sign-stable
        p_ref, m, v = sd0[k].clone(), torch.zeros_like(g), torch.zeros_like(g)
        R.adam_step(p_ref, g_ref.clone(), m, v, step=0, lr=1e-4, weight_decay=1e-4)
        sel = (g_ref + 1e-4 * sd0[k]).abs() > 1e-5 * scale + 1e-6
        if not noise and bool(sel.any()):
            assert float((p.detach().cpu() - p_ref)[sel].abs().max()) < 1e-6, k
        stable += int(sel.sum())
        total += sel.numel()
    assert stable > 0.9 * total
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert max_abs(v, sd[k]) < 1e-5, k
    assert opt.state[id(next(model.parameters()))]["step"] == 1
    # a batch without any image of the level is skipped, like train_dehazing.py:75-76
    assert T.dehazing_train_step(model, crit, opt, {"hazy": hazy, "clear": clear,
                                                    "intensity": torch.zeros(6, dtype=torch.long)}, 1, DEV) is None


def test_dehazing_driver_validation_checkpoints_and_resume(tmp_path):
    """train_dehazing_model: epochs of train + validation (device PSNR / SSIM), the reference's checkpoint key sets
    (train_dehazing.py:196-203,208-215), and --resume continuing from the latest checkpoint with the optimiser state."""
    cfg = _cfg()
    cfg["dehazing"]["checkpoint_dir"] = str(tmp_path / "dehazing")
    torch.manual_seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, losses = T.train_dehazing_model(cfg, "medium", steps=3, epochs=5, val_steps=2)
    assert len(losses) >= 3 and all(l == l for l in losses)
    ck_dir = tmp_path / "dehazing" / "medium"
    best = torch.load(ck_dir / "best_model.pth", map_location="cpu")
    per5 = torch.load(ck_dir / "checkpoint_epoch_5.pth", map_location="cpu")
    want = {"epoch", "model_state_dict", "optimizer_state_dict", "val_psnr", "val_ssim", "val_loss"}
    assert want <= set(best) and want <= set(per5) and per5["epoch"] == 4
    assert 0.0 < best["val_psnr"] < 100.0 and -1.0 <= best["val_ssim"] <= 1.0
    assert set(per5["model_state_dict"]) == set(model.state_dict())
    # the optimiser state is torch-loadable
    tm = [torch.nn.Parameter(v.clone()) for v in model.parameters()]
    topt = torch.optim.Adam(tm, lr=1.0)
    topt.load_state_dict(per5["optimizer_state_dict"])
    nsteps = {int(s["step"]) for s in topt.state_dict()["state"].values()}
    assert len(nsteps) == 1 and nsteps.pop() == len(losses)
    # resume: continues at epoch 6 from checkpoint_epoch_5 (one more epoch), parameters start where they were saved
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model2, losses2 = T.train_dehazing_model(cfg, "medium", steps=3, epochs=6, val_steps=2, resume=True)
    assert 1 <= len(losses2) <= 3
    with pytest.raises(FileNotFoundError):
        T.train_dehazing_model(cfg, "low", steps=1, epochs=1, resume=True)


def test_joint_driver_validation_and_checkpoint_keys(tmp_path):
    """train_joint_model: train + validate_joint (train_joint.py:173-236 on device) + checkpoint dicts with exactly the
    reference's keys (train_joint.py:272-283) + resume."""
    cfg = _cfg()
    cfg["joint_training"]["checkpoint_dir"] = str(tmp_path / "joint")
    cfg["joint_training"]["epochs"] = 1
    torch.manual_seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        system, hist = T.train_joint_model(cfg, steps_per_epoch=2, val_steps=1)
    assert len(hist) == 1 and hist[0]["val_samples"] == 4
    for k in ("train_loss", "val_loss", "val_dehaze_loss", "val_class_loss", "val_psnr", "val_ssim"):
        assert hist[0][k] == hist[0][k]
    ck = torch.load(tmp_path / "joint" / "best_model.pth", map_location="cpu")
    ref_keys = {"epoch", "router_state_dict", "low_model_state_dict", "medium_model_state_dict", "high_model_state_dict",
                "classifier_state_dict", "optimizer_state_dict", "val_psnr", "val_ssim", "val_loss"}
    assert ref_keys <= set(ck) and set(ck) - ref_keys == {"scheduler_state_dict"}
    # duplicated parameters are packed once in the optimiser state, like torch does
    n_unique = len({id(p) for p in system["router"].parameters()})
    assert len(ck["optimizer_state_dict"]["state"]) <= n_unique
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cfg["joint_training"]["epochs"] = 2
        system2, hist2 = T.train_joint_model(cfg, steps_per_epoch=1, val_steps=1, resume=True)
    assert [h["epoch"] for h in hist2] == [1]
    high_p = next(system2["models"]["high"].parameters())
    assert system2["optimizer"].state[id(high_p)]["step"] == 2 * (2 + 1)      # 2 steps restored + 1, listed twice


def test_complex_eval_forward_large_frame_properties():
    """Size-independent properties at a BASELINE-sized frame (1x3x512x1024): output in [0,1], batch
    independence in eval mode (image i of a batch == the same image alone), determinism."""
    torch.manual_seed(4)
    m = A.HighIntensityDehazeModel().to(DEV).eval()
    hazy, _, _ = R.synthetic_batch(2, 512, 1024, seed=9)
    x = hazy.to(DEV)
    with torch.no_grad():
        both = m(x)
        one = m(x[1:2].contiguous())
        again = m(x)
    assert float(both.min()) >= 0.0 and float(both.max()) <= 1.0
    assert torch.equal(both, again)
    assert max_abs(both[1:2], one) < 1e-6


def test_evaluate_joint_model_writes_the_reference_results_json(tmp_path):
    """evaluation/evaluate.py:94-177 (image-quality part): route synthetic test batches through the joint model,
    accumulate PSNR / SSIM / LPIPS per intensity category on the device, save joint_model_results.json."""
    import json
    cfg = _cfg()
    cfg["evaluation"] = {"results_dir": str(tmp_path / "results")}
    cfg["joint_training"]["checkpoint_dir"] = str(tmp_path / "nojoint")
    torch.manual_seed(6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = T.evaluate_joint_model(cfg, steps=2)
    saved = json.load(open(tmp_path / "results" / "joint_model_results.json"))
    assert saved == res and set(saved) <= {"low_intensity", "medium_intensity", "high_intensity"} and len(saved) >= 2
    n = sum(v["samples"] for v in saved.values())
    assert n == 8
    for v in saved.values():
        assert set(v) == {"psnr", "ssim", "lpips", "samples"}
        assert 0.0 < v["psnr"] < 100.0 and -1.0 <= v["ssim"] <= 1.0 and v["lpips"] == v["lpips"]


def test_evaluate_detection_writes_coco_tables(tmp_path):
    """evaluation/evaluate.py:179-383: detector on hazy and on routed frames, detections above the score threshold through
    DetectionMetrics (COCO box mAP per fog intensity) when an annotation file exists, the three results files."""
    import json
    cfg = _cfg()
    cfg["dataset"] = {"batch_size": 2, "img_size": 64, "test_path": str(tmp_path / "data")}
    cfg["evaluation"] = {"results_dir": str(tmp_path / "results")}
    cfg["joint_training"]["checkpoint_dir"] = str(tmp_path / "nojoint")
    ann = tmp_path / "data" / "annotations"
    ann.mkdir(parents=True)
    (ann / "instances.json").write_text(json.dumps({
        "images": [{"id": 100 + i, "file_name": f"synthetic_{i}"} for i in range(2)],
        "annotations": [{"id": 1, "image_id": 100, "category_id": 3, "bbox": [4, 4, 40, 40], "area": 1600, "iscrowd": 0},
                        {"id": 2, "image_id": 101, "category_id": 5, "bbox": [10, 8, 30, 50], "area": 1500, "iscrowd": 0}],
        "categories": [{"id": i, "name": str(i)} for i in range(1, 92)]}))
    torch.manual_seed(8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = T.evaluate_detection(cfg, steps=1, score_threshold=0.0)      # random-init detector: keep every detection
    assert set(res) == {"counts", "hazy", "dehazed", "weights"}
    # which weights produced the numbers is part of the result: no joint / detector checkpoint exists here
    assert res["weights"] == {"joint_checkpoint": None, "detector_checkpoint": None, "detector_random_init": True}
    saved = json.load(open(tmp_path / "results" / "detection_results.json"))
    assert saved["counts"] == res["counts"] and sum(res["counts"]["hazy"].values()) == sum(1 for r in saved["detections"]
                                                                                          if r["source"] == "hazy")
    for tag in ("hazy", "dehazed"):
        table = json.load(open(tmp_path / "results" / f"{tag}_detection_results.json"))
        assert table == res[tag] and "overall" in table
        if sum(res["counts"][tag].values()):
            assert set(table["overall"]) == {"mAP", "mAP_50", "mAP_75", "mAP_small", "mAP_medium", "mAP_large", "AR_1", "AR_10",
                                             "AR_100", "AR_small", "AR_medium", "AR_large"}
            assert all(-1.0 <= v <= 1.0 for v in table["overall"].values())
            assert set(table) - {"overall"} <= {"low", "medium", "high"}
    # without an annotation file: detections only
    cfg["dataset"]["test_path"] = str(tmp_path / "nodata")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res2 = T.evaluate_detection(cfg, steps=1)
    assert set(res2) == {"counts", "weights"}


def test_comprehensive_evaluation_files_and_keys(tmp_path):
    """evaluation/evaluate.py:32-92,464-540: baseline branches by ground-truth intensity, the routed system, the detector stage, and
    comprehensive_results.json with the reference's key structure."""
    import json
    cfg = _cfg()
    cfg["dataset"] = {"batch_size": 6, "img_size": 64, "test_path": str(tmp_path / "nodata")}
    cfg["evaluation"] = {"results_dir": str(tmp_path / "results"), "visualization_dir": str(tmp_path / "vis")}
    cfg["joint_training"]["checkpoint_dir"] = str(tmp_path / "nojoint")
    torch.manual_seed(9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = T.run_comprehensive_evaluation(cfg, steps=2, use_lpips=False)
    saved = json.load(open(tmp_path / "results" / "comprehensive_results.json"))
    assert set(saved) == {"baseline", "joint", "detection", "comparison"}
    assert set(saved["comparison"]) == {"baseline_avg_psnr", "joint_avg_psnr", "psnr_improvement"}
    assert {"hazy", "dehazed", "improvement_percent"} <= set(saved["detection"])
    base = json.load(open(tmp_path / "results" / "baseline_results.json"))
    assert base == saved["baseline"] == res["baseline"] and sum(v["samples"] for v in base.values()) == 12
    for v in base.values():
        assert set(v) == {"psnr", "ssim", "samples"} and 0.0 < v["psnr"] < 100.0
    assert abs(saved["comparison"]["psnr_improvement"] - (saved["comparison"]["joint_avg_psnr"] -
                                                          saved["comparison"]["baseline_avg_psnr"])) < 1e-9
