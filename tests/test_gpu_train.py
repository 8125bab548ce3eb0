"""GPU checks of the composed training steps (T1/T2 in SURVEY.md 8a)."""
import copy
import warnings

import pytest
import torch
import torch.nn.functional as F

import adam_dehaze_amd as A
from adam_dehaze_amd import train as T
from oracle import ref_cpu as R
from tests._util import max_abs, rel_err

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
    batch = next(T.synthetic_loader(4, 32, 1, seed=3))
    hazy, clear, labels = batch["hazy"], batch["clear"], batch["intensity"]

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
    for batch in T.synthetic_loader(4, 32, 3, seed=5):
        losses.append(float(T.joint_train_step(system, batch)["loss"]))
    assert all(l == l and l < 1e3 for l in losses)
    assert not torch.equal(before, high_p.detach())
    assert opt.state[id(high_p)]["step"] == 6 and opt.state[id(clf_p)]["step"] == 3


def test_branch_training_step_reduces_l1():
    cfg = _cfg()
    torch.manual_seed(3)
    model, losses = T.train_dehazing_model(cfg, "medium", steps=6)
    assert len(losses) >= 3 and all(l == l for l in losses)


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
