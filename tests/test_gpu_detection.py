"""Detector stage (BASELINE config 5; /root/reference models/detection.py:7-140) on the GPU against the oracle's
restatement of torchvision's Faster R-CNN inference path.  PARITY UNPINNED: torchvision is absent from /root/reference and from
this image, so both sides restate its published algorithm; weights are seeded random (FrozenBatchNorm buffers randomised so the
folds are not the identity).  Stage by stage -- so that a tie in a sort cannot hide an arithmetic difference -- then end to end."""
import warnings

import pytest
import torch

import adam_dehaze_amd as A
from adam_dehaze_amd import _hip as H
from adam_dehaze_amd import detection as D
from adam_dehaze_amd.engine import Act, Engine
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KW = dict(min_size=160, max_size=256, rpn_pre_nms_top_n=200, rpn_post_nms_top_n=60, box_score_thresh=0.012)


def _model(seed=0):
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = D.DetectionModel(num_classes=91, pretrained=False, **KW)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, buf in m.named_buffers():
            if name.endswith(("running_var", ".weight")) and buf.dim() == 1:
                buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
            elif buf.dim() == 1:
                buf.copy_(0.1 * torch.randn(buf.shape, generator=g))
        # spread the class scores and the box deltas so that thresholds and NMS have something to decide
        m.model.roi_heads.box_predictor.cls_score.weight.mul_(30.0)
        m.model.roi_heads.box_predictor.bbox_pred.weight.mul_(5.0)
        m.model.rpn.head.bbox_pred.weight.mul_(3.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.to(DEV).eval(), sd


def _nchw(a: Act):
    return a.t[..., :a.C].permute(0, 3, 1, 2).contiguous().cpu()


def _iou(a, b):
    x1, y1 = torch.max(a[:, None, 0], b[None, :, 0]), torch.max(a[:, None, 1], b[None, :, 1])
    x2, y2 = torch.min(a[:, None, 2], b[None, :, 2]), torch.min(a[:, None, 3], b[None, :, 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    aa, ab = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]), (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter).clamp(min=1e-9)


def test_nms_kernel_matches_the_sequential_definition():
    g = torch.Generator().manual_seed(3)
    for M, ngroups in ((1, 1), (70, 1), (300, 4), (1500, 5)):
        c = torch.rand(M, 2, generator=g) * 200
        wh = torch.rand(M, 2, generator=g) * 60 + 2
        boxes = torch.cat([c, c + wh], dim=1)
        scores = torch.rand(M, generator=g)
        groups = torch.randint(0, ngroups, (M,), generator=g)
        want = R.det_nms(boxes, scores, groups, 0.5)
        got = D.FasterRCNN.nms(boxes.to(DEV), scores.to(DEV), groups.to(DEV), 0.5).cpu()
        assert torch.equal(got, want), (M, ngroups)


def test_nms_beyond_the_kernel_limit_is_exact_and_no_proposals_give_empty_detections(monkeypatch):
    """more candidates than one launch of the mask kernel takes (16384) used to be truncated to the best-scoring
    16384; now every group goes through on its own and an oversized group in score-ordered chunks (suppressed first by what
    earlier chunks kept) -- the greedy algorithm exactly.  Checked against the sequential definition with the limit lowered
    (several groups, and ONE group larger than the limit), and at a real size with two different chunkings agreeing.  And a
    frame whose RPN filter leaves no proposal returns empty detections instead of failing in adh_roi_align_fpn."""
    g = torch.Generator().manual_seed(5)

    def case(M, ngroups):
        c = torch.rand(M, 2, generator=g) * 300
        wh = torch.rand(M, 2, generator=g) * 60 + 2
        return torch.cat([c, c + wh], dim=1), torch.rand(M, generator=g), torch.randint(0, ngroups, (M,), generator=g)
    monkeypatch.setattr(D.FasterRCNN, "NMS_LIMIT", 256)
    for M, ngroups in ((1500, 5), (1500, 1), (257, 1)):
        boxes, scores, groups = case(M, ngroups)
        want = R.det_nms(boxes, scores, groups, 0.5)
        got = D.FasterRCNN.nms(boxes.to(DEV), scores.to(DEV), groups.to(DEV), 0.5).cpu()
        assert torch.equal(got, want), (M, ngroups)
    boxes, scores, groups = case(20000, 3)
    monkeypatch.setattr(D.FasterRCNN, "NMS_LIMIT", 16384)
    a = D.FasterRCNN.nms(boxes.to(DEV), scores.to(DEV), groups.to(DEV), 0.5).cpu()
    monkeypatch.setattr(D.FasterRCNN, "NMS_LIMIT", 3000)
    b = D.FasterRCNN.nms(boxes.to(DEV), scores.to(DEV), groups.to(DEV), 0.5).cpu()
    assert torch.equal(a, b) and a.numel() > 100
    # no proposals
    torch.manual_seed(0)
    det = D.FasterRCNN(num_classes=5).to(DEV).eval()
    monkeypatch.setattr(D.FasterRCNN, "proposals", lambda self, rpn_out, image_size: [torch.zeros((0, 4), device=DEV) for _ in range(2)])
    out = det(torch.rand(2, 3, 64, 96, device=DEV))
    assert len(out) == 2 and all(d["boxes"].shape == (0, 4) and d["scores"].numel() == 0 and d["labels"].dtype == torch.int64 for d in out)


def test_roi_align_fpn_matches_torchvision_semantics():
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(2, 256, h, w, generator=g) for h, w in ((40, 56), (20, 28), (10, 14), (5, 7))]
    image_size = (160, 224)
    b = torch.rand(40, 2, generator=g) * torch.tensor([200.0, 140.0])
    wh = torch.rand(40, 2, generator=g) ** 2 * torch.tensor([260.0, 200.0]) + 1.0      # some reach past the image, some are tiny
    props = [torch.cat([b[:20], b[:20] + wh[:20]], dim=1) - 10.0, torch.cat([b[20:], b[20:] + wh[20:]], dim=1)]
    props[1][0] = torch.tensor([-50.0, -50.0, 300.0, 260.0])                               # sqrt(area) 329 -> pyramid level '2'
    props[1][1] = torch.tensor([-200.0, -200.0, 400.0, 350.0])                             # sqrt(area) 574 -> level '3'
    # oracle pooling (through det_box_head's level mapping), read back from its roi_align directly
    rois = torch.cat([torch.cat([torch.full((p.shape[0], 1), float(i)), p], dim=1) for i, p in enumerate(props)])
    s = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lv = (torch.floor(4 + torch.log2(s / 224) + torch.tensor(1e-6)).clamp(2, 5) - 2).long()
    assert set(lv.tolist()) == {0, 1, 2, 3}
    want = torch.zeros(40, 256, 7, 7)
    for l in range(4):
        idx = torch.where(lv == l)[0]
        if idx.numel():
            want[idx] = R.det_roi_align(feats[l], rois[idx], 2.0 ** round(__import__("math").log2(feats[l].shape[-2] / 160.0)))
    L = H.FpnLevels()
    L.nlevels = 4
    dev_feats = [f.permute(0, 2, 3, 1).contiguous().to(DEV) for f in feats]
    for i, f in enumerate(dev_feats):
        L.f[i], L.H[i], L.W[i], L.cs[i], L.scale[i] = f.data_ptr(), f.shape[1], f.shape[2], f.shape[3], 2.0 ** round(__import__("math").log2(f.shape[1] / 160.0))
    out = torch.empty(40, 256 * 49, device=DEV)
    import ctypes
    H.call("adh_roi_align_fpn", ctypes.byref(L), rois.to(DEV).contiguous().data_ptr(), 40, 256, out.data_ptr())
    torch.cuda.synchronize()
    assert float((out.cpu().view(40, 256, 7, 7) - want).abs().max()) < 2e-5


def test_faster_rcnn_stages_vs_oracle():
    m, sd = _model(0)
    det = m.model
    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(9))
    eng = Engine(torch.device(DEV), record=False)
    # (a) transform
    batch, image_size = det.transform(eng, x.to(DEV))
    ob, osz = R.det_transform(x, KW["min_size"], KW["max_size"])
    assert image_size == osz and (batch.Hh, batch.Ww) == tuple(ob.shape[-2:])
    assert float((batch.t[..., :3].permute(0, 3, 1, 2).cpu() - ob).abs().max()) < 2e-5
    # (b) backbone + FPN
    feats = det.features(eng, batch)
    ofeats = R.det_backbone(ob, sd)
    assert len(feats) == 5
    for f, of in zip(feats, ofeats):
        assert tuple(_nchw(f).shape) == tuple(of.shape)
        assert float((_nchw(f) - of).abs().max()) < 3e-4 * max(1.0, float(of.abs().max()))
    # (c) RPN head, anchors, decode, clip
    rpn = det.rpn_outputs(eng, feats, image_size, (batch.Hh, batch.Ww))
    orpn = R.det_rpn(ofeats, sd, image_size, ob.shape[-2:])
    for (b, l), (ob_, ol) in zip(rpn, orpn):
        assert float((l.cpu() - ol).abs().max()) < 1e-3 * max(1.0, float(ol.abs().max()))
        assert float((b.cpu() - ob_).abs().max()) < 2e-2           # pixels, after exp() of deltas up to the clamp
    # (d) proposals: same count, every box matched
    props = det.proposals(rpn, image_size)
    oprops = R.det_proposals(orpn, KW["rpn_pre_nms_top_n"], KW["rpn_post_nms_top_n"], 0.7)
    for p, op in zip(props, oprops):
        assert abs(p.shape[0] - op.shape[0]) <= 2
        assert float((_iou(p.cpu(), op).max(dim=1).values > 0.98).float().mean()) > 0.95
    # (e) box head on the ORACLE's proposals (so the comparison is arithmetic, not ordering)
    cls, reg, rois = det.box_head_outputs(eng, feats, [p.to(DEV) for p in oprops], image_size)
    ocls, oreg, orois = R.det_box_head(ofeats, oprops, sd, image_size)
    Rn = orois.shape[0]
    gcls, greg = cls.t.reshape(-1, cls.t.shape[-1])[:Rn, :91].cpu(), reg.t.reshape(-1, reg.t.shape[-1])[:Rn, :364].cpu()
    assert float((gcls - ocls).abs().max()) < 5e-4 * max(1.0, float(ocls.abs().max()))
    assert float((greg - oreg).abs().max()) < 5e-4 * max(1.0, float(oreg.abs().max()))
    # (f) post-processing on the ORACLE's head outputs
    ocls_a = Act(torch.zeros(1, (Rn + 31) // 32, 32, 96, device=DEV), 91, needs_grad=False)
    oreg_a = Act(torch.zeros(1, (Rn + 31) // 32, 32, 364, device=DEV), 364, needs_grad=False)
    ocls_a.t.view(-1, 96)[:Rn, :91] = ocls.to(DEV)
    oreg_a.t.view(-1, 364)[:Rn] = oreg.to(DEV)
    dets = det.detections(ocls_a, oreg_a, orois.to(DEV), image_size, 2)
    odets = R.det_postprocess(ocls, oreg, orois, image_size, 2, KW["box_score_thresh"], 0.5, 100)
    total = 0
    for d, od in zip(dets, odets):
        assert d["boxes"].shape[0] == od["boxes"].shape[0]
        total += d["boxes"].shape[0]
        assert torch.equal(d["labels"].cpu(), od["labels"])
        assert float((d["scores"].cpu() - od["scores"]).abs().max()) < 1e-5 if od["scores"].numel() else True
        assert float((d["boxes"].cpu() - od["boxes"]).abs().max()) < 1e-3 if od["boxes"].numel() else True
    assert total >= 20, "the seeded weights must produce detections for this test to mean anything"


def test_detection_model_end_to_end_and_integrated_system():
    m, sd = _model(1)
    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        dets = m(list(x.to(DEV)))                                  # the reference passes a list of images
    odets = R.faster_rcnn_forward(x, sd, KW["min_size"], KW["max_size"], KW["rpn_pre_nms_top_n"], KW["rpn_post_nms_top_n"], 0.7,
                                  KW["box_score_thresh"], 0.5, 100)
    assert len(dets) == 2 and set(dets[0]) == {"boxes", "labels", "scores"}
    matched = total = 0
    for d, od in zip(dets, odets):
        assert d["labels"].dtype == torch.int64 and d["boxes"].shape[1] == 4
        assert bool((d["scores"][:-1] >= d["scores"][1:]).all())
        total += od["boxes"].shape[0]
        if od["boxes"].numel() and d["boxes"].numel():
            iou = _iou(od["boxes"], d["boxes"].cpu())
            best = iou.argmax(dim=1)
            ok = (iou.max(dim=1).values > 0.95) & (d["labels"].cpu()[best] == od["labels"]) & \
                 ((d["scores"].cpu()[best] - od["scores"]).abs() < 1e-3)
            matched += int(ok.sum())
    assert total >= 10 and matched >= 0.9 * total, (matched, total)
    f = D.filter_detections(dets, 0.02)
    assert set(f[0]) == {"boxes_xywh", "labels", "scores"} and bool((f[0]["scores"] > 0.02).all())
    # the integrated system (models/detection.py:73-125): (detections, dehazed images), detector frozen
    torch.manual_seed(2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        models = {"low": A.LightweightDehazeModel(base_channels=8, n_blocks=1), "medium": A.MediumIntensityDehazeModel(base_channels=8),
                  "high": A.HighIntensityDehazeModel(base_channels=16)}

    class Router(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.m = models["low"]

        def forward(self, images):
            return self.m(images), {"weights": None}
    system = D.create_integrated_system(Router().to(DEV).eval(), m)
    with torch.no_grad():
        results, dehazed = system(x.to(DEV))
    assert dehazed.shape == x.shape and len(results) == 2 and "boxes" in results[0]
    assert all(not p.requires_grad for p in system.detection_model.parameters())
    mean, std = torch.tensor(R.DET_IMAGE_MEAN).view(1, 3, 1, 1), torch.tensor(R.DET_IMAGE_STD).view(1, 3, 1, 1)
    oint = R.faster_rcnn_forward((dehazed.cpu() - mean) / std, sd, KW["min_size"], KW["max_size"], KW["rpn_pre_nms_top_n"],
                                 KW["rpn_post_nms_top_n"], 0.7, KW["box_score_thresh"], 0.5, 100)
    assert abs(results[0]["boxes"].shape[0] - oint[0]["boxes"].shape[0]) <= max(3, oint[0]["boxes"].shape[0] // 5)
    with pytest.raises(NotImplementedError):
        m(list(x.to(DEV)), targets=[{}])
    with pytest.raises(ValueError):
        D.DetectionModel(model_name="mask_rcnn_resnet50_fpn", pretrained=False)
    with pytest.raises(ValueError):
        D.DetectionModel(model_name="yolo", pretrained=False)
    cfg = {"detection": {"model": "faster_rcnn_resnet50_fpn", "pretrained": False}}
    assert isinstance(D.create_detection_model(cfg), D.DetectionModel)
