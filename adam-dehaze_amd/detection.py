"""Detector stage of the evaluate path (/root/reference models/detection.py, evaluation/evaluate.py:288-344; BASELINE config 5)
on the HIP engine -- smallest honest slice.

The reference's detector is torchvision's `fasterrcnn_resnet50_fpn` wholesale (models/detection.py:23-29), a third-party
dependency that is neither in /root/reference nor in this image.  This module keeps the reference's surface
(`DetectionModel(num_classes, model_name, pretrained)`, `IntegratedDetectionSystem.forward -> (detections, dehazed_images)`,
`create_detection_model`, `create_integrated_system`) and restates torchvision's inference path with torchvision's parameter
names (`model.backbone.body.*`, `model.backbone.fpn.*`, `model.rpn.head.*`, `model.roi_heads.*`), so that a real checkpoint loads
on a machine that has one:
  GeneralizedRCNNTransform (normalise, resize to min 800 / max 1333, pad to /32)  ->  ResNet-50 (FrozenBatchNorm folded into the
  conv epilogues) + FPN  ->  RPN (head, anchors, decode, top-1000 per level, NMS 0.7, top 1000)  ->  MultiScaleRoIAlign 7x7  ->
  TwoMLPHead + FastRCNNPredictor  ->  softmax / decode / score > 0.05 / per-class NMS 0.5 / top 100  ->  boxes in the input frame.
Convolutions and the two fully connected layers run on the engine's MFMA kernels; FPN merge, anchor decode, NMS, RoIAlign and the
box post-processing are csrc/detect.hip; torch only orders (topk / sort) and indexes.  Inference only: `targets` (training the
detector) raises NotImplementedError; the reference freezes the detector anyway (models/detection.py:91-93).
PARITY UNPINNED: checked against oracle/ref_cpu.py's restatement with seeded random weights (tests/test_gpu_detection.py).
Other `model_name`s of the reference's switch (mobilenet / Mask R-CNN) raise ValueError like any unknown name.
"""
from __future__ import annotations

import ctypes as C
import math
import warnings
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _hip as H
from .engine import Act, BNState, Engine
from .layers import ConvParams, Seq

IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ASPECT_RATIOS = (0.5, 1.0, 2.0)


class _FrozenBN(nn.Module):
    """torchvision.ops.misc.FrozenBatchNorm2d: four buffers, no gradient, eps 1e-5."""

    def __init__(self, c: int):
        super().__init__()
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))

    def state(self) -> BNState:
        return BNState(self.weight, self.bias, self.running_mean, self.running_var, None)


def _conv(cin, cout, k, bias=False):
    p = ConvParams(cin, cout, k, bias=bias)
    p.weight.requires_grad_(False)
    if p.bias is not None:
        p.bias.requires_grad_(False)
    return p


class _Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5: the stride sits on the 3x3 convolution)."""

    def __init__(self, cin, width, stride, downsample):
        super().__init__()
        self.stride = stride
        self.conv1, self.bn1 = _conv(cin, width, 1), _FrozenBN(width)
        self.conv2, self.bn2 = _conv(width, width, 3), _FrozenBN(width)
        self.conv3, self.bn3 = _conv(width, width * 4, 1), _FrozenBN(width * 4)
        self.downsample = Seq([(0, _conv(cin, width * 4, 1)), (1, _FrozenBN(width * 4))]) if downsample else None

    def run(self, eng: Engine, x: Act) -> Act:
        h = eng.conv(x, self.conv1.weight, None, self.bn1.state(), k=1, stride=1, pad=0, relu=True)
        h = eng.conv(h, self.conv2.weight, None, self.bn2.state(), k=3, stride=self.stride, pad=1, relu=True)
        idt = x
        if self.downsample is not None:
            idt = eng.conv(x, self.downsample.at(0).weight, None, self.downsample.at(1).state(), k=1, stride=self.stride, pad=0, relu=False)
        return eng.conv(h, self.conv3.weight, None, self.bn3.state(), k=1, stride=1, pad=0, relu=True, residual=idt)


class _ResNet50Body(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.bn1 = _conv(3, 64, 7), _FrozenBN(64)
        cin = 64
        for li, (width, n) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), start=1):
            blocks = []
            for bi in range(n):
                stride = 2 if (bi == 0 and li > 1) else 1
                blocks.append((bi, _Bottleneck(cin, width, stride, downsample=(bi == 0))))
                cin = width * 4
            setattr(self, f"layer{li}", Seq(blocks))

    def run(self, eng: Engine, x8: Act) -> List[Act]:
        h = eng.conv(x8, self.conv1.weight, None, self.bn1.state(), k=7, stride=2, pad=3, relu=True)
        h = eng.maxpool(h, 3, 2, 1)
        feats = []
        for li in range(1, 5):
            layer = getattr(self, f"layer{li}")
            for bi in range(len(layer)):
                h = layer.at(bi).run(eng, h)
            feats.append(h)
        return feats


class _FPN(nn.Module):
    """torchvision.ops.FeaturePyramidNetwork (>= 0.13 key names: inner_blocks.i.0.*, layer_blocks.i.0.*) + LastLevelMaxPool."""

    def __init__(self, in_channels=(256, 512, 1024, 2048), out_channels=256):
        super().__init__()
        self.inner_blocks = Seq([(i, Seq([(0, _conv(c, out_channels, 1, bias=True))])) for i, c in enumerate(in_channels)])
        self.layer_blocks = Seq([(i, Seq([(0, _conv(out_channels, out_channels, 3, bias=True))])) for i in range(len(in_channels))])

    def run(self, eng: Engine, feats: List[Act]) -> List[Act]:
        n = len(feats)
        inner = lambda i, x: eng.conv(x, self.inner_blocks.at(i).at(0).weight, self.inner_blocks.at(i).at(0).bias, None, k=1, stride=1,
                                      pad=0, relu=False)
        layer = lambda i, x: eng.conv(x, self.layer_blocks.at(i).at(0).weight, self.layer_blocks.at(i).at(0).bias, None, k=3, stride=1,
                                      pad=1, relu=False)
        last = inner(n - 1, feats[-1])
        outs = [layer(n - 1, last)]
        for i in range(n - 2, -1, -1):
            lat = inner(i, feats[i])
            H.call("adh_upsample_nearest_add", last.t.data_ptr(), last.cs, last.Hh, last.Ww, lat.t.data_ptr(), lat.cs, lat.N, lat.Hh,
                   lat.Ww, lat.C)
            last = lat
            outs.insert(0, layer(i, last))
        outs.append(eng.maxpool(outs[-1], 1, 2, 0))      # LastLevelMaxPool: F.max_pool2d(x, 1, 2, 0)
        return outs


class _Backbone(nn.Module):
    def __init__(self):
        super().__init__()
        self.body = _ResNet50Body()
        self.fpn = _FPN()


class _RPNHead(nn.Module):
    def __init__(self, c=256, A=3):
        super().__init__()
        self.conv = Seq([(0, Seq([(0, _conv(c, c, 3, bias=True))]))])     # torchvision >= 0.13: head.conv.0.0.*
        self.cls_logits = _conv(c, A, 1, bias=True)
        self.bbox_pred = _conv(c, 4 * A, 1, bias=True)


class _RPN(nn.Module):
    def __init__(self):
        super().__init__()
        self.head = _RPNHead()


class _Linear(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(fout), requires_grad=False)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.uniform_(self.bias, -1.0 / math.sqrt(fin), 1.0 / math.sqrt(fin))


class _BoxHead(nn.Module):
    def __init__(self, fin=256 * 49, rep=1024):
        super().__init__()
        self.fc6, self.fc7 = _Linear(fin, rep), _Linear(rep, rep)


class FastRCNNPredictor(nn.Module):
    """torchvision.models.detection.faster_rcnn.FastRCNNPredictor (the reference replaces it: models/detection.py:27-29)."""

    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.cls_score, self.bbox_pred = _Linear(in_channels, num_classes), _Linear(in_channels, num_classes * 4)


class _RoIHeads(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.box_head = _BoxHead()
        self.box_predictor = FastRCNNPredictor(1024, num_classes)


def base_anchors(size: float) -> torch.Tensor:
    """anchor_utils.AnchorGenerator.generate_anchors for one level: round([-w, -h, w, h] / 2) per aspect ratio."""
    ar = torch.tensor(ASPECT_RATIOS, dtype=torch.float32)
    h_r = torch.sqrt(ar)
    w_r = 1.0 / h_r
    ws = (w_r[:, None] * torch.tensor([float(size)])[None, :]).view(-1)
    hs = (h_r[:, None] * torch.tensor([float(size)])[None, :]).view(-1)
    return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()


def resized_size(h: int, w: int, min_size: int, max_size: int):
    """GeneralizedRCNNTransform._resize_image: scale = min(min_size / min(h, w), max_size / max(h, w)); F.interpolate with
    recompute_scale_factor=True gives floor(h * scale), floor(w * scale) (float32 scale, like torch computes it)."""
    scale = torch.min(torch.tensor(float(min_size)) / torch.tensor(float(min(h, w))),
                      torch.tensor(float(max_size)) / torch.tensor(float(max(h, w)))).item()
    return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))


class FasterRCNN(nn.Module):
    """torchvision.models.detection.FasterRCNN (ResNet-50 FPN), inference path."""

    def __init__(self, num_classes=91, min_size=800, max_size=1333, rpn_pre_nms_top_n=1000, rpn_post_nms_top_n=1000,
                 rpn_nms_thresh=0.7, box_score_thresh=0.05, box_nms_thresh=0.5, box_detections_per_img=100):
        super().__init__()
        self.backbone = _Backbone()
        self.rpn = _RPN()
        self.roi_heads = _RoIHeads(num_classes)
        self.num_classes = num_classes
        self.min_size, self.max_size = min_size, max_size
        self.rpn_pre_nms_top_n, self.rpn_post_nms_top_n, self.rpn_nms_thresh = rpn_pre_nms_top_n, rpn_post_nms_top_n, rpn_nms_thresh
        self.box_score_thresh, self.box_nms_thresh, self.box_detections_per_img = box_score_thresh, box_nms_thresh, box_detections_per_img
        for p in self.parameters():
            p.requires_grad_(False)

    # ------------------------------------------------------------------ pieces (also driven one by one by the tests)
    def transform(self, eng: Engine, x: torch.Tensor, pre_affine=None):
        """normalise + resize + pad: [N,3,H,W] -> (NHWC8 batch Act, (h, w) of the resized images).  `pre_affine` = (mean, std) of a
        normalisation the caller wants applied BEFORE the detector's own (IntegratedDetectionSystem): both are one affine map."""
        N, _, Hh, Ww = x.shape
        mean, std = IMAGE_MEAN, IMAGE_STD
        if pre_affine is not None:      # ((x - m1) / s1 - m2) / s2 = (x - (m1 + m2 s1)) / (s1 s2)
            m1, s1 = pre_affine
            mean = tuple(a1 + m2 * b1 for a1, b1, m2 in zip(m1, s1, IMAGE_MEAN))
            std = tuple(b1 * s2 for b1, s2 in zip(s1, IMAGE_STD))
        a = eng.image_normalize_to_nhwc8(x, mean, std, {})
        oh, ow = resized_size(Hh, Ww, self.min_size, self.max_size)
        if (oh, ow) != (Hh, Ww):
            a = eng.bilinear(a, oh, ow, align_corners=False)
        ph, pw = (oh + 31) // 32 * 32, (ow + 31) // 32 * 32
        if (ph, pw) != (oh, ow):
            buf = torch.zeros((N, ph, pw, 8), device=x.device, dtype=torch.float32)
            buf[:, :oh, :ow] = a.t                       # batch_images: zero padding to a multiple of 32 (a device copy)
            a = Act(buf, 8, needs_grad=False)
        return a, (oh, ow)

    def features(self, eng: Engine, batch: Act) -> List[Act]:
        return self.backbone.fpn.run(eng, self.backbone.body.run(eng, batch))

    def rpn_outputs(self, eng: Engine, feats: List[Act], image_size, padded_size):
        """Per level: (boxes [N, HWA, 4] decoded + clipped, objectness logits [N, HWA])."""
        head = self.rpn.head
        outs = []
        for lvl, f in enumerate(feats):
            c = head.conv.at(0).at(0)
            t = eng.conv(f, c.weight, c.bias, None, k=3, stride=1, pad=1, relu=True)
            cls = eng.conv(t, head.cls_logits.weight, head.cls_logits.bias, None, k=1, stride=1, pad=0, relu=False)
            reg = eng.conv(t, head.bbox_pred.weight, head.bbox_pred.bias, None, k=1, stride=1, pad=0, relu=False)
            A = len(ASPECT_RATIOS)
            sh, sw = padded_size[0] // f.Hh, padded_size[1] // f.Ww
            base = base_anchors(ANCHOR_SIZES[lvl]).to(f.t.device).contiguous()
            n_anchors = f.Hh * f.Ww * A
            boxes = torch.empty((f.N, n_anchors, 4), device=f.t.device, dtype=torch.float32)
            logits = torch.empty((f.N, n_anchors), device=f.t.device, dtype=torch.float32)
            H.call("adh_rpn_decode", cls.t.data_ptr(), cls.cs, reg.t.data_ptr(), reg.cs, f.N, f.Hh, f.Ww, A, sh, sw, base.data_ptr(),
                   float(image_size[0]), float(image_size[1]), boxes.data_ptr(), logits.data_ptr())
            outs.append((boxes, logits))
        return outs

    @staticmethod
    def nms(boxes: torch.Tensor, scores: torch.Tensor, groups: torch.Tensor, thr: float) -> torch.Tensor:
        """torchvision.ops.batched_nms: indices of the kept boxes, by descending score."""
        M = boxes.shape[0]
        if M == 0:
            return torch.empty(0, dtype=torch.int64, device=boxes.device)
        order = torch.sort(scores, descending=True, stable=True).indices
        if M <= FasterRCNN.NMS_LIMIT:
            return order[FasterRCNN._nms_sorted(boxes[order], groups[order], thr)]
        # More candidates than the mask kernel takes at once (this used to drop everything beyond the 16384 best-scoring
        # boxes, silently differing from torchvision.batched_nms -- reachable with flat softmax scores: R * (NC - 1) ~ 90 k box-head
        # candidates per image before the score filter).  Batched NMS is independent per group, so each group goes through on its own;
        # a group that is itself too large is cut into score-ordered chunks: a chunk's boxes are first suppressed by the boxes ALREADY
        # KEPT (all of higher score), then by each other -- exactly the greedy algorithm.
        kept = []
        gs = groups[order]
        for gid in torch.unique(gs).tolist():
            idx = order[gs == gid]                      # this group's boxes, by descending score
            mine: List[torch.Tensor] = []
            for lo in range(0, idx.numel(), FasterRCNN.NMS_LIMIT):
                cand = idx[lo:lo + FasterRCNN.NMS_LIMIT]
                for prev in mine:                       # (chunked: at most NMS_LIMIT x NMS_LIMIT IoUs at a time)
                    if cand.numel() == 0:
                        break
                    cand = cand[(FasterRCNN._iou(boxes[cand], boxes[prev]) <= thr).all(dim=1)]
                if cand.numel():
                    z = torch.zeros(cand.numel(), dtype=torch.int32, device=boxes.device)
                    mine.append(cand[FasterRCNN._nms_sorted(boxes[cand], z, thr)])
            kept.extend(mine)
        keep = torch.cat(kept)
        return keep[torch.sort(scores[keep], descending=True, stable=True).indices]

    NMS_LIMIT = 16384      # boxes adh_nms_sorted takes in one launch

    @staticmethod
    def _iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """torchvision.ops.box_iou."""
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(a[:, None, :2], b[None, :, :2])
        rb = torch.min(a[:, None, 2:], b[None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        return inter / (area_a[:, None] + area_b[None, :] - inter)

    @staticmethod
    def _nms_sorted(b: torch.Tensor, g: torch.Tensor, thr: float) -> torch.Tensor:
        """Boolean keep mask of score-sorted boxes `b` with group ids `g` (one launch of the mask kernel + scan)."""
        M = b.shape[0]
        b = b.contiguous()
        g = g.to(torch.int32).contiguous()
        words = H.value("adh_nms_words", M)
        mask = torch.empty(M * words, device=b.device, dtype=torch.int64)
        keep = torch.empty(M, device=b.device, dtype=torch.int32)
        H.call("adh_nms_sorted", b.data_ptr(), g.data_ptr(), M, float(thr), mask.data_ptr(), keep.data_ptr())
        return keep.bool()

    def proposals(self, rpn_out, image_size) -> List[torch.Tensor]:
        """rpn.filter_proposals: per level top-k by logit, sigmoid, small-box removal, NMS per level, first post_nms_top_n."""
        N = rpn_out[0][0].shape[0]
        res = []
        for n in range(N):
            bs, ss, ls = [], [], []
            for lvl, (boxes, logits) in enumerate(rpn_out):
                k = min(self.rpn_pre_nms_top_n, logits.shape[1])
                top = torch.topk(logits[n], k).indices
                bs.append(boxes[n][top])
                ss.append(torch.sigmoid(logits[n][top]))
                ls.append(torch.full((k,), lvl, dtype=torch.int32, device=boxes.device))
            b, s, l = torch.cat(bs), torch.cat(ss), torch.cat(ls)
            ok = ((b[:, 2] - b[:, 0]) >= 1e-3) & ((b[:, 3] - b[:, 1]) >= 1e-3) & (s >= 0.0)
            b, s, l = b[ok], s[ok], l[ok]
            keep = self.nms(b, s, l, self.rpn_nms_thresh)[:self.rpn_post_nms_top_n]
            res.append(b[keep])
        return res

    def box_head_outputs(self, eng: Engine, feats: List[Act], props: List[torch.Tensor], image_size):
        """MultiScaleRoIAlign + TwoMLPHead + FastRCNNPredictor -> (class logits [R, NC], box deltas [R, 4 NC], rois [R, 5])."""
        dev = feats[0].t.device
        rois = torch.cat([torch.cat([torch.full((p.shape[0], 1), float(i), device=dev), p], dim=1) for i, p in enumerate(props)]).contiguous()
        R = rois.shape[0]
        L = H.FpnLevels()
        L.nlevels = 4
        for i in range(4):
            f = feats[i]
            L.f[i], L.H[i], L.W[i], L.cs[i] = f.t.data_ptr(), f.Hh, f.Ww, f.cs
            # poolers.py _infer_scale: 2 ** round(log2(feature size / image size))
            L.scale[i] = 2.0 ** round(math.log2(f.Hh / float(image_size[0])))
        Rp = (R + 255) // 256 * 256          # RoIs laid out as one [Rp / 32, 32]-pixel "image": full MFMA tiles for the linear layers
        pooled = torch.zeros((Rp, 256 * 49), device=dev, dtype=torch.float32)
        H.call("adh_roi_align_fpn", C.byref(L), rois.data_ptr(), R, 256, pooled.data_ptr())
        bh, bp = self.roi_heads.box_head, self.roi_heads.box_predictor

        def linear(x: Act, lin: _Linear, relu: bool) -> Act:
            return eng.conv(x, lin.weight.view(lin.weight.shape[0], lin.weight.shape[1], 1, 1), lin.bias, None, k=1, stride=1, pad=0, relu=relu)
        h = linear(Act(pooled.view(1, Rp // 32, 32, 256 * 49), needs_grad=False), bh.fc6, True)
        h = linear(h, bh.fc7, True)
        cls = linear(h, bp.cls_score, False)
        reg = linear(h, bp.bbox_pred, False)
        return cls, reg, rois

    def detections(self, cls: Act, reg: Act, rois: torch.Tensor, image_size, n_images: int) -> List[Dict[str, torch.Tensor]]:
        """roi_heads.postprocess_detections."""
        dev = rois.device
        R, NC = rois.shape[0], self.num_classes
        img = rois[:, 0].to(torch.int32).contiguous()
        props = rois[:, 1:].contiguous()
        hw = torch.tensor([[float(image_size[0]), float(image_size[1])]] * n_images, device=dev)
        boxes = torch.empty((R, NC - 1, 4), device=dev)
        scores = torch.empty((R, NC - 1), device=dev)
        valid = torch.empty((R, NC - 1), device=dev, dtype=torch.int32)
        H.call("adh_box_postprocess", cls.t.data_ptr(), cls.cs, reg.t.data_ptr(), reg.cs, props.data_ptr(), hw.data_ptr(), img.data_ptr(), R,
               NC, float(self.box_score_thresh), 1e-2, boxes.data_ptr(), scores.data_ptr(), valid.data_ptr())
        labels = torch.arange(1, NC, device=dev).expand(R, NC - 1)
        out = []
        for n in range(n_images):
            sel = (img == n)[:, None] & valid.bool()
            b, s, l = boxes[sel], scores[sel], labels[sel]
            keep = self.nms(b, s, l, self.box_nms_thresh)[:self.box_detections_per_img]
            out.append({"boxes": b[keep], "labels": l[keep], "scores": s[keep]})
        return out

    @torch.no_grad()
    def forward(self, images, targets=None, pre_affine=None):
        if targets is not None:
            raise NotImplementedError("training the detector (targets) is outside this slice; the reference freezes it "
                                      "(models/detection.py:91-93)")
        if isinstance(images, (list, tuple)):
            if len({tuple(i.shape) for i in images}) != 1:
                raise RuntimeError("this slice batches images of one size (the evaluation loader resizes to 512 x 512, data/dataset.py:257)")
            x = torch.stack(list(images))
        else:
            x = images
        H.require_cuda(x, "detector input")
        x = x.float().contiguous()
        N, _, Hh, Ww = x.shape
        eng = Engine(x.device, record=False)
        batch, image_size = self.transform(eng, x, pre_affine)
        feats = self.features(eng, batch)
        props = self.proposals(self.rpn_outputs(eng, feats, image_size, (batch.Hh, batch.Ww)), image_size)
        if sum(int(p_.shape[0]) for p_ in props) == 0:    # no proposal survived the RPN filter: empty detections, as torchvision returns
            return [{"boxes": torch.zeros((0, 4), device=x.device), "labels": torch.zeros(0, dtype=torch.int64, device=x.device),
                     "scores": torch.zeros(0, device=x.device)} for _ in range(N)]
        cls, reg, rois = self.box_head_outputs(eng, feats, props, image_size)
        dets = self.detections(cls, reg, rois, image_size, N)
        # transform.postprocess: back to the input frame
        rh, rw = float(Hh) / float(image_size[0]), float(Ww) / float(image_size[1])
        scale = torch.tensor([rw, rh, rw, rh], device=x.device)
        for d in dets:
            d["boxes"] = d["boxes"] * scale
        return dets


class DetectionModel(nn.Module):
    """Object detection model with support for multiple architectures (models/detection.py:7-71)."""

    def __init__(self, num_classes=91, model_name="faster_rcnn_resnet50_fpn", pretrained=True, **detector_kwargs):
        super().__init__()
        self.model_name = model_name
        self.num_classes = num_classes
        if model_name == "faster_rcnn_resnet50_fpn":
            self.model = FasterRCNN(num_classes=num_classes, **detector_kwargs)
        elif model_name in ("faster_rcnn_mobilenet_v3_large_fpn", "mask_rcnn_resnet50_fpn"):
            raise ValueError(f"detection model '{model_name}' needs torchvision (not part of this build); only "
                             "'faster_rcnn_resnet50_fpn' is built natively")
        else:
            raise ValueError(f"Unsupported detection model: {model_name}")
        if pretrained:
            warnings.warn("pretrained detector weights cannot be downloaded in this environment: the detector is RANDOMLY "
                          "initialised until a torchvision fasterrcnn_resnet50_fpn state_dict is loaded (key names match)")

    def forward(self, images, targets=None, pre_affine=None):
        return self.model(images, targets, pre_affine)


def filter_detections(dets: Sequence[Dict[str, torch.Tensor]], score_threshold: float = 0.5):
    """The `score > 0.5` filter of evaluation/evaluate.py:327,343 + COCO [x, y, w, h] boxes, per image."""
    out = []
    for d in dets:
        k = d["scores"] > score_threshold
        b = d["boxes"][k]
        out.append({"boxes_xywh": torch.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], dim=1) if b.numel() else b,
                    "labels": d["labels"][k], "scores": d["scores"][k]})
    return out


class IntegratedDetectionSystem(nn.Module):
    """Dehazing router -> ImageNet normalisation -> detector (models/detection.py:73-125).  Returns
    (detection_results, dehazed_images) like the reference."""

    def __init__(self, dehazing_model, detection_model):
        super().__init__()
        self.dehazing_model = dehazing_model
        self.detection_model = detection_model
        for param in self.detection_model.parameters():
            param.requires_grad = False

    def forward(self, images, targets=None):
        dehazed_images, dehazing_info = self.dehazing_model(images)
        # the reference normalises here AND the detector's own transform normalises again (models/detection.py:112-119 +
        # torchvision GeneralizedRCNNTransform): kept -- it is what the reference computes -- as ONE affine map inside the
        # detector's first kernel (`pre_affine`)
        detection_results = self.detection_model(dehazed_images, targets, pre_affine=(IMAGE_MEAN, IMAGE_STD))
        return detection_results, dehazed_images


def create_detection_model(config):
    """models/detection.py:127-133."""
    return DetectionModel(num_classes=91, model_name=config["detection"]["model"], pretrained=config["detection"]["pretrained"])


def create_integrated_system(dehazing_router, detection_model):
    """models/detection.py:135-140."""
    return IntegratedDetectionSystem(dehazing_model=dehazing_router, detection_model=detection_model)
