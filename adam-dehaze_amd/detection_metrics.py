"""COCO bounding-box detection metrics for the detector stage: `DetectionMetrics` with the surface of
/root/reference evaluation/metrics.py:126-270 (`add_detection_result`, `evaluate`, `evaluate_by_category`, `print_results`,
`save_results`, the twelve result keys) as fed by evaluation/evaluate.py:241-344.

The reference delegates the arithmetic to pycocotools (`COCO`, `COCO.loadRes`, `COCOeval(..., 'bbox')`: evaluate ->
accumulate -> summarize), a third-party package that is absent from this image.  This module restates the published
COCOeval algorithm for boxes (pycocotools 2.0.x, cocoeval.py: greedy per-threshold matching of score-sorted detections,
crowd / area-range ignore rules, 101-point interpolated precision, the twelve summary statistics) in numpy; host code, as in
the reference.  PARITY UNPINNED: neither pycocotools nor reference-held vectors exist here; tests/test_detection_metrics.py
pins it to hand-computed cases instead.  Behaviours kept on purpose: detections whose image id is not in the annotation file
raise AssertionError("Results do not correspond to current coco set") as `COCO.loadRes` does (this is what the reference's
generated empty annotation file leads to as soon as one detection passes the score threshold); a ground-truth annotation with
id 0 makes its matches look unmatched (`dtMatches == 0` is COCOeval's "no match" test); statistics without any valid entry
are -1.
"""
from __future__ import annotations

import json
import os
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np

_AREA_RNG = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]   # all, small, medium, large
_AREA_LBL = ["all", "small", "medium", "large"]
_MAX_DETS = [1, 10, 100]
RESULT_KEYS = ["mAP", "mAP_50", "mAP_75", "mAP_small", "mAP_medium", "mAP_large", "AR_1", "AR_10", "AR_100", "AR_small",
               "AR_medium", "AR_large"]


def box_iou_xywh(dt: np.ndarray, gt: np.ndarray, iscrowd: np.ndarray) -> np.ndarray:
    """IoU matrix [D, G] of [x, y, w, h] boxes (maskUtils.iou on boxes): for a crowd ground truth the union is the
    detection's own area."""
    dt = np.asarray(dt, dtype=np.float64).reshape(-1, 4)
    gt = np.asarray(gt, dtype=np.float64).reshape(-1, 4)
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    dx1, dy1, dx2, dy2 = dt[:, 0:1], dt[:, 1:2], dt[:, 0:1] + dt[:, 2:3], dt[:, 1:2] + dt[:, 3:4]
    gx1, gy1, gx2, gy2 = gt[:, 0], gt[:, 1], gt[:, 0] + gt[:, 2], gt[:, 1] + gt[:, 3]
    w = np.clip(np.minimum(dx2, gx2) - np.maximum(dx1, gx1), 0, None)
    h = np.clip(np.minimum(dy2, gy2) - np.maximum(dy1, gy1), 0, None)
    inter = w * h
    da = (dt[:, 2] * dt[:, 3])[:, None]
    ga = (gt[:, 2] * gt[:, 3])[None, :]
    union = np.where(np.asarray(iscrowd, dtype=bool)[None, :], da, da + ga - inter)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(union > 0, inter / union, 0.0)


class _CocoBoxEval:
    """COCOeval(cocoGt, cocoDt, 'bbox') restated: evaluate() / accumulate() / summarize()."""

    def __init__(self, images: Sequence[int], categories: Sequence[int], gts: List[dict], dts: List[dict],
                 iou_thrs: Optional[Sequence[float]] = None):
        self.img_ids = sorted(set(images))
        self.cat_ids = sorted(set(categories))
        self.iou_thrs = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True) \
            if iou_thrs is None else np.array(iou_thrs, dtype=np.float64)
        self.rec_thrs = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
        self._gts: Dict[tuple, List[dict]] = defaultdict(list)
        self._dts: Dict[tuple, List[dict]] = defaultdict(list)
        for g in gts:
            g = dict(g)
            g["iscrowd"] = int(g.get("iscrowd", 0))
            g["ignore"] = g["iscrowd"]                                   # _prepare: ignore := iscrowd
            if "area" not in g:
                g["area"] = float(g["bbox"][2]) * float(g["bbox"][3])
            self._gts[g["image_id"], g["category_id"]].append(g)
        for d in dts:
            self._dts[d["image_id"], d["category_id"]].append(d)
        self.eval_imgs: List[Optional[dict]] = []
        self.stats = np.full(12, -1.0)

    # ---------------------------------------------------------------- evaluate
    def _ious(self, img, cat):
        gt, dt = self._gts[img, cat], self._dts[img, cat]
        if not gt and not dt:
            return []
        order = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in order][:_MAX_DETS[-1]]
        return box_iou_xywh([d["bbox"] for d in dt], [g["bbox"] for g in gt], [g["iscrowd"] for g in gt])

    def _evaluate_img(self, img, cat, a_rng, max_det, ious):
        gt, dt = self._gts[img, cat], self._dts[img, cat]
        if not gt and not dt:
            return None
        g_ignore = np.array([1 if (g["ignore"] or g["area"] < a_rng[0] or g["area"] > a_rng[1]) else 0 for g in gt], dtype=np.int64)
        gtind = np.argsort(g_ignore, kind="mergesort")                  # evaluated ground truth first
        gt = [gt[i] for i in gtind]
        g_ignore = g_ignore[gtind]
        dtind = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in dtind[:max_det]]
        iscrowd = [g["iscrowd"] for g in gt]
        ious = ious[:, gtind] if len(ious) > 0 else ious
        T, G, D = len(self.iou_thrs), len(gt), len(dt)
        gtm = np.zeros((T, G))
        dtm = np.zeros((T, D))
        dt_ig = np.zeros((T, D))
        if G and D:
            for ti, t in enumerate(self.iou_thrs):
                for di in range(D):
                    iou = min(t, 1 - 1e-10)
                    m = -1
                    for gi in range(G):
                        if gtm[ti, gi] > 0 and not iscrowd[gi]:
                            continue                                    # already matched (a crowd may match again)
                        if m > -1 and g_ignore[m] == 0 and g_ignore[gi] == 1:
                            break                                       # a regular match exists and only ignored ones follow
                        if ious[di, gi] < iou:
                            continue
                        iou = ious[di, gi]
                        m = gi
                    if m == -1:
                        continue
                    dt_ig[ti, di] = g_ignore[m]
                    dtm[ti, di] = gt[m]["id"]
                    gtm[ti, m] = dt[di]["id"]
        # unmatched detections outside the area range are ignored
        a = np.array([d["area"] < a_rng[0] or d["area"] > a_rng[1] for d in dt]).reshape((1, D))
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(a, T, 0)))
        return {"dtMatches": dtm, "dtScores": [d["score"] for d in dt], "gtIgnore": g_ignore, "dtIgnore": dt_ig}

    def evaluate(self):
        ious = {(i, c): self._ious(i, c) for i in self.img_ids for c in self.cat_ids}
        self.eval_imgs = [self._evaluate_img(i, c, a, _MAX_DETS[-1], ious[i, c])
                          for c in self.cat_ids for a in _AREA_RNG for i in self.img_ids]

    # ---------------------------------------------------------------- accumulate
    def accumulate(self):
        T, R, K, A, M = len(self.iou_thrs), len(self.rec_thrs), len(self.cat_ids), len(_AREA_RNG), len(_MAX_DETS)
        I = len(self.img_ids)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        for k in range(K):
            for a in range(A):
                base = (k * A + a) * I
                E0 = [e for e in self.eval_imgs[base:base + I] if e is not None]
                if not E0:
                    continue
                for m, max_det in enumerate(_MAX_DETS):
                    scores = np.concatenate([np.asarray(e["dtScores"][0:max_det], dtype=np.float64) for e in E0])
                    inds = np.argsort(-scores, kind="mergesort")
                    scores_sorted = scores[inds]
                    dtm = np.concatenate([e["dtMatches"][:, 0:max_det] for e in E0], axis=1)[:, inds]
                    dt_ig = np.concatenate([e["dtIgnore"][:, 0:max_det] for e in E0], axis=1)[:, inds]
                    gt_ig = np.concatenate([e["gtIgnore"] for e in E0])
                    npig = np.count_nonzero(gt_ig == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dt_ig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                    tp_sum = np.cumsum(tps, axis=1).astype(dtype=np.float64)
                    fp_sum = np.cumsum(fps, axis=1).astype(dtype=np.float64)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        q = np.zeros((R,))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):                  # precision envelope
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        inds_r = np.searchsorted(rc, self.rec_thrs, side="left")
                        for ri, pi in enumerate(inds_r):
                            if pi >= nd:
                                break                                   # recall levels never reached keep precision 0
                            q[ri] = pr[pi]
                        precision[t, :, k, a, m] = q
                    del scores_sorted
        self.precision, self.recall = precision, recall

    # ---------------------------------------------------------------- summarize
    def _summarize(self, ap: bool, iou_thr: Optional[float] = None, area: str = "all", max_dets: int = 100) -> float:
        aind = [_AREA_LBL.index(area)]
        mind = [_MAX_DETS.index(max_dets)]
        s = self.precision if ap else self.recall
        if iou_thr is not None:
            t = np.where(iou_thr == self.iou_thrs)[0]
            s = s[t]
        s = s[:, :, :, aind, mind] if ap else s[:, :, aind, mind]
        valid = s[s > -1]
        return -1.0 if len(valid) == 0 else float(np.mean(valid))

    def summarize(self) -> np.ndarray:
        st = np.zeros(12)
        st[0] = self._summarize(True)
        st[1] = self._summarize(True, iou_thr=0.5, max_dets=_MAX_DETS[2])
        st[2] = self._summarize(True, iou_thr=0.75, max_dets=_MAX_DETS[2])
        st[3] = self._summarize(True, area="small", max_dets=_MAX_DETS[2])
        st[4] = self._summarize(True, area="medium", max_dets=_MAX_DETS[2])
        st[5] = self._summarize(True, area="large", max_dets=_MAX_DETS[2])
        st[6] = self._summarize(False, max_dets=_MAX_DETS[0])
        st[7] = self._summarize(False, max_dets=_MAX_DETS[1])
        st[8] = self._summarize(False, max_dets=_MAX_DETS[2])
        st[9] = self._summarize(False, area="small", max_dets=_MAX_DETS[2])
        st[10] = self._summarize(False, area="medium", max_dets=_MAX_DETS[2])
        st[11] = self._summarize(False, area="large", max_dets=_MAX_DETS[2])
        self.stats = st
        return st


class DetectionMetrics:
    """evaluation/metrics.py:126-270.  `annotation_file`: COCO-format JSON (images / annotations / categories)."""

    def __init__(self, annotation_file):
        with open(annotation_file) as f:
            data = json.load(f)
        self.images = [im["id"] for im in data.get("images", [])]
        self.categories = [c["id"] for c in data.get("categories", [])]
        self.annotations = list(data.get("annotations", []))
        self.results: List[dict] = []
        self.category_results = defaultdict(list)

    def add_detection_result(self, image_id, category_id, bbox, score, category=None):
        """bbox = [x, y, width, height]; `category`: optional grouping key (the fog intensity in evaluate.py:318-343)."""
        result = {"image_id": image_id, "category_id": category_id, "bbox": [float(v) for v in bbox], "score": float(score)}
        self.results.append(result)
        if category:
            self.category_results[category].append(result)

    def _load_res(self) -> List[dict]:
        """COCO.loadRes for box results: ids 1.., area = w * h, iscrowd = 0; every image id must be annotated."""
        assert set(r["image_id"] for r in self.results) == (set(r["image_id"] for r in self.results) & set(self.images)), \
            "Results do not correspond to current coco set"
        dts = []
        for i, r in enumerate(self.results):
            d = dict(r)
            d["area"] = d["bbox"][2] * d["bbox"][3]
            d["id"] = i + 1
            d["iscrowd"] = 0
            dts.append(d)
        return dts

    def evaluate(self, iou_thresholds=None):
        if not self.results:
            print("No detection results to evaluate")
            return {}
        ev = _CocoBoxEval(self.images, self.categories, self.annotations, self._load_res(),
                          iou_thrs=iou_thresholds if iou_thresholds else None)
        ev.evaluate()
        ev.accumulate()
        stats = ev.summarize()
        return {k: float(v) for k, v in zip(RESULT_KEYS, stats)}

    def evaluate_by_category(self, iou_thresholds=None):
        results_by_category = {"overall": self.evaluate(iou_thresholds)}
        for category, category_results in self.category_results.items():
            backup = self.results.copy()
            self.results = category_results
            results_by_category[category] = self.evaluate(iou_thresholds)
            self.results = backup
        return results_by_category

    def print_results(self, results=None):
        if results is None or not results:
            print("No detection results to evaluate")
            return {"mAP": 0.0, "mAP_50": 0.0, "mAP_75": 0.0, "mAP_small": 0.0, "mAP_medium": 0.0, "mAP_large": 0.0}
        print("Object Detection Evaluation Results:")
        print(f"  mAP (IoU=0.5:0.95): {results['mAP']:.4f}")
        print(f"  mAP (IoU=0.5): {results['mAP_50']:.4f}")
        print(f"  mAP (IoU=0.75): {results['mAP_75']:.4f}")
        print(f"  mAP (small objects): {results['mAP_small']:.4f}")
        print(f"  mAP (medium objects): {results['mAP_medium']:.4f}")
        print(f"  mAP (large objects): {results['mAP_large']:.4f}")
        return results

    def save_results(self, results, output_path):
        d = os.path.dirname(output_path)
        if d:
            os.makedirs(d, exist_ok=True)
        with open(output_path, "w") as f:
            json.dump(results, f, indent=2)
        print(f"Results saved to {output_path}")
