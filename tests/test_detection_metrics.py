"""Known-answer tests of the COCO box evaluation restated in adam-dehaze_amd/detection_metrics.py (pycocotools is absent: the
expected values below are derived by hand from the published algorithm -- greedy matching of score-sorted detections per IoU
threshold, precision envelope, 101 recall levels -- and written out in the comments).  CPU only."""
import json

import numpy as np
import pytest

from adam_dehaze_amd.detection_metrics import RESULT_KEYS, DetectionMetrics, box_iou_xywh
from adam_dehaze_amd.metrics import DetectionMetrics as ReExported


def _ann_file(tmp_path, images, anns, cats=(1, 2)):
    p = tmp_path / "instances.json"
    p.write_text(json.dumps({"images": [{"id": i, "file_name": f"{i}.png"} for i in images], "annotations": anns,
                             "categories": [{"id": c, "name": str(c)} for c in cats]}))
    return str(p)


def _gt(i, img, cat, box, crowd=0, area=None):
    return {"id": i, "image_id": img, "category_id": cat, "bbox": box, "iscrowd": crowd,
            "area": box[2] * box[3] if area is None else area}


def test_reexport_and_keys():
    assert ReExported is DetectionMetrics
    assert RESULT_KEYS[:3] == ["mAP", "mAP_50", "mAP_75"] and len(RESULT_KEYS) == 12


def test_box_iou_regular_and_crowd():
    iou = box_iou_xywh([[0, 0, 10, 10]], [[5, 0, 10, 10], [0, 0, 20, 20]], [0, 1])
    assert iou[0, 0] == pytest.approx(50 / 150)       # intersection 5 x 10 over 100 + 100 - 50
    assert iou[0, 1] == pytest.approx(1.0)            # crowd: intersection over the DETECTION's area
    assert box_iou_xywh([], [[0, 0, 1, 1]], [0]).shape == (0, 1)


def test_perfect_detections_score_one(tmp_path):
    anns = [_gt(1, 10, 1, [0, 0, 50, 50]), _gt(2, 10, 2, [100, 100, 120, 120]), _gt(3, 11, 1, [5, 5, 20, 20])]
    m = DetectionMetrics(_ann_file(tmp_path, [10, 11], anns))
    for a in anns:
        m.add_detection_result(a["image_id"], a["category_id"], a["bbox"], 0.9, "low")
    r = m.evaluate()
    assert list(r) == RESULT_KEYS
    # areas: 2500 (medium), 14400 (large), 400 (small): one perfectly detected box in each range
    for k in RESULT_KEYS:
        assert r[k] == pytest.approx(1.0), k
    by = m.evaluate_by_category()
    assert set(by) == {"overall", "low"} and by["low"] == by["overall"]


def test_hand_computed_precision_recall(tmp_path):
    """One image, one category, two ground-truth boxes A, B (area 100 x 100 = large).  Detections by score:
         0.9  = A exactly                 (IoU 1.0)
         0.8  = far away                  (false positive)
         0.7  = B shifted by 25 of 100 px (IoU = 75 / 125 = 0.6)
       IoU thresholds 0.50, 0.55, 0.60 (the shifted box matches: min(t, 1 - 1e-10) <= 0.6): TP FP TP -> precision 1, 1/2, 2/3 ->
       envelope 1, 2/3, 2/3; recall 1/2, 1/2, 1.  101 recall levels: r <= 0.5 (51 levels) read precision 1, r > 0.5 (50 levels) 2/3.
       Thresholds 0.65 .. 0.95 (7 of them): TP FP FP -> recall stops at 1/2: 51 levels at precision 1, 50 at 0."""
    A, B = [0, 0, 100, 100], [300, 300, 100, 100]
    m = DetectionMetrics(_ann_file(tmp_path, [1], [_gt(1, 1, 1, A), _gt(2, 1, 1, B)]))
    m.add_detection_result(1, 1, A, 0.9)
    m.add_detection_result(1, 1, [600, 600, 120, 120], 0.8)     # (large, like A and B: it counts in the 'large' range too)
    m.add_detection_result(1, 1, [325, 300, 100, 100], 0.7)
    r = m.evaluate()
    ap_lo = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101
    ap_hi = 51.0 / 101
    assert r["mAP_50"] == pytest.approx(ap_lo, abs=1e-12)
    assert r["mAP_75"] == pytest.approx(ap_hi, abs=1e-12)
    assert r["mAP"] == pytest.approx((3 * ap_lo + 7 * ap_hi) / 10, abs=1e-12)
    assert r["mAP_large"] == pytest.approx(r["mAP"], abs=1e-12)
    assert r["mAP_small"] == -1.0 and r["mAP_medium"] == -1.0          # no ground truth in those ranges
    # AR: recall at the end of the list, averaged over thresholds.  maxDets = 1 keeps only the 0.9 detection: recall 1/2 everywhere;
    # maxDets = 10 / 100: 1.0 at three thresholds, 1/2 at seven
    assert r["AR_1"] == pytest.approx(0.5)
    assert r["AR_10"] == pytest.approx((3 * 1.0 + 7 * 0.5) / 10)
    assert r["AR_100"] == r["AR_10"] == r["AR_large"]
    # a custom threshold list without 0.5 / 0.75 leaves those two statistics empty (-1), as COCOeval's lookup does
    r2 = m.evaluate(iou_thresholds=[0.6, 0.9])
    assert r2["mAP_50"] == -1.0 and r2["mAP_75"] == -1.0
    assert r2["mAP"] == pytest.approx((ap_lo + ap_hi) / 2, abs=1e-12)


def test_crowd_and_area_rules(tmp_path):
    """A crowd region is never a miss and absorbs any number of detections without false positives; a small detection without a
    match is ignored when the 'large' range is evaluated (and counts as a false positive in 'all')."""
    big, crowd = [0, 0, 200, 200], [400, 0, 300, 300]
    m = DetectionMetrics(_ann_file(tmp_path, [1], [_gt(1, 1, 1, big), _gt(2, 1, 1, crowd, crowd=1)]))
    m.add_detection_result(1, 1, big, 0.9)
    m.add_detection_result(1, 1, [410, 10, 100, 100], 0.85)      # inside the crowd region: IoU (over its own area) 1
    m.add_detection_result(1, 1, [500, 100, 100, 100], 0.8)      # inside the crowd region too
    m.add_detection_result(1, 1, [0, 500, 10, 10], 0.95)         # small stray box, best score
    r = m.evaluate()
    # 'all': FP (0.95), TP (0.9), ignored, ignored -> precision 0, 1/2 -> envelope 1/2, 1/2; recall 0, 1: every level reads 1/2
    assert r["mAP"] == pytest.approx(0.5)
    # 'large': the stray small box is outside the range and unmatched -> ignored: TP only
    assert r["mAP_large"] == pytest.approx(1.0)
    assert r["AR_100"] == pytest.approx(1.0) and r["AR_1"] == pytest.approx(0.0)     # maxDets = 1 keeps only the stray box


def test_max_dets_and_score_ties_are_stable(tmp_path):
    """100 detections at most per image and category are evaluated (score order, mergesort = insertion order for ties)."""
    gts = [_gt(i + 1, 1, 1, [20.0 * i, 0, 10, 10]) for i in range(120)]
    m = DetectionMetrics(_ann_file(tmp_path, [1], gts))
    for g in gts:
        m.add_detection_result(1, 1, g["bbox"], 0.5)             # all tied
    r = m.evaluate()
    assert r["AR_100"] == pytest.approx(100 / 120)
    assert r["AR_10"] == pytest.approx(10 / 120) and r["AR_1"] == pytest.approx(1 / 120)
    # precision is 1 up to recall 100/120 = 0.8333: levels 0 .. 0.83 (84 of 101) read 1, the rest 0
    assert r["mAP_50"] == pytest.approx(84 / 101)


def test_empty_and_foreign_results(tmp_path, capsys):
    f = _ann_file(tmp_path, [], [], cats=(1, 2, 3, 4, 5))        # the reference's generated dummy annotation file
    m = DetectionMetrics(f)
    assert m.evaluate() == {}
    assert "No detection results to evaluate" in capsys.readouterr().out
    assert m.print_results({})["mAP"] == 0.0
    m.add_detection_result(7, 3, [0, 0, 5, 5], 0.9, "high")
    with pytest.raises(AssertionError, match="Results do not correspond to current coco set"):
        m.evaluate()
    out = tmp_path / "res" / "hazy_detection_results.json"
    m.save_results({"overall": {}}, str(out))
    assert json.loads(out.read_text()) == {"overall": {}}


def test_images_without_detections_count_as_misses(tmp_path):
    gts = [_gt(1, 1, 1, [0, 0, 40, 40]), _gt(2, 2, 1, [0, 0, 40, 40])]
    m = DetectionMetrics(_ann_file(tmp_path, [1, 2, 3], gts))
    m.add_detection_result(1, 1, [0, 0, 40, 40], 0.9)
    r = m.evaluate()
    # recall stops at 1/2: 51 levels at precision 1
    assert r["mAP"] == pytest.approx(51 / 101) and r["AR_100"] == pytest.approx(0.5)
    assert np.isfinite(list(r.values())).all()
