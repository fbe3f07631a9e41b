"""Known-answer tests of `datr_amd.evaluation.BoxEvaluator`: small detection sets whose twelve COCO
box statistics are worked out BY HAND here from the published protocol -- the rules of pycocotools'
`COCOeval.evaluateImg / accumulate / summarize`, which the reference runs through
/root/reference/datasets/coco_eval.py:22-70 (`CocoEvaluator`) and engine.py:349-523 (`evaluate`): greedy matching in
score order against ground truth sorted non-ignored first, `iou >= min(t, 1 - 1e-10)`, crowd boxes matched by
intersection / detection area and re-usable, unmatched detections outside the area range ignored, precision made
monotone from the right, sampled at the 101 recall points 0, 0.01, ..., 1 with `searchsorted(..., side="left")`,
categories without ground truth left out of the means.

Nothing here imports tests/coco_bruteforce.py (the builder's second restatement): every expected number below is an
explicit fraction with its derivation beside it.  pycocotools itself is absent from this image, so the evaluator's
parity stays "unpinned" in the sense of the task (no vector produced by the reference's own dependency exists);
these cases pin the protocol's arithmetic.

stats = [AP, AP50, AP75, AP small, AP medium, AP large, AR@1, AR@10, AR@100, AR small, AR medium, AR large]
"""
import numpy as np
import pytest
import torch

from datr_amd.evaluation import BoxEvaluator


def gt(image, xywh, cat, crowd=0, area=None):
    return {"image_id": image, "bbox": list(map(float, xywh)), "category_id": cat, "iscrowd": crowd,
            "area": float(xywh[2] * xywh[3] if area is None else area)}


def dets(rows):
    """rows: (x, y, w, h, score, label) -> PostProcess-style dict (xyxy boxes)."""
    a = np.asarray(rows, dtype=np.float64).reshape(-1, 6)
    xyxy = np.stack([a[:, 0], a[:, 1], a[:, 0] + a[:, 2], a[:, 1] + a[:, 3]], 1)
    return {"boxes": torch.tensor(xyxy, dtype=torch.float32), "scores": torch.tensor(a[:, 4], dtype=torch.float32),
            "labels": torch.tensor(a[:, 5], dtype=torch.int64)}


def evaluate(anns, preds):
    images = sorted({a["image_id"] for a in anns} | set(preds))
    ds = {"images": [{"id": i} for i in images], "annotations": anns,
          "categories": [{"id": c} for c in sorted({a["category_id"] for a in anns})]}
    ev = BoxEvaluator(ds)
    ev.update(preds)
    ev.accumulate()
    return ev.summarize(verbose=False)


BIG = [0, 0, 100, 100]          # area 10 000: "large" (> 96^2 = 9 216)
FAR = [1000, 1000, 100, 100]    # overlaps nothing


def test_101_point_interpolation_with_three_ground_truth_boxes():
    """One class, three large boxes; detections in score order: TP, FP, TP, FP, TP (exact boxes / far away).
    tp = 1 1 2 2 3, fp = 0 1 1 2 2  ->  precision 1, 1/2, 2/3, 1/2, 3/5; monotone from the right: 1, 2/3, 2/3, 3/5, 3/5;
    recall 1/3, 1/3, 2/3, 2/3, 1.  Recall points 0 .. 0.33 (34 of them) see precision 1, 0.34 .. 0.66 (33) see 2/3,
    0.67 .. 1 (34) see 3/5:  AP_t = (34 + 33 * 2/3 + 34 * 3/5) / 101 = 76.4 / 101 at every threshold (IoU is 1 or 0)."""
    g = [gt(1, [0, 0, 100, 100], 1), gt(1, [200, 0, 100, 100], 1), gt(1, [400, 0, 100, 100], 1)]
    d = dets([[0, 0, 100, 100, .9, 1], FAR[:4] + [.8, 1], [200, 0, 100, 100, .7, 1],
              [1000, 0, 100, 100, .6, 1], [400, 0, 100, 100, .5, 1]])
    s = evaluate(g, {1: d})
    ap = 76.4 / 101
    want = [ap, ap, ap, -1, -1, ap, 1 / 3, 1, 1, -1, -1, 1]
    # AR@1: the best detection alone recalls one of three boxes; AR@10 / @100: all three
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-12)


def test_iou_exactly_at_the_threshold_counts():
    """Detection [0, 0, 100, 50] on ground truth [0, 0, 100, 100]: intersection 5 000, union 10 000, IoU = 0.5 exactly:
    a match at threshold 0.50 (`iou >= t`), none at 0.55 .. 0.95.  AP50 = 1, AP75 = 0, AP = 1/10; AR = 1/10.
    The detection's own area (5 000) is "medium", the ground truth's (10 000) "large": in the LARGE range the match at
    0.50 counts (AP large = 1/10) and where it is unmatched it lies outside the range and is ignored; in the MEDIUM
    range there is no ground truth at all (-1)."""
    s = evaluate([gt(1, BIG, 1)], {1: dets([[0, 0, 100, 50, .9, 1]])})
    want = [.1, 1, 0, -1, -1, .1, .1, .1, .1, -1, -1, .1]
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-12)


def test_crowd_region_swallows_a_detection_only_while_it_matches():
    """Ground truth: a normal box and a crowd box [50, 300, 100, 100].  Detections: A (0.9) = [0, 300, 100, 100], whose
    overlap with the crowd box is 50 x 100 = half of A's own area -> crowd IoU = intersection / detection area = 0.5;
    B (0.8) = the normal box exactly.
      t = 0.50: A matches the crowd box -> ignored (neither TP nor FP); B is a TP: tp = 0 1, fp = 0 0, precision
                0 / eps = 0, 1 -> monotone 1, 1; recall 0, 1 -> AP_t = 1.
      t > 0.50: A matches nothing, lies inside the area range -> FP first: precision 0, 1/2 -> monotone 1/2, 1/2 -> 1/2.
    AP = (1 + 9 * 1/2) / 10 = 0.55, AP50 = 1, AP75 = 1/2.  The crowd box is never ground truth to be recalled: AR = 1."""
    g = [gt(1, BIG, 1), gt(1, [50, 300, 100, 100], 1, crowd=1)]
    s = evaluate(g, {1: dets([[0, 300, 100, 100, .9, 1], BIG + [.8, 1]])})
    want = [.55, 1, .5, -1, -1, .55, 0, 1, 1, -1, -1, 1]
    # AR@1: only A is considered -- ignored at 0.50, a false positive above: recall 0 at every threshold
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-12)


def test_two_detections_may_share_a_crowd_box_but_not_a_normal_one():
    """Crowd box [0, 0, 200, 200] with two detections inside it (each IoU_crowd = 1): both ignored at every threshold.
    Normal box [500, 0, 100, 100] with two exact detections: the better one is the TP, the second a FP behind full
    recall.  tp = . . 1 1 (ignored ones contribute nothing), fp = . . 0 1 -> precision at the recall-1 point is 1:
    AP = 1 at every threshold."""
    g = [gt(1, [0, 0, 200, 200], 1, crowd=1), gt(1, [500, 0, 100, 100], 1)]
    d = dets([[10, 10, 50, 50, .95, 1], [100, 100, 60, 60, .9, 1], [500, 0, 100, 100, .8, 1], [500, 0, 100, 100, .7, 1]])
    s = evaluate(g, {1: d})
    want = [1, 1, 1, -1, -1, 1, 0, 1, 1, -1, -1, 1]
    # AR@1: the single best detection sits in the crowd region (ignored): nothing recalled
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-12)


def test_max_dets_limits_recall_not_the_summarised_precision():
    """Three boxes; detections by score: FP, TP, TP, TP.  With at most 1 detection per image only the FP is seen:
    AR@1 = 0.  At 10 / 100: tp = 0 1 2 3, fp = 1 1 1 1, precision 0, 1/2, 2/3, 3/4 -> monotone 3/4 everywhere,
    recall 0, 1/3, 2/3, 1: every recall point reads 3/4.  AP = 3/4 at every threshold; AR@10 = AR@100 = 1."""
    g = [gt(1, [0, 0, 100, 100], 1), gt(1, [200, 0, 100, 100], 1), gt(1, [400, 0, 100, 100], 1)]
    d = dets([FAR + [.9, 1], [0, 0, 100, 100, .8, 1], [200, 0, 100, 100, .7, 1], [400, 0, 100, 100, .6, 1]])
    s = evaluate(g, {1: d})
    want = [.75, .75, .75, -1, -1, .75, 0, 1, 1, -1, -1, 1]
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-12)


def test_area_ranges_ignore_out_of_range_ground_truth_and_its_matches():
    """Ground truth: small 30 x 30 (900 < 32^2), medium 50 x 50 (2 500), large 200 x 200.  Detections: the small box
    exactly (0.9); the large box shifted by 18 px in x: intersection 182 x 200, union 218 x 200, IoU = 182 / 218 =
    0.8349 (0.8): a match at thresholds 0.50 .. 0.80 (seven of ten).  The medium box is missed.
      all areas, t <= 0.80: tp = 1 2, fp = 0 0; recall 1/3, 2/3 -> precision 1 up to recall 0.66 (67 points), 0 beyond:
                            67 / 101.   t > 0.80: tp = 1 1, fp = 0 1; recall 1/3: 34 / 101.
                            AP = (7 * 67 + 3 * 34) / 1010 = 571 / 1010; AP50 = AP75 = 67 / 101; AR = (7 * 2/3 + 3 * 1/3) / 10.
      small:  the medium / large boxes are ignored, and so is the large detection (matched to an ignored box, or
              unmatched and out of range): AP = AR = 1.
      medium: one box, never detected; the other detections are ignored as above: AP = AR = 0.
      large:  AP = AR = 7 / 10 (the small detection is out of range and unmatched: ignored)."""
    g = [gt(1, [0, 0, 30, 30], 1), gt(1, [100, 0, 50, 50], 1), gt(1, [300, 0, 200, 200], 1)]
    s = evaluate(g, {1: dets([[0, 0, 30, 30, .9, 1], [318, 0, 200, 200, .8, 1]])})
    ar = (7 * 2 / 3 + 3 * 1 / 3) / 10
    want = [571 / 1010, 67 / 101, 67 / 101, 1, 0, .7, ((7 + 3) * 1 / 3) / 10, ar, ar, 1, 0, .7]
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-12)


def test_categories_are_averaged_and_those_without_ground_truth_left_out():
    """Category 1: one box, detected (AP 1).  Category 2: one box, its only detection carries label 3 (AP 0).
    Category 3: detections but no ground truth -> no entry in the means (-1 slots are skipped, they are not zeros).
    AP = AR = (1 + 0) / 2."""
    g = [gt(1, BIG, 1), gt(1, [300, 0, 100, 100], 2)]
    s = evaluate(g, {1: dets([BIG + [.9, 1], [300, 0, 100, 100, .8, 3]])})
    want = [.5, .5, .5, -1, -1, .5, .5, .5, .5, -1, -1, .5]
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-12)


def test_scores_decide_across_images_and_ties_keep_image_order():
    """Two images, one box each, one class.  Image 1: a false positive (0.9) and the true positive (0.3); image 2: the
    true positive (0.6).  Pooled by score: FP (0.9), TP (0.6), TP (0.3): tp = 0 1 2, fp = 1 1 1, precision 0, 1/2, 2/3
    -> monotone 2/3 everywhere, recall 0, 1/2, 1: AP = 2/3.  AR@1 = 1/2 (image 1's best detection is the FP)."""
    g = [gt(1, BIG, 1), gt(2, BIG, 1)]
    s = evaluate(g, {1: dets([FAR + [.9, 1], BIG + [.3, 1]]), 2: dets([BIG + [.6, 1]])})
    want = [2 / 3, 2 / 3, 2 / 3, -1, -1, 2 / 3, .5, 1, 1, -1, -1, 1]
    np.testing.assert_allclose(s, want, rtol=0, atol=1e-12)
    # equal scores: the stable sort keeps image order, so swapping which image holds the false positive changes nothing
    # as long as the pooled order of TP / FP is the same
    s2 = evaluate(g, {1: dets([BIG + [.5, 1]]), 2: dets([BIG + [.5, 1]])})
    np.testing.assert_allclose(s2[:3], [1, 1, 1], rtol=0, atol=1e-12)


def test_a_better_overlap_wins_and_a_free_box_is_preferred_to_an_ignored_one():
    """One detection overlapping two boxes takes the one with the larger IoU; a detection that could match a normal box
    (IoU 0.6) and a crowd box (IoU_crowd 1.0) takes the NORMAL one: ground truth is scanned non-ignored first and the
    scan stops at the first ignored box once a non-ignored match exists."""
    # detection = [0, 0, 100, 100]; normal box [0, 25, 100, 100]: intersection 75 x 100, union 125 x 100 -> 0.6;
    # crowd box [0, 0, 300, 300] contains the detection: intersection / detection area = 1
    g = [gt(1, [0, 25, 100, 100], 1), gt(1, [0, 0, 300, 300], 1, crowd=1)]
    s = evaluate(g, {1: dets([BIG + [.9, 1]])})
    # TP at 0.50, 0.55, 0.60; above, it falls to the crowd box and is ignored (no FP): AP_t = 1 / 1 / 1 / 0 ...
    want_ap = 3 / 10
    np.testing.assert_allclose(s[:3], [want_ap, 1, 0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(s[8], 3 / 10, rtol=0, atol=1e-12)


@pytest.mark.parametrize("n", [1, 7])
def test_no_detections_at_all(n):
    """Ground truth without any detection: precision 0 at every recall point (AP 0), recall 0."""
    g = [gt(i + 1, BIG, 1) for i in range(n)]
    ev = BoxEvaluator({"images": [{"id": i + 1} for i in range(n)], "annotations": g, "categories": [{"id": 1}]})
    ev.update({i + 1: {"boxes": torch.zeros(0, 4), "scores": torch.zeros(0), "labels": torch.zeros(0, dtype=torch.int64)}
               for i in range(n)})
    ev.accumulate()
    s = ev.summarize(verbose=False)
    np.testing.assert_allclose(s, [0, 0, 0, -1, -1, 0, 0, 0, 0, -1, -1, 0], rtol=0, atol=1e-12)
