"""CPU: the COCO box-mAP protocol restated in datr_amd/evaluation.py (the reference
calls pycocotools through /root/reference/datasets/coco_eval.py:22-70, which is not installable
here).  Hand-computed cases + protocol properties."""
import numpy as np
import torch

from datr_amd.evaluation import BoxEvaluator


def coco(images, anns):
    return {"images": [{"id": i} for i in images],
            "annotations": [{"image_id": i, "bbox": b, "category_id": c, "iscrowd": cr, "area": b[2] * b[3]}
                            for i, b, c, cr in anns],
            "categories": [{"id": c} for c in sorted({a[2] for a in anns})]}


def pred(boxes_xywh, scores, labels):
    b = torch.tensor(boxes_xywh, dtype=torch.float32).reshape(-1, 4)
    xyxy = torch.stack([b[:, 0], b[:, 1], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]], 1)
    return {"boxes": xyxy, "scores": torch.tensor(scores, dtype=torch.float32),
            "labels": torch.tensor(labels, dtype=torch.int64)}


def run(ds, preds):
    ev = BoxEvaluator(ds)
    ev.update(preds)
    ev.accumulate()
    return ev.summarize(verbose=False)


def test_perfect_detections_score_one():
    ds = coco([1, 2], [(1, [10, 10, 50, 60], 1, 0), (1, [100, 40, 120, 120], 2, 0), (2, [5, 5, 200, 150], 1, 0)])
    s = run(ds, {1: pred([[10, 10, 50, 60], [100, 40, 120, 120]], [0.9, 0.8], [1, 2]),
                 2: pred([[5, 5, 200, 150]], [0.7], [1])})
    assert abs(s[0] - 1) < 1e-9 and abs(s[1] - 1) < 1e-9 and abs(s[8] - 1) < 1e-9
    assert s[3] == -1.0                      # no small (< 32^2) ground truth -> undefined


def test_hand_computed_ap50():
    """One class, 2 ground-truth boxes.  Detections by score: TP, FP, TP.  Precision envelope:
    recall <= 0.5 -> 1.0, 0.5 < recall <= 1.0 -> 2/3.  101-point AP = (51 * 1 + 50 * 2/3) / 101."""
    ds = coco([1], [(1, [0, 0, 100, 100], 1, 0), (1, [200, 200, 100, 100], 1, 0)])
    s = run(ds, {1: pred([[0, 0, 100, 100], [500, 500, 50, 50], [200, 200, 100, 100]], [0.9, 0.8, 0.7], [1, 1, 1])})
    assert abs(s[1] - (51 + 50 * 2 / 3) / 101) < 1e-9
    # the same boxes shifted so that IoU = 0.6: counted at thresholds 0.5, 0.55, 0.6 only
    ds = coco([1], [(1, [0, 0, 100, 100], 1, 0)])
    s = run(ds, {1: pred([[0, 25, 100, 100]], [0.9], [1])})       # inter 75*100, union 125*100 -> 0.6
    assert abs(s[1] - 1) < 1e-9 and abs(s[2] - 0) < 1e-9 and abs(s[0] - 3 / 10) < 1e-9


def test_crowd_and_duplicate_rules():
    # a second detection of an already matched box is a false positive ...
    ds = coco([1], [(1, [0, 0, 100, 100], 1, 0)])
    s = run(ds, {1: pred([[0, 0, 100, 100], [1, 1, 100, 100]], [0.9, 0.8], [1, 1])})
    assert abs(s[1] - 1) < 1e-9              # the FP comes after full recall: AP unaffected
    s = run(ds, {1: pred([[1, 1, 100, 100], [0, 0, 100, 100]], [0.4, 0.9], [1, 1])})
    assert abs(s[1] - 1) < 1e-9              # order in the input does not matter, score does
    # ... but detections inside a crowd region are ignored, not false positives
    ds = coco([1], [(1, [0, 0, 100, 100], 1, 0), (1, [300, 300, 200, 200], 1, 1)])
    s = run(ds, {1: pred([[320, 320, 40, 40], [330, 330, 50, 50], [0, 0, 100, 100]], [0.95, 0.9, 0.5], [1, 1, 1])})
    assert abs(s[1] - 1) < 1e-9
    # wrong category = miss + false positive
    ds = coco([1], [(1, [0, 0, 100, 100], 1, 0), (1, [0, 0, 100, 100], 2, 0)])
    s = run(ds, {1: pred([[0, 0, 100, 100]], [0.9], [2])})
    assert abs(s[1] - 0.5) < 1e-9            # class 2 AP 1, class 1 AP 0


def test_max_dets_and_area_ranges():
    ds = coco([1], [(1, [0, 0, 10, 10], 1, 0), (1, [100, 100, 50, 50], 1, 0), (1, [300, 300, 200, 200], 1, 0)])
    s = run(ds, {1: pred([[0, 0, 10, 10], [100, 100, 50, 50], [300, 300, 200, 200]], [0.9, 0.8, 0.7], [1, 1, 1])})
    assert abs(s[3] - 1) < 1e-9 and abs(s[4] - 1) < 1e-9 and abs(s[5] - 1) < 1e-9    # small / medium / large
    assert abs(s[6] - 1 / 3) < 1e-9 and abs(s[8] - 1) < 1e-9                          # AR@1 = 1 of 3 boxes


def test_ground_truth_from_targets_and_evaluate_loop():
    """engine.evaluate with base_ds=None on a stub model: ground truth comes from the targets
    (normalised cxcywh * orig_size), predictions from PostProcess."""
    from datr_amd.detector import PostProcess
    from datr_amd.engine import evaluate
    from datr_amd.nested import NestedTensor

    class Model(torch.nn.Module):
        def forward(self, samples, targets=None):
            logits = torch.full((1, 5, 3), -8.0)
            logits[0, 0, 1] = 4.0
            logits[0, 1, 2] = 3.0
            boxes = torch.tensor([[[0.25, 0.25, 0.5, 0.5], [0.75, 0.75, 0.2, 0.2], [0.5, 0.5, 0.1, 0.1],
                                   [0.5, 0.5, 0.1, 0.1], [0.5, 0.5, 0.1, 0.1]]])
            return {"pred_logits": logits, "pred_boxes": boxes}

    class Crit(torch.nn.Module):
        weight_dict = {"loss_ce": 1.0}

        def forward(self, outputs, targets):
            return {"loss_ce": torch.tensor(0.5), "class_error": torch.tensor(10.0)}
    tgt = {"image_id": torch.tensor([7]), "orig_size": torch.tensor([200, 400]),
           "boxes": torch.tensor([[0.25, 0.25, 0.5, 0.5], [0.75, 0.75, 0.2, 0.2]]),
           "labels": torch.tensor([1, 2])}
    loader = [(NestedTensor(torch.zeros(1, 3, 8, 8), torch.zeros(1, 8, 8, dtype=torch.bool)), None, [tgt])]
    stats, ev = evaluate(Model(), Crit(), {"bbox": PostProcess(num_select=5)}, loader, None,
                         torch.device("cpu"), logger=object())
    assert abs(stats["coco_eval_bbox"][1] - 1) < 1e-6 and abs(stats["loss"] - 0.5) < 1e-9
    assert ev.gt[7][0]["bbox"] == [0.0, 0.0, 200.0, 100.0]


def _random_case(rng):
    """A few images with clustered ground truth (classes 1..3, some crowd boxes, small / medium /
    large areas) and detections that are jittered copies, duplicates, wrong-class copies and random
    boxes, with scores from a coarse grid (ties)."""
    images = list(range(1, int(rng.integers(1, 4)) + 1))
    anns, preds = [], {}
    for img in images:
        boxes = []
        for _ in range(int(rng.integers(0, 6))):
            side = float(rng.choice([12.0, 24.0, 40.0, 70.0, 110.0, 200.0])) * float(rng.uniform(0.8, 1.25))
            w, h = side * float(rng.uniform(0.6, 1.6)), side
            x, y = float(rng.uniform(0, 400)), float(rng.uniform(0, 300))
            cat, crowd = int(rng.integers(1, 4)), int(rng.random() < 0.15)
            anns.append((img, [x, y, w, h], cat, crowd))
            boxes.append(([x, y, w, h], cat))
        db, ds, dl = [], [], []
        for _ in range(int(rng.integers(0, 13))):
            if boxes and rng.random() < 0.75:
                (x, y, w, h), cat = boxes[int(rng.integers(0, len(boxes)))]
                j = float(rng.choice([0.0, 0.05, 0.15, 0.3, 0.6]))
                b = [x + j * w * float(rng.uniform(-1, 1)), y + j * h * float(rng.uniform(-1, 1)),
                     w * (1 + j * float(rng.uniform(-0.5, 0.5))), h * (1 + j * float(rng.uniform(-0.5, 0.5)))]
                if rng.random() < 0.15:
                    cat = int(rng.integers(1, 4))
            else:
                b = [float(rng.uniform(0, 400)), float(rng.uniform(0, 300)), float(rng.uniform(5, 150)),
                     float(rng.uniform(5, 150))]
                cat = int(rng.integers(1, 4))
            db.append(b)
            ds.append(float(rng.integers(1, 11)) / 10.0)
            dl.append(cat)
        if db or rng.random() < 0.8:
            preds[img] = pred(db, ds, dl) if db else pred(np.zeros((0, 4)).tolist(), [], [])
    return images, anns, preds


def test_matches_an_independent_restatement_on_1000_random_image_sets():
    """BoxEvaluator against tests/coco_bruteforce.py -- an independently written
    restatement of the published protocol -- on 1 000 seeded random image sets with crowd boxes, all
    three area ranges, tied scores, duplicates, wrong-class and empty detections: all 12 summary
    numbers agree to 1e-9, with and without categories."""
    import coco_bruteforce as bf
    rng = np.random.default_rng(20260930)
    checked = nonzero = 0
    for case in range(1000):
        images, anns, preds = _random_case(rng)
        ds = coco(images, anns)
        gts = {}
        for img, b, c, cr in anns:
            gts.setdefault(img, []).append((b, c, cr, b[2] * b[3]))
        dts = {}
        for img, p in preds.items():
            bx = p["boxes"].numpy()             # float32: width / height are formed in float32 (coco_eval.py:252-254)
            dts[img] = [([float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])],
                         float(s), int(l)) for b, s, l in zip(bx, p["scores"].numpy(), p["labels"].numpy())]
        for use_cats in ((True, False) if case % 10 == 0 else (True,)):
            ev = BoxEvaluator(ds, use_cats=use_cats)
            ev.update(preds)
            ev.accumulate()
            mine = ev.summarize(verbose=False)
            want = bf.coco_stats(list(preds.keys()), gts, dts, use_cats=use_cats)
            assert np.allclose(mine, want, rtol=0, atol=1e-9), (case, use_cats, mine, want)
            checked += 1
            nonzero += mine[0] > 0
    assert checked >= 1000 and nonzero > 600
