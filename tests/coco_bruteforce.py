"""Test-side, independently written restatement of the published COCO bounding-box evaluation
protocol (the algorithm of pycocotools' COCOeval.evaluateImg / accumulate / summarize, which the
reference reaches through /root/reference/datasets/coco_eval.py:22-70; pycocotools itself is not
installable here).  Deliberately structured differently from datr_amd/evaluation.py -- plain
python loops over one (category, area range, maxDets, IoU threshold) cell at a time, dict-based
matching, no shared helpers -- so that an indexing or ordering slip in either shows up as a
disagreement.  Used by tests/test_evaluation_cpu.py on random image sets and by the GPU test of
engine.evaluate."""
import numpy as np

# The published parameter grids are DEFINED through np.linspace (cocoeval.py Params.setDetParams), and its
# rounding matters: linspace(0, 1, 101)[70] is 0.7000000000000001, so a recall of exactly 7/10 does not
# reach the 71st sampling point -- a first version of this file used r / 100.0 and disagreed there.
IOUS = [float(v) for v in np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)]
RECS = [float(v) for v in np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)]
AREAS = [(0.0, 1e10), (0.0, 1024.0), (1024.0, 9216.0), (9216.0, 1e10)]
MAXDETS = [1, 10, 100]


def _iou(d, g, crowd):
    ix = min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0])
    iy = min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1])
    if ix <= 0 or iy <= 0:
        return 0.0
    inter = ix * iy
    union = d[2] * d[3] if crowd else d[2] * d[3] + g[2] * g[3] - inter
    return inter / union if union > 0 else 0.0


def _cell(images, gts, dts, cat, area, maxdet, thr):
    """AP and recall of one cell, or (None, None) when the cell has no countable ground truth.
    gts[img] = list of (xywh, cat, crowd, area); dts[img] = list of (xywh, score, cat) in input order."""
    lo, hi = area
    pooled = []          # (score, image order, rank in image, matched, ignored)
    npos = 0
    for order_i, img in enumerate(images):
        g = [x for x in gts.get(img, []) if cat is None or x[1] == cat]
        d = [x for x in dts.get(img, []) if cat is None or x[2] == cat]
        if not g and not d:
            continue
        d = sorted(enumerate(d), key=lambda kv: -kv[1][1])      # stable: ties keep input order
        d = [x for _, x in d][:100][:maxdet]
        ign = [bool(x[2]) or x[3] < lo or x[3] > hi for x in g]
        g_sorted = [i for i in range(len(g)) if not ign[i]] + [i for i in range(len(g)) if ign[i]]
        npos += sum(1 for i in range(len(g)) if not ign[i])
        taken = {}
        for rank, (box, score, _) in enumerate(d):
            best_iou, best = min(thr, 1 - 1e-10), None
            for gi in g_sorted:
                crowd = bool(g[gi][2])
                if gi in taken and not crowd:
                    continue
                if best is not None and not ign[best] and ign[gi]:
                    break                      # a real match is never traded for an ignored box
                v = _iou(box, g[gi][0], crowd)
                if v < best_iou:
                    continue
                best_iou, best = v, gi
            if best is not None:
                taken[best] = rank
                pooled.append((score, order_i, rank, True, ign[best]))
            else:
                a = box[2] * box[3]
                pooled.append((score, order_i, rank, False, a < lo or a > hi))
    if npos == 0:
        return None, None
    pooled.sort(key=lambda x: (-x[0], x[1], x[2]))            # mergesort over the per-image concatenation
    tp = fp = 0
    rec, prec = [], []
    for _, _, _, matched, ignored in pooled:
        if not ignored:
            if matched:
                tp += 1
            else:
                fp += 1
        rec.append(tp / npos)
        prec.append(tp / (tp + fp + 2.220446049250313e-16))
    for i in range(len(prec) - 1, 0, -1):
        if prec[i] > prec[i - 1]:
            prec[i - 1] = prec[i]
    total = 0.0
    for rt in RECS:
        # first index with recall >= rt  (np.searchsorted(rec, rt, side='left'))
        lo_i, hi_i = 0, len(rec)
        while lo_i < hi_i:
            mid = (lo_i + hi_i) // 2
            if rec[mid] < rt:
                lo_i = mid + 1
            else:
                hi_i = mid
        total += prec[lo_i] if lo_i < len(prec) else 0.0
    return total / 101.0, (rec[-1] if rec else 0.0)


def coco_stats(images, gts, dts, use_cats=True):
    """The 12 numbers of COCOeval.summarize() for bounding boxes."""
    images = sorted(set(images))
    if use_cats:
        cats = sorted({x[1] for i in images for x in gts.get(i, [])} | {x[2] for i in images for x in dts.get(i, [])})
    else:
        cats = [None]

    def mean_over(cells):
        vals = [v for v in cells if v is not None]
        return sum(vals) / len(vals) if vals else -1.0

    def ap(thrs, area, md):
        return mean_over([_cell(images, gts, dts, c, AREAS[area], MAXDETS[md], t)[0] for t in thrs for c in cats])

    def ar(area, md):
        return mean_over([_cell(images, gts, dts, c, AREAS[area], MAXDETS[md], t)[1] for t in IOUS for c in cats])
    return [ap(IOUS, 0, 2), ap([IOUS[0]], 0, 2), ap([IOUS[5]], 0, 2), ap(IOUS, 1, 2), ap(IOUS, 2, 2), ap(IOUS, 3, 2),
            ar(0, 0), ar(0, 1), ar(0, 2), ar(1, 2), ar(2, 2), ar(3, 2)]
