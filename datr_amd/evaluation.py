"""Box mAP evaluation for the eval path.

The reference's `evaluate()` (/root/reference/engine.py:349-523) hands PostProcess results to
`datasets/coco_eval.py::CocoEvaluator`, a thin wrapper over **pycocotools** `COCOeval`
(un-vendored third-party code; absent from this image and the GPU box).  `BoxEvaluator` restates
the published COCO bounding-box protocol that wrapper runs -- the numbers the reference reports
as `stats['coco_eval_bbox']` (12 values, [1] = mAP50, the figure DATR's tables quote):

  * per image and category, detections sorted by descending score (stable), at most 100;
    IoU in xywh with `iscrowd` ground truth scored as intersection / detection area;
  * for each IoU threshold 0.50:0.05:0.95 a detection greedily takes the best still-free
    ground truth with IoU >= threshold (crowd boxes may be taken repeatedly; a match with a
    non-ignored box is never traded for an ignored one), ground truth outside the area range
    or crowd is "ignored", unmatched detections outside the area range are ignored;
  * precision/recall accumulated over images per category / area range / maxDets (1, 10, 100),
    precision made monotone and sampled at 101 recall points; AP = mean over the sampled
    precisions of categories that have ground truth.
Parity unpinned against pycocotools itself (not installable here): tests check hand-computed
cases and protocol properties (tests/test_evaluation_cpu.py).

Ground truth comes either from a COCO-format dict (`{"images", "annotations", "categories"}`,
what the reference's `base_ds` holds) or, with `base_ds=None`, from the targets seen during the
evaluation loop (normalised cxcywh boxes scaled by `orig_size`).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional

import numpy as np
import torch

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)
MAX_DETS = (1, 10, 100)
AREA_RNG = ((0.0, 1e10), (0.0, 32.0 ** 2), (32.0 ** 2, 96.0 ** 2), (96.0 ** 2, 1e10))
AREA_LBL = ("all", "small", "medium", "large")


def _iou_xywh(dt: np.ndarray, gt: np.ndarray, crowd: np.ndarray) -> np.ndarray:
    """[D,4] x [G,4] xywh -> [D,G]; for crowd ground truth the union is the detection's area."""
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    dx1, dy1, dx2, dy2 = dt[:, 0], dt[:, 1], dt[:, 0] + dt[:, 2], dt[:, 1] + dt[:, 3]
    gx1, gy1, gx2, gy2 = gt[:, 0], gt[:, 1], gt[:, 0] + gt[:, 2], gt[:, 1] + gt[:, 3]
    iw = (np.minimum(dx2[:, None], gx2[None]) - np.maximum(dx1[:, None], gx1[None])).clip(min=0)
    ih = (np.minimum(dy2[:, None], gy2[None]) - np.maximum(dy1[:, None], gy1[None])).clip(min=0)
    inter = iw * ih
    da = (dt[:, 2] * dt[:, 3])[:, None]
    ga = (gt[:, 2] * gt[:, 3])[None]
    union = np.where(crowd[None].astype(bool), da, da + ga - inter)
    return np.where(union > 0, inter / np.where(union > 0, union, 1.0), 0.0)


class BoxEvaluator:
    def __init__(self, base_ds: Optional[dict] = None, use_cats: bool = True):
        self.use_cats = use_cats
        self.gt: Dict[int, List[dict]] = defaultdict(list)     # image id -> annotations
        self.dt: Dict[int, dict] = {}                          # image id -> {"boxes","scores","labels"}
        self.img_ids: List[int] = []
        self.have_base = base_ds is not None
        if base_ds is not None:
            for a in base_ds["annotations"]:
                self.gt[int(a["image_id"])].append(
                    {"bbox": [float(v) for v in a["bbox"]], "category_id": int(a["category_id"]),
                     "iscrowd": int(a.get("iscrowd", 0)),
                     "area": float(a.get("area", a["bbox"][2] * a["bbox"][3]))})
        self.stats = None
        self.eval = None

    # -- feeding ----------------------------------------------------------------------------------
    def add_ground_truth(self, targets) -> None:
        """Ground truth from the loop's targets (used when no COCO dict was given)."""
        for t in targets:
            img = int(t["image_id"].reshape(-1)[0])
            if img in self.gt and self.gt[img]:
                continue
            h, w = [float(v) for v in t["orig_size"].reshape(-1)[:2]]
            boxes = t["boxes"].detach().float().cpu()
            cx, cy, bw, bh = boxes.unbind(-1) if len(boxes) else (torch.zeros(0),) * 4
            xywh = torch.stack([(cx - bw / 2) * w, (cy - bh / 2) * h, bw * w, bh * h], -1) \
                if len(boxes) else torch.zeros(0, 4)
            crowd = t.get("iscrowd", torch.zeros(len(boxes), dtype=torch.long)).cpu()
            area = t.get("area")
            self.gt[img] = [
                {"bbox": xywh[i].tolist(), "category_id": int(t["labels"][i]), "iscrowd": int(crowd[i]),
                 "area": float(area[i]) if area is not None and len(area) == len(boxes)
                 else float(xywh[i, 2] * xywh[i, 3])}
                for i in range(len(boxes))]

    def update(self, predictions: Dict[int, dict]) -> None:
        """predictions: image id -> {"scores" [K], "labels" [K], "boxes" [K,4] xyxy absolute} (the
        output of PostProcess, engine.py:414-419)."""
        for img, p in predictions.items():
            img = int(img)
            boxes = p["boxes"].detach().float().cpu().numpy().reshape(-1, 4)
            # width / height in float32 as the reference's convert_to_xywh does on the float32 tensors
            # (datasets/coco_eval.py:252-254), then doubles: `.tolist()` there, and the C routine
            # behind COCOeval's IoU works in double
            xywh = np.stack([boxes[:, 0], boxes[:, 1], boxes[:, 2] - boxes[:, 0],
                             boxes[:, 3] - boxes[:, 1]], 1).astype(np.float64) if len(boxes) else np.zeros((0, 4))
            self.dt[img] = {"boxes": xywh,
                            "scores": p["scores"].detach().float().cpu().numpy().reshape(-1),
                            "labels": p["labels"].detach().cpu().numpy().reshape(-1).astype(np.int64)}
            self.img_ids.append(img)

    def synchronize_between_processes(self) -> None:
        """Gather every rank's detections (and loop-collected ground truth) on all ranks."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        payload = (self.dt, None if self.have_base else dict(self.gt), self.img_ids)
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, payload)
        for dt, gt, ids in gathered:
            self.dt.update(dt)
            if gt:
                for k, v in gt.items():
                    if not self.gt.get(k):
                        self.gt[k] = v
        self.img_ids = sorted({i for _, _, ids in gathered for i in ids})

    # -- protocol ---------------------------------------------------------------------------------
    def _evaluate_image(self, img: int, cat: Optional[int]):
        """-> per area range: (dt scores [D], dtm [T,D] bool, dt ignore [T,D] bool, gt ignore [G])"""
        gts = [g for g in self.gt.get(img, []) if cat is None or g["category_id"] == cat]
        d = self.dt.get(img)
        if d is None:
            sel = np.zeros(0, dtype=np.int64)
            d = {"boxes": np.zeros((0, 4)), "scores": np.zeros(0), "labels": np.zeros(0, np.int64)}
        else:
            sel = np.arange(len(d["scores"])) if cat is None else np.nonzero(d["labels"] == cat)[0]
        if len(gts) == 0 and len(sel) == 0:
            return None
        order = sel[np.argsort(-d["scores"][sel], kind="mergesort")][:MAX_DETS[-1]]
        dbox, dscore = d["boxes"][order], d["scores"][order]
        darea = dbox[:, 2] * dbox[:, 3]
        gbox = np.array([g["bbox"] for g in gts], dtype=np.float64).reshape(-1, 4)
        gcrowd = np.array([g["iscrowd"] for g in gts], dtype=np.int64)
        garea = np.array([g["area"] for g in gts], dtype=np.float64)
        ious_all = _iou_xywh(dbox, gbox, gcrowd)
        out = []
        T, D, G = len(IOU_THRS), len(dbox), len(gbox)
        for lo, hi in AREA_RNG:
            gig = (gcrowd == 1) | (garea < lo) | (garea > hi)
            gorder = np.argsort(gig, kind="mergesort")          # non-ignored first
            gig_s, crowd_s = gig[gorder], gcrowd[gorder]
            ious = ious_all[:, gorder]
            gtm = -np.ones((T, G), dtype=np.int64)
            dtm = np.zeros((T, D), dtype=bool)
            dig = np.zeros((T, D), dtype=bool)
            for ti, thr in enumerate(IOU_THRS):
                for di in range(D):
                    best, m = min(thr, 1 - 1e-10), -1
                    for gi in range(G):
                        if gtm[ti, gi] >= 0 and not crowd_s[gi]:
                            continue
                        if m > -1 and not gig_s[m] and gig_s[gi]:
                            break
                        if ious[di, gi] < best:
                            continue
                        best, m = ious[di, gi], gi
                    if m == -1:
                        continue
                    dig[ti, di] = gig_s[m]
                    dtm[ti, di] = True
                    gtm[ti, m] = di
            outside = (darea < lo) | (darea > hi)
            dig = dig | (~dtm & outside[None])
            out.append((dscore, dtm, dig, gig_s))
        return out

    def accumulate(self) -> None:
        imgs = sorted(set(self.img_ids))
        if self.use_cats:
            cats = sorted({g["category_id"] for i in imgs for g in self.gt.get(i, [])} |
                          {int(c) for i in imgs if i in self.dt for c in self.dt[i]["labels"]})
        else:
            cats = [None]
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(cats), len(AREA_RNG), len(MAX_DETS)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        for k, cat in enumerate(cats):
            per_img = [e for e in (self._evaluate_image(i, cat) for i in imgs) if e is not None]
            for a in range(A):
                for mi, maxdet in enumerate(MAX_DETS):
                    if not per_img:
                        continue
                    scores = np.concatenate([e[a][0][:maxdet] for e in per_img])
                    order = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([e[a][1][:, :maxdet] for e in per_img], 1)[:, order]
                    dig = np.concatenate([e[a][2][:, :maxdet] for e in per_img], 1)[:, order]
                    gig = np.concatenate([e[a][3] for e in per_img])
                    npig = int(np.count_nonzero(~gig))
                    if npig == 0:
                        continue
                    tps = np.cumsum(dtm & ~dig, 1, dtype=np.float64)
                    fps = np.cumsum(~dtm & ~dig, 1, dtype=np.float64)
                    for t in range(T):
                        tp, fp = tps[t], fps[t]
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        recall[t, k, a, mi] = rc[-1] if nd else 0.0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        inds = np.searchsorted(rc, REC_THRS, side="left")
                        q = np.zeros(R)
                        for ri, pi in enumerate(inds):
                            if pi < nd:
                                q[ri] = pr[pi]
                        precision[t, :, k, a, mi] = q
        self.eval = {"precision": precision, "recall": recall, "categories": cats}

    def summarize(self, verbose: bool = True) -> List[float]:
        p, r = self.eval["precision"], self.eval["recall"]

        def mean(x):
            x = x[x > -1]
            return float(x.mean()) if x.size else -1.0

        def ap(iou=None, area=0, mdet=2):
            s = p[:, :, :, area, mdet] if iou is None else p[np.isclose(IOU_THRS, iou), :, :, area, mdet]
            return mean(s)

        def ar(area=0, mdet=2):
            return mean(r[:, :, area, mdet])
        self.stats = [ap(), ap(0.5), ap(0.75), ap(area=1), ap(area=2), ap(area=3),
                      ar(mdet=0), ar(mdet=1), ar(mdet=2), ar(area=1), ar(area=2), ar(area=3)]
        if verbose:
            names = ["AP@[.50:.95]", "AP@.50", "AP@.75", "AP small", "AP medium", "AP large",
                     "AR maxDets=1", "AR maxDets=10", "AR maxDets=100", "AR small", "AR medium", "AR large"]
            print("IoU metric: bbox")
            for n, v in zip(names, self.stats):
                print(f" {n:<16s} = {v:0.3f}")
        return self.stats
