"""Hungarian matching between predictions and ground-truth boxes.

Mirror of /root/reference/models/dino/matcher.py (`HungarianMatcher` :24-95, `build_matcher`
:176-190): cost = cost_bbox * L1 + cost_class * (focal_pos - focal_neg) + cost_giou * (-GIoU)
with alpha = 0.25, gamma = 2, eps 1e-8 inside the logs and no normalisation; one rectangular
assignment per image on the [num_queries, T_i] block.

The assignment itself is the reference's third-party call,
`scipy.optimize.linear_sum_assignment` (matcher.py:20,94; SciPy un-vendored, unpinned in
requirements.txt:7).  On the host `solve_lsap` keeps that exact solver.  On the device
`solve_lsap_device` runs the same algorithm (csrc/lsap.hip: SciPy's shortest-augmenting-path
solver restated, same scan order and tie rule, double arithmetic on the fp32 costs) for every
problem of the step in one launch and leaves the indices on the device: the reference's seven
device->host synchronisations per step (matcher.py:91) disappear and index selection stays
bit-exact for a given cost matrix (BASELINE north_star: "bit-exact query/box index selection";
tests/test_lsap_gpu.py compares against SciPy, ties included).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

from .boxes import box_cxcywh_to_xyxy, boxes_are_valid, generalized_box_iou


def solve_lsap(cost: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rectangular linear-sum assignment of a CPU cost matrix -> (row_idx, col_idx) int64."""
    rows, cols = linear_sum_assignment(cost)
    return torch.as_tensor(rows, dtype=torch.int64), torch.as_tensor(cols, dtype=torch.int64)


_OFFSETS = {}


def _device_offsets(sizes, device) -> torch.Tensor:
    """[B+1] int32 prefix sums of the per-image box counts, cached per (counts, device)."""
    key = (tuple(sizes), str(device))
    t = _OFFSETS.get(key)
    if t is None:
        acc = [0]
        for n in sizes:
            acc.append(acc[-1] + int(n))
        t = torch.tensor(acc, dtype=torch.int32).to(device)
        if len(_OFFSETS) > 256:
            _OFFSETS.clear()
        _OFFSETS[key] = t
    return t


def device_lsap_supported(nq: int, sizes) -> bool:
    return nq <= 1024 and (not sizes or max(sizes) < nq)


def solve_lsap_device(C: torch.Tensor, sizes, transposed: bool = False):
    """C [G, B, nq, sum(sizes)] fp32 on the device (or, with `transposed`, [G, B, sum(sizes), nq]
    contiguous) -> (q_idx, t_idx, status): q_idx / t_idx [G, sum(sizes)] int64 on the device
    (image b's pairs at columns offsets[b]:offsets[b+1], query indices ascending -- what SciPy
    returns), status [G*B] int32 (0 = ok).  No host sync."""
    from . import _native
    if transposed:
        G, B, Tsum, nq = C.shape
    else:
        G, B, nq, Tsum = C.shape
    dev = C.device
    q_idx = torch.zeros(G, Tsum, dtype=torch.int64, device=dev)
    t_idx = torch.zeros(G, Tsum, dtype=torch.int64, device=dev)
    status = torch.zeros(max(G * B, 1), dtype=torch.int32, device=dev)
    if G * B == 0 or Tsum == 0:
        return q_idx, t_idx, status
    Ct = C.contiguous() if transposed else C.transpose(-1, -2).contiguous()
    offsets = _device_offsets(sizes, dev)
    with _native.on_device(dev):
        rc = _native.lib.datr_lsap_f32(Ct.data_ptr(), offsets.data_ptr(), G, B, Tsum, nq,
                                       max(sizes), q_idx.data_ptr(), t_idx.data_ptr(),
                                       status.data_ptr(), _native.current_stream_ptr(dev))
    _native.check(rc, "lsap")
    return q_idx, t_idx, status


FUSED_COST = True      # device cost matrix in one launch (csrc/match_cost.hip)


class IndexSets(list):
    """list (per prediction set) of per-image (query_idx, box_idx) pairs, as the reference's
    matcher returns them, plus `.packed`: the same indices as two [G, sum_i T_i] device tensors
    when they came from the device solver."""
    packed = None


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1,
                 focal_alpha=0.25):
        super().__init__()
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.focal_alpha = focal_alpha

    @torch.no_grad()
    def cost_matrix(self, outputs, targets) -> torch.Tensor:
        """[bs, num_queries, sum_i T_i] cost on the predictions' device."""
        bs, nq = outputs["pred_logits"].shape[:2]
        prob = outputs["pred_logits"].flatten(0, 1).sigmoid()
        boxes = outputs["pred_boxes"].flatten(0, 1)
        tgt_ids = torch.cat([v["labels"] for v in targets])
        tgt_bbox = torch.cat([v["boxes"] for v in targets])
        alpha, gamma = self.focal_alpha, 2.0
        neg = (1 - alpha) * (prob ** gamma) * (-(1 - prob + 1e-8).log())
        pos = alpha * ((1 - prob) ** gamma) * (-(prob + 1e-8).log())
        cost_class = pos[:, tgt_ids] - neg[:, tgt_ids]
        cost_bbox = torch.cdist(boxes, tgt_bbox, p=1)
        xyxy, tgt_xyxy = box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tgt_bbox)
        if boxes.is_cuda:
            # box_ops.py:52-53 asserts non-degenerate boxes, which costs two device->host
            # syncs; on the device the same condition is folded into `poison` instead
            self._boxes_ok = boxes_are_valid(xyxy) & boxes_are_valid(tgt_xyxy)
            cost_giou = -generalized_box_iou(xyxy, tgt_xyxy, check=False)
        else:
            self._boxes_ok = None
            cost_giou = -generalized_box_iou(xyxy, tgt_xyxy)
        C = self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou
        return C.view(bs, nq, -1)

    # NaN on the device if any assignment problem of the last call was rejected (NaN / -inf
    # costs -- SciPy raises ValueError for those), else 0: the criterion adds it to the loss so
    # that a bad matching fails loudly (non-finite loss stops training, engine.py) without a
    # host synchronisation.  None after host-side matching.
    poison = None
    _boxes_ok = None

    @torch.no_grad()
    def cost_matrix_transposed(self, logits, boxes, targets):
        """[sets, sum_i T_i, nq] cost for `sets` blocks of nq queries (logits [sets, nq, C], boxes
        [sets, nq, 4] on the device) in ONE launch (csrc/match_cost.hip): the same entries as
        cost_matrix(), stored targets x queries as the device solver reads them.  Also sets
        `_boxes_ok`."""
        from . import _native
        sets, nq, C = logits.shape
        tgt_ids = torch.cat([v["labels"] for v in targets]).contiguous()
        tgt_bbox = torch.cat([v["boxes"] for v in targets]).float().contiguous()
        T = tgt_ids.shape[0]
        logits, boxes = logits.float().contiguous(), boxes.float().contiguous()
        cost_t = torch.empty(sets, T, nq, dtype=torch.float32, device=logits.device)
        ok = torch.ones(1, dtype=torch.int32, device=logits.device)
        with _native.on_device(logits.device):
            rc = _native.lib.datr_match_cost_f32(
                logits.data_ptr(), boxes.data_ptr(), tgt_ids.data_ptr(), tgt_bbox.data_ptr(), sets, nq,
                T, C, float(self.cost_class), float(self.cost_bbox), float(self.cost_giou),
                float(self.focal_alpha), cost_t.data_ptr(), ok.data_ptr(),
                _native.current_stream_ptr(logits.device))
        _native.check(rc, "match_cost")
        self._boxes_ok = ok[0] != 0
        return cost_t

    def _indices_from_device(self, C, sizes, transposed=False):
        q_idx, t_idx, status = solve_lsap_device(C, sizes, transposed)
        bad = status.any() if self._boxes_ok is None else (status.any() | ~self._boxes_ok)
        self.poison = torch.where(bad, float("nan"), 0.0)
        out = IndexSets()
        out.packed = (q_idx, t_idx)           # [G, sum T] each, (g, image, ascending query) order
        for g in range(C.shape[0]):
            per_image, off = [], 0
            for n in sizes:
                per_image.append((q_idx[g, off:off + n], t_idx[g, off:off + n]))
                off += n
            out.append(per_image)
        return out

    @torch.no_grad()
    def forward(self, outputs, targets) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        C = self.cost_matrix(outputs, targets)
        sizes = [len(v["boxes"]) for v in targets]
        if C.is_cuda and device_lsap_supported(C.shape[1], sizes):
            return self._indices_from_device(C[None], sizes)[0]
        self.poison = None
        C = C.cpu()
        return [solve_lsap(c[i]) for i, c in enumerate(C.split(sizes, -1))]

    @torch.no_grad()
    def forward_many(self, outputs_list, targets):
        """Match several prediction sets of identical shape (the final layer, the auxiliary
        layers, the two-stage output) against the same targets: the G cost matrices are built by
        one batched evaluation and cross to the host in ONE copy (the reference pays a
        device->host sync per set, matcher.py:91, i.e. 7 per step).  Returns a list (per set) of
        the per-image (row_idx, col_idx) pairs -- identical to calling forward() on each."""
        G = len(outputs_list)
        bs, nq = outputs_list[0]["pred_logits"].shape[:2]
        sizes = [len(v["boxes"]) for v in targets]
        if FUSED_COST and outputs_list[0]["pred_logits"].is_cuda and sum(sizes) > 0 \
                and device_lsap_supported(nq, sizes):
            Ct = self.cost_matrix_transposed(
                torch.cat([o["pred_logits"] for o in outputs_list], 0),
                torch.cat([o["pred_boxes"] for o in outputs_list], 0), targets)
            return self._indices_from_device(Ct.view(G, bs, -1, nq), sizes, transposed=True)
        stacked = {"pred_logits": torch.cat([o["pred_logits"] for o in outputs_list], 0),
                   "pred_boxes": torch.cat([o["pred_boxes"] for o in outputs_list], 0)}
        C = self.cost_matrix(stacked, targets).view(G, bs, nq, -1)
        sizes = [len(v["boxes"]) for v in targets]
        if C.is_cuda and device_lsap_supported(nq, sizes):
            return self._indices_from_device(C, sizes)
        self.poison = None
        C = C.cpu()
        return [[solve_lsap(c[i]) for i, c in enumerate(C[g].split(sizes, -1))] for g in range(G)]


def build_matcher(args):
    if args.matcher_type != "HungarianMatcher":
        raise NotImplementedError(f"Unknown args.matcher_type: {args.matcher_type}")
    return HungarianMatcher(cost_class=args.set_cost_class, cost_bbox=args.set_cost_bbox,
                            cost_giou=args.set_cost_giou, focal_alpha=args.focal_alpha)
