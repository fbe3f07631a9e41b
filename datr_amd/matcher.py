"""Hungarian matching between predictions and ground-truth boxes.

Mirror of /root/reference/models/dino/matcher.py (`HungarianMatcher` :24-95, `build_matcher`
:176-190): cost = cost_bbox * L1 + cost_class * (focal_pos - focal_neg) + cost_giou * (-GIoU)
with alpha = 0.25, gamma = 2, eps 1e-8 inside the logs and no normalisation; one rectangular
assignment per image on the [num_queries, T_i] block.

The assignment itself is the reference's third-party call,
`scipy.optimize.linear_sum_assignment` (matcher.py:20,94; SciPy un-vendored, unpinned in
requirements.txt:7).  `solve_lsap` keeps that exact solver so index selection is bit-exact for
a given cost matrix (BASELINE north_star: "bit-exact query/box index selection").
"""
from __future__ import annotations

from typing import List, Tuple

import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

from .boxes import box_cxcywh_to_xyxy, generalized_box_iou


def solve_lsap(cost: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rectangular linear-sum assignment of a CPU cost matrix -> (row_idx, col_idx) int64."""
    rows, cols = linear_sum_assignment(cost)
    return torch.as_tensor(rows, dtype=torch.int64), torch.as_tensor(cols, dtype=torch.int64)


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
        cost_giou = -generalized_box_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tgt_bbox))
        C = self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou
        return C.view(bs, nq, -1)

    @torch.no_grad()
    def forward(self, outputs, targets) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        C = self.cost_matrix(outputs, targets).cpu()
        sizes = [len(v["boxes"]) for v in targets]
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
        stacked = {"pred_logits": torch.cat([o["pred_logits"] for o in outputs_list], 0),
                   "pred_boxes": torch.cat([o["pred_boxes"] for o in outputs_list], 0)}
        C = self.cost_matrix(stacked, targets).view(G, bs, nq, -1).cpu()
        sizes = [len(v["boxes"]) for v in targets]
        return [[solve_lsap(c[i]) for i, c in enumerate(C[g].split(sizes, -1))] for g in range(G)]


def build_matcher(args):
    if args.matcher_type != "HungarianMatcher":
        raise NotImplementedError(f"Unknown args.matcher_type: {args.matcher_type}")
    return HungarianMatcher(cost_class=args.set_cost_class, cost_bbox=args.set_cost_bbox,
                            cost_giou=args.set_cost_giou, focal_alpha=args.focal_alpha)
