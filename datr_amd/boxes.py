"""Box utilities on the hot path (mirror of /root/reference/util/box_ops.py:9-63).

Quirks kept on purpose: `box_iou` adds 1e-6 to the union and `generalized_box_iou` adds 1e-6
to the hull area (box_ops.py:37,63); degenerate boxes assert (box_ops.py:52-53).
"""
from __future__ import annotations

import torch
from torch import Tensor


def box_cxcywh_to_xyxy(x: Tensor) -> Tensor:
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x: Tensor) -> Tensor:
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def box_area(boxes: Tensor) -> Tensor:
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_iou(boxes1: Tensor, boxes2: Tensor):
    """Pairwise IoU [N,M] and union [N,M] of xyxy boxes."""
    area1, area2 = box_area(boxes1), box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    return inter / (union + 1e-6), union


def boxes_are_valid(boxes: Tensor) -> Tensor:
    """0-d bool tensor: every xyxy box has x2 >= x1 and y2 >= y1 (the condition box_ops.py:52-53
    asserts), left on the boxes' device."""
    return (boxes[:, 2:] >= boxes[:, :2]).all()


def generalized_box_iou(boxes1: Tensor, boxes2: Tensor, check: bool = True) -> Tensor:
    """Pairwise GIoU [N,M] of xyxy boxes.  `check=False` skips the reference's asserts (each one
    a device->host sync); callers that do so test `boxes_are_valid` on the device instead."""
    if check:
        assert boxes_are_valid(boxes1)
        assert boxes_are_valid(boxes2)
    iou, union = box_iou(boxes1, boxes2)
    lt = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    hull = wh[:, :, 0] * wh[:, :, 1]
    return iou - (hull - union) / (hull + 1e-6)


def generalized_box_iou_pairs(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """GIoU of box i in boxes1 with box i in boxes2 ([n,4] xyxy each) -> [n]: the diagonal of
    generalized_box_iou(boxes1, boxes2) with the same operations per element, without building
    the n x n matrix the reference takes the diagonal of (models/dino/dino.py:563-565)."""
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    wh = (torch.min(boxes1[:, 2:], boxes2[:, 2:]) - torch.max(boxes1[:, :2], boxes2[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area1 + area2 - inter
    iou = inter / (union + 1e-6)
    hull_wh = (torch.max(boxes1[:, 2:], boxes2[:, 2:]) - torch.min(boxes1[:, :2], boxes2[:, :2])).clamp(min=0)
    hull = hull_wh[:, 0] * hull_wh[:, 1]
    return iou - (hull - union) / (hull + 1e-6)
