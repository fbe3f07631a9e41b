"""Pseudo-label plumbing of the teacher-student stage.

Mirror of /root/reference/models/dino/self_training_utils.py: `get_unlabel_img` (:15-20),
`get_pseudo_label_via_threshold` (:23-52), `deal_pesudo_label` (:54-70),
`rescale_pseudo_targets` (:72-96, class-aware NMS 0.7, keep 100), `spilt_output` (:98-106),
`get_valid_output` (:109-146).  Names keep the reference's spelling.  The per-class threshold
lookup stays on the device (the reference round-trips labels through numpy, :36) and
`batched_nms` is restated here because torchvision is not a dependency of this package.
"""
from __future__ import annotations

import numpy as np
import torch

from . import boxes as box_ops


def get_unlabel_img(nestedtensor):
    images, _ = nestedtensor.decompose()
    return images[images.shape[0] // 2:]


def get_pseudo_label_via_threshold(results, threshold=0.8):
    """results: PostProcess output per image; threshold: scalar or per-class array.
    -> (indices of images with >= 1 pseudo label, {i: labels}, {i: boxes}, {i: scores})."""
    idx_list, labels_d, boxes_d, scores_d = [], {}, {}, {}
    for n, r in enumerate(results):
        thr = torch.as_tensor(np.asarray(threshold), dtype=r["scores"].dtype,
                              device=r["scores"].device)
        per_box = thr[r["labels"]] if thr.dim() > 0 else thr
        keep = r["scores"] >= per_box
        if bool(keep.any()):
            idx_list.append(n)
            labels_d[n], boxes_d[n], scores_d[n] = r["labels"][keep], r["boxes"][keep], r["scores"][keep]
    return idx_list, labels_d, boxes_d, scores_d


def deal_pesudo_label(unlabel_target_list, idx_list, pesudo_labels_dict, pesudo_boxes_dict,
                      scores_dcit):
    out = {}
    for i in idx_list:
        t = unlabel_target_list[i]
        d = {"labels": pesudo_labels_dict[i], "boxes": pesudo_boxes_dict[i], "scores": scores_dcit[i]}
        for k in ("image_id", "area", "iscrowd", "orig_size", "size"):
            if k in t:
                d[k] = t[k]
        out[i] = d
    return out


def nms_host(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS on xyxy boxes -> kept indices by decreasing score, ties by index (IoU without the
    reference box_ops' +1e-6: this restates torchvision.ops.nms, which the reference calls).
    Host formulation for CPU tensors; the checker of the device kernel (tests/test_selftrain_gpu.py)."""
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    order = scores.argsort(descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(b[:, None, :2], b[None, :, :2])
    rb = torch.min(b[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    iou = (inter / (area[:, None] + area[None, :] - inter)).cpu()       # [n, n], one copy
    n = b.shape[0]
    alive = torch.ones(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        alive &= ~(iou[i] > iou_threshold)
        alive[i] = False
    return order[torch.as_tensor(keep, dtype=torch.int64, device=boxes.device)]


def batched_nms(boxes, scores, idxs, iou_threshold):
    """Class-aware NMS: boxes of different classes never suppress each other (coordinate
    offset trick, as torchvision.ops.boxes.batched_nms).  Device tensors go through ONE launch of
    csrc/nms.hip (rank, offsets, IoU and the greedy scan on the device; the same float32
    arithmetic, so the kept indices equal the host formulation's bit for bit)."""
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    if boxes.is_cuda:
        import ctypes  # noqa: F401
        from . import _native
        n = boxes.shape[0]
        b = boxes.detach().contiguous().float()
        s = scores.detach().contiguous().float()
        lab = idxs.contiguous().to(torch.int64)
        keep = torch.empty(n, dtype=torch.int64, device=boxes.device)
        count = torch.zeros(1, dtype=torch.int32, device=boxes.device)
        with _native.on_device(boxes.device):
            rc = _native.lib.datr_nms_f32(b.data_ptr(), s.data_ptr(), lab.data_ptr(), n, float(iou_threshold),
                                          keep.data_ptr(), count.data_ptr(),
                                          _native.current_stream_ptr(boxes.device))
        _native.check(rc, "nms")
        return keep[:int(count.item())]
    offsets = idxs.to(boxes) * (boxes.max() + 1)
    return nms_host(boxes + offsets[:, None], scores, iou_threshold)


def rescale_pseudo_targets(unlabel_samples_img, unlabel_pseudo_targets, nms_th=0.7):
    _, _, H, W = unlabel_samples_img.shape
    for k, t in unlabel_pseudo_targets.items():
        h_real, w_real = [float(v) for v in t["size"].cpu()]
        b = box_ops.box_cxcywh_to_xyxy(t["boxes"])
        b[:, [0, 2]] = b[:, [0, 2]] * W
        b[:, [1, 3]] = b[:, [1, 3]] * H
        keep = batched_nms(b, t["scores"], t["labels"], nms_th)[:100]
        b, t["scores"], t["labels"] = b[keep], t["scores"][keep], t["labels"][keep]
        b = box_ops.box_xyxy_to_cxcywh(b)
        b[:, [0, 2]] = b[:, [0, 2]] / w_real
        b[:, [1, 3]] = b[:, [1, 3]] / h_real
        t["boxes"] = b
    return unlabel_pseudo_targets


def spilt_output(output_dict):
    source, pseudo = {}, {}
    for k, v in output_dict.items():
        (pseudo if "target" in k else source)[k] = v
    return source, pseudo


def get_valid_output(target_outputs, target_pseudo_labels_dict, idx):
    pick = lambda d: {"pred_logits": d["pred_logits"][idx, :, :], "pred_boxes": d["pred_boxes"][idx, :, :]}
    valid = {}
    for k, v in target_outputs.items():
        if "pred" in k:
            valid[k] = v[idx, :, :]
        elif "aux_outputs_target" in k:
            valid[k] = [pick(d) for d in v]
        elif "interm_outputs" in k:
            valid[k] = pick(v)
    return valid, list(target_pseudo_labels_dict.values())
