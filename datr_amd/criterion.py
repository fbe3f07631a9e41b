"""Training criterion of DATR/DINO: Hungarian-matched focal / L1 / GIoU losses for the final,
auxiliary, two-stage and de-noising outputs, plus the three domain-adaptation losses.

Mirror of /root/reference/models/dino/dino.py `SetCriterion` (:486-941) and of
/root/reference/models/dino/utils.py `sigmoid_focal_loss` (:79-104).  Loss-dict keys, their
order of creation and every normaliser follow the reference:
  * focal:  sum over queries of the per-query class mean, / num_boxes, * num_queries;
  * DN:     analytic indices, normaliser num_boxes * num_dn_groups;
  * num_boxes = clamp(all_reduce(sum T_i) / world, min = 1);
  * loss_da: mean BCE on source tokens (label 0) + mean BCE on target tokens (label 1);
  * loss_proto_da: BCE masked by class presence but averaged over all 2C entries;
  * loss_contrast_da: cross-entropy with soft (masked identity) targets on cosine logits.

Differences from the reference that do not change values: prediction sets of identical shape
(final + 5 auxiliary + two-stage; the 6 de-noising sets) are evaluated as one batched family
(`_family_losses`) -- one fused focal-loss launch, pair-wise GIoU, ONE device->host copy for all
7 Hungarian cost matrices (`HungarianMatcher.forward_many`) instead of 7.  The reference's
degenerate-box asserts (box_ops.py:52-53) remain in the matcher; inside the loss they are
redundant for matched sets (same boxes) and a NaN box trips the engine's non-finite guard.
`loss_labels` / `loss_boxes` / `loss_cardinality` / `get_loss` keep the reference's per-set API.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from . import boxes as box_ops
from .focal import MAX_BOX_LOSS_PAIRS, box_loss_sums
from .focal import sigmoid_focal_loss_sums as focal_loss_sums
from .nested import accuracy, get_world_size, is_dist_avail_and_initialized


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2):
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


_STATIC = {}


FUSED_BOX_LOSS = True


def weighted_total(loss_dict, weight_dict):
    """sum_k loss_dict[k] * weight_dict[k] over the keys present in both, in loss_dict's order
    (/root/reference/engine.py:77: `sum(loss_dict[k] * weight_dict[k] for k in ...)`), evaluated
    as ONE stacked dot product: 3 launches instead of 2 x 82, and 3 instead of ~330 autograd nodes
    in backward.  d total / d loss_k = weight_k exactly, as in the reference's chain of adds."""
    keys = [k for k in loss_dict.keys() if k in weight_dict]
    if not keys:
        return 0
    vals = torch.stack([loss_dict[k] for k in keys])
    w = _cached(("w", tuple(keys), tuple(float(weight_dict[k]) for k in keys), str(vals.device),
                 vals.dtype), lambda: torch.tensor([float(weight_dict[k]) for k in keys],
                                                   dtype=vals.dtype).to(vals.device))
    return (vals * w).sum()


def _cached(key, make):
    t = _STATIC.get(key)
    if t is None:
        if len(_STATIC) > 512:
            _STATIC.clear()
        t = _STATIC[key] = make()
    return t


def _device_lengths(counts, device):
    """float tensor of the per-image box counts, uploaded once per distinct count tuple (a
    per-step upload from a python list is a synchronising copy)."""
    return _cached(("len", counts, str(device)),
                   lambda: torch.tensor(counts, dtype=torch.float32).to(device))


def _static_index_columns(G, counts, device):
    """(set index, image index, box offset) of every matched pair in (set, image, rank) order."""
    def make():
        b = torch.tensor([i for i, n in enumerate(counts) for _ in range(n)], dtype=torch.int64)
        off, acc = [], 0
        for n in counts:
            off += [acc] * n
            acc += n
        off = torch.tensor(off, dtype=torch.int64)
        g = torch.arange(G, dtype=torch.int64).repeat_interleave(len(b))
        return g.to(device), b.repeat(G).to(device), off.repeat(G).to(device)
    return _cached(("cols", G, counts, str(device)), make)


FUSED_CONTRAST = __import__("os").environ.get("DATR_FUSED_CONTRAST", "1") != "0"       # A/B switch


class _ContrastLoss(torch.autograd.Function):
    """loss_contrast_da as one launch (csrc/prototypes.hip::contrast_loss_kernel): the scalar loss and, since it is
    a scalar, its gradients with respect to both prototype sets, which backward only scales."""

    @staticmethod
    def forward(ctx, q_s, q_t, g, m_s, m_t):
        from . import _native
        K, C = q_s.shape
        loss = torch.empty(1, device=q_s.device, dtype=torch.float32)
        dq_s, dq_t = torch.empty_like(q_s), torch.empty_like(q_t)
        with _native.on_device(q_s.device):
            rc = _native.lib.datr_contrast_loss_f32(q_s.data_ptr(), q_t.data_ptr(), g.data_ptr(), m_s.data_ptr(),
                                                    m_t.data_ptr(), K, C, 1e-12, loss.data_ptr(), dq_s.data_ptr(),
                                                    dq_t.data_ptr(), _native.current_stream_ptr(q_s.device))
        _native.check(rc, "contrast_loss")
        ctx.save_for_backward(dq_s, dq_t)
        return loss[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        dq_s, dq_t = ctx.saved_tensors
        return dq_s * go, dq_t * go, None, None, None


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, focal_alpha, losses):
        super().__init__()
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.losses = losses
        self.focal_alpha = focal_alpha

    # ---- matched losses ---------------------------------------------------------------------
    def loss_labels(self, outputs, targets, indices, num_boxes, log=True):
        src_logits = outputs["pred_logits"]
        idx = self._get_src_permutation_idx(indices)
        matched_cls = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
        target_classes = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64,
                                    device=src_logits.device)
        target_classes[idx] = matched_cls
        onehot = torch.zeros([src_logits.shape[0], src_logits.shape[1], src_logits.shape[2] + 1],
                             dtype=src_logits.dtype, device=src_logits.device)
        onehot.scatter_(2, target_classes.unsqueeze(-1), 1)
        onehot = onehot[:, :, :-1]
        loss_ce = sigmoid_focal_loss(src_logits, onehot, num_boxes, alpha=self.focal_alpha,
                                     gamma=2) * src_logits.shape[1]
        losses = {"loss_ce": loss_ce}
        if log:
            losses["class_error"] = 100 - accuracy(src_logits[idx], matched_cls)[0]
        return losses

    @torch.no_grad()
    def loss_cardinality(self, outputs, targets, indices, num_boxes):
        logits = outputs["pred_logits"]
        tgt_lengths = torch.as_tensor([len(v["labels"]) for v in targets], device=logits.device)
        card_pred = (logits.argmax(-1) != logits.shape[-1] - 1).sum(1)
        return {"cardinality_error": F.l1_loss(card_pred.float(), tgt_lengths.float())}

    def loss_boxes(self, outputs, targets, indices, num_boxes):
        idx = self._get_src_permutation_idx(indices)
        src_boxes = outputs["pred_boxes"][idx]
        target_boxes = torch.cat([t["boxes"][i] for t, (_, i) in zip(targets, indices)], dim=0)
        l1 = F.l1_loss(src_boxes, target_boxes, reduction="none")
        losses = {"loss_bbox": l1.sum() / num_boxes}
        giou = torch.diag(box_ops.generalized_box_iou(box_ops.box_cxcywh_to_xyxy(src_boxes),
                                                      box_ops.box_cxcywh_to_xyxy(target_boxes)))
        losses["loss_giou"] = (1 - giou).sum() / num_boxes
        with torch.no_grad():
            losses["loss_xy"] = l1[..., :2].sum() / num_boxes
            losses["loss_hw"] = l1[..., 2:].sum() / num_boxes
        return losses

    @staticmethod
    def _get_src_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        src_idx = torch.cat([src for (src, _) in indices])
        return batch_idx, src_idx

    @staticmethod
    def _get_tgt_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(tgt, i) for i, (_, tgt) in enumerate(indices)])
        tgt_idx = torch.cat([tgt for (_, tgt) in indices])
        return batch_idx, tgt_idx

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        table = {"labels": self.loss_labels, "cardinality": self.loss_cardinality,
                 "boxes": self.loss_boxes}
        assert loss in table, f"do you really want to compute {loss} loss?"
        return table[loss](outputs, targets, indices, num_boxes, **kwargs)

    # ---- domain-adaptation losses -----------------------------------------------------------
    def loss_da(self, outputs):
        ready = getattr(outputs, "_ready_event", None)
        if ready is not None:       # produced on a side stream (detector.py, overlap_d_img)
            torch.cuda.current_stream().wait_event(ready)
        B = outputs.shape[0]
        assert B % 2 == 0
        src, tgt = outputs[:B // 2], outputs[B // 2:]
        return (F.binary_cross_entropy_with_logits(src, torch.zeros_like(src))
                + F.binary_cross_entropy_with_logits(tgt, torch.ones_like(tgt)))

    def loss_proto_da(self, outputs):
        protos = outputs["da_protos"]
        assert protos.shape[0] % 2 == 0
        cm_s, cm_t = outputs["class_map_source"], outputs["class_map_target"]
        C = cm_s.shape[0]
        domain = torch.empty_like(protos)
        domain[:C] = 0
        domain[C:] = 1
        loss = F.binary_cross_entropy_with_logits(protos, domain, reduction="none")
        return (loss * torch.cat([cm_s, cm_t], dim=0).unsqueeze(1)).mean()

    def loss_contrast_da(self, outputs):
        q_s, q_t = outputs["output_source"], outputs["outputs_target"]
        m_s, m_t = outputs["query_mask_source"], outputs["query_mask_target"]
        g = outputs["global_proto"]
        assert not g.requires_grad and not m_s.requires_grad and not m_t.requires_grad
        assert q_s.requires_grad and q_t.requires_grad
        C = q_s.shape[0]
        if FUSED_CONTRAST and q_s.is_cuda and q_s.dtype == torch.float32 and q_s.shape == q_t.shape == g.shape \
                and q_s.shape[1] == 256 and C <= 16 and not torch.is_autocast_enabled():
            return _ContrastLoss.apply(q_s.contiguous(), q_t.contiguous(), g.contiguous(), m_s.contiguous().float(),
                                       m_t.contiguous().float())
        g = F.normalize(g, dim=1).permute(1, 0).contiguous()
        logits_s = F.normalize(q_s, dim=1).mm(g)
        logits_t = F.normalize(q_t, dim=1).mm(g)
        eye = torch.eye(C, device=logits_s.device)
        ce = nn.CrossEntropyLoss()
        return ce(logits_s, eye * m_s) + ce(logits_t, eye * m_t)

    # ---- orchestration ----------------------------------------------------------------------
    def _dn_indices(self, targets, single_pad, groups, device):
        pos, neg = [], []
        for t in targets:
            n = len(t["labels"])
            if n > 0:
                tgt_idx = torch.arange(n, device=device).unsqueeze(0).repeat(groups, 1)
                out_idx = (torch.arange(groups, device=device) * single_pad).unsqueeze(1) + tgt_idx
                tgt_idx, out_idx = tgt_idx.flatten(), out_idx.flatten()
            else:
                out_idx = tgt_idx = torch.tensor([], dtype=torch.long, device=device)
            pos.append((out_idx, tgt_idx))
            neg.append((out_idx + single_pad // 2, tgt_idx))
        return pos, neg

    def _zero_dn(self, device, suffix=""):
        z = lambda: torch.as_tensor(0.0, device=device)
        return {k + suffix: z() for k in ("loss_bbox_dn", "loss_giou_dn", "loss_ce_dn", "loss_xy_dn",
                                          "loss_hw_dn", "cardinality_error_dn")}

    def _dn_columns(self, targets, single_pad, groups, G, device):
        """(set, image, query, target) index columns of the de-noising family: they depend on the ground-truth
        counts and the de-noising layout only, so they are built once per (counts, layout) -- the generic
        path issued ~60 small launches per step for them (arange / repeat / full / add / cat)."""
        counts = tuple(len(t["labels"]) for t in targets)

        def make():
            pos, _ = self._dn_indices(targets, single_pad, groups, device)
            gi, bi, qi, ti, off = [], [], [], [], 0
            for g in range(G):
                off = 0
                for b, (src, tgt) in enumerate(pos):
                    n = src.numel()
                    if n:
                        gi.append(torch.full((n,), g, dtype=torch.int64, device=device))
                        bi.append(torch.full((n,), b, dtype=torch.int64, device=device))
                        qi.append(src)
                        ti.append(tgt + off)
                    off += counts[b]
            if not gi:
                z = torch.zeros(0, dtype=torch.int64, device=device)
                return z, z, z, z
            return torch.cat(gi), torch.cat(bi), torch.cat(qi), torch.cat(ti)
        return _cached(("dn_cols", G, counts, single_pad, groups, str(device)), make)

    def _family_losses(self, outs, targets, indices_per_out, num_boxes, log_first, columns=None):
        """All requested losses for a FAMILY of prediction sets with identical shapes (e.g. the
        final + auxiliary + two-stage outputs, or the de-noising outputs of every layer) in one
        batched evaluation.  Same arithmetic per element as get_loss() on each set; the focal
        term runs in the fused HIP kernel (datr_amd.focal), the matched-pair GIoU is evaluated
        pair-wise instead of as the diagonal of an n x n matrix (dino.py:563-565).
        Returns one dict per set, keys in the reference's creation order."""
        G = len(outs)
        logits = torch.stack([o["pred_logits"] for o in outs])          # [G, B, Q, C]
        boxes = torch.stack([o["pred_boxes"] for o in outs])            # [G, B, Q, 4]
        _, B, Q, C = logits.shape
        device = logits.device
        counts = [len(t["labels"]) for t in targets]
        offsets = [0]
        for c in counts:
            offsets.append(offsets[-1] + c)
        labels_cat = torch.cat([t["labels"] for t in targets])
        boxes_cat = torch.cat([t["boxes"] for t in targets])
        packed = getattr(indices_per_out, "packed", None)
        if columns is not None:
            g_idx, b_idx, q_idx, t_idx = columns
        elif packed is not None:
            # indices straight from the device solver, already in (set, image, query) order: the
            # set / image / box-offset columns are static per (G, counts) and cached
            g_idx, b_idx, off_idx = _static_index_columns(G, tuple(counts), device)
            q_idx = packed[0].reshape(-1)
            t_idx = packed[1].reshape(-1) + off_idx
        else:
            gi, bi, qi, ti = [], [], [], []
            for g, indices in enumerate(indices_per_out):
                for b, (src, tgt) in enumerate(indices):
                    n = src.numel()
                    if n == 0:
                        continue
                    gi.append(torch.full((n,), g, dtype=torch.int64, device=src.device))
                    bi.append(torch.full((n,), b, dtype=torch.int64, device=src.device))
                    qi.append(src)
                    ti.append(tgt + offsets[b])
            if gi:
                packed = torch.stack([torch.cat(gi), torch.cat(bi), torch.cat(qi), torch.cat(ti)])
                packed = packed.to(device, non_blocking=True)
                g_idx, b_idx, q_idx, t_idx = packed[0], packed[1], packed[2], packed[3]
            else:
                g_idx = b_idx = q_idx = t_idx = torch.zeros(0, dtype=torch.int64, device=device)
        n_first = offsets[-1] if len(indices_per_out) else 0     # pairs of set 0 come first

        res = [dict() for _ in range(G)]
        zeros_g = lambda: torch.zeros(G, dtype=logits.dtype, device=device)
        for loss in self.losses:
            if loss == "labels":
                matched_cls = labels_cat[t_idx]
                target_classes = torch.full((G, B, Q), self.num_classes, dtype=torch.int64, device=device)
                target_classes[g_idx, b_idx, q_idx] = matched_cls
                sums = focal_loss_sums(logits.reshape(G, B * Q, C), target_classes.view(G, B * Q),
                                       self.focal_alpha, 2.0)
                loss_ce = sums / Q / num_boxes * Q
                # unbind, not loss_ce[g]: ONE backward node (a stack of the G scalar gradients)
                # instead of G select-backwards (zeros + scatter + accumulate each)
                for g, v in enumerate(loss_ce.unbind(0)):
                    res[g]["loss_ce"] = v
                if log_first:
                    res[0]["class_error"] = 100 - accuracy(
                        logits[0][b_idx[:n_first], q_idx[:n_first]], matched_cls[:n_first])[0]
            elif loss == "boxes":
                src = boxes[g_idx, b_idx, q_idx]
                tgt = boxes_cat[t_idx]
                if FUSED_BOX_LOSS and src.is_cuda and src.dtype == torch.float32 \
                        and 0 < src.shape[0] <= MAX_BOX_LOSS_PAIRS and G <= 64:
                    # one launch each way instead of ~125 (csrc/box_loss.hip)
                    sums = box_loss_sums(src, tgt, g_idx, G) / num_boxes
                    lb, lg, lx, lh = sums.unbind(0)
                    for g, (vb, vg, vx, vh) in enumerate(zip(lb.unbind(0), lg.unbind(0),
                                                             lx.detach().unbind(0), lh.detach().unbind(0))):
                        res[g]["loss_bbox"], res[g]["loss_giou"] = vb, vg
                        res[g]["loss_xy"], res[g]["loss_hw"] = vx, vh
                    continue
                l1 = F.l1_loss(src, tgt, reduction="none")
                giou = box_ops.generalized_box_iou_pairs(box_ops.box_cxcywh_to_xyxy(src),
                                                         box_ops.box_cxcywh_to_xyxy(tgt))
                per_g = lambda v: zeros_g().index_add_(0, g_idx, v)
                loss_bbox = per_g(l1.sum(-1)) / num_boxes
                loss_giou = per_g(1 - giou) / num_boxes
                with torch.no_grad():
                    loss_xy = per_g(l1[..., :2].sum(-1)) / num_boxes
                    loss_hw = per_g(l1[..., 2:].sum(-1)) / num_boxes
                for g, (lb, lg, lx, lh) in enumerate(zip(loss_bbox.unbind(0), loss_giou.unbind(0),
                                                         loss_xy.unbind(0), loss_hw.unbind(0))):
                    res[g]["loss_bbox"], res[g]["loss_giou"] = lb, lg
                    res[g]["loss_xy"], res[g]["loss_hw"] = lx, lh
            elif loss == "cardinality":
                with torch.no_grad():
                    lengths = _device_lengths(tuple(counts), device)
                    card_pred = (logits.argmax(-1) != C - 1).sum(-1).float()         # [G, B]
                    card_err = (card_pred - lengths[None]).abs().mean(-1)
                for g, v in enumerate(card_err.unbind(0)):
                    res[g]["cardinality_error"] = v
            else:
                raise AssertionError(f"do you really want to compute {loss} loss?")
        return res

    _nb_prefetch = None

    def prefetch_num_boxes(self, targets, device):
        """Optional, data parallel only: start the `num_boxes` all-reduce (dino.py:766-767) BEFORE
        the forward pass -- it depends on the targets alone -- so that the criterion does not stall
        the compute stream on a collective's latency in the middle of the step.  Every rank must
        call it (or none); the next forward() with the same box count consumes it."""
        if not is_dist_avail_and_initialized():
            return
        num_boxes = float(sum(len(t["labels"]) for t in targets))
        nb = torch.full((1,), num_boxes, dtype=torch.float, device=device)
        self._nb_prefetch = (num_boxes, nb, dist.all_reduce(nb, async_op=True))

    def forward(self, outputs, targets, return_indices=False, target_domain_flag=False):
        sfx = "_target" if target_domain_flag else ""
        if target_domain_flag:
            outputs.update({"pred_boxes": outputs.pop("pred_boxes_target")})
            outputs.update({"pred_logits": outputs.pop("pred_logits_target")})
            device = outputs["pred_logits"].device
        else:
            device = next(iter(outputs.values())).device

        # every prediction set that gets matched: final, auxiliary layers, two-stage, encoder
        final = {"pred_logits": outputs["pred_logits"], "pred_boxes": outputs["pred_boxes"]}
        aux = list(outputs.get("aux_outputs" + sfx, []))
        interm = outputs.get("interm_outputs" + sfx)
        enc = list(outputs.get("enc_outputs" + sfx, []))
        matched = [final] + aux + ([interm] if interm is not None else []) + enc

        if len(targets) > 0:
            same = all(o["pred_logits"].shape == final["pred_logits"].shape for o in matched)
            if same:
                all_indices = self.matcher.forward_many(matched, targets)
            else:
                all_indices = [self.matcher(o, targets) for o in matched]
            indices = all_indices[0]
            num_boxes = float(sum(len(t["labels"]) for t in targets))
        else:           # no pseudo labels on this rank: keep the collective below in lock-step
            indices, num_boxes = None, 1.0

        if is_dist_avail_and_initialized():
            pre, self._nb_prefetch = self._nb_prefetch, None
            if pre is not None and indices is not None and pre[0] == num_boxes and pre[1].device == device:
                pre[2].wait()               # issued before the forward pass: long complete
                nb = pre[1]
            else:
                # torch.full = a fill kernel with a scalar argument; as_tensor([x], device=...) would
                # be a pageable host->device copy, i.e. a host synchronisation in every step
                nb = torch.full((1,), num_boxes, dtype=torch.float, device=device)
                dist.all_reduce(nb)
            if indices is None:
                nb = nb - 1
            num_boxes = torch.clamp(nb / get_world_size(), min=1)[0]   # stays on device: no sync
        else:
            if indices is None:
                num_boxes -= 1
            num_boxes = max(num_boxes / get_world_size(), 1.0)
        if indices is None:
            return {}

        if all(o["pred_logits"].shape == final["pred_logits"].shape for o in matched):
            fam = self._family_losses(matched, targets, all_indices, num_boxes,
                                      log_first=not target_domain_flag)
        else:
            fam = [self._family_losses([o], targets, [idx], num_boxes,
                                       log_first=(i == 0 and not target_domain_flag))[0]
                   for i, (o, idx) in enumerate(zip(matched, all_indices))]
        final_l, aux_l = fam[0], fam[1:1 + len(aux)]
        interm_l = fam[1 + len(aux)] if interm is not None else None
        enc_l = fam[len(fam) - len(enc):] if enc else []

        losses = {}
        use_dn = False
        if not target_domain_flag:
            dn_meta = outputs["dn_meta"]
            use_dn = bool(self.training and dn_meta and "output_known_lbs_bboxes" in dn_meta)
            if use_dn:
                known = dn_meta["output_known_lbs_bboxes"]
                groups, pad_size = dn_meta["num_dn_group"], dn_meta["pad_size"]
                assert pad_size % groups == 0
                dn_sets = [{"pred_logits": known["pred_logits"], "pred_boxes": known["pred_boxes"]}]
                dn_sets += list(known.get("aux_outputs", []))
                cols = self._dn_columns(targets, pad_size // groups, groups, len(dn_sets), device)
                dn_l = self._family_losses(dn_sets, targets, [None] * len(dn_sets),
                                           num_boxes * groups, log_first=False, columns=cols)
                losses.update({k + "_dn": v for k, v in dn_l[0].items()})
            else:
                losses.update(self._zero_dn(device))
            losses.update(final_l)

        for idx, l_dict in enumerate(aux_l):
            l_dict = {k: v for k, v in l_dict.items() if k != "class_error"}
            losses.update({k + f"_{idx}": v for k, v in l_dict.items()})
            if not target_domain_flag:
                if use_dn:
                    losses.update({k + f"_dn_{idx}": v for k, v in dn_l[1 + idx].items()})
                else:
                    losses.update(self._zero_dn(device, f"_{idx}"))
        if interm_l is not None:
            losses.update({k + "_interm": v for k, v in interm_l.items() if k != "class_error"})
        for i, l_dict in enumerate(enc_l):
            losses.update({k + f"_enc_{i}": v for k, v in l_dict.items() if k != "class_error"})

        poison = getattr(self.matcher, "poison", None)
        if poison is not None and "loss_ce" in losses:
            # NaN if the device matcher rejected a cost matrix (NaN costs / degenerate boxes: the
            # reference raises there), else + 0.0 -- fails loudly without a host sync
            losses["loss_ce"] = losses["loss_ce"] + poison

        if "da_output" in outputs:
            da = outputs["da_output"]
            losses["loss_backbone_DA"] = self.loss_da(da["backbone_DA"])
            losses["loss_proto_DA"] = self.loss_proto_da(da["proto_DA"])
            losses["loss_global_proto_DA"] = self.loss_contrast_da(da["global_proto_DA"])

        if return_indices:
            # reference order: auxiliary layers, two-stage, (encoder), then the final layer
            return losses, all_indices[1:] + [all_indices[0]]
        return losses
