"""Contrastive de-noising (CDN) query construction for DINO training.

Mirror of /root/reference/models/dino/dn_components.py (`prepare_for_cdn` :20-137,
`dn_post_process` :140-154).  Differences that do not change results:
  * group counts and pad sizes come from the python lengths of the target lists, so there is
    no device->host sync (the reference calls int(max(sum(ones))) on device tensors, :36-44);
  * the self-attention mask is built with one comparison of group ids instead of the
    reference's python loop (:117-124) -- same boolean matrix;
  * the four random draws (:64-66, :84-85) happen in the reference's order with the
    reference's shapes/dtypes, or are taken from `noise` when given (golden-vector tests
    capture the reference's draws, since generators differ between devices).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .nested import inverse_sigmoid


def prepare_for_cdn(dn_args, training, num_queries, num_classes, hidden_dim, label_enc,
                    noise: Optional[Dict[str, torch.Tensor]] = None):
    if not training:
        return None, None, None, None
    targets, dn_number, label_noise_ratio, box_noise_scale = dn_args
    device = label_enc.weight.device
    counts = [int(t["labels"].shape[0]) for t in targets]
    batch_size = len(targets)
    max_gt = max(counts) if counts else 0

    dn_number = dn_number * 2
    if max_gt == 0:
        dn_number = 1
    elif dn_number >= 100:
        dn_number = dn_number // (max_gt * 2)
    elif dn_number < 1:
        dn_number = 1
    if dn_number == 0:
        dn_number = 1
    groups = dn_number
    total = sum(counts)

    labels = torch.cat([t["labels"] for t in targets])
    boxes = torch.cat([t["boxes"] for t in targets])
    single_pad = max_gt
    pad_size = int(single_pad * 2 * groups)
    static = _static_parts(tuple(counts), groups, num_queries, device)

    known_labels = labels.repeat(2 * groups, 1).view(-1)
    known_bid = static["known_bid"]
    known_bboxs = boxes.repeat(2 * groups, 1)
    noised_labels = known_labels.clone()
    noised_boxes = known_bboxs.clone()

    if label_noise_ratio > 0:
        if noise is not None:       # replayed draws (parity tests): the reference's exact sequence
            p = noise["label_p"].to(device)
            chosen = torch.nonzero(p < (label_noise_ratio * 0.5)).view(-1)
            noised_labels.scatter_(0, chosen, noise["new_label"].to(device))
        else:
            # same distribution as dn_components.py:60-63 (flip each label with probability
            # ratio/2 to a uniform class) without `nonzero`, whose data-dependent size costs a
            # device->host sync; the random stream is not the reference's either way
            p = torch.rand_like(noised_labels.float())
            new_label = torch.randint_like(noised_labels, 0, num_classes)
            noised_labels = torch.where(p < (label_noise_ratio * 0.5), new_label, noised_labels)

    negative_idx = static["negative_idx"]

    if box_noise_scale > 0:
        corners = torch.zeros_like(known_bboxs)
        corners[:, :2] = known_bboxs[:, :2] - known_bboxs[:, 2:] / 2
        corners[:, 2:] = known_bboxs[:, :2] + known_bboxs[:, 2:] / 2
        half = torch.zeros_like(known_bboxs)
        half[:, :2] = known_bboxs[:, 2:] / 2
        half[:, 2:] = known_bboxs[:, 2:] / 2
        if noise is not None:
            rand_sign = noise["rand_sign"].to(device)
            rand_part = noise["rand_part"].to(device).clone()
        else:
            rand_sign = torch.randint_like(known_bboxs, low=0, high=2, dtype=torch.float32) * 2.0 - 1.0
            rand_part = torch.rand_like(known_bboxs)
        rand_part[negative_idx] += 1.0       # negatives are pushed 1..2 half-sizes away
        rand_part *= rand_sign
        corners = corners + torch.mul(rand_part, half) * box_noise_scale
        corners = corners.clamp(min=0.0, max=1.0)
        noised_boxes[:, :2] = (corners[:, :2] + corners[:, 2:]) / 2
        noised_boxes[:, 2:] = corners[:, 2:] - corners[:, :2]

    label_embed = label_enc(noised_labels.long())
    bbox_embed = inverse_sigmoid(noised_boxes)

    input_query_label = torch.zeros(batch_size, pad_size, hidden_dim, device=device)
    input_query_bbox = torch.zeros(batch_size, pad_size, 4, device=device)
    if total > 0:
        input_query_label[(known_bid, static["slot"])] = label_embed
        input_query_bbox[(known_bid, static["slot"])] = bbox_embed

    dn_meta = {"pad_size": pad_size, "num_dn_group": groups}
    return input_query_label, input_query_bbox, static["attn_mask"], dn_meta


_STATIC = {}


def _static_parts(counts, groups, num_queries, device):
    """Index tensors and the attention mask of prepare_for_cdn depend only on the per-image box
    counts, the number of groups and the number of queries: built once per distinct key (about
    40 small launches per step otherwise).  The mask is shared between steps: read-only."""
    key = (counts, groups, num_queries, str(device))
    parts = _STATIC.get(key)
    if parts is not None:
        return parts
    total, single_pad = sum(counts), (max(counts) if counts else 0)
    pad_size = int(single_pad * 2 * groups)
    batch_idx = torch.cat([torch.full((c,), i, dtype=torch.int64, device=device)
                           for i, c in enumerate(counts)]) if counts else \
        torch.zeros(0, dtype=torch.int64, device=device)
    known_bid = batch_idx.repeat(2 * groups, 1).view(-1)
    positive_idx = torch.arange(total, device=device).unsqueeze(0).repeat(groups, 1)
    positive_idx = positive_idx + (torch.arange(groups, device=device) * total * 2).unsqueeze(1)
    negative_idx = positive_idx.flatten() + total
    if total > 0:
        within = torch.cat([torch.arange(c, device=device) for c in counts])
        slot = torch.cat([within + single_pad * i for i in range(2 * groups)]).long()
    else:
        slot = torch.zeros(0, dtype=torch.int64, device=device)
    tgt_size = pad_size + num_queries
    attn_mask = torch.zeros(tgt_size, tgt_size, dtype=torch.bool, device=device)
    attn_mask[pad_size:, :pad_size] = True            # matching queries never see DN queries
    if pad_size > 0:
        gid = torch.arange(pad_size, device=device) // max(2 * single_pad, 1)
        attn_mask[:pad_size, :pad_size] = gid[:, None] != gid[None, :]   # groups are mutually blind
    if len(_STATIC) > 64:
        _STATIC.clear()
    parts = _STATIC[key] = {"known_bid": known_bid, "negative_idx": negative_idx, "slot": slot,
                            "attn_mask": attn_mask}
    return parts


def dn_post_process(outputs_class, outputs_coord, dn_meta, aux_loss, _set_aux_loss):
    """Split the first pad_size (de-noising) queries off and park them in dn_meta."""
    if dn_meta and dn_meta["pad_size"] > 0:
        pad = dn_meta["pad_size"]
        known_class, known_coord = outputs_class[:, :, :pad, :], outputs_coord[:, :, :pad, :]
        outputs_class, outputs_coord = outputs_class[:, :, pad:, :], outputs_coord[:, :, pad:, :]
        out = {"pred_logits": known_class[-1], "pred_boxes": known_coord[-1]}
        if aux_loss:
            out["aux_outputs"] = _set_aux_loss(known_class, known_coord)
        dn_meta["output_known_lbs_bboxes"] = out
    return outputs_class, outputs_coord
