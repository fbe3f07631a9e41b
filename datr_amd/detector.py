"""DINO detector with DATR's domain-adaptation branch, the box post-processor and the builder.

Mirror of /root/reference/models/dino/dino.py: `DINO` (:43-484), `PostProcess` (:944-996),
`build_dino` (:999-1143) registered under 'dino' in `MODULE_BUILD_FUNCS`
(/root/reference/models/registry.py:12-57).  Attribute names equal the reference's, so
`state_dict()` keys and shapes match SURVEY.md A.2 and the published checkpoints load.

Behaviour kept from the reference:
  * a training batch of 2B images is B source images followed by B target images; the main
    transformer pass (with de-noising queries) runs on the source half, a second pass without
    DN queries runs on the target half, the image discriminator sees all 2B images;
  * `D_img`, `Proto_D`, `global_proto`, `Amount` always exist; the two prototype tensors are
    NOT part of the state_dict (non-persistent buffers here, plain attributes there) and are
    updated twice per training forward (source, then target);
  * `hs[0] += label_enc.weight[0,0] * 0.0` keeps label_enc in the graph without targets.
One addition that is NOT in the reference: `model.domain_adaptation = False` turns the
training forward into the plain (source-only) DINO step that BASELINE configs 1-2 name.
"""
from __future__ import annotations

import copy
import math
from typing import List

import torch
import torch.nn.functional as F
from torch import nn

from . import boxes as box_ops
from .backbone import build_backbone
from .criterion import SetCriterion
from .denoising import dn_post_process, prepare_for_cdn
from .fused import GroupNormNHWC, refine_boxes
from .domain import (FCDiscriminator_img, decompose_features, get_prototype_class_wise,
                     grad_reverse)
from .matcher import build_matcher
from .nested import NestedTensor, inverse_sigmoid, nested_tensor_from_tensor_list
from .registry import MODULE_BUILD_FUNCS
from .transformer import MLP, build_deformable_transformer


OVERLAP_D_IMG = __import__("os").environ.get("DATR_OVERLAP_D_IMG", "1") != "0"     # A/B switch
_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        from .dist import EXTRA_STREAMS
        EXTRA_STREAMS.append(_SIDE_STREAMS[key])
    return _SIDE_STREAMS[key]


class DINO(nn.Module):
    def __init__(self, backbone, transformer, num_classes, num_queries, aux_loss=False,
                 iter_update=False, query_dim=2, random_refpoints_xy=False, fix_refpoints_hw=-1,
                 num_feature_levels=1, nheads=8, two_stage_type="no", two_stage_add_query_num=0,
                 dec_pred_class_embed_share=True, dec_pred_bbox_embed_share=True,
                 two_stage_class_embed_share=True, two_stage_bbox_embed_share=True,
                 decoder_sa_type="sa", num_patterns=0, dn_number=100, dn_box_noise_scale=0.4,
                 dn_label_noise_ratio=0.5, dn_labelbook_size=100):
        super().__init__()
        assert query_dim == 4 and iter_update, "query_dim == 4 with iterative update only"
        assert two_stage_type in ("no", "standard"), f"unknown param {two_stage_type} of two_stage_type"
        assert decoder_sa_type == "sa", "only decoder_sa_type='sa' is on the hot path"
        if two_stage_add_query_num or random_refpoints_xy or int(fix_refpoints_hw) != -1:
            raise NotImplementedError("refpoint-embedding options are off in every DA config")
        self.num_queries = num_queries
        self.transformer = transformer
        self.num_classes = num_classes
        self.hidden_dim = hidden_dim = transformer.d_model
        self.num_feature_levels = num_feature_levels
        self.nheads = nheads
        self.label_enc = nn.Embedding(dn_labelbook_size + 1, hidden_dim)
        self.query_dim = query_dim
        self.random_refpoints_xy = random_refpoints_xy
        self.fix_refpoints_hw = fix_refpoints_hw
        self.num_patterns = num_patterns
        self.dn_number = dn_number
        self.dn_box_noise_scale = dn_box_noise_scale
        self.dn_label_noise_ratio = dn_label_noise_ratio
        self.dn_labelbook_size = dn_labelbook_size

        # domain adaptation: image-level discriminator, prototype discriminator, running prototypes
        self.D_img = FCDiscriminator_img(256)
        self.register_buffer("global_proto", torch.zeros(num_classes, 256), persistent=False)
        self.register_buffer("Amount", torch.zeros(num_classes), persistent=False)
        self.Proto_D = MLP(hidden_dim, hidden_dim, 1, 3)
        self.domain_adaptation = True
        # one encoder call for source+target (same values up to fp32 GEMM blocking); False
        # reproduces the reference's two separate transformer calls exactly
        self.merge_encoder_passes = True
        self.merge_decoder_passes = True      # with merge_encoder_passes: one decoder pass too
        self.dn_noise_override = None        # tests inject the reference's RNG draws here

        if num_feature_levels > 1:
            projs = []
            for in_channels in backbone.num_channels:
                projs.append(nn.Sequential(nn.Conv2d(in_channels, hidden_dim, kernel_size=1),
                                           GroupNormNHWC(32, hidden_dim)))
            for _ in range(num_feature_levels - len(backbone.num_channels)):
                projs.append(nn.Sequential(
                    nn.Conv2d(in_channels, hidden_dim, kernel_size=3, stride=2, padding=1),
                    GroupNormNHWC(32, hidden_dim)))
                in_channels = hidden_dim
            self.input_proj = nn.ModuleList(projs)
        else:
            assert two_stage_type == "no", "two_stage_type should be no if num_feature_levels=1 !!!"
            self.input_proj = nn.ModuleList([nn.Sequential(
                nn.Conv2d(backbone.num_channels[-1], hidden_dim, kernel_size=1),
                GroupNormNHWC(32, hidden_dim))])

        self.backbone = backbone
        self.aux_loss = aux_loss
        self.box_pred_damping = None
        self.iter_update = iter_update
        self.dec_pred_class_embed_share = dec_pred_class_embed_share
        self.dec_pred_bbox_embed_share = dec_pred_bbox_embed_share

        _class_embed = nn.Linear(hidden_dim, num_classes)
        _bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        prior_prob = 0.01
        _class_embed.bias.data = torch.ones(num_classes) * (-math.log((1 - prior_prob) / prior_prob))
        nn.init.constant_(_bbox_embed.layers[-1].weight.data, 0)
        nn.init.constant_(_bbox_embed.layers[-1].bias.data, 0)
        n_dec = transformer.num_decoder_layers
        self.bbox_embed = nn.ModuleList(
            [_bbox_embed if dec_pred_bbox_embed_share else copy.deepcopy(_bbox_embed)
             for _ in range(n_dec)])
        self.class_embed = nn.ModuleList(
            [_class_embed if dec_pred_class_embed_share else copy.deepcopy(_class_embed)
             for _ in range(n_dec)])
        self.transformer.decoder.bbox_embed = self.bbox_embed
        self.transformer.decoder.class_embed = self.class_embed

        self.two_stage_type = two_stage_type
        self.two_stage_add_query_num = two_stage_add_query_num
        if two_stage_type != "no":
            if two_stage_bbox_embed_share:
                assert dec_pred_class_embed_share and dec_pred_bbox_embed_share
                self.transformer.enc_out_bbox_embed = _bbox_embed
            else:
                self.transformer.enc_out_bbox_embed = copy.deepcopy(_bbox_embed)
            if two_stage_class_embed_share:
                assert dec_pred_class_embed_share and dec_pred_bbox_embed_share
                self.transformer.enc_out_class_embed = _class_embed
            else:
                self.transformer.enc_out_class_embed = copy.deepcopy(_class_embed)
            self.refpoint_embed = None
        self.decoder_sa_type = decoder_sa_type
        self.label_embedding = None
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)

    # ------------------------------------------------------------------------------------------
    def _heads(self, hs, reference):
        """Per decoder layer: boxes = sigmoid(delta + logit(reference)), class logits.
        When the per-layer heads alias one module (dec_pred_*_embed_share, dino.py:155-166) all
        layers go through it as ONE batched call -- same weights, same arithmetic per row."""
        n = len(hs)
        hs_all = torch.stack(list(hs))                               # [n_dec, B, Q, d]
        if all(m is self.bbox_embed[0] for m in self.bbox_embed[:n]):
            delta = self.bbox_embed[0](hs_all)
        else:
            delta = torch.stack([h(x) for h, x in zip(self.bbox_embed, hs)])
        coords = refine_boxes(delta, torch.stack(list(reference[:-1])))
        if all(m is self.class_embed[0] for m in self.class_embed[:n]):
            classes = self.class_embed[0](hs_all)
        else:
            classes = torch.stack([h(x) for h, x in zip(self.class_embed, hs)])
        return classes, coords

    def _input_proj(self, lvl, src):
        """input_proj[lvl](src) (dino.py:111-126: 1x1 conv + GroupNorm); a channels_last device
        input takes the convolution as a GEMM with the bias in its epilogue (datr_amd.pointwise)."""
        from . import pointwise
        conv, norm = self.input_proj[lvl][0], self.input_proj[lvl][1]
        if conv.kernel_size == (1, 1) and conv.stride == (1, 1) and len(self.input_proj[lvl]) == 2:
            y = pointwise.conv1x1(src, conv.weight, conv.bias)
            if y is not None:
                return norm(y)
        if conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1) and len(self.input_proj[lvl]) == 2:
            # the extra pyramid level (dino.py:120-124): own stride-2 kernels, bias in the epilogue
            from . import strided
            y = strided.conv3x3_s2(src, conv.weight, None, conv.bias)
            if y is not None:
                return norm(y)
        return self.input_proj[lvl](src)

    def side_stream_parameters(self):
        """Parameters whose gradients are produced on the side stream (the image-level
        discriminator when OVERLAP_D_IMG): datr_amd.dist.GradAllReducer buckets them apart."""
        d = getattr(self, "D_img", None)
        return list(d.parameters()) if (d is not None and OVERLAP_D_IMG) else []

    @torch.jit.unused
    def _set_aux_loss(self, outputs_class, outputs_coord):
        return [{"pred_logits": a, "pred_boxes": b}
                for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]

    def forward(self, samples: NestedTensor, targets: List = None, self_training_flag=False):
        if isinstance(samples, (list, torch.Tensor)):
            samples = nested_tensor_from_tensor_list(samples)
        self.transformer.no_padding = getattr(samples, "padded", None) is False
        features, poss = self.backbone(samples)

        srcs, masks = [], []
        for lvl, feat in enumerate(features):
            src, mask = feat.decompose()
            assert mask is not None
            srcs.append(self._input_proj(lvl, src))
            masks.append(mask)
        for lvl in range(len(srcs), self.num_feature_levels):
            src = self._input_proj(lvl, features[-1].tensors if lvl == len(features) else srcs[-1])
            mask = F.interpolate(samples.mask[None].float(), size=src.shape[-2:]).to(torch.bool)[0]
            poss.append(self.backbone[1](NestedTensor(src, mask, getattr(samples, "padded", None))).to(src.dtype))
            srcs.append(src)
            masks.append(mask)

        da = self.training and self.domain_adaptation
        if da:
            (srcs, masks, poss, srcs_all, masks_all, poss_all,
             srcs_target, masks_target, poss_target) = decompose_features(srcs, masks, poss)

        if self.dn_number > 0 or targets is not None:
            input_query_label, input_query_bbox, attn_mask, dn_meta = prepare_for_cdn(
                dn_args=(targets, self.dn_number, self.dn_label_noise_ratio, self.dn_box_noise_scale),
                training=self.training, num_queries=self.num_queries,
                num_classes=self.num_classes, hidden_dim=self.hidden_dim,
                label_enc=self.label_enc, noise=self.dn_noise_override)
        else:
            assert targets is None
            input_query_bbox = input_query_label = attn_mask = dn_meta = None

        merged = da and self.merge_encoder_passes
        # The image-level discriminator only needs the projected backbone features.  On the device
        # it runs on a SIDE STREAM that starts when the main stream reaches the decoder, so its
        # dense convolutions fill the gaps between the decoder's thousands of tiny kernels
        # (forward here; autograd replays the same stream assignment in backward, where D_img's
        # nodes -- created after the decoder's -- are scheduled right before the decoder's).
        overlap_d_img = da and OVERLAP_D_IMG and srcs_all[0].is_cuda
        if overlap_d_img:
            decoder_start = torch.cuda.Event()
        if merged:
            # the encoder is per-sample: run it ONCE on all 2B images (the reference runs it
            # separately for the source and the target half, dino.py:291,380 -- same values,
            # half the launches, larger GEMMs), then decode each half on its own
            enc_all = self.transformer.encode(srcs_all, masks_all, poss_all)
            half = srcs_all[0].shape[0] // 2
            enc_src = self.transformer.slice_encoded(enc_all, slice(0, half))
            enc_tgt = self.transformer.slice_encoded(enc_all, slice(half, None))
            if overlap_d_img:
                decoder_start.record()
            if self.merge_decoder_passes:
                # ONE decoder pass for all 2B images as well.  The target images get the source's
                # de-noising slots as placeholders: the attention mask already forbids matching
                # queries to look at DN slots (dn_components.py:117-124), every other decoder
                # operation is per query, so the target's 900 matching queries come out exactly
                # as in a separate mask-free pass (dino.py:380-381); the placeholder outputs are
                # dropped.  Half the launches of the launch-bound decoder, larger GEMMs.
                if input_query_bbox is not None:
                    q_bbox = torch.cat([input_query_bbox, input_query_bbox.detach()], 0)
                    q_label = torch.cat([input_query_label, input_query_label.detach()], 0)
                    dn_pad = input_query_bbox.shape[1]
                else:
                    q_bbox = q_label = None
                    dn_pad = 0
                hs_a, ref_a, hs_enc_a, ref_enc_a, ibp_a = self.transformer.decode(
                    enc_all, q_bbox, q_label, attn_mask)
                hs, reference = [h[:half] for h in hs_a], [r[:half] for r in ref_a]
                hs_enc = None if hs_enc_a is None else hs_enc_a[:, :half]
                ref_enc = None if ref_enc_a is None else ref_enc_a[:, :half]
                init_box_proposal = ibp_a[:half]
                target_pass = ([h[half:, dn_pad:] for h in hs_a], [r[half:, dn_pad:] for r in ref_a],
                               None if hs_enc_a is None else hs_enc_a[:, half:],
                               None if ref_enc_a is None else ref_enc_a[:, half:], ibp_a[half:])
            else:
                target_pass = None
                hs, reference, hs_enc, ref_enc, init_box_proposal = self.transformer.decode(
                    enc_src, input_query_bbox, input_query_label, attn_mask)
        else:
            if overlap_d_img:
                decoder_start.record()
            hs, reference, hs_enc, ref_enc, init_box_proposal = self.transformer(
                srcs, masks, input_query_bbox, poss, input_query_label, attn_mask)
        hs[0] = hs[0] + self.label_enc.weight[0, 0] * 0.0

        outputs_class, outputs_coord_list = self._heads(hs, reference)
        if self.dn_number > 0 and dn_meta is not None:
            outputs_class, outputs_coord_list = dn_post_process(
                outputs_class, outputs_coord_list, dn_meta, self.aux_loss, self._set_aux_loss)
        out = {"pred_logits": outputs_class[-1], "pred_boxes": outputs_coord_list[-1]}
        if self.aux_loss:
            out["aux_outputs"] = self._set_aux_loss(outputs_class, outputs_coord_list)

        if hs_enc is not None:
            interm_class = self.transformer.enc_out_class_embed(hs_enc[-1])
            out["interm_outputs"] = {"pred_logits": interm_class, "pred_boxes": ref_enc[-1]}
            out["interm_outputs_for_matching_pre"] = {"pred_logits": interm_class,
                                                      "pred_boxes": init_box_proposal}
        out["dn_meta"] = dn_meta

        if da:
            da_output = {}
            # 1. image-level alignment: per-pixel domain logits on every level, all 2B images
            if overlap_d_img:
                main = torch.cuda.current_stream()
                side = _side_stream(srcs_all[0].device)
                side.wait_event(decoder_start)
                with torch.cuda.stream(side):
                    for src in srcs_all:
                        src.record_stream(side)
                    d_out = self.D_img.reversed_pyramid(srcs_all)
                    backbone_da = torch.cat([o.flatten(2).transpose(1, 2) for o in d_out], dim=1)
                    backbone_da.record_stream(main)
                    done = torch.cuda.Event()
                    done.record(side)
                # consumers (SetCriterion.loss_da) make their stream wait for this event
                backbone_da._ready_event = done
                da_output["backbone_DA"] = backbone_da
            else:
                d_out = self.D_img.reversed_pyramid(srcs_all)
                da_output["backbone_DA"] = torch.cat([o.flatten(2).transpose(1, 2) for o in d_out], dim=1)

            # 2. class-wise query prototypes, source domain
            pad = dn_meta["pad_size"] if dn_meta is not None else 0
            proto_s, present_s, g_proto, g_amount, _ = get_prototype_class_wise(
                hs[-1][:, pad:, :], out["pred_logits"], self.num_classes,
                global_proto=self.global_proto.detach(), global_amount=self.Amount)
            self.global_proto, self.Amount = g_proto, g_amount

            # second transformer pass: target half, no DN queries, no attention mask
            if merged and target_pass is not None:
                hs_t, reference_t, hs_enc_t, ref_enc_t, init_box_proposal_t = target_pass
            elif merged:
                hs_t, reference_t, hs_enc_t, ref_enc_t, init_box_proposal_t = \
                    self.transformer.decode(enc_tgt, None, None, None)
            else:
                hs_t, reference_t, hs_enc_t, ref_enc_t, init_box_proposal_t = self.transformer(
                    srcs_target, masks_target, None, poss_target, None, None)
            out_t = hs_t[-1]
            proto_t, present_t, g_proto, g_amount, _ = get_prototype_class_wise(
                out_t, self.class_embed[-1](out_t), self.num_classes,
                global_proto=self.global_proto.detach(), global_amount=self.Amount)
            self.global_proto, self.Amount = g_proto, g_amount

            protos = torch.cat([proto_s, proto_t], dim=0)
            da_output["proto_DA"] = {"da_protos": self.Proto_D(grad_reverse(protos)),
                                     "class_map_source": present_s,
                                     "class_map_target": present_t}
            da_output["global_proto_DA"] = {"output_source": proto_s, "outputs_target": proto_t,
                                            "query_mask_source": present_s,
                                            "query_mask_target": present_t,
                                            "global_proto": self.global_proto}
            out["da_output"] = da_output

            if self_training_flag:       # decode the target pass too (pseudo-label supervision)
                hs_t[0] = hs_t[0] + self.label_enc.weight[0, 0] * 0.0
                cls_t, coord_t = self._heads(hs_t, reference_t)
                out["pred_logits_target"] = cls_t[-1]
                out["pred_boxes_target"] = coord_t[-1]
                if self.aux_loss:
                    out["aux_outputs_target"] = self._set_aux_loss(cls_t, coord_t)
                if hs_enc_t is not None:
                    interm_class_t = self.transformer.enc_out_class_embed(hs_enc_t[-1])
                    out["interm_outputs_target"] = {"pred_logits": interm_class_t,
                                                    "pred_boxes": ref_enc_t[-1]}
                    out["interm_outputs_for_matching_pre_target"] = {
                        "pred_logits": interm_class_t, "pred_boxes": init_box_proposal_t}
        return out


def _nms(boxes, scores, iou_threshold):
    """Greedy NMS (the reference calls torchvision.ops.nms, dino.py:22,990; off by default)."""
    order = scores.argsort(descending=True)
    keep = []
    while order.numel() > 0:
        i = order[0]
        keep.append(i)
        if order.numel() == 1:
            break
        iou, _ = box_ops.box_iou(boxes[i][None], boxes[order[1:]])
        order = order[1:][iou[0] <= iou_threshold]
    return torch.stack(keep) if keep else torch.empty(0, dtype=torch.long, device=boxes.device)


class PostProcess(nn.Module):
    """Top-k over the flattened (query, class) scores -> {scores, labels, boxes} per image."""

    def __init__(self, num_select=100, nms_iou_threshold=-1) -> None:
        super().__init__()
        self.num_select = num_select
        self.nms_iou_threshold = nms_iou_threshold

    @torch.no_grad()
    def forward(self, outputs, target_sizes, not_to_xyxy=False, test=False):
        logits, bbox = outputs["pred_logits"], outputs["pred_boxes"]
        assert len(logits) == len(target_sizes) and target_sizes.shape[1] == 2
        C = logits.shape[2]
        from .fused import topk_rows
        scores, topk = topk_rows(logits.sigmoid().view(logits.shape[0], -1), self.num_select)
        query_idx = torch.div(topk, C, rounding_mode="floor")
        labels = topk % C
        boxes = bbox if not_to_xyxy else box_ops.box_cxcywh_to_xyxy(bbox)
        if test:
            assert not not_to_xyxy
            boxes[:, :, 2:] = boxes[:, :, 2:] - boxes[:, :, :2]
        boxes = torch.gather(boxes, 1, query_idx.unsqueeze(-1).repeat(1, 1, 4))
        img_h, img_w = target_sizes.unbind(1)
        boxes = boxes * torch.stack([img_w, img_h, img_w, img_h], dim=1)[:, None, :]
        if self.nms_iou_threshold > 0:
            keep = [_nms(b, s, self.nms_iou_threshold) for b, s in zip(boxes, scores)]
            return [{"scores": s[i], "labels": l[i], "boxes": b[i]}
                    for s, l, b, i in zip(scores, labels, boxes, keep)]
        return [{"scores": s, "labels": l, "boxes": b} for s, l, b in zip(scores, labels, boxes)]


def build_weight_dict(args):
    """Loss weights (dino.py:1072-1127): base ce/bbox/giou, DA terms, DN copies, one copy per
    auxiliary decoder layer, and the `_interm` copies for the two-stage outputs."""
    weight_dict = {"loss_ce": args.cls_loss_coef, "loss_bbox": args.bbox_loss_coef,
                   "loss_giou": args.giou_loss_coef}
    clean_wo_dn = copy.deepcopy(weight_dict)
    weight_dict["loss_backbone_DA"] = args.da_backbone_loss_coef
    weight_dict["loss_proto_DA"] = args.da_proto_loss_coef
    weight_dict["loss_global_proto_DA"] = args.da_global_proto_coef
    weight_dict["loss_self_training"] = args.self_training_loss_coef
    if args.use_dn:
        weight_dict["loss_ce_dn"] = args.cls_loss_coef
        weight_dict["loss_bbox_dn"] = args.bbox_loss_coef
        weight_dict["loss_giou_dn"] = args.giou_loss_coef
    clean = copy.deepcopy(weight_dict)
    if args.aux_loss:
        for i in range(args.dec_layers - 1):
            weight_dict.update({k + f"_{i}": v for k, v in clean.items()})
    if args.two_stage_type != "no":
        no_box = getattr(args, "no_interm_box_loss", False)
        coeff = {"loss_ce": 1.0, "loss_bbox": 0.0 if no_box else 1.0,
                 "loss_giou": 0.0 if no_box else 1.0}
        interm = getattr(args, "interm_loss_coef", 1.0)
        weight_dict.update({k + "_interm": v * interm * coeff[k] for k, v in clean_wo_dn.items()})
    return weight_dict


@MODULE_BUILD_FUNCS.registe_with_name(module_name="dino")
def build_dino(args):
    """args: any attribute bag with the fields of SURVEY.md A.1 (see datr_amd.config)."""
    if getattr(args, "masks", False):
        raise NotImplementedError("segmentation heads are off in every DA config (masks=False)")
    num_classes = args.num_classes
    device = torch.device(args.device)
    backbone = build_backbone(args)
    transformer = build_deformable_transformer(args)
    model = DINO(
        backbone, transformer, num_classes=num_classes, num_queries=args.num_queries,
        aux_loss=True, iter_update=True, query_dim=4,
        random_refpoints_xy=args.random_refpoints_xy, fix_refpoints_hw=args.fix_refpoints_hw,
        num_feature_levels=args.num_feature_levels, nheads=args.nheads,
        dec_pred_class_embed_share=getattr(args, "dec_pred_class_embed_share", True),
        dec_pred_bbox_embed_share=getattr(args, "dec_pred_bbox_embed_share", True),
        two_stage_type=args.two_stage_type,
        two_stage_bbox_embed_share=args.two_stage_bbox_embed_share,
        two_stage_class_embed_share=args.two_stage_class_embed_share,
        decoder_sa_type=args.decoder_sa_type, num_patterns=args.num_patterns,
        dn_number=args.dn_number if args.use_dn else 0,
        dn_box_noise_scale=args.dn_box_noise_scale,
        dn_label_noise_ratio=args.dn_label_noise_ratio,
        dn_labelbook_size=getattr(args, "dn_labelbook_size", num_classes))
    matcher = build_matcher(args)
    criterion = SetCriterion(num_classes, matcher=matcher, weight_dict=build_weight_dict(args),
                             focal_alpha=args.focal_alpha,
                             losses=["labels", "boxes", "cardinality"])
    criterion.to(device)
    postprocessors = {"bbox": PostProcess(num_select=args.num_select,
                                          nms_iou_threshold=args.nms_iou_threshold)}
    return model, criterion, postprocessors
