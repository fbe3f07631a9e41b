"""Frozen configuration of the hot path: the Cityscapes -> Foggy Cityscapes burn-in config
(/root/reference/config/DA/Cityscapes2FoggyCityscapes/DINO_4scale_C2F.py) with the launch
script's overrides (/root/reference/scripts/DINO_train.sh:4-6: embed_init_tgt=TRUE,
dn_box_noise_scale=1.0, use_ema=False) and the CLI defaults build_dino reads
(/root/reference/main.py:28-76).  The reference's mmcv-style config loader is out of scope
(SURVEY.md section 2.1); `build_dino` takes any attribute bag, so a plain Namespace does."""
from __future__ import annotations

import argparse

C2F = dict(
    num_classes=9, lr=1e-4, param_dict_type="default", lr_backbone=1e-5, batch_size=2,
    weight_decay=1e-4, epochs=36, lr_drop=30, clip_max_norm=0.1, onecyclelr=False,
    modelname="dino", frozen_weights=None, backbone="resnet50", use_checkpoint=True,
    dilation=False, position_embedding="sine", pe_temperatureH=20, pe_temperatureW=20,
    return_interm_indices=[1, 2, 3], backbone_freeze_keywords=None,
    enc_layers=6, dec_layers=6, unic_layers=0, pre_norm=False, dim_feedforward=2048,
    hidden_dim=256, dropout=0.0, nheads=8, num_queries=900, query_dim=4, num_patterns=0,
    random_refpoints_xy=False, fix_refpoints_hw=-1, use_deformable_box_attn=False,
    box_attn_type="roi_align", dec_layer_number=None, num_feature_levels=4, enc_n_points=4,
    dec_n_points=4, decoder_layer_noise=False, dln_xy_noise=0.2, dln_hw_noise=0.2,
    add_channel_attention=False, add_pos_value=False, two_stage_type="standard",
    two_stage_pat_embed=0, two_stage_add_query_num=0, two_stage_bbox_embed_share=False,
    two_stage_class_embed_share=False, two_stage_learn_wh=False, two_stage_default_hw=0.05,
    two_stage_keep_all_tokens=False, num_select=300, transformer_activation="relu",
    batch_norm_type="FrozenBatchNorm2d", masks=False, aux_loss=True,
    set_cost_class=2.0, set_cost_bbox=5.0, set_cost_giou=2.0, cls_loss_coef=1.0,
    mask_loss_coef=1.0, dice_loss_coef=1.0, bbox_loss_coef=5.0, giou_loss_coef=2.0,
    enc_loss_coef=1.0, interm_loss_coef=1.0, no_interm_box_loss=False, focal_alpha=0.25,
    da_backbone_loss_coef=0.1, da_proto_loss_coef=0.1, da_global_proto_coef=0.1,
    decoder_sa_type="sa", matcher_type="HungarianMatcher", decoder_module_seq=["sa", "ca", "ffn"],
    nms_iou_threshold=-1, dec_pred_bbox_embed_share=True, dec_pred_class_embed_share=True,
    use_dn=True, dn_number=100, dn_box_noise_scale=1.0, dn_label_noise_ratio=0.5,
    embed_init_tgt=True, dn_labelbook_size=9, match_unstable_error=True,
    use_ema=False, ema_decay=0.9997, ema_epoch=0, use_detached_boxes_dec_out=False,
    burn_epochs=40, strong_aug=True, pseudo_label_threshold=0.3, ema_decay_teacher=0.9997,
    ema_decay_best_model=0.9, self_training_loss_coef=1.0,
    # CLI defaults
    device="cuda", amp=False, debug=False, dataset_file="city2foggy", seed=42,
)


def c2f_args(**overrides) -> argparse.Namespace:
    cfg = dict(C2F)
    cfg.update(overrides)
    return argparse.Namespace(**cfg)


def sim10k_args(**overrides) -> argparse.Namespace:
    """Sim10k -> Cityscapes (/root/reference/config/DA/Sim10k2Cityscapes/DINO_4scale_sim2cityscapes.py): the C2F
    values with `num_classes = dn_labelbook_size = 2` (:3,109 -- one foreground class, `car`)."""
    return c2f_args(**{**dict(num_classes=2, dn_labelbook_size=2, dataset_file="sim2city"), **overrides})


def bdd_args(**overrides) -> argparse.Namespace:
    """Cityscapes -> BDD100K-daytime (/root/reference/config/DA/Cityscapes2BDD100k/DINO_4scale_city2BDD100k.py):
    identical to C2F on the hot path (the files differ only in the data-transform base they name)."""
    return c2f_args(**{**dict(dataset_file="city2bdd"), **overrides})


def get_param_dict(args, model_without_ddp):
    """'default' grouping (/root/reference/util/get_param_dicts.py:23-31): everything whose
    name contains "backbone" trains at lr_backbone, the rest at lr."""
    if getattr(args, "param_dict_type", "default") != "default":
        raise NotImplementedError("only param_dict_type='default' is used by the DA configs")
    named = list(model_without_ddp.named_parameters())
    return [
        {"params": [p for n, p in named if "backbone" not in n and p.requires_grad]},
        {"params": [p for n, p in named if "backbone" in n and p.requires_grad],
         "lr": args.lr_backbone},
    ]
