"""Deformable transformer of DINO: 6 MSDA encoder layers, two-stage query selection,
6 decoder layers with iterative box refinement.

Mirror of /root/reference/models/dino/deformable_transformer.py (`DeformableTransformer`
:25-426, `TransformerEncoder` :434-577, `TransformerDecoder` :579-763, encoder/decoder
layers :765-994, `build_deformable_transformer` :1004-1068) and the helpers of
/root/reference/models/dino/utils.py (`gen_encoder_output_proposals` :15-61, `MLP` :107-119,
`gen_sineembed_for_position` :138-163).  Only the options the DA configs use are
implemented (deformable encoder+decoder, two_stage_type 'standard' or 'no', decoder
self-attention 'sa', module order sa -> ca -> ffn); anything else raises at build time.
Module / parameter names equal the reference's so state_dicts interchange.
"""
from __future__ import annotations

import copy
import math
import weakref
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .fused import FastLinear
from .fused import attention_qk_d32, fan_out, layer_norm, layer_norm_class_max, linear_relu, refine_boxes
from .fused import linear as fused_linear
from .msda import MSDeformAttn, value_projections
from .nested import inverse_sigmoid


class MLP(nn.Module):
    """Linear -> ReLU -> ... -> Linear (no activation after the last layer)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(FastLinear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            # hidden layers: linear + ReLU as one node (bias + ReLU in the GEMM epilogue on the device)
            x = linear_relu(x, layer.weight, layer.bias) if i < self.num_layers - 1 else layer(x)
        return x


FUSED_ADD_NORM = __import__("os").environ.get("DATR_FUSED_ADD_NORM", "1") != "0"   # A/B switch
_REF_POINTS = {}      # (level shapes, batch, device) -> encoder reference points of an unpadded batch
_UNIT_RATIOS = {}     # (batch, levels, device) -> the all-ones valid ratios of an unpadded batch


def _add_norm(x, branch, dropout, norm):
    """norm(x + dropout(branch)) (deformable_transformer.py:796-806, :856-893); on the device the
    add and the LayerNorm are one pass each way (datr_amd/fused.py::add_layer_norm)."""
    if FUSED_ADD_NORM and x.is_cuda and not (dropout.training and dropout.p > 0):
        from .fused import add_layer_norm
        return add_layer_norm(branch, x, norm)
    return norm(x + dropout(branch))


FUSED_FFN = __import__("os").environ.get("DATR_FUSED_FFN", "1") != "0"     # A/B switch
FUSED_SELF_ATTN = __import__("os").environ.get("DATR_FUSED_SELF_ATTN", "1") != "0"   # A/B switch
OWN_ATTENTION = __import__("os").environ.get("DATR_OWN_ATTENTION", "1") != "0"       # A/B switch
BATCH_FIRST_DECODER = __import__("os").environ.get("DATR_BATCH_FIRST_DECODER", "1") != "0"   # A/B switch


def _plain_mha(m: nn.MultiheadAttention) -> bool:
    return (m._qkv_same_embed_dim and m.in_proj_bias is not None and not m.batch_first
            and m.bias_k is None and not m.add_zero_attn and not (m.training and m.dropout > 0))


def _self_attention(mha: nn.MultiheadAttention, qk_in: Tensor, v_in: Tensor, attn_mask, batch_first=False) -> Tensor:
    """`mha(qk_in, qk_in, v_in, attn_mask=attn_mask, need_weights=False)[0]` for [L, N, E] inputs
    ([N, L, E] with batch_first: same values, the result in the layout of the inputs) --
    the decoder's self-attention call (deformable_transformer.py:880-884: query = key = tgt + pos,
    value = tgt) -- with the same parameters and the same operations per element, arranged for
    fewer launches: the query and key projections share their input, so they are ONE GEMM
    (nn.MultiheadAttention runs three because value differs); linears go through fused.linear
    (column-sum bias gradients); the in_proj parameters are split, not sliced (one backward
    node).  Scaled dot-product attention itself is the own MFMA kernel pair (csrc/mha_fwd.hip,
    csrc/mha_bwd.hip) for head_dim 32 in fp32, PyTorch's otherwise; the kernels take strides, so the
    merged query / key projection goes in whole and its gradient comes back whole (no split / cat)."""
    if batch_first:
        N, L, E = qk_in.shape
    else:
        L, N, E = qk_in.shape
    H = mha.num_heads
    hd = E // H
    w_qk, w_v = mha.in_proj_weight.split([2 * E, E], 0)
    b_qk, b_v = mha.in_proj_bias.split([2 * E, E], 0)
    qk = fused_linear(qk_in, w_qk, b_qk)                                 # layout of the input, 2 E wide
    v = fused_linear(v_in, w_v, b_v)
    if attn_mask is not None and attn_mask.dtype == torch.bool:
        attn_mask = torch.zeros(attn_mask.shape, dtype=qk.dtype, device=qk.device) \
            .masked_fill_(attn_mask, float("-inf"))
    if batch_first:
        qk, v = qk.transpose(0, 1), v.transpose(0, 1)                   # [L, N, .] views of batch-major memory
    if OWN_ATTENTION and hd == 32 and qk.dtype == torch.float32 and all(
            x.stride(0) % 4 == 0 and x.stride(1) % 4 == 0 for x in (qk, v)):
        # own MFMA forward / backward (csrc/mha_fwd.hip, mha_bwd.hip); the output comes in the memory
        # order of the inputs, i.e. the layout out_proj reads
        out = attention_qk_d32(qk, v, None if attn_mask is None else attn_mask.contiguous(), H)
        if batch_first:
            out = out.transpose(0, 1)
        out = out.reshape(L * N, E)
    else:
        q, k = qk.split(E, dim=-1)
        # [L, N, E] -> [N, H, L, hd] (views, as F.multi_head_attention_forward arranges them)
        q, k, v = (x.reshape(L, N * H, hd).transpose(0, 1).reshape(N, H, L, hd) for x in (q, k, v))
        out = F.scaled_dot_product_attention(
            q, k, v, None if attn_mask is None else attn_mask.view(1, 1, L, L), 0.0, False)
        out = (out.permute(0, 2, 1, 3) if batch_first else out.permute(2, 0, 1, 3)).reshape(L * N, E)
    return fused_linear(out, mha.out_proj.weight, mha.out_proj.bias).view(qk_in.shape)


def _ffn(x, linear1, activation, dropout, linear2):
    """linear2(dropout(activation(linear1(x)))) (deformable_transformer.py:803-806, :879-883).
    ReLU without active dropout on the device takes the fused FFN (datr_amd/fused.py)."""
    if FUSED_FFN and activation is F.relu and x.is_cuda and not (dropout.training and dropout.p > 0):
        from .fused import ffn_relu
        return ffn_relu(x, linear1, linear2)
    return linear2(dropout(activation(linear1(x))))


FUSED_FFN_BLOCK = __import__("os").environ.get("DATR_FUSED_FFN_BLOCK", "1") != "0"   # A/B switch


FUSED_NEXT_QUERY = __import__("os").environ.get("DATR_FUSED_NEXT_QUERY", "1") != "0"   # A/B switch


def _ffn_block(x, linear1, activation, dropout_a, linear2, dropout_b, norm, next_pos=None):
    """norm(x + dropout_b(linear2(dropout_a(activation(linear1(x)))))): the FFN sub-block; one autograd
    node on the device (fused.ffn_add_norm) when both fused halves apply, else their composition.
    With `next_pos` the fused node may return the NEXT encoder layer's three handles (query = output +
    next_pos, value input, residual) as a tuple instead of the output."""
    if (FUSED_FFN_BLOCK and FUSED_FFN and FUSED_ADD_NORM and activation is F.relu and x.is_cuda
            and not (dropout_a.training and dropout_a.p > 0) and not (dropout_b.training and dropout_b.p > 0)):
        from .fused import ffn_add_norm
        if next_pos is not None and FUSED_NEXT_QUERY and x.requires_grad:
            out = ffn_add_norm(x, linear1, linear2, norm, next_pos=next_pos)
            if out is not None:
                return out
        out = ffn_add_norm(x, linear1, linear2, norm)
        if out is not None:
            return out
    return _add_norm(x, _ffn(x, linear1, activation, dropout_a, linear2), dropout_b, norm)


def _activation(name: str):
    table = {"relu": F.relu, "gelu": F.gelu, "glu": F.glu, "selu": F.selu}
    if name not in table:
        raise RuntimeError(f"activation should be relu/gelu, not {name}.")
    return table[name]


_DIM_T = {}


def _sine_temperatures(device) -> Tensor:
    t = _DIM_T.get(device)
    if t is None:
        k = torch.arange(128, dtype=torch.float32, device=device)
        t = _DIM_T[device] = 10000 ** (2 * torch.div(k, 2, rounding_mode="floor") / 128)
    return t


def gen_sineembed_for_position(pos_tensor: Tensor) -> Tensor:
    """[nq, bs, 2|4] normalised (x, y[, w, h]) -> [nq, bs, 128 * last_dim] sine embedding in
    the order (y, x[, w, h]); temperature 10000, 128 features per coordinate.  Device float32
    boxes that carry no gradient (the decoder's case: boxes are detached between layers) take
    the one-launch kernel csrc/sine_embed.hip; everything else the torch formulation."""
    if pos_tensor.size(-1) not in (2, 4):
        raise ValueError(f"Unknown pos_tensor shape(-1):{pos_tensor.size(-1)}")
    dim_t = _sine_temperatures(pos_tensor.device)
    if pos_tensor.is_cuda and pos_tensor.dtype == torch.float32 and not pos_tensor.requires_grad:
        from . import _native
        nq, bs, nc = pos_tensor.shape
        pos = pos_tensor.contiguous()
        out = torch.empty(nq, bs, 128 * nc, device=pos.device, dtype=torch.float32)
        with _native.on_device(pos.device):
            rc = _native.lib.datr_sine_embed_f32(pos.data_ptr(), dim_t.data_ptr(), nq * bs, nc,
                                                 out.data_ptr(), _native.current_stream_ptr(pos.device))
        _native.check(rc, "sine_embed")
        return out

    def embed(coord):
        p = (coord * (2 * math.pi))[:, :, None] / dim_t
        return torch.stack((p[:, :, 0::2].sin(), p[:, :, 1::2].cos()), dim=3).flatten(2)

    parts = [embed(pos_tensor[:, :, 1]), embed(pos_tensor[:, :, 0])]
    if pos_tensor.size(-1) == 4:
        parts += [embed(pos_tensor[:, :, 2]), embed(pos_tensor[:, :, 3])]
    return torch.cat(parts, dim=2)


_PROPOSALS = {}


def gen_encoder_output_proposals(memory: Tensor, memory_padding_mask: Tensor,
                                 spatial_shapes, learnedwh=None, no_padding: bool = False, fill_memory: bool = True):
    """One anchor per encoder token: centre = pixel centre / valid extent, size 0.05 * 2^level;
    anchors outside (0.01, 0.99) or on padding become +inf in logit space and their memory is
    zeroed.  `spatial_shapes` is a list of (H, W) python ints or an int64 tensor.
    `no_padding=True` (the caller knows the mask is all False, without looking at it): anchors
    and validity depend on the shapes only and are cached -- 75 small launches per call.
    `fill_memory=False`: returns (invalid [N, S, 1] bool, proposals) and leaves the zeroing to the caller."""
    N = memory.shape[0]
    shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist()
              if isinstance(spatial_shapes, Tensor) else spatial_shapes)]
    key = (N, tuple(shapes), str(memory.device)) if no_padding and learnedwh is None else None
    cached = _PROPOSALS.get(key) if key is not None else None
    if cached is None:
        proposals = []
        cur = 0
        for lvl, (H, W) in enumerate(shapes):
            m = memory_padding_mask[:, cur:cur + H * W].view(N, H, W, 1)
            valid_H = torch.sum(~m[:, :, 0, 0], 1)
            valid_W = torch.sum(~m[:, 0, :, 0], 1)
            gy, gx = torch.meshgrid(
                torch.linspace(0, H - 1, H, dtype=torch.float32, device=memory.device),
                torch.linspace(0, W - 1, W, dtype=torch.float32, device=memory.device),
                indexing="ij")
            grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
            scale = torch.cat([valid_W.unsqueeze(-1), valid_H.unsqueeze(-1)], 1).view(N, 1, 1, 2)
            grid = (grid.unsqueeze(0).expand(N, -1, -1, -1) + 0.5) / scale
            if learnedwh is not None:
                wh = torch.ones_like(grid) * learnedwh.sigmoid() * (2.0 ** lvl)
            else:
                wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
            proposals.append(torch.cat((grid, wh), -1).view(N, -1, 4))
            cur += H * W
        output_proposals = torch.cat(proposals, 1)
        ok = ((output_proposals > 0.01) & (output_proposals < 0.99)).all(-1, keepdim=True)
        output_proposals = torch.log(output_proposals / (1 - output_proposals))
        # masked_fill(padding) then masked_fill(~ok) == one masked_fill with the union
        invalid = memory_padding_mask.unsqueeze(-1) | ~ok
        output_proposals = output_proposals.masked_fill(invalid, float("inf"))
        if key is not None:
            if len(_PROPOSALS) > 16:
                _PROPOSALS.clear()
            _PROPOSALS[key] = (output_proposals, invalid)
    else:
        output_proposals, invalid = cached
    if not fill_memory:
        return invalid, output_proposals
    output_memory = memory.masked_fill(invalid, 0.0)
    return output_memory, output_proposals


class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4,
                 n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _activation(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index,
                key_padding_mask=None, next_pos=None):
        """`src`: the tokens, or the (query, value input, residual) handles the previous layer's FFN node
        returned (query already = tokens + pos).  `next_pos`: the position table of the next layer, when
        there is one -- the result may then be such a triple (TransformerEncoder.forward chains them)."""
        if isinstance(src, tuple):
            q, s_v, s_r = src
        else:
            # three consumers of src (query, value projection, residual): one alias each, their gradients
            # meet in one pass (fused.fan_out) instead of two pairwise adds over the token tensor
            s_q, s_v, s_r = fan_out(src, 3)
            q = s_q if pos is None else s_q + pos
        src = _add_norm(s_r, self.self_attn(q, reference_points, s_v, spatial_shapes,
                                            level_start_index, key_padding_mask),
                        self.dropout1, self.norm1)
        return _ffn_block(src, self.linear1, self.activation, self.dropout2, self.linear2, self.dropout3, self.norm2,
                          next_pos=next_pos)


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None, d_model=256, num_queries=300):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)]) \
            if num_layers > 0 else []
        self.num_layers = num_layers
        self.norm = norm
        self.d_model = d_model
        self.num_queries = num_queries

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel centres of every level, normalised by the valid extent, then re-scaled into
        every level's frame: [N, S, L, 2]."""
        shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist()
                  if isinstance(spatial_shapes, Tensor) else spatial_shapes)]
        per_level = []
        for lvl, (H, W) in enumerate(shapes):
            ry, rx = torch.meshgrid(
                torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device),
                torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device),
                indexing="ij")
            ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
            rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
            per_level.append(torch.stack((rx, ry), -1))
        ref = torch.cat(per_level, 1)
        return ref[:, :, None] * valid_ratios[:, None]

    def forward(self, src, pos, spatial_shapes, level_start_index, valid_ratios,
                key_padding_mask, ref_token_index=None, ref_token_coord=None,
                shapes_list=None):
        assert ref_token_index is None
        output = src
        if self.num_layers > 0:
            shp = shapes_list if shapes_list is not None else spatial_shapes
            if key_padding_mask is None and shapes_list is not None and src.is_cuda and not valid_ratios.requires_grad:
                # unpadded batch (known on the host): the valid ratios are all 1 and the reference points
                # depend on the geometry only -- ~40 small launches per step, computed once per shape
                key = (tuple(shapes_list), int(src.shape[0]), str(src.device))
                reference_points = _REF_POINTS.get(key)
                if reference_points is None:
                    if len(_REF_POINTS) >= 16:
                        _REF_POINTS.clear()
                    reference_points = _REF_POINTS[key] = self.get_reference_points(shp, valid_ratios, device=src.device)
            else:
                reference_points = self.get_reference_points(shp, valid_ratios, device=src.device)
        # the position table feeds every layer: one alias per layer, one gradient sum
        pos_l = fan_out(pos, len(self.layers)) if (pos is not None and 2 <= len(self.layers) <= 8) else None
        for li, layer in enumerate(self.layers):
            # the layer's last node also writes the next layer's query (tokens + its position table)
            nxt = None
            if pos is not None and li + 1 < len(self.layers):
                nxt = pos if pos_l is None else pos_l[li + 1]
            output = layer(src=output, pos=pos if pos_l is None else pos_l[li], reference_points=reference_points,
                           spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                           key_padding_mask=key_padding_mask, next_pos=nxt)
        if self.norm is not None:
            output = self.norm(output)
        return output, None, None


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4,
                 n_heads=8, n_points=4, decoder_sa_type="sa", module_seq=("sa", "ca", "ffn")):
        super().__init__()
        if sorted(module_seq) != ["ca", "ffn", "sa"]:
            raise ValueError(f"module_seq must be a permutation of sa/ca/ffn, got {module_seq}")
        if decoder_sa_type != "sa":
            raise NotImplementedError("only decoder_sa_type='sa' is on the hot path")
        self.module_seq = list(module_seq)
        self.decoder_sa_type = decoder_sa_type
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _activation(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.label_embedding = None
        self.key_aware_type = None
        self.key_aware_proj = None

    def forward_ffn(self, tgt):
        return _ffn_block(tgt, self.linear1, self.activation, self.dropout3, self.linear2, self.dropout4, self.norm3)

    def forward_sa(self, tgt, query_pos, attn_mask, batch_first=False):
        q = k = tgt if query_pos is None else tgt + query_pos
        if FUSED_SELF_ATTN and tgt.is_cuda and _plain_mha(self.self_attn):
            tgt2 = _self_attention(self.self_attn, q, tgt, attn_mask, batch_first)
        elif batch_first:
            tgt2 = self.self_attn(q.transpose(0, 1), k.transpose(0, 1), tgt.transpose(0, 1), attn_mask=attn_mask,
                                  need_weights=False)[0].transpose(0, 1)
        else:
            tgt2 = self.self_attn(q, k, tgt, attn_mask=attn_mask, need_weights=False)[0]
        return _add_norm(tgt, tgt2, self.dropout2, self.norm2)

    def forward_ca(self, tgt, query_pos, reference_points, memory, spatial_shapes,
                   level_start_index, key_padding_mask, value=None, grad_slot=None, batch_first=False):
        q = tgt if query_pos is None else tgt + query_pos
        kw = {} if value is None else {"value": value, "grad_slot": grad_slot}
        if batch_first:        # tgt, reference_points are [bs, nq, .]: the layout the attention module works in
            tgt2 = self.cross_attn(q, reference_points, memory.transpose(0, 1), spatial_shapes, level_start_index,
                                   key_padding_mask, **kw)
        else:
            tgt2 = self.cross_attn(q.transpose(0, 1), reference_points.transpose(0, 1).contiguous(),
                                   memory.transpose(0, 1), spatial_shapes, level_start_index,
                                   key_padding_mask, **kw).transpose(0, 1)
        return _add_norm(tgt, tgt2, self.dropout1, self.norm1)

    def forward(self, tgt, tgt_query_pos=None, tgt_query_sine_embed=None,
                tgt_key_padding_mask=None, tgt_reference_points=None, memory=None,
                memory_key_padding_mask=None, memory_level_start_index=None,
                memory_spatial_shapes=None, memory_pos=None, self_attn_mask=None,
                cross_attn_mask=None, memory_value=None, memory_grad_slot=None, batch_first=False):
        """batch_first: tgt / tgt_query_pos / tgt_reference_points are [bs, nq, .] instead of the reference's
        [nq, bs, .] (same values; every op but the self-attention is per row)."""
        for name in self.module_seq:
            if name == "ffn":
                tgt = self.forward_ffn(tgt)
            elif name == "ca":
                tgt = self.forward_ca(tgt, tgt_query_pos, tgt_reference_points, memory,
                                      memory_spatial_shapes, memory_level_start_index,
                                      memory_key_padding_mask, memory_value, memory_grad_slot, batch_first)
            else:
                tgt = self.forward_sa(tgt, tgt_query_pos, self_attn_mask, batch_first)
        return tgt


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False,
                 d_model=256, query_dim=4, num_feature_levels=1,
                 use_detached_boxes_dec_out=False):
        super().__init__()
        assert return_intermediate, "support return_intermediate only"
        assert query_dim in (2, 4)
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)]) \
            if num_layers > 0 else []
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate
        self.query_dim = query_dim
        self.num_feature_levels = num_feature_levels
        self.use_detached_boxes_dec_out = use_detached_boxes_dec_out
        self.ref_point_head = MLP(query_dim // 2 * d_model, d_model, d_model, 2)
        self.query_pos_sine_scale = None
        self.query_scale = None
        self.bbox_embed = None          # set by DINO: shared box-refinement heads
        self.class_embed = None
        self.d_model = d_model
        self.ref_anchor_head = None
        self.rm_detach = None

    def forward(self, tgt, memory, tgt_mask: Optional[Tensor] = None,
                memory_mask: Optional[Tensor] = None,
                tgt_key_padding_mask: Optional[Tensor] = None,
                memory_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None,
                refpoints_unsigmoid: Optional[Tensor] = None,
                level_start_index: Optional[Tensor] = None,
                spatial_shapes: Optional[Tensor] = None, valid_ratios: Optional[Tensor] = None):
        output = tgt
        intermediate = []
        if tgt_mask is not None and tgt_mask.dtype == torch.bool:
            # nn.MultiheadAttention turns a boolean mask into an additive one in every layer
            # (F._canonical_mask: zeros + masked_fill -inf); do it once, and once per mask object
            # (prepare_for_cdn hands out the same cached mask every step)
            cached = getattr(self, "_additive_mask", None)
            if cached is None or cached[0] is not tgt_mask:
                cached = (tgt_mask, torch.zeros(tgt_mask.shape, dtype=tgt.dtype, device=tgt_mask.device)
                          .masked_fill_(tgt_mask, float("-inf")))
                self._additive_mask = cached
            tgt_mask = cached[1]
        # On the device the layers run BATCH-first ([bs, nq, C]: what the caller's tensors are in memory --
        # it hands over transposed views -- and what the cross-attention works in), so no sub-block has
        # to re-lay its input or output; the reference's [nq, bs, C] order otherwise.  Per-row values are
        # identical; the results are handed back batch-first either way.
        bf = BATCH_FIRST_DECODER and tgt.is_cuda and tgt.dim() == 3 and refpoints_unsigmoid.dim() == 3
        if bf:
            output = output.transpose(0, 1)
            output = output if output.is_contiguous() else output.contiguous()
            refpoints_unsigmoid = refpoints_unsigmoid.transpose(0, 1)
        reference_points = refpoints_unsigmoid.sigmoid()
        ref_points = [reference_points]
        # The memory feeds every layer's value projection.  All six projections as ONE autograd node
        # (msda.value_projections): the attention calls leave their value gradients in one shared
        # buffer and the data / weight / bias gradients of the six projections are one GEMM / GEMM /
        # column sum.  Otherwise one alias per layer, one gradient sum (fan_out).
        batched = None
        if memory_key_padding_mask is None and memory.dim() == 3 \
                and all(isinstance(getattr(l, "cross_attn", None), MSDeformAttn) for l in self.layers):
            mem_nsc = memory.transpose(0, 1)
            if mem_nsc.is_contiguous():
                batched = value_projections(mem_nsc, [l.cross_attn for l in self.layers])
        if batched is None:
            mem_l = fan_out(memory, len(self.layers)) if 2 <= len(self.layers) <= 8 else (memory,) * len(self.layers)
        else:
            mem_l = (memory,) * len(self.layers)
        for layer_id, layer in enumerate(self.layers):
            vr = torch.cat([valid_ratios, valid_ratios], -1) if reference_points.shape[-1] == 4 else valid_ratios
            ref_in = reference_points[:, :, None] * (vr[:, None] if bf else vr[None, :])
            query_sine_embed = gen_sineembed_for_position(ref_in[:, :, 0, :])
            query_pos = self.ref_point_head(query_sine_embed)
            output = layer(tgt=output, tgt_query_pos=query_pos,
                           tgt_query_sine_embed=query_sine_embed,
                           tgt_key_padding_mask=tgt_key_padding_mask,
                           tgt_reference_points=ref_in, memory=mem_l[layer_id],
                           memory_key_padding_mask=memory_key_padding_mask,
                           memory_level_start_index=level_start_index,
                           memory_spatial_shapes=spatial_shapes, memory_pos=pos,
                           self_attn_mask=tgt_mask, cross_attn_mask=memory_mask, batch_first=bf,
                           **({} if batched is None else {"memory_value": batched[0][layer_id],
                                                          "memory_grad_slot": (batched[1], layer_id)}))
            if self.bbox_embed is not None:
                # iterative refinement: the next layer starts from this layer's box, detached
                new_ref = refine_boxes(self.bbox_embed[layer_id](output), reference_points)
                reference_points = new_ref.detach()
                ref_points.append(reference_points if self.use_detached_boxes_dec_out else new_ref)
            intermediate.append(layer_norm(output, self.norm))
        if bf:
            return [intermediate, ref_points]
        return [[x.transpose(0, 1) for x in intermediate],
                [r.transpose(0, 1) for r in ref_points]]


# DATR_SELECTED_ROWS_BWD=0: differentiate enc_output over all encoder tokens, as the reference does
SELECTED_ROWS_BACKWARD = __import__("os").environ.get("DATR_SELECTED_ROWS_BWD", "1") != "0"
FUSED_CLASS_SCORES = __import__("os").environ.get("DATR_FUSED_CLASS_SCORES", "1") != "0"   # A/B switch
_POS_TABLES = {}          # ids of the per-level position embeddings -> (weak refs, flattened [N, S, C] table)


class _AddLevelEmbed(torch.autograd.Function):
    """pos [N, S, C] (no gradient) + level_embed[level of token] -> [N, S, C]; sizes = tokens per level."""

    @staticmethod
    def forward(ctx, pos, level_embed, sizes):
        ctx.sizes = sizes
        rows = torch.cat([level_embed[l].expand(n, -1) for l, n in enumerate(sizes)], 0)   # no host -> device copy
        return pos + rows.unsqueeze(0)

    @staticmethod
    def backward(ctx, d):
        from .fused import column_sums
        per_token = d.sum(0) if d.shape[0] > 1 else d[0]                 # [S, C]
        per_token = per_token.contiguous()
        out, start = [], 0
        for n in ctx.sizes:
            out.append(column_sums(per_token[start:start + n]))
            start += n
        return None, torch.stack(out, 0), None


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_queries=300, num_encoder_layers=6,
                 num_unicoder_layers=0, num_decoder_layers=6, dim_feedforward=2048, dropout=0.0,
                 activation="relu", normalize_before=False, return_intermediate_dec=False,
                 query_dim=4, num_patterns=0, modulate_hw_attn=False, deformable_encoder=False,
                 deformable_decoder=False, num_feature_levels=1, enc_n_points=4, dec_n_points=4,
                 use_deformable_box_attn=False, box_attn_type="roi_align",
                 learnable_tgt_init=False, decoder_query_perturber=None,
                 add_channel_attention=False, add_pos_value=False, random_refpoints_xy=False,
                 two_stage_type="no", two_stage_pat_embed=0, two_stage_add_query_num=0,
                 two_stage_learn_wh=False, two_stage_keep_all_tokens=False, dec_layer_number=None,
                 rm_enc_query_scale=True, rm_dec_query_scale=True, rm_self_attn_layers=None,
                 key_aware_type=None, layer_share_type=None, rm_detach=None,
                 decoder_sa_type="ca", module_seq=("sa", "ca", "ffn"), embed_init_tgt=False,
                 use_detached_boxes_dec_out=False):
        super().__init__()
        unsupported = dict(
            use_deformable_box_attn=use_deformable_box_attn, add_channel_attention=add_channel_attention,
            two_stage_pat_embed=two_stage_pat_embed, two_stage_add_query_num=two_stage_add_query_num,
            two_stage_keep_all_tokens=two_stage_keep_all_tokens, dec_layer_number=dec_layer_number,
            rm_self_attn_layers=rm_self_attn_layers, key_aware_type=key_aware_type,
            layer_share_type=layer_share_type, rm_detach=rm_detach,
            decoder_query_perturber=decoder_query_perturber, num_patterns=num_patterns,
            random_refpoints_xy=random_refpoints_xy)
        for k, v in unsupported.items():
            if v:
                raise NotImplementedError(f"DeformableTransformer option {k}={v!r} is off the hot path")
        assert deformable_encoder and deformable_decoder, "only the deformable enc/dec is built"
        assert query_dim == 4 and learnable_tgt_init
        assert two_stage_type in ("no", "standard"), f"unknown param {two_stage_type} of two_stage_type"
        self.num_feature_levels = num_feature_levels
        self.num_encoder_layers = num_encoder_layers
        self.num_unicoder_layers = num_unicoder_layers
        self.num_decoder_layers = num_decoder_layers
        self.deformable_encoder, self.deformable_decoder = True, True
        self.two_stage_keep_all_tokens = False
        self.num_queries = num_queries
        self.random_refpoints_xy = False
        self.use_detached_boxes_dec_out = use_detached_boxes_dec_out
        self.decoder_sa_type = decoder_sa_type
        self.d_model, self.nhead, self.dec_layers = d_model, nhead, num_decoder_layers
        self.num_patterns = 0

        enc_layer = DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, enc_n_points)
        self.encoder = TransformerEncoder(enc_layer, num_encoder_layers,
                                          nn.LayerNorm(d_model) if normalize_before else None,
                                          d_model=d_model, num_queries=num_queries)
        dec_layer = DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, dec_n_points,
                                                      decoder_sa_type=decoder_sa_type,
                                                      module_seq=module_seq)
        self.decoder = TransformerDecoder(dec_layer, num_decoder_layers, nn.LayerNorm(d_model),
                                          return_intermediate=return_intermediate_dec,
                                          d_model=d_model, query_dim=query_dim,
                                          num_feature_levels=num_feature_levels,
                                          use_detached_boxes_dec_out=use_detached_boxes_dec_out)
        self.level_embed = None
        if num_feature_levels > 1 and num_encoder_layers > 0:
            self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.learnable_tgt_init = True
        self.embed_init_tgt = embed_init_tgt
        if (two_stage_type != "no" and embed_init_tgt) or two_stage_type == "no":
            self.tgt_embed = nn.Embedding(num_queries, d_model)
            nn.init.normal_(self.tgt_embed.weight.data)
        else:
            self.tgt_embed = None
        self.two_stage_type = two_stage_type
        self.two_stage_pat_embed = 0
        self.two_stage_add_query_num = 0
        self.two_stage_learn_wh = two_stage_learn_wh
        if two_stage_type == "standard":
            self.enc_output = FastLinear(d_model, d_model)
            self.enc_output_norm = nn.LayerNorm(d_model)
            self.two_stage_wh_embedding = nn.Embedding(1, 2) if two_stage_learn_wh else None
        if two_stage_type == "no":
            self.refpoint_embed = nn.Embedding(num_queries, 4)
        self.enc_out_class_embed = None   # set by DINO
        self.enc_out_bbox_embed = None
        self.dec_layer_number = None
        self._reset_parameters()
        self.rm_self_attn_layers = None
        self.rm_detach = None
        self._meta_cache = {}

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        if self.num_feature_levels > 1 and self.level_embed is not None:
            nn.init.normal_(self.level_embed)
        if self.two_stage_learn_wh:
            nn.init.constant_(self.two_stage_wh_embedding.weight, math.log(0.05 / (1 - 0.05)))

    def select_queries(self, scores: Tensor) -> Tensor:
        """Indices [N, num_queries] of the highest-scoring encoder tokens
        (deformable_transformer.py:342: `torch.topk(scores, num_queries, dim=1)[1]`), in ONE
        defined order -- descending score, equal scores by ascending token index
        (`fused.topk_rows`, csrc/topk.hip) -- because `torch.topk` fixes neither the order nor,
        at the 900th score, the membership of ties.  A method of its own because it is THE
        discontinuity of the forward pass: parity tests pin it as a function and may substitute
        the reference's selection when comparing what comes after it."""
        from .fused import topk_rows
        return topk_rows(scores, self.num_queries)[1]

    @staticmethod
    def get_valid_ratio(mask):
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)

    def _level_meta(self, shapes_list, device):
        """int64 device tensors [L,2] / [L] for the MSDA op, cached per geometry so the step
        issues no per-call host->device copies."""
        key = (tuple(shapes_list), str(device))
        hit = self._meta_cache.get(key)
        if hit is None:
            spatial_shapes = torch.as_tensor(shapes_list, dtype=torch.long, device=device)
            level_start_index = torch.cat((spatial_shapes.new_zeros((1,)),
                                           spatial_shapes.prod(1).cumsum(0)[:-1]))
            if len(self._meta_cache) > 64:
                self._meta_cache.clear()
            hit = self._meta_cache[key] = (spatial_shapes, level_start_index)
        return hit

    # Set by the caller (DINO.forward) when the batch is KNOWN on the host to contain no padded
    # pixel (all images of equal size: `NestedTensor.padded is False`): the attention layers then
    # skip `value.masked_fill(mask, 0)` (ms_deform_attn.py:101-102), a no-op on an all-False mask
    # that still costs a pass over value per layer.  Proposals / valid ratios keep using the mask.
    no_padding = False

    def _level_positions(self, pos_embeds):
        """cat_l(flatten(pos_l) + level_embed[l]) -> [N, S, C]
        (/root/reference/models/dino/deformable_transformer.py:283-286).  On the device with
        gradient-free position embeddings (the sine embedding) the flattened table is one cat -- kept
        across steps when the embeddings are the cached per-shape tensors of an unpadded batch -- and
        the level embedding is added by ONE broadcast add whose backward is a sum over the batch and
        one deterministic column-sum launch per level (fused.column_sums), instead of four adds, a cat
        and four ATen reductions over [N, tokens_l, C]."""
        with_level = self.num_feature_levels > 1 and self.level_embed is not None
        fast = (with_level and pos_embeds[0].is_cuda and pos_embeds[0].dtype == torch.float32
                and not any(p.requires_grad for p in pos_embeds))
        if not fast:
            flat = [p.flatten(2).transpose(1, 2) for p in pos_embeds]
            if with_level:
                flat = [f + self.level_embed[lvl].view(1, 1, -1) for lvl, f in enumerate(flat)]
            return torch.cat(flat, 1)
        # keyed by the identity of the embedding tensors (weak references: a dead tensor's id may be reused)
        tables = _POS_TABLES
        key = tuple(id(p) for p in pos_embeds)
        hit = tables.get(key) if self.no_padding else None
        if hit is not None and all(r() is p for r, p in zip(hit[0], pos_embeds)):
            cat = hit[1]
        else:
            cat = torch.cat([p.flatten(2).transpose(1, 2) for p in pos_embeds], 1).contiguous()
            if self.no_padding:
                if len(tables) >= 8:
                    tables.clear()
                tables[key] = (tuple(weakref.ref(p) for p in pos_embeds), cat)
        sizes = [int(p.shape[2] * p.shape[3]) for p in pos_embeds]
        return _AddLevelEmbed.apply(cat, self.level_embed, sizes)

    def encode(self, srcs, masks, pos_embeds):
        """Flatten the pyramid and run the deformable encoder.  Every encoder operation is
        per-sample, so the source and target halves of a DATR batch can share one call."""
        src_flatten, mask_flatten, lvl_pos_embed_flatten, shapes_list = [], [], [], []
        for lvl, (src, mask, pos_embed) in enumerate(zip(srcs, masks, pos_embeds)):
            bs, c, h, w = src.shape
            shapes_list.append((h, w))
            src_flatten.append(src.flatten(2).transpose(1, 2))
            mask_flatten.append(mask.flatten(1))
            lvl_pos_embed_flatten.append(pos_embed)
        src_flatten = torch.cat(src_flatten, 1)
        mask_flatten = torch.cat(mask_flatten, 1)
        lvl_pos_embed_flatten = self._level_positions(lvl_pos_embed_flatten)
        spatial_shapes, level_start_index = self._level_meta(shapes_list, src_flatten.device)
        if self.no_padding and src_flatten.is_cuda:
            # no padded pixel anywhere: valid_W / W = valid_H / H = 1 exactly (get_valid_ratio would compute that
            # with 9 launches per level)
            key = (int(src_flatten.shape[0]), len(masks), str(src_flatten.device))
            valid_ratios = _UNIT_RATIOS.get(key)
            if valid_ratios is None:
                valid_ratios = _UNIT_RATIOS[key] = torch.ones(key[0], key[1], 2, dtype=torch.float32, device=src_flatten.device)
        else:
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
        memory, _, _ = self.encoder(src_flatten, pos=lvl_pos_embed_flatten,
                                    level_start_index=level_start_index,
                                    spatial_shapes=spatial_shapes, valid_ratios=valid_ratios,
                                    key_padding_mask=None if self.no_padding else mask_flatten,
                                    shapes_list=shapes_list)
        return {"memory": memory, "mask": mask_flatten, "pos": lvl_pos_embed_flatten,
                "shapes_list": shapes_list, "spatial_shapes": spatial_shapes,
                "level_start_index": level_start_index, "valid_ratios": valid_ratios}

    @staticmethod
    def slice_encoded(enc, sl):
        """The batch slice `sl` of an encode() result (e.g. the source or the target half)."""
        out = dict(enc)
        for k in ("memory", "mask", "pos", "valid_ratios"):
            out[k] = enc[k][sl]
        return out

    def decode(self, enc, refpoint_embed, tgt, attn_mask=None):
        """Two-stage query selection + decoder on an encode() result."""
        memory, mask_flatten, lvl_pos_embed_flatten = enc["memory"], enc["mask"], enc["pos"]
        shapes_list, spatial_shapes = enc["shapes_list"], enc["spatial_shapes"]
        level_start_index, valid_ratios = enc["level_start_index"], enc["valid_ratios"]
        bs = memory.shape[0]

        if self.two_stage_type == "standard":
            input_hw = self.two_stage_wh_embedding.weight[0] if self.two_stage_learn_wh else None
            # Only the selected 900 tokens per image carry gradient back into enc_output / its norm: the
            # class scores of all tokens feed the (non-differentiable) top-k and nothing else reads the
            # unselected rows (deformable_transformer.py:338-350).  On the device, in training, the pass
            # over ALL tokens therefore runs without autograd, and the selected rows are projected and
            # normalised again WITH autograd -- 3 600 rows instead of 88 892 through the backward of the
            # projection (weight gradient, data gradient), the LayerNorm and the masking.  Same values
            # up to the GEMM's rounding for a different row count; the host path keeps the reference's order.
            sparse = (memory.is_cuda and torch.is_grad_enabled() and memory.requires_grad and input_hw is None
                      and SELECTED_ROWS_BACKWARD)
            with torch.set_grad_enabled(torch.is_grad_enabled() and not sparse):
                scores = None
                head, proj = self.enc_out_class_embed, self.enc_output
                if sparse and FUSED_CLASS_SCORES and isinstance(proj, nn.Linear) and proj.bias is not None:
                    # training on the device: nothing but the top-k reads the zeroed / projected / normalised
                    # tokens and their logits, so they are never written: the projection runs on `memory` as it
                    # is (the projection of a zeroed token is the bias: the kernel substitutes it row-wise) and
                    # LayerNorm + class head + max are one pass (fused.layer_norm_class_max)
                    invalid, output_proposals = gen_encoder_output_proposals(
                        memory, mask_flatten, shapes_list, input_hw, no_padding=self.no_padding, fill_memory=False)
                    scores = layer_norm_class_max(proj(memory), self.enc_output_norm, head,
                                                  row_mask=invalid, row_fill=proj.bias)
                    output_memory = memory                 # (device / dtype of what follows)
                if scores is None:
                    output_memory, output_proposals = gen_encoder_output_proposals(
                        memory, mask_flatten, shapes_list, input_hw, no_padding=self.no_padding)
                    output_memory = layer_norm(proj(output_memory), self.enc_output_norm)
                    scores = head(output_memory).max(-1)[0]
                topk_idx = self.select_queries(scores)
            selected_proposals = torch.gather(output_proposals, 1, topk_idx.unsqueeze(-1).repeat(1, 1, 4))
            if sparse:
                rows = torch.gather(memory, 1, topk_idx.unsqueeze(-1).repeat(1, 1, self.d_model))
                rows = rows.masked_fill(torch.isinf(selected_proposals[..., :1]), 0.0)     # invalid anchors
                tgt_undetach = layer_norm(self.enc_output(rows), self.enc_output_norm)
            else:
                tgt_undetach = torch.gather(output_memory, 1,
                                            topk_idx.unsqueeze(-1).repeat(1, 1, self.d_model))
            if output_memory.is_cuda:
                # The reference runs the box head on ALL tokens and gathers the selected 900 per image
                # (deformable_transformer.py:339-343); the head is a per-token MLP and nothing else reads
                # the unselected rows, so on the device it runs on the gathered rows only: 3 600 instead
                # of 88 892 rows through three layers, forward and backward (same values up to the
                # GEMM's rounding for a different row count; the host path keeps the reference's order).
                refpoint_embed_undetach = self.enc_out_bbox_embed(tgt_undetach) + selected_proposals   # logits
            else:
                enc_coord = self.enc_out_bbox_embed(output_memory) + output_proposals
                refpoint_embed_undetach = torch.gather(enc_coord, 1, topk_idx.unsqueeze(-1).repeat(1, 1, 4))
            refpoint_embed_ = refpoint_embed_undetach.detach()
            init_box_proposal = selected_proposals.sigmoid()
            if self.embed_init_tgt:
                tgt_ = self.tgt_embed.weight[:, None, :].repeat(1, bs, 1).transpose(0, 1)
            else:
                tgt_ = tgt_undetach.detach()
            if refpoint_embed is not None:
                refpoint_embed = torch.cat([refpoint_embed, refpoint_embed_], dim=1)
                tgt = torch.cat([tgt, tgt_], dim=1)
            else:
                refpoint_embed, tgt = refpoint_embed_, tgt_
        else:
            tgt_ = self.tgt_embed.weight[:, None, :].repeat(1, bs, 1).transpose(0, 1)
            refpoint_embed_ = self.refpoint_embed.weight[:, None, :].repeat(1, bs, 1).transpose(0, 1)
            if refpoint_embed is not None:
                refpoint_embed = torch.cat([refpoint_embed, refpoint_embed_], dim=1)
                tgt = torch.cat([tgt, tgt_], dim=1)
            else:
                refpoint_embed, tgt = refpoint_embed_, tgt_
            init_box_proposal = refpoint_embed_.sigmoid()

        hs, references = self.decoder(
            tgt=tgt.transpose(0, 1), memory=memory.transpose(0, 1),
            memory_key_padding_mask=None if self.no_padding else mask_flatten,
            pos=lvl_pos_embed_flatten.transpose(0, 1),
            refpoints_unsigmoid=refpoint_embed.transpose(0, 1),
            level_start_index=level_start_index, spatial_shapes=spatial_shapes,
            valid_ratios=valid_ratios, tgt_mask=attn_mask)

        if self.two_stage_type == "standard":
            hs_enc = tgt_undetach.unsqueeze(0)
            ref_enc = refpoint_embed_undetach.sigmoid().unsqueeze(0)
        else:
            hs_enc = ref_enc = None
        return hs, references, hs_enc, ref_enc, init_box_proposal

    def forward(self, srcs, masks, refpoint_embed, pos_embeds, tgt, attn_mask=None):
        """Reference signature (deformable_transformer.py:256): encoder, selection, decoder."""
        return self.decode(self.encode(srcs, masks, pos_embeds), refpoint_embed, tgt, attn_mask)


def build_deformable_transformer(args):
    if getattr(args, "decoder_layer_noise", False):
        raise NotImplementedError("decoder_layer_noise is off in every DA config")
    return DeformableTransformer(
        d_model=args.hidden_dim, dropout=args.dropout, nhead=args.nheads,
        num_queries=args.num_queries, dim_feedforward=args.dim_feedforward,
        num_encoder_layers=args.enc_layers, num_unicoder_layers=args.unic_layers,
        num_decoder_layers=args.dec_layers, normalize_before=args.pre_norm,
        return_intermediate_dec=True, query_dim=args.query_dim,
        activation=args.transformer_activation, num_patterns=args.num_patterns,
        modulate_hw_attn=True, deformable_encoder=True, deformable_decoder=True,
        num_feature_levels=args.num_feature_levels, enc_n_points=args.enc_n_points,
        dec_n_points=args.dec_n_points, use_deformable_box_attn=args.use_deformable_box_attn,
        box_attn_type=getattr(args, "box_attn_type", "roi_align"), learnable_tgt_init=True,
        decoder_query_perturber=None, add_channel_attention=args.add_channel_attention,
        add_pos_value=getattr(args, "add_pos_value", False),
        random_refpoints_xy=args.random_refpoints_xy, two_stage_type=args.two_stage_type,
        two_stage_pat_embed=args.two_stage_pat_embed,
        two_stage_add_query_num=args.two_stage_add_query_num,
        two_stage_learn_wh=args.two_stage_learn_wh,
        two_stage_keep_all_tokens=args.two_stage_keep_all_tokens,
        dec_layer_number=args.dec_layer_number, rm_self_attn_layers=None, key_aware_type=None,
        layer_share_type=None, rm_detach=None, decoder_sa_type=args.decoder_sa_type,
        module_seq=args.decoder_module_seq, embed_init_tgt=args.embed_init_tgt,
        use_detached_boxes_dec_out=getattr(args, "use_detached_boxes_dec_out", False))
