"""
DETR-style fusion encoder and box-refining decoder of the SAM3 image model, on PyTorch-ROCm.

Restates, for the training step of row a14:
  * ``sam3/model/model_misc.py``: ``inverse_sigmoid`` :20-28, ``MultiheadAttentionWrapper`` :31-34, ``DotProductScoring``
    :37-91, ``MLP`` :149-185, ``TransformerWrapper`` :122-146 (xavier re-initialisation), ``gen_sineembed_for_position``
    :225-262;
  * ``sam3/model/encoder.py``: ``TransformerEncoderLayer`` :12-252, ``TransformerEncoderFusion`` :448-575 in the
    single-feature-level, no-padding-mask configuration ``model_builder.py:115-150`` builds;
  * ``sam3/model/decoder.py``: ``TransformerDecoderLayer`` :30-187, ``TransformerDecoder`` :190-611 (DAC one-to-many
    query doubling, presence token, log-scale box-relative position bias, iterative box refinement).

Module, parameter and output-dictionary names are the reference's (state dicts load ``strict=True``; the LoRA
injectors gate on ``transformer.encoder`` / ``transformer.decoder`` and match ``linear1`` / ``linear2`` / ``out_proj``
by name).  Tensors are sequence-first ``[L, B, C]`` unless a layer is built ``batch_first``.

Per-layer activation checkpointing follows the reference default (on while training) and can be switched off
(``use_act_checkpoint=False``): 288 GB of HBM3E hold these activations at batch 8 many times over.
"""
from __future__ import annotations

import copy
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from .matcher import box_cxcywh_to_xyxy

__all__ = ["inverse_sigmoid", "MultiheadAttention", "MLP", "DotProductScoring", "TransformerWrapper",
           "TransformerEncoderLayer", "TransformerEncoderFusion", "TransformerDecoderLayer", "TransformerDecoder",
           "gen_sineembed_for_position", "clones"]


def inverse_sigmoid(x: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class MultiheadAttention(nn.MultiheadAttention):
    """``nn.MultiheadAttention`` (same parameters, same state-dict keys, same arithmetic) evaluated without the detours
    of the stock forward: the three input projections are plain 2-D GEMMs on the tensors as they arrive (the stock
    ``batch_first`` path transposes first and then projects the transposed VIEW, which runs as a 5184-batch bmm, and it
    projects q / k / v separately whenever ``query is key`` but ``value`` differs -- the fusion encoder's case), heads are
    strided views for SDPA (no permute copies), the averaged attention map is never built, and SDPA tries the backend
    measured fastest on MI355X for these head-dim-32 shapes first (tools/sdpa_probe_detr.py: "efficient" 2.2 ms vs
    "flash" 2.4 ms fwd+bwd at [8, 8, 5184, 32]; with an additive bias only "efficient" applies).
    ``out_proj`` is read through ``.weight`` / ``.bias`` exactly as the stock module does, so an adapter wrapped around it
    by the package injector stays unused as it does in the reference."""

    def forward(self, query, key, value, key_padding_mask=None, need_weights=False, attn_mask=None,
                average_attn_weights=True, is_causal=False):
        if (not self._qkv_same_embed_dim or self.bias_k is not None or self.add_zero_attn or self.in_proj_bias is None
                or not query.is_cuda):
            return super().forward(query, key, value, key_padding_mask=key_padding_mask, need_weights=False,
                                   attn_mask=attn_mask, average_attn_weights=average_attn_weights, is_causal=is_causal)
        E, H = self.embed_dim, self.num_heads
        D = E // H
        w, b = self.in_proj_weight, self.in_proj_bias
        if query is key and key is value:
            q, k, v = F.linear(query, w, b).chunk(3, dim=-1)
        elif key is value:
            q = F.linear(query, w[:E], b[:E])
            k, v = F.linear(key, w[E:], b[E:]).chunk(2, dim=-1)
        elif query is key:
            q, k = F.linear(query, w[:2 * E], b[:2 * E]).chunk(2, dim=-1)
            v = F.linear(value, w[2 * E:], b[2 * E:])
        else:
            q, k, v = F.linear(query, w[:E], b[:E]), F.linear(key, w[E:2 * E], b[E:2 * E]), F.linear(value, w[2 * E:], b[2 * E:])
        if self.batch_first:                                   # [B, L, E] -> [B, H, L, D] (views)
            B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
            q, k, v = (t.reshape(B, t.shape[1], H, D).transpose(1, 2) for t in (q, k, v))
        else:                                                  # [L, B, E] -> [B, H, L, D] (views)
            B, Lq, Lk = q.shape[1], q.shape[0], k.shape[0]
            q, k, v = (t.reshape(t.shape[0], B, H, D).permute(1, 2, 0, 3) for t in (q, k, v))
        mask = None
        if attn_mask is not None:
            mask = attn_mask
            if mask.dtype != torch.bool and mask.dtype != q.dtype:
                mask = mask.to(q.dtype)
            mask = mask.view(B, H, Lq, Lk) if mask.dim() == 3 else mask          # [B*H, Lq, Lk] or [Lq, Lk]
        if key_padding_mask is not None:
            pad = key_padding_mask.view(B, 1, 1, Lk)
            if pad.dtype == torch.bool:
                pad = torch.zeros_like(pad, dtype=q.dtype).masked_fill(pad, float("-inf"))
            if mask is None:
                mask = pad
            else:
                if mask.dtype == torch.bool:                   # nn.MultiheadAttention: True = masked
                    mask = torch.zeros_like(mask, dtype=q.dtype).masked_fill(mask, float("-inf"))
                mask = mask + pad
        elif mask is not None and mask.dtype == torch.bool:
            mask = ~mask                                       # nn.MultiheadAttention: True = masked; SDPA: True = attend
        from torch.nn.attention import SDPBackend, sdpa_kernel
        with sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION, SDPBackend.MATH], set_priority=True):
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=self.dropout if self.training else 0.0,
                                               is_causal=bool(is_causal and mask is None))
        if self.batch_first:
            o = o.transpose(1, 2).reshape(B, Lq, E)
        else:
            o = o.permute(2, 0, 1, 3).reshape(Lq, B, E)
        return F.linear(o, self.out_proj.weight, self.out_proj.bias), None


def clones(module: nn.Module, n: int) -> nn.ModuleList:
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def _activation(name: str):
    try:
        return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[name]
    except KeyError:
        raise RuntimeError(f"activation should be relu/gelu, not {name}.") from None


def _maybe_checkpoint(enabled: bool, fn, *args):
    return checkpoint(fn, *args, use_reentrant=False) if enabled else fn(*args)


class MLP(nn.Module):
    """``num_layers`` Linears with ReLU (+dropout) between them, optional residual and output norm."""

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_layers: int, dropout: float = 0.0,
                 residual: bool = False, out_norm: Optional[nn.Module] = None):
        super().__init__()
        if residual and input_dim != output_dim:
            raise ValueError("residual is only supported if input_dim == output_dim")
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.drop = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        self.residual = residual
        self.out_norm = out_norm or nn.Identity()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = x
        for layer in self.layers[:-1]:
            h = self.drop(F.relu(layer(h)))
        h = self.layers[-1](h)
        return self.out_norm(h + x if self.residual else h)


def masked_mean(seq: torch.Tensor, pad_mask: torch.Tensor) -> torch.Tensor:
    """Mean over the valid positions of ``seq[L, B, C]`` (``pad_mask[B, L]``, True = padding); at least one counted."""
    valid = (~pad_mask).to(seq.dtype).permute(1, 0)[..., None]
    return (seq * valid).sum(dim=0) / valid.sum(dim=0).clamp(min=1.0)


class DotProductScoring(nn.Module):
    """Query logits as a scaled dot product between projected decoder states and the projected, mean-pooled prompt."""

    def __init__(self, d_model: int, d_proj: int, prompt_mlp: Optional[nn.Module] = None, clamp_logits: bool = True,
                 clamp_max_val: float = 12.0):
        super().__init__()
        self.d_proj = d_proj
        self.prompt_mlp = prompt_mlp
        self.prompt_proj = nn.Linear(d_model, d_proj)
        self.hs_proj = nn.Linear(d_model, d_proj)
        self.scale = float(1.0 / math.sqrt(d_proj))
        self.clamp_logits = clamp_logits
        self.clamp_max_val = clamp_max_val

    def mean_pool_text(self, prompt, prompt_mask):
        return masked_mean(prompt, prompt_mask)

    def forward(self, hs: torch.Tensor, prompt: torch.Tensor, prompt_mask: torch.Tensor) -> torch.Tensor:
        """hs [layers, B, Q, C], prompt [L, B, C], prompt_mask [B, L] -> [layers, B, Q, 1]."""
        assert hs.dim() == 4 and prompt.dim() == 3 and prompt_mask.dim() == 2
        if self.prompt_mlp is not None:
            prompt = self.prompt_mlp(prompt)
        pooled = self.prompt_proj(masked_mean(prompt, prompt_mask))
        # [layers, B, Q, C] . [B, C]: written as a multiply + row sum in fp32, not as a batched matrix-VECTOR product --
        # rocBLAS' path for the N = 1 batched GEMM of its backward ([48, 256, 400] x [48, 400, 1]) spends 12.7 ms on the
        # host per call with the device idle (profiles/r02g_hostops.txt)
        scores = (self.hs_proj(hs).float() * pooled.float()[None, :, None, :]).sum(-1, keepdim=True) * self.scale
        if self.clamp_logits:
            scores = scores.clamp(min=-self.clamp_max_val, max=self.clamp_max_val)
        return scores


class TransformerWrapper(nn.Module):
    """Holds encoder + decoder; re-initialises every matrix below it with xavier-uniform except box heads, query
    embeddings and reference points (model_misc.py:136-146)."""

    def __init__(self, encoder, decoder, d_model: int):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder
        self.num_queries = decoder.num_queries if decoder is not None else None
        self.d_model = d_model
        for name, p in self.named_parameters():
            if p.dim() > 1 and not any(s in name for s in ("box_embed", "query_embed", "reference_points")):
                nn.init.xavier_uniform_(p)


_SINE_FREQ: Dict[Tuple, torch.Tensor] = {}


def gen_sineembed_for_position(pos: torch.Tensor, num_feats: int = 256) -> torch.Tensor:
    """[Q, B, 2|4] normalised (x, y[, w, h]) -> [Q, B, 2|4 x num_feats/2] sine code ordered (y, x, w, h).
    All coordinates go through one set of operators (the decoder calls this once per layer: operator count is host
    time there); the frequency table is built once per device."""
    assert num_feats % 2 == 0
    half = num_feats // 2
    key = (half, str(pos.device))
    freq = _SINE_FREQ.get(key)
    if freq is None or freq.is_inference():
        with torch.inference_mode(False):       # a table first built under inference_mode could not be saved for backward later
            k = torch.arange(half, dtype=torch.float32, device=pos.device)
            freq = _SINE_FREQ[key] = 10000 ** (2 * torch.div(k, 2, rounding_mode="floor") / half)
    n = pos.size(-1)
    if n == 4:
        ordered = torch.stack((pos[:, :, 1], pos[:, :, 0], pos[:, :, 2], pos[:, :, 3]), dim=2)
    elif n == 2:
        ordered = torch.stack((pos[:, :, 1], pos[:, :, 0]), dim=2)
    else:
        raise ValueError(f"Unknown pos_tensor shape(-1):{n}")
    a = (ordered * (2 * math.pi))[..., None] / freq                                  # [Q, B, n, half]
    return torch.stack((a[..., 0::2].sin(), a[..., 1::2].cos()), dim=4).flatten(2)


# ===================================================================================================== encoder ==
class TransformerEncoderLayer(nn.Module):
    """Self-attention over ``tgt``, cross-attention to ``memory``, FFN; pre- or post-norm.  Used as the fusion-encoder
    layer (tgt = image tokens, memory = prompt tokens) and as the geometry-encoder layer (tgt = prompt tokens,
    memory = image tokens)."""

    def __init__(self, activation: str, cross_attention: nn.Module, d_model: int, dim_feedforward: int, dropout: float,
                 pos_enc_at_attn: bool, pos_enc_at_cross_attn_keys: bool, pos_enc_at_cross_attn_queries: bool,
                 pre_norm: bool, self_attention: nn.Module):
        super().__init__()
        self.d_model, self.dim_feedforward, self.dropout_value = d_model, dim_feedforward, dropout
        self.self_attn = self_attention
        self.cross_attn_image = cross_attention
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        self.activation_str = activation
        self.activation = _activation(activation)
        self.pre_norm = pre_norm
        self.pos_enc_at_attn = pos_enc_at_attn
        self.pos_enc_at_cross_attn_queries = pos_enc_at_cross_attn_queries
        self.pos_enc_at_cross_attn_keys = pos_enc_at_cross_attn_keys
        self.layer_idx = None

    def _ffn(self, x):
        return self.linear2(self.dropout(self.activation(self.linear1(x))))

    def forward(self, tgt, memory, dac: bool = False, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        def self_block(x):
            qk = x + query_pos if self.pos_enc_at_attn else x
            return self.self_attn(qk, qk, value=x, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]

        def cross_block(x):
            q = x + query_pos if self.pos_enc_at_cross_attn_queries else x
            k = memory + pos if self.pos_enc_at_cross_attn_keys else memory
            return self.cross_attn_image(query=q, key=k, value=memory, attn_mask=memory_mask,
                                         key_padding_mask=memory_key_padding_mask)[0]

        if not self.pre_norm:
            tgt = self.norm1(tgt + self.dropout1(self_block(tgt)))
            tgt = self.norm2(tgt + self.dropout2(cross_block(tgt)))
            return self.norm3(tgt + self.dropout3(self._ffn(tgt)))
        rest = None
        if dac:                         # self-attention only over the first (one-to-one) half
            assert tgt.shape[0] % 2 == 0
            half = tgt.shape[0] // 2
            tgt, rest = tgt[:half], tgt[half:]
        tgt = tgt + self.dropout1(self_block(self.norm1(tgt)))
        if rest is not None:
            tgt = torch.cat((tgt, rest), dim=0)
        tgt = tgt + self.dropout2(cross_block(self.norm2(tgt)))
        return tgt + self.dropout3(self._ffn(self.norm3(tgt)))


class TransformerEncoderFusion(nn.Module):
    """Stack of :class:`TransformerEncoderLayer` run batch-first over the flattened image tokens with the prompt as
    cross-attention memory.  ``forward`` takes the sequence-first lists ``Sam3Image`` holds and returns the dict the
    decoder / segmentation head read (``memory`` [HW, B, C], ``pos_embed``, ``padding_mask`` None, level bookkeeping)."""

    def __init__(self, layer: nn.Module, num_layers: int, d_model: int, num_feature_levels: int, frozen: bool = False,
                 use_act_checkpoint: bool = False, add_pooled_text_to_img_feat: bool = True,
                 pool_text_with_mask: bool = False):
        super().__init__()
        self.layers = clones(layer, num_layers)
        self.num_layers, self.num_feature_levels = num_layers, num_feature_levels
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model)) if num_feature_levels > 1 else None
        if frozen:
            for p in self.parameters():
                p.requires_grad_(False)
        self.use_act_checkpoint = use_act_checkpoint
        for i, lay in enumerate(self.layers):
            lay.layer_idx = i
        self.add_pooled_text_to_img_feat = add_pooled_text_to_img_feat
        if add_pooled_text_to_img_feat:
            self.text_pooling_proj = nn.Linear(d_model, d_model)
        self.pool_text_with_mask = pool_text_with_mask

    def forward(self, src: List[torch.Tensor], prompt: torch.Tensor, src_key_padding_mask=None,
                src_pos: Optional[List[torch.Tensor]] = None, prompt_key_padding_mask=None, prompt_pos=None,
                feat_sizes=None, encoder_extra_kwargs: Optional[Dict] = None) -> Dict:
        assert len(src) == self.num_feature_levels, "must be equal to num_feature_levels"
        assert src_key_padding_mask is None or all(m is None for m in src_key_padding_mask), \
            "image padding masks are not part of the training path"
        device = src[0].device
        if self.add_pooled_text_to_img_feat:
            pooled = (masked_mean(prompt, prompt_key_padding_mask) if self.pool_text_with_mask else prompt.mean(dim=0))
            pooled = self.text_pooling_proj(pooled)
            src = [x + pooled[None] for x in src]
        # [HW, B, C] per level -> [B, sum HW, C]; level embedding only with several levels
        tokens, codes = [], []
        for lvl, (x, p) in enumerate(zip(src, src_pos)):
            p = p.transpose(0, 1)
            if self.level_embed is not None:
                p = p + self.level_embed[lvl].view(1, 1, -1)
            tokens.append(x.transpose(0, 1))
            codes.append(p)
        out = torch.cat(tokens, dim=1)
        query_pos = torch.cat(codes, dim=1)
        shapes = torch.tensor([tuple(s) for s in feat_sizes], dtype=torch.long, device=device)
        level_start = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        valid_ratios = torch.ones((out.shape[0], self.num_feature_levels, 2), device=device)
        memory_bf = prompt.transpose(0, 1)
        ckpt = self.training and self.use_act_checkpoint and torch.is_grad_enabled()
        extra = encoder_extra_kwargs or {}
        for layer in self.layers:
            def run(t, m, mkpm, qp, layer=layer):
                return layer(t, m, memory_key_padding_mask=mkpm, query_pos=qp, **extra)
            out = _maybe_checkpoint(ckpt, run, out, memory_bf, prompt_key_padding_mask, query_pos)
        # sequence-first and CONTIGUOUS: the decoder and the mask head feed these to Linears six times over; on a transposed
        # view every such Linear runs as a 5184-batch bmm and every add as a strided elementwise kernel
        return {"memory": out.transpose(0, 1).contiguous(), "padding_mask": None,
                "pos_embed": query_pos.transpose(0, 1).contiguous(),
                "memory_text": prompt, "level_start_index": level_start, "spatial_shapes": shapes,
                "valid_ratios": valid_ratios}


# ===================================================================================================== decoder ==
class TransformerDecoderLayer(nn.Module):
    """Query self-attention (one-to-one half + presence token under DAC), text cross-attention, image cross-attention
    with an additive per-head bias, FFN; post-norm throughout."""

    def __init__(self, activation: str, d_model: int, dim_feedforward: int, dropout: float, cross_attention: nn.Module,
                 n_heads: int, use_text_cross_attention: bool = False):
        super().__init__()
        drop = (lambda: nn.Dropout(dropout)) if dropout > 0 else nn.Identity
        self.cross_attn = cross_attention
        self.dropout1 = drop()
        self.norm1 = nn.LayerNorm(d_model)
        self.use_text_cross_attention = use_text_cross_attention
        if use_text_cross_attention:
            self.ca_text = MultiheadAttention(d_model, n_heads, dropout=dropout)
            self.catext_dropout = drop()
            self.catext_norm = nn.LayerNorm(d_model)
        self.self_attn = MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = drop()
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.activation = _activation(activation)
        self.dropout3 = drop()
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.dropout4 = drop()
        self.norm3 = nn.LayerNorm(d_model)

    def forward_ffn(self, tgt):
        with torch.amp.autocast(device_type="cuda", enabled=False):
            h = self.linear2(self.dropout3(self.activation(self.linear1(tgt))))
        return self.norm3(tgt + self.dropout4(h))

    def forward(self, tgt, tgt_query_pos, memory_text, text_attention_mask, memory, memory_pos, cross_attn_mask,
                presence_token=None, dac: bool = False, dac_use_selfatt_ln: bool = True, self_attn_mask=None,
                memory_key_padding_mask=None, mask_has_presence_row: bool = False, memory_key=None):
        """tgt / tgt_query_pos [Q, B, C]; memory [HW, B, C]; cross_attn_mask [B*heads, Q (+1), HW] additive
        (``mask_has_presence_row``: the caller already put the presence token's all-zero row in front)."""
        o2m = None
        o2o, o2o_pos = tgt, tgt_query_pos
        if dac:
            assert tgt.shape[0] % 2 == 0
            half = tgt.shape[0] // 2
            o2o, o2o_pos, o2m = tgt[:half], tgt_query_pos[:half], tgt[half:]
        if presence_token is not None:
            zero = torch.zeros_like(presence_token)
            o2o = torch.cat([presence_token, o2o], dim=0)
            o2o_pos = torch.cat([zero, o2o_pos], dim=0)
            tgt_query_pos = torch.cat([zero, tgt_query_pos], dim=0)
        qk = o2o + o2o_pos
        o2o = o2o + self.dropout2(self.self_attn(qk, qk, o2o, attn_mask=self_attn_mask, need_weights=False)[0])
        if dac:
            if not dac_use_selfatt_ln:
                o2o = self.norm2(o2o)
            tgt = torch.cat((o2o, o2m), dim=0)
            if dac_use_selfatt_ln:
                tgt = self.norm2(tgt)
        else:
            tgt = self.norm2(o2o)

        if self.use_text_cross_attention:
            h = self.ca_text(tgt + tgt_query_pos, memory_text, memory_text, key_padding_mask=text_attention_mask,
                             need_weights=False)[0]
            tgt = self.catext_norm(tgt + self.catext_dropout(h))

        if presence_token is not None and not mask_has_presence_row:
            # the presence token attends to the image without a position bias
            cross_attn_mask = torch.cat([torch.zeros_like(cross_attn_mask[:, :1, :]), cross_attn_mask], dim=1)
        # memory_key: memory + memory_pos, the same for all layers -- computed once by the decoder
        h = self.cross_attn(query=tgt + tgt_query_pos, key=memory + memory_pos if memory_key is None else memory_key,
                            value=memory,
                            attn_mask=cross_attn_mask,
                            key_padding_mask=(memory_key_padding_mask.transpose(0, 1)
                                              if memory_key_padding_mask is not None else None))[0]
        tgt = self.norm1(tgt + self.dropout1(h))
        tgt = self.forward_ffn(tgt)
        if presence_token is not None:
            return tgt[1:], tgt[:1]
        return tgt, None


class TransformerDecoder(nn.Module):
    def __init__(self, d_model: int, frozen: bool, interaction_layer, layer, num_layers: int, num_queries: int,
                 return_intermediate: bool, box_refine: bool = False, num_o2m_queries: int = 0, dac: bool = False,
                 boxRPB: str = "none", instance_query: bool = False, num_instances: int = 1,
                 dac_use_selfatt_ln: bool = True, use_act_checkpoint: bool = False, presence_token: bool = False,
                 clamp_presence_logits: bool = True, clamp_presence_logit_max_val: float = 10.0,
                 use_normed_output_consistently: bool = True, resolution: Optional[int] = None,
                 stride: Optional[int] = None):
        super().__init__()
        assert interaction_layer is None and not instance_query, "tracking / instance queries are outside the path"
        assert return_intermediate, "support return_intermediate only"
        assert box_refine, "support box refine only"
        assert boxRPB in ("none", "log", "linear", "both")
        self.d_model = d_model
        self.layers = clones(layer, num_layers)
        self.fine_layers = [None] * num_layers
        self.num_layers, self.num_queries, self.dac = num_layers, num_queries, dac
        self.num_o2m_queries = num_queries if dac else num_o2m_queries
        total_queries = num_queries if dac else num_queries + num_o2m_queries
        self.norm = nn.LayerNorm(d_model)
        self.return_intermediate, self.box_refine = return_intermediate, box_refine
        self.bbox_embed = MLP(d_model, d_model, 4, 3)
        self.query_embed = nn.Embedding(total_queries, d_model)
        self.instance_query_embed = self.instance_norm = self.instance_bbox_embed = None
        self.use_instance_query, self.num_instances = instance_query, num_instances
        self.use_normed_output_consistently = use_normed_output_consistently
        nn.init.constant_(self.bbox_embed.layers[-1].weight, 0)
        nn.init.constant_(self.bbox_embed.layers[-1].bias, 0)
        self.reference_points = nn.Embedding(num_queries, 4)
        self.boxRPB = boxRPB
        if boxRPB != "none":
            heads = self.layers[0].cross_attn.num_heads
            n_in = 4 if boxRPB == "both" else 2
            self.boxRPB_embed_x = MLP(n_in, d_model, heads, 2)
            self.boxRPB_embed_y = MLP(n_in, d_model, heads, 2)
        self._coords: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.roi_pooler = None
        if frozen:
            for p in self.parameters():
                p.requires_grad_(False)
        self.presence_token = None
        self.clamp_presence_logits = clamp_presence_logits
        self.clamp_presence_logit_max_val = clamp_presence_logit_max_val
        if presence_token:
            self.presence_token = nn.Embedding(1, d_model)
            self.presence_token_head = MLP(d_model, d_model, 1, 3)
            self.presence_token_out_norm = nn.LayerNorm(d_model)
        self.ref_point_head = MLP(2 * d_model, d_model, d_model, 2)
        self.dac_use_selfatt_ln = dac_use_selfatt_ln
        self.use_act_checkpoint = use_act_checkpoint
        nn.init.normal_(self.query_embed.weight)
        for i, lay in enumerate(self.layers):
            lay.layer_idx = i

    def _grid(self, H: int, W: int, device):
        key = (H, W, str(device))
        if key not in self._coords or self._coords[key][0].is_inference():
            with torch.inference_mode(False):   # see gen_sineembed_for_position
                self._coords[key] = (torch.arange(0, H, device=device, dtype=torch.float32) / H,
                                     torch.arange(0, W, device=device, dtype=torch.float32) / W)
        return self._coords[key]

    def _get_rpb_matrix(self, reference_boxes: torch.Tensor, feat_size, presence_row: bool = False) -> torch.Tensor:
        """Box-relative position bias: for every query box, the signed (log-scaled) offsets of each token row / column
        from the box's two edges go through a small MLP per axis; the per-head bias of token (y, x) is the sum of its
        row and column terms.  [Q, B, 4] -> [B, heads, Q (+1), H*W]; with ``presence_row`` an all-zero row for the
        presence token comes first.  The [B, heads, Q, H, W] tensor (266 MB at batch 8 in bf16) is written ONCE, by the
        broadcast add of the two small per-axis terms already laid out head-major -- the reference builds it query-major,
        permutes + copies it, and concatenates the presence row with another full copy (decoder.py:395-407, :147-151)."""
        H, W = int(feat_size[0]), int(feat_size[1])
        if self._rpb_kernel_applies(reference_boxes):
            return self._rpb_kernel(reference_boxes, H, W, presence_row)
        xyxy = box_cxcywh_to_xyxy(reference_boxes.float()).transpose(0, 1)      # [B, Q, 4]
        ys, xs = self._grid(H, W, reference_boxes.device)
        dy = ys.view(1, 1, H, 1) - xyxy[:, :, None, 1::2]                       # [B, Q, H, 2] (to y0, y1)
        dx = xs.view(1, 1, W, 1) - xyxy[:, :, None, 0::2]                       # [B, Q, W, 2] (to x0, x1)

        def log_scale(d):
            d = d * 8
            return torch.sign(d) * torch.log2(torch.abs(d) + 1.0) / math.log2(8)

        if self.boxRPB == "log":
            dx, dy = log_scale(dx), log_scale(dy)
        elif self.boxRPB == "both":
            dx, dy = torch.cat([dx, log_scale(dx)], dim=-1), torch.cat([dy, log_scale(dy)], dim=-1)
        ckpt = self.training and self.use_act_checkpoint and torch.is_grad_enabled()
        wd = self.boxRPB_embed_x.layers[0].weight.dtype        # the bias is made in the dtype of the cross-attention that takes it
        bx = _maybe_checkpoint(ckpt, self.boxRPB_embed_x, dx.to(wd))           # [B, Q, W, heads]
        by = _maybe_checkpoint(ckpt, self.boxRPB_embed_y, dy.to(wd))           # [B, Q, H, heads]
        # head-major COPIES of the two small terms: as permuted views the broadcast add below would inherit their
        # heads-innermost strides and the flatten would then copy the full bias (a 53 MB round trip per layer)
        by = by.permute(0, 3, 1, 2).contiguous()                               # [B, heads, Q, H]  (small)
        bx = bx.permute(0, 3, 1, 2).contiguous()                               # [B, heads, Q, W]
        if presence_row:
            by, bx = F.pad(by, (0, 0, 1, 0)), F.pad(bx, (0, 0, 1, 0))
        return (by.unsqueeze(-1) + bx.unsqueeze(-2)).flatten(3)                # [B, heads, Q(+1), H*W], contiguous

    def _rpb_kernel_applies(self, reference_boxes: torch.Tensor) -> bool:
        """``sam3_rpb_bias_fwd`` (include/sam3_seg_amd.h) covers the training path: boxes detached (they are, between
        decoder layers), the two per-axis MLPs frozen plain 2 -> hidden -> heads stacks in bf16 / fp32, log or plain
        offsets.  Anything else (trainable or adapted MLPs, "both", autocast, CPU) keeps the operator formulation."""
        if not reference_boxes.is_cuda or reference_boxes.requires_grad or self.boxRPB not in ("log", "linear"):
            return False
        if os.environ.get("SAM3_RPB_KERNEL", "1") == "0":         # validation aid: keep the operator chain
            return False
        if torch.is_autocast_enabled():
            return False
        for mlp in (self.boxRPB_embed_x, self.boxRPB_embed_y):
            if not (isinstance(mlp, MLP) and len(mlp.layers) == 2 and not mlp.residual
                    and isinstance(mlp.drop, nn.Identity) and isinstance(mlp.out_norm, nn.Identity)):
                return False
            l1, l2 = mlp.layers
            if type(l1) is not nn.Linear or type(l2) is not nn.Linear or l1.in_features != 2 or l2.out_features > 16:
                return False
            ps = (l1.weight, l1.bias, l2.weight, l2.bias)
            if any(p is None or p.requires_grad or p.dtype != l1.weight.dtype or not p.is_cuda for p in ps):
                return False
            if l1.weight.dtype not in (torch.bfloat16, torch.float32):
                return False
        return self.boxRPB_embed_x.layers[0].weight.dtype == self.boxRPB_embed_y.layers[0].weight.dtype

    def _rpb_kernel(self, reference_boxes: torch.Tensor, H: int, W: int, presence_row: bool) -> torch.Tensor:
        import ctypes
        from . import _ffi
        lib = _ffi.load()
        boxes = reference_boxes.detach().float().contiguous()                   # [Q, B, 4]
        Q, B = boxes.shape[:2]
        lx, ly = self.boxRPB_embed_x.layers, self.boxRPB_embed_y.layers
        heads, hidden = lx[1].out_features, lx[0].out_features
        wd = lx[0].weight.dtype
        out = torch.empty(B, heads, Q + int(presence_row), H * W, device=boxes.device, dtype=wd)
        keep = [t.detach().contiguous() for l in (lx, ly) for t in (l[0].weight, l[0].bias, l[1].weight, l[1].bias)]
        ptrs_x = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in keep[:4]])
        ptrs_y = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in keep[4:]])
        rc = lib.sam3_rpb_bias_fwd(boxes.data_ptr(), ptrs_x, ptrs_y, out.data_ptr(), B, Q, H, W, hidden, heads,
                                   int(presence_row), int(self.boxRPB == "log"), 0 if wd == torch.bfloat16 else 1,
                                   ctypes.c_void_p(torch.cuda.current_stream(boxes.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sam3_rpb_bias_fwd failed ({rc}): {lib.sam3_seg_last_error().decode()}")
        return out

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, reference_boxes=None, level_start_index=None,
                spatial_shapes=None, valid_ratios=None, memory_text=None, text_attention_mask=None,
                apply_dac: Optional[bool] = None, is_instance_prompt: bool = False,
                feat_sizes: Optional[Sequence] = None):
        """Returns (hs [layers, Q, B, C], reference boxes per layer [layers, Q, B, 4], presence logits
        [layers, 1, B] or None, presence features [1, B, C] or None)."""
        if memory_mask is not None:
            assert self.boxRPB == "none", "a memory_mask together with boxRPB is not implemented"
        assert not is_instance_prompt
        apply_dac = self.dac if apply_dac is None else apply_dac
        if apply_dac:
            assert tgt.shape[0] == self.num_queries
            tgt = tgt.repeat(2, 1, 1)
            if reference_boxes is not None:
                assert reference_boxes.shape[0] == self.num_queries
                reference_boxes = reference_boxes.repeat(2, 1, 1)
        bs = tgt.shape[1]
        wd = self.norm.weight.dtype          # layer dtype (bf16 layout); box arithmetic itself stays fp32 throughout
        if reference_boxes is None:
            reference_boxes = self.reference_points.weight.float().unsqueeze(1).repeat(2 if apply_dac else 1, bs, 1).sigmoid()
        reference_boxes = reference_boxes.float()
        ref_per_layer = [reference_boxes]
        hs_per_layer, presence_logits = [], []
        presence_feats = None
        output = tgt
        presence = self.presence_token.weight[None].expand(1, bs, -1) if self.presence_token is not None else None
        ckpt = self.training and self.use_act_checkpoint and torch.is_grad_enabled()
        ratios4 = torch.cat([valid_ratios, valid_ratios], -1)[None, :].float()   # [1, B, levels, 4]
        memory_key = memory + pos if pos is not None else memory                  # the cross-attention key of every layer
        feat_hw = None
        if self.boxRPB != "none":
            assert spatial_shapes.shape[0] == 1, "only single scale support implemented"
            if feat_sizes is not None:      # the caller's host-side (H, W): no device read, the host keeps running ahead
                assert len(feat_sizes) == 1
                feat_hw = (int(feat_sizes[0][0]), int(feat_sizes[0][1]))
            else:
                feat_hw = tuple(int(v) for v in spatial_shapes[0].tolist())       # one host read per forward, not per layer
        for idx, layer in enumerate(self.layers):
            ref_in = reference_boxes[:, :, None] * ratios4                        # [Q, B, levels, 4]
            query_pos = self.ref_point_head(gen_sineembed_for_position(ref_in[:, :, 0, :], self.d_model).to(wd))
            if self.boxRPB != "none":
                memory_mask = self._get_rpb_matrix(reference_boxes, feat_hw, presence_row=presence is not None)
                memory_mask = memory_mask.flatten(0, 1)                           # [B*heads, Q (+1), HW]

            def run(out, qpos, mtext, tmask, mem, mpos, cmask, ptok, mkey, layer=layer):
                return layer(out, qpos, mtext, tmask, mem, mpos, cmask, presence_token=ptok, dac=apply_dac,
                             dac_use_selfatt_ln=self.dac_use_selfatt_ln, self_attn_mask=tgt_mask,
                             memory_key_padding_mask=memory_key_padding_mask,
                             mask_has_presence_row=self.boxRPB != "none" and ptok is not None, memory_key=mkey)
            output, presence = _maybe_checkpoint(ckpt, run, output, query_pos, memory_text, text_attention_mask, memory,
                                                 pos, memory_mask, presence, memory_key)
            normed = self.norm(output)
            delta = self.bbox_embed(normed if self.use_normed_output_consistently else output).float()
            refined = (delta + inverse_sigmoid(reference_boxes)).sigmoid()
            reference_boxes = refined.detach()
            if idx != self.num_layers - 1:
                ref_per_layer.append(refined)
            hs_per_layer.append(normed)
            if presence is not None:
                # (the reference calls an out-of-place clamp here and drops its result, decoder.py:575-579: the logits
                # leave unclamped, and so do these)
                presence_logits.append(self.presence_token_head(self.presence_token_out_norm(presence)).squeeze(-1).float())
                presence_feats = presence.clone()
        return (torch.stack(hs_per_layer), torch.stack(ref_per_layer),
                torch.stack(presence_logits) if presence is not None else None, presence_feats)
