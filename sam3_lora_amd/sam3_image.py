"""
The SAM3 image model around the adapted ViT trunk, and its builder (section 3.3).

Restates ``sam3/model/sam3_image.py:35-597`` (``Sam3Image``: prompt encoding, fusion encoder, decoder, score / box
heads, mask head, training-time matching, ``back_convert``), ``SAM3Output`` (``sam3/model/model_misc.py:265-428``) and
``build_sam3_image_model`` (``sam3/model_builder.py:557-637`` with the component factories :58-324, :478-512) for the
text-prompted detection / segmentation training step that ``train_sam3_lora_native.py:887-943`` runs.  Everything
outside the ViT trunk's adapted Linears is stock PyTorch-ROCm.

What is kept exactly: module tree and parameter names (a reference checkpoint's ``detector.*`` tensors load
``strict=True``; both LoRA injectors see the reference's names -- pinned against ``tests/golden/sam3_state_keys.json``
and ``sam3_linears.json``), the output dictionary (``pred_logits``, ``pred_boxes``, ``pred_boxes_xyxy``,
``pred_masks``, ``presence_logit_dec``, the ``*_o2m`` twins, ``semantic_seg``, ``queries``, ``aux_outputs``,
``indices``, ``encoder_hidden_states``, ``prev_encoder_out``) and the arithmetic (pinned against a tiny-configuration
end-to-end fixture produced by the reference's own classes, ``tests/golden/e2e_tiny.npz``).

What differs on purpose:
  * no checkpoint download: weights come from ``checkpoint_path`` or a seeded random initialisation (there is no
    network on the target machines; SURVEY F8 for the two tensors the reference leaves uninitialised);
  * ``device`` is honoured everywhere (the reference hard-codes ``"cuda"`` in two constructors);
  * matching inside ``forward`` is optional (``match_in_forward``): the native training loop matches again right
    after the forward with an identical matcher (``train_sam3_lora_native.py:914-927``) and overwrites the indices, so
    the trainer builds the model with it off and the twelve host assignments per step become six (SURVEY f-2);
  * activation checkpointing of the DETR / text / mask-head layers is a setting (``act_checkpoint``), not a constant.
"""
from __future__ import annotations

import os
from contextlib import AbstractContextManager
from enum import Enum, auto
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from .matcher import BinaryHungarianMatcherV2, box_cxcywh_to_xyxy
from .sam3_detr import (MLP, DotProductScoring, MultiheadAttention, TransformerDecoder, TransformerDecoderLayer,
                        TransformerEncoderFusion, TransformerEncoderLayer, TransformerWrapper, inverse_sigmoid)
from .sam3_geometry import Prompt, SequenceGeometryEncoder
from .sam3_neck import PositionEmbeddingSine, SAM3VLBackbone, Sam3DualViTDetNeck
from .sam3_seghead import PixelDecoder, UniversalSegmentationHead
from .sam3_text import SimpleTokenizer, VETextEncoder
from .vit import ViT

__all__ = ["SAM3Output", "Sam3Image", "build_sam3_image_model", "SAM3_CONFIG", "TINY_CONFIG", "load_detector_checkpoint"]


class SAM3Output(list):
    """``[[step dict, ...] per stage]`` with a switchable iteration view (all steps per stage / last step per stage /
    flattened)."""

    class IterMode(Enum):
        ALL_STEPS_PER_STAGE = auto()
        LAST_STEP_PER_STAGE = auto()
        FLATTENED = auto()

    def __init__(self, output: Optional[List[List[Dict]]] = None, iter_mode: "SAM3Output.IterMode" = None,
                 loss_stages: Optional[List[int]] = None):
        super().__init__()
        if output is not None:
            assert isinstance(output, list) and len(output) > 0 and isinstance(output[0], list), \
                "Expected output to be a list of lists"
        self.output = output if output is not None else []
        self.iter_mode = iter_mode or SAM3Output.IterMode.ALL_STEPS_PER_STAGE
        assert isinstance(self.iter_mode, SAM3Output.IterMode)
        self.loss_stages = loss_stages

    def _view(self) -> list:
        M = SAM3Output.IterMode
        if self.iter_mode is M.ALL_STEPS_PER_STAGE:
            return self.output
        if self.iter_mode is M.LAST_STEP_PER_STAGE:
            return [stage[-1] for stage in self.output]
        return [step for stage in self.output for step in stage]

    def __iter__(self):
        return iter(self._view())

    def __len__(self):
        return len(self._view())

    def __getitem__(self, index):
        assert isinstance(index, int), f"index should be an integer. Got {type(index)}"
        return self._view()[index]

    def append(self, item: list):
        assert isinstance(item, list), f"Only list items are supported. Got {type(item)}"
        self.output.append(item)

    def __repr__(self):
        return repr(self.output)

    class _IterationMode(AbstractContextManager):
        def __init__(self, model_output: "SAM3Output", iter_mode: "SAM3Output.IterMode"):
            self._out, self._saved, self._new = model_output, model_output.iter_mode, iter_mode

        def __enter__(self) -> "SAM3Output":
            self._out.iter_mode = self._new
            return self._out

        def __exit__(self, exc_type, exc_value, traceback):
            self._out.iter_mode = self._saved
            return None

    @staticmethod
    def iteration_mode(model_output: "SAM3Output", iter_mode: "SAM3Output.IterMode") -> "_IterationMode":
        return SAM3Output._IterationMode(model_output, iter_mode)


def _spread(out: Dict, name: str, per_layer: torch.Tensor, with_aux: bool) -> None:
    """Last decoder layer under ``name``; earlier layers under ``aux_outputs[i][name]`` when ``with_aux``."""
    out[name] = per_layer[-1]
    if with_aux:
        aux = out.setdefault("aux_outputs", [{} for _ in range(len(per_layer) - 1)])
        assert len(aux) == len(per_layer) - 1
        for slot, value in zip(aux, per_layer[:-1]):
            slot[name] = value


class Sam3Image(nn.Module):
    TEXT_ID_FOR_TEXT, TEXT_ID_FOR_VISUAL, TEXT_ID_FOR_GEOMETRIC = 0, 1, 2

    def __init__(self, backbone: SAM3VLBackbone, transformer: TransformerWrapper, input_geometry_encoder: nn.Module,
                 segmentation_head: Optional[nn.Module] = None, num_feature_levels: int = 1,
                 o2m_mask_predict: bool = True, dot_prod_scoring: Optional[nn.Module] = None,
                 use_instance_query: bool = True, multimask_output: bool = True,
                 use_act_checkpoint_seg_head: bool = True, matcher=None, supervise_joint_box_scores: bool = False,
                 detach_presence_in_joint_score: bool = False, inst_interactive_predictor=None,
                 match_in_forward: bool = True, **unused):
        super().__init__()
        assert dot_prod_scoring is not None
        assert inst_interactive_predictor is None, "the interactive (SAM-1 task) predictor is outside the training path"
        self.backbone = backbone
        self.geometry_encoder = input_geometry_encoder
        self.transformer = transformer
        self.hidden_dim = transformer.d_model
        self.num_feature_levels = num_feature_levels
        self.segmentation_head = segmentation_head
        self.o2m_mask_predict = o2m_mask_predict
        self.dot_prod_scoring = dot_prod_scoring
        self.instance_dot_prod_scoring = None
        self.use_dot_prod_scoring = True
        self.use_act_checkpoint_seg_head = use_act_checkpoint_seg_head
        self.matcher = matcher
        self.match_in_forward = match_in_forward
        # set by the training loop to ITS matcher (trainer.match_all_steps collects): the matching of the final + auxiliary
        # outputs then starts right after the decoder, before the mask head is queued.  Not a sub-module (plain attribute).
        object.__setattr__(self, "prefetch_matcher", None)
        self.num_interactive_steps_val = 0
        self.supervise_joint_box_scores = supervise_joint_box_scores
        self.detach_presence_in_joint_score = detach_presence_in_joint_score
        dec = transformer.decoder
        assert dec.num_o2m_queries == (dec.num_queries if dec.dac else 0)
        self.dac = dec.dac
        self.use_instance_query, self.multimask_output = use_instance_query, multimask_output
        self.inst_interactive_predictor = None

    @property
    def device(self):
        return next(self.parameters()).device

    def set_prefetch_matcher(self, matcher) -> None:
        """Start the training loop's matching inside ``forward`` right after the decoder (``matcher.launch``, or the loss
        wrapper's ``launch_matching`` which covers the one-to-many indices too); the loop's ``match_all_steps`` -- called
        with the same object -- collects.  ``None`` switches it off."""
        object.__setattr__(self, "prefetch_matcher", matcher)

    # ----------------------------------------------------------------------------------------------- pieces --
    def _get_img_feats(self, backbone_out: Dict, img_ids: torch.Tensor):
        """Per-prompt image tokens: ``([HW, N, C] per level, position codes likewise, (H, W) per level)``."""
        assert "backbone_fpn" in backbone_out, "image features are computed once per batch in forward()"
        cached = backbone_out.get("_token_cache")
        if cached is not None and cached[0] is img_ids:
            return (backbone_out,) + cached[1]
        feats = backbone_out["backbone_fpn"][-self.num_feature_levels:]
        codes = backbone_out["vision_pos_enc"][-self.num_feature_levels:]
        sizes = [c.shape[-2:] for c in codes]
        # contiguous [HW, N, C]: these feed LayerNorm / Linear / attention of the geometry and fusion encoders; as
        # permuted views of [N, C, HW] every one of those would run on its strided slow path.  Built once per forward.
        pick = (lambda t: t) if backbone_out.get("_img_ids_are_arange") else (lambda t: t[img_ids])
        tokens = [pick(f).flatten(2).permute(2, 0, 1).contiguous() for f in feats]
        token_pos = [pick(c).flatten(2).permute(2, 0, 1).contiguous() for c in codes]
        backbone_out["_token_cache"] = (img_ids, (tokens, token_pos, sizes))
        return backbone_out, tokens, token_pos, sizes

    def _encode_prompt(self, backbone_out, find_input, geometric_prompt, encode_text: bool = True):
        text = backbone_out["language_features"][:, find_input.text_ids]
        text_mask = backbone_out["language_mask"][find_input.text_ids]
        backbone_out, tokens, token_pos, sizes = self._get_img_feats(backbone_out, find_input.img_ids)
        geo, geo_mask = self.geometry_encoder(geo_prompt=geometric_prompt, img_feats=tokens, img_sizes=sizes,
                                              img_pos_embeds=token_pos)
        if encode_text:
            return torch.cat([text, geo], dim=0), torch.cat([text_mask, geo_mask], dim=1), backbone_out
        return geo, geo_mask, backbone_out

    def _run_encoder(self, backbone_out, find_input, prompt, prompt_mask):
        backbone_out, tokens, token_pos, sizes = self._get_img_feats(backbone_out, find_input.img_ids)
        mem = self.transformer.encoder(src=list(tokens), src_key_padding_mask=None, src_pos=list(token_pos),
                                       prompt=prompt, prompt_pos=torch.zeros_like(prompt),
                                       prompt_key_padding_mask=prompt_mask, feat_sizes=sizes)
        encoder_out = {"encoder_hidden_states": mem["memory"], "pos_embed": mem["pos_embed"],
                       "padding_mask": mem["padding_mask"], "level_start_index": mem["level_start_index"],
                       "spatial_shapes": mem["spatial_shapes"], "valid_ratios": mem["valid_ratios"],
                       "vis_feat_sizes": sizes, "prompt_before_enc": prompt,
                       "prompt_after_enc": mem.get("memory_text", prompt), "prompt_mask": prompt_mask}
        return backbone_out, encoder_out

    def _run_decoder(self, out: Dict, encoder_out: Dict, prompt, prompt_mask):
        memory = out["encoder_hidden_states"]
        dec = self.transformer.decoder
        tgt = dec.query_embed.weight.unsqueeze(1).repeat(1, memory.shape[1], 1)
        hs, ref_boxes, presence, presence_feats = dec(
            tgt=tgt, memory=memory, memory_key_padding_mask=encoder_out["padding_mask"], pos=encoder_out["pos_embed"],
            reference_boxes=None, level_start_index=encoder_out["level_start_index"],
            spatial_shapes=encoder_out["spatial_shapes"], valid_ratios=encoder_out["valid_ratios"], tgt_mask=None,
            memory_text=prompt, text_attention_mask=prompt_mask, apply_dac=dec.dac and self.training,
            feat_sizes=encoder_out.get("vis_feat_sizes"))
        hs = hs.transpose(1, 2)                         # [layers, B, Q, C]
        ref_boxes = ref_boxes.transpose(1, 2)
        if presence is not None:
            presence = presence.transpose(1, 2)         # [layers, B, 1]
        out["presence_feats"] = presence_feats
        self._update_scores_and_boxes(out, hs, ref_boxes, prompt, prompt_mask, dec_presence_out=presence)
        return out, hs

    def _update_scores_and_boxes(self, out, hs, reference_boxes, prompt, prompt_mask, dec_presence_out=None):
        dac = self.transformer.decoder.dac and self.training
        n_o2o = hs.size(2) // 2 if dac else hs.size(2)
        n_o2m = hs.size(2) - n_o2o
        aux = self.training
        out["queries"] = hs[-1][:, :n_o2o]
        logits = self.dot_prod_scoring(hs, prompt, prompt_mask)
        # scores and boxes leave in fp32 whatever the layer dtype: they feed the matcher's cost and the box losses
        boxes = (inverse_sigmoid(reference_boxes.float()) + self.transformer.decoder.bbox_embed(hs).float()).sigmoid()
        boxes_xyxy = box_cxcywh_to_xyxy(boxes)
        if dec_presence_out is not None:
            _spread(out, "presence_logit_dec", dec_presence_out, aux)
        if self.supervise_joint_box_scores:
            assert dec_presence_out is not None
            p = dec_presence_out.clone().sigmoid()
            if self.detach_presence_in_joint_score:
                p = p.detach()
            logits = inverse_sigmoid(logits.sigmoid() * p.unsqueeze(2)).clamp(min=-10.0, max=10.0)
        _spread(out, "pred_logits", logits[:, :, :n_o2o], aux)
        _spread(out, "pred_boxes", boxes[:, :, :n_o2o], aux)
        _spread(out, "pred_boxes_xyxy", boxes_xyxy[:, :, :n_o2o], aux)
        if n_o2m > 0 and self.training:
            _spread(out, "pred_logits_o2m", logits[:, :, n_o2o:], aux)
            _spread(out, "pred_boxes_o2m", boxes[:, :, n_o2o:], aux)
            _spread(out, "pred_boxes_xyxy_o2m", boxes_xyxy[:, :, n_o2o:], aux)

    def _run_segmentation_heads(self, out, backbone_out, img_ids, encoder_hidden_states, prompt, prompt_mask, hs):
        if self.segmentation_head is None:
            backbone_out.pop("backbone_fpn", None)
            return
        dac = self.transformer.decoder.dac and self.training
        n_o2o = hs.size(2) // 2 if dac else hs.size(2)
        n_o2m = hs.size(2) - n_o2o
        queries = hs if self.o2m_mask_predict else hs[:, :, :n_o2o]
        head = self.segmentation_head

        def run(feats, q, ids, enc, p, pm):
            return head(backbone_feats=feats, obj_queries=q, image_ids=ids, encoder_hidden_states=enc, prompt=p,
                        prompt_mask=pm)
        if backbone_out.get("_img_ids_are_arange"):
            img_ids = None                       # the head's per-prompt gather of the FPN levels is the identity
        args = (backbone_out["backbone_fpn"], queries, img_ids, encoder_hidden_states, prompt, prompt_mask)
        if self.training and self.use_act_checkpoint_seg_head and torch.is_grad_enabled():
            seg = checkpoint(run, *args, use_reentrant=False)
        else:
            seg = run(*args)
        for k, v in seg.items():
            if k in head.instance_keys:
                out[k] = v[:, :n_o2o]
                if self.o2m_mask_predict and n_o2m > 0:
                    out[f"{k}_o2m"] = v[:, n_o2o:]
            else:
                out[k] = v

    # ---------------------------------------------------------------------------------------------- forward --
    def forward_grounding(self, backbone_out, find_input, find_target, geometric_prompt: Prompt) -> Dict:
        prompt, prompt_mask, backbone_out = self._encode_prompt(backbone_out, find_input, geometric_prompt)
        backbone_out, encoder_out = self._run_encoder(backbone_out, find_input, prompt, prompt_mask)
        out = {"encoder_hidden_states": encoder_out["encoder_hidden_states"],
               "prev_encoder_out": {"encoder_out": encoder_out, "backbone_out": backbone_out}}
        out, hs = self._run_decoder(out, encoder_out, prompt, prompt_mask)
        if self.training and self.prefetch_matcher is not None and find_target is not None:
            # scores and boxes are final here: start the matcher's cost + device->host copy now, so that the host-side
            # assignment later overlaps with the mask head's device work instead of following it
            m = self.prefetch_matcher
            tg = self.back_convert(find_target)
            out["_match_handle"] = (m, m.launch_matching(out, tg) if hasattr(m, "launch_matching")
                                    else m.launch([out] + list(out.get("aux_outputs", ())), tg))
        self._run_segmentation_heads(out, backbone_out, find_input.img_ids, out["encoder_hidden_states"], prompt,
                                     prompt_mask, hs)
        if self.training and self.match_in_forward and self.matcher is not None:
            self._compute_matching(out, self.back_convert(find_target))
        return out

    def forward(self, input) -> SAM3Output:
        device = self.device
        backbone_out = {"img_batch_all_stages": input.img_batch}
        # images arrive fp32 from the collator; the trunk computes in its weights' dtype (bf16 in the MI355X layout)
        images = input.img_batch.to(next(self.backbone.vision_backbone.parameters()).dtype)
        backbone_out.update(self.backbone.forward_image(images))
        assert len(input.find_inputs) == 1, "image training has exactly one find stage"
        backbone_out.update(self.backbone.forward_text(input.find_text_batch, device=device))
        stages = SAM3Output(iter_mode=SAM3Output.IterMode.LAST_STEP_PER_STAGE)
        find_input, find_target = input.find_inputs[0], input.find_targets[0]
        backbone_out["_img_ids_are_arange"] = (bool(getattr(find_input, "img_ids_are_arange", False))
                                               and len(find_input.img_ids) == images.shape[0])
        if find_input.input_points is not None and find_input.input_points.numel() > 0:
            print("Warning: Point prompts are ignored in PCS.")
        prompt = Prompt(box_embeddings=find_input.input_boxes, box_mask=find_input.input_boxes_mask,
                        box_labels=find_input.input_boxes_label)
        stages.append([self.forward_grounding(backbone_out, find_input, find_target, prompt.clone())])
        return stages

    def _compute_matching(self, out, targets):
        out["indices"] = self.matcher(out, targets)
        for aux in out.get("aux_outputs", []):
            aux["indices"] = self.matcher(aux, targets)

    def back_convert(self, targets) -> Dict:
        boxes = targets.boxes.view(-1, 4)
        out = {"boxes": boxes, "boxes_xyxy": box_cxcywh_to_xyxy(boxes), "boxes_padded": targets.boxes_padded,
               "positive_map": targets.boxes.new_ones(len(targets.boxes), 1), "num_boxes": targets.num_boxes,
               "masks": targets.segments, "semantic_masks": targets.semantic_segments,
               "is_valid_mask": targets.is_valid_segment, "is_exhaustive": targets.is_exhaustive,
               "object_ids_packed": targets.object_ids, "object_ids_padded": targets.object_ids_padded}
        host_counts = getattr(targets, "num_boxes_host", None)     # the collator's host copy: matching needs the counts
        if host_counts is not None and len(host_counts) == len(targets.num_boxes):     # on the host, without a device read
            out["num_boxes_host"] = tuple(host_counts)
        return out


# ===================================================================================================== builder ==
SAM3_CONFIG = dict(
    vit=dict(img_size=1008, pretrain_img_size=336, patch_size=14, embed_dim=1024, depth=32, num_heads=16,
             mlp_ratio=4.625, drop_path_rate=0.1, window_size=24, global_att_blocks=(7, 15, 23, 31)),
    d_model=256, heads=8, ffn=2048, dropout=0.1, enc_layers=6, dec_layers=6, num_queries=200, geo_layers=3,
    text=dict(width=1024, heads=16, layers=24, context_length=32, vocab_size=49408), scoring_hidden=2048, roi_size=7)

# the configuration of the end-to-end parity fixture: same structure, every width small (tests/golden/make_e2e_golden.py)
TINY_CONFIG = dict(
    vit=dict(img_size=112, pretrain_img_size=56, patch_size=14, embed_dim=64, depth=4, num_heads=2, mlp_ratio=4.625,
             drop_path_rate=0.0, window_size=4, global_att_blocks=(1, 3)),
    d_model=32, heads=2, ffn=64, dropout=0.0, enc_layers=2, dec_layers=3, num_queries=10, geo_layers=2,
    text=dict(width=48, heads=2, layers=2, context_length=8, vocab_size=64), scoring_hidden=64, roi_size=3)


def build_sam3_image_model(bpe_path: Optional[str] = None, device=None, eval_mode: bool = True,
                           checkpoint_path: Optional[str] = None, load_from_HF: bool = False,
                           enable_segmentation: bool = True, enable_inst_interactivity: bool = False,
                           compile: bool = False, config: Optional[Dict] = None, tokenizer=None,
                           match_in_forward: bool = True, act_checkpoint: bool = True, seed: Optional[int] = None):
    """Signature of ``sam3.model_builder.build_sam3_image_model`` plus ``config`` (widths; default = the one SAM3 size),
    ``tokenizer`` (callable replacing the BPE tokenizer), ``match_in_forward``, ``act_checkpoint`` and ``seed``.
    ``load_from_HF`` / ``compile`` / ``enable_inst_interactivity`` must stay off: no network, no tracing compiler, and
    the interactive predictor is outside the training path."""
    if load_from_HF and checkpoint_path is None:
        raise RuntimeError("load_from_HF: there is no network on the target machines; pass checkpoint_path=<sam3.pt> "
                           "or train from the seeded random initialisation")
    if compile or enable_inst_interactivity:
        raise NotImplementedError("compile / enable_inst_interactivity are not part of the LoRA training path")
    cfg = dict(SAM3_CONFIG if config is None else config)
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if seed is not None:
        torch.manual_seed(seed)
    d, heads, ffn, p = cfg["d_model"], cfg["heads"], cfg["ffn"], cfg["dropout"]
    res, stride = cfg["vit"]["img_size"], cfg["vit"]["patch_size"]

    def attn(batch_first=False, dropout=p):
        return MultiheadAttention(num_heads=heads, dropout=dropout, embed_dim=d, batch_first=batch_first)

    trunk = ViT(**cfg["vit"], use_act_checkpoint=act_checkpoint)
    neck = Sam3DualViTDetNeck(trunk=trunk, position_encoding=PositionEmbeddingSine(num_pos_feats=d, normalize=True),
                              d_model=d, scale_factors=[4.0, 2.0, 1.0, 0.5])
    if tokenizer is None:
        if bpe_path is None:
            bpe_path = os.environ.get("SAM3_BPE_PATH") or os.path.join("sam3", "assets", "bpe_simple_vocab_16e6.txt.gz")
        tokenizer = SimpleTokenizer(bpe_path=bpe_path)
    text = VETextEncoder(tokenizer=tokenizer, d_model=d, use_act_checkpoint=act_checkpoint, **cfg["text"])
    backbone = SAM3VLBackbone(visual=neck, text=text, scalp=1)

    enc_layer = TransformerEncoderLayer(activation="relu", d_model=d, dim_feedforward=ffn, dropout=p,
                                        pos_enc_at_attn=True, pos_enc_at_cross_attn_keys=False,
                                        pos_enc_at_cross_attn_queries=False, pre_norm=True,
                                        self_attention=attn(batch_first=True), cross_attention=attn(batch_first=True))
    encoder = TransformerEncoderFusion(layer=enc_layer, num_layers=cfg["enc_layers"], d_model=d, num_feature_levels=1,
                                       frozen=False, use_act_checkpoint=act_checkpoint,
                                       add_pooled_text_to_img_feat=False, pool_text_with_mask=True)
    dec_layer = TransformerDecoderLayer(activation="relu", d_model=d, dim_feedforward=ffn, dropout=p,
                                        cross_attention=attn(), n_heads=heads, use_text_cross_attention=True)
    decoder = TransformerDecoder(layer=dec_layer, num_layers=cfg["dec_layers"], num_queries=cfg["num_queries"],
                                 return_intermediate=True, box_refine=True, num_o2m_queries=0, dac=True, boxRPB="log",
                                 d_model=d, frozen=False, interaction_layer=None, dac_use_selfatt_ln=True,
                                 resolution=res, stride=stride, use_act_checkpoint=act_checkpoint, presence_token=True)
    transformer = TransformerWrapper(encoder=encoder, decoder=decoder, d_model=d)
    scoring = DotProductScoring(d_model=d, d_proj=d, prompt_mlp=MLP(d, cfg["scoring_hidden"], d, 2, dropout=p,
                                                                    residual=True, out_norm=nn.LayerNorm(d)))
    seg_head = None
    if enable_segmentation:
        seg_head = UniversalSegmentationHead(hidden_dim=d, upsampling_stages=3, aux_masks=False, presence_head=False,
                                             dot_product_scorer=None, act_ckpt=act_checkpoint,
                                             cross_attend_prompt=attn(dropout=0),
                                             pixel_decoder=PixelDecoder(d, 3, interpolation_mode="nearest"))
    geo_layer = TransformerEncoderLayer(activation="relu", d_model=d, dim_feedforward=ffn, dropout=p,
                                        pos_enc_at_attn=False, pre_norm=True, self_attention=attn(),
                                        pos_enc_at_cross_attn_queries=False, pos_enc_at_cross_attn_keys=True,
                                        cross_attention=attn())
    geometry = SequenceGeometryEncoder(pos_enc=PositionEmbeddingSine(num_pos_feats=d, normalize=True),
                                       encode_boxes_as_points=False, points_direct_project=True, points_pool=True,
                                       points_pos_enc=True, boxes_direct_project=True, boxes_pool=True,
                                       boxes_pos_enc=True, d_model=d, num_layers=cfg["geo_layers"], layer=geo_layer,
                                       use_act_ckpt=act_checkpoint, add_cls=True, add_post_encode_proj=True,
                                       roi_size=cfg["roi_size"])
    matcher = None if eval_mode else BinaryHungarianMatcherV2(focal=True, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0,
                                                              alpha=0.25, gamma=2, stable=False)
    model = Sam3Image(backbone=backbone, transformer=transformer, input_geometry_encoder=geometry,
                      segmentation_head=seg_head, num_feature_levels=1, o2m_mask_predict=True, dot_prod_scoring=scoring,
                      use_instance_query=False, multimask_output=True, matcher=matcher,
                      use_act_checkpoint_seg_head=act_checkpoint, match_in_forward=match_in_forward)
    if checkpoint_path is not None:
        load_detector_checkpoint(model, checkpoint_path)
    model = model.to(device)
    if eval_mode:
        model.eval()
    return model


def load_detector_checkpoint(model: nn.Module, checkpoint_path: str) -> None:
    """``sam3.pt`` layout (model_builder.py:523-545): tensors of the image model live under ``detector.``."""
    blob = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    if "model" in blob and isinstance(blob["model"], dict):
        blob = blob["model"]
    state = {k.replace("detector.", ""): v for k, v in blob.items() if "detector" in k} or blob
    missing, _ = model.load_state_dict(state, strict=False)
    if missing:
        print(f"loaded {checkpoint_path} and found missing and/or unexpected keys:\nmissing_keys={missing}")
