#!/usr/bin/env python3
"""
End-to-end golden step (SURVEY section 7-1(iv), section 8c "Model step"): the REFERENCE's own parametric classes --
ViT, Sam3DualViTDetNeck, VETextEncoder, TransformerEncoderFusion, TransformerDecoder, SequenceGeometryEncoder,
UniversalSegmentationHead, Sam3Image, collate_fn_api, BinaryHungarianMatcherV2, Sam3LossWrapper, the root LoRA
injector -- imported from /root/reference and assembled exactly as ``sam3/model_builder.py:58-324,478-512`` assembles
them, at the tiny widths of e2e_case_defs.TINY, dropout / DropPath 0, CPU fp32.  Build container only.

    python tests/golden/make_e2e_golden.py        ->  e2e_tiny.npz   (all weights stored)
    python tests/golden/make_e2e_golden.py tiny --yardstick  ->  only ref_autocast_bf16.json["tiny"] (the reference's own
                                                      autocast(bf16) deviation from its fp32 forward; see main())
    python tests/golden/make_e2e_golden.py wide   ->  e2e_wide.npz   (e2e_case_defs.WIDE: 256-wide trunk x 8 blocks at
                                                      224^2, rank-16 adapters; weights = e2e_case_defs.seeded_parameter,
                                                      so only buffers, batch and outputs are stored)

Stores: the state dict, the collated batch, the training-mode forward (every output tensor, aux outputs, matcher
indices), the eval-mode forward, and -- with root LoRA injected and B seeded non-zero -- the loss dictionary, A/B
gradients, A/B after one AdamW step and the loss curve of LR/WD/STEPS in e2e_case_defs (the loop of
``train_sam3_lora_native.py:887-943`` re-enacted around the imported classes).
"""
import contextlib
import functools
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = "/root/reference"
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn

import sam3_manifest
import e2e_case_defs as D
from loss_case_defs import CLI_LOSS_CFG


def build_reference_tiny(c=None):
    from sam3.model.decoder import TransformerDecoder, TransformerDecoderLayer
    from sam3.model.encoder import TransformerEncoderFusion, TransformerEncoderLayer
    from sam3.model.geometry_encoders import SequenceGeometryEncoder
    from sam3.model.maskformer_segmentation import PixelDecoder, UniversalSegmentationHead
    from sam3.model.model_misc import DotProductScoring, MLP, MultiheadAttentionWrapper as MHA, TransformerWrapper
    from sam3.model.necks import Sam3DualViTDetNeck
    from sam3.model.position_encoding import PositionEmbeddingSine
    from sam3.model.sam3_image import Sam3Image
    from sam3.model.text_encoder_ve import VETextEncoder
    from sam3.model.vitdet import ViT
    from sam3.model.vl_combiner import SAM3VLBackbone
    from sam3.train.matcher import BinaryHungarianMatcherV2
    c = c or D.TINY
    d, h, ffn, p = c["d_model"], c["heads"], c["ffn"], c["dropout"]
    vit = ViT(norm_layer="LayerNorm", qkv_bias=True, use_abs_pos=True, tile_abs_pos=True, rel_pos_blocks=(),
              use_rope=True, use_interp_rope=True, pretrain_use_cls_token=True, retain_cls_token=False, ln_pre=True,
              ln_post=False, return_interm_layers=False, bias_patch_embed=False, compile_mode=None, **c["vit"])
    pe = lambda: PositionEmbeddingSine(num_pos_feats=d, normalize=True, scale=None, temperature=10000)
    neck = Sam3DualViTDetNeck(position_encoding=pe(), d_model=d, scale_factors=[4.0, 2.0, 1.0, 0.5], trunk=vit,
                              add_sam2_neck=False)
    text = VETextEncoder(tokenizer=D.toy_tokenizer_32 if c["text"]["context_length"] == 32 else D.toy_tokenizer, d_model=d, **c["text"])
    backbone = SAM3VLBackbone(visual=neck, text=text, scalp=1)
    enc_layer = TransformerEncoderLayer(activation="relu", d_model=d, dim_feedforward=ffn, dropout=p,
                                        pos_enc_at_attn=True, pos_enc_at_cross_attn_keys=False,
                                        pos_enc_at_cross_attn_queries=False, pre_norm=True,
                                        self_attention=MHA(num_heads=h, dropout=p, embed_dim=d, batch_first=True),
                                        cross_attention=MHA(num_heads=h, dropout=p, embed_dim=d, batch_first=True))
    encoder = TransformerEncoderFusion(layer=enc_layer, num_layers=c["enc_layers"], d_model=d, num_feature_levels=1,
                                       frozen=False, use_act_checkpoint=True, add_pooled_text_to_img_feat=False,
                                       pool_text_with_mask=True)
    dec_layer = TransformerDecoderLayer(activation="relu", d_model=d, dim_feedforward=ffn, dropout=p,
                                        cross_attention=MHA(num_heads=h, dropout=p, embed_dim=d), n_heads=h,
                                        use_text_cross_attention=True)
    decoder = TransformerDecoder(layer=dec_layer, num_layers=c["dec_layers"], num_queries=c["num_queries"],
                                 return_intermediate=True, box_refine=True, num_o2m_queries=0, dac=True, boxRPB="log",
                                 d_model=d, frozen=False, interaction_layer=None, dac_use_selfatt_ln=True,
                                 resolution=c["vit"]["img_size"], stride=c["vit"]["patch_size"], use_act_checkpoint=True,
                                 presence_token=True)
    transformer = TransformerWrapper(encoder=encoder, decoder=decoder, d_model=d)
    scoring = DotProductScoring(d_model=d, d_proj=d, prompt_mlp=MLP(input_dim=d, hidden_dim=c["scoring_hidden"],
                                                                    output_dim=d, num_layers=2, dropout=p,
                                                                    residual=True, out_norm=nn.LayerNorm(d)))
    seg = UniversalSegmentationHead(hidden_dim=d, upsampling_stages=3, aux_masks=False, presence_head=False,
                                    dot_product_scorer=None, act_ckpt=True,
                                    cross_attend_prompt=MHA(num_heads=h, dropout=0, embed_dim=d),
                                    pixel_decoder=PixelDecoder(num_upsampling_stages=3, interpolation_mode="nearest",
                                                               hidden_dim=d, compile_mode=None))
    geo_layer = TransformerEncoderLayer(activation="relu", d_model=d, dim_feedforward=ffn, dropout=p,
                                        pos_enc_at_attn=False, pre_norm=True,
                                        self_attention=MHA(num_heads=h, dropout=p, embed_dim=d, batch_first=False),
                                        pos_enc_at_cross_attn_queries=False, pos_enc_at_cross_attn_keys=True,
                                        cross_attention=MHA(num_heads=h, dropout=p, embed_dim=d, batch_first=False))
    geometry = SequenceGeometryEncoder(pos_enc=pe(), encode_boxes_as_points=False, points_direct_project=True,
                                       points_pool=True, points_pos_enc=True, boxes_direct_project=True, boxes_pool=True,
                                       boxes_pos_enc=True, d_model=d, num_layers=c["geo_layers"], layer=geo_layer,
                                       use_act_ckpt=True, add_cls=True, add_post_encode_proj=True, roi_size=c["roi_size"])
    matcher = BinaryHungarianMatcherV2(focal=True, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, alpha=0.25, gamma=2,
                                       stable=False)
    return Sam3Image(backbone=backbone, transformer=transformer, input_geometry_encoder=geometry,
                     segmentation_head=seg, num_feature_levels=1, o2m_mask_predict=True, dot_prod_scoring=scoring,
                     use_instance_query=False, multimask_output=True, inst_interactive_predictor=None, matcher=matcher)


def reference_batch(res=None, samples=None):
    from sam3.train.data.collator import collate_fn_api
    from sam3.train.data.sam3_image_dataset import Datapoint, FindQueryLoaded, Image, InferenceMetadata, Object
    res = res or D.RES
    imgs = D.make_images_res(res)
    dps = []
    for i, ((text, boxes), img) in enumerate(zip(samples or D.SAMPLES, imgs)):
        objs = [Object(bbox=torch.tensor(b, dtype=torch.float32), area=b[2] * b[3], object_id=j, segment=D.box_mask_res(b, res))
                for j, b in enumerate(boxes)]
        q = FindQueryLoaded(query_text=text, image_id=0, object_ids_output=list(range(len(objs))), is_exhaustive=True,
                            query_processing_order=0,
                            inference_metadata=InferenceMetadata(coco_image_id=i, original_image_id=i,
                                                                 original_category_id=0, original_size=(res, res),
                                                                 object_id=-1, frame_index=-1))
        dps.append(Datapoint(find_queries=[q], images=[Image(data=img, objects=objs, size=(res, res))]))
    return collate_fn_api(dps, dict_key="input", with_seg_masks=True)["input"]


def np_(t):
    return t.detach().cpu().numpy().copy()


OUT_KEYS = ["pred_logits", "pred_boxes", "pred_boxes_xyxy", "pred_masks", "presence_logit_dec", "pred_logits_o2m",
            "pred_boxes_o2m", "pred_boxes_xyxy_o2m", "pred_masks_o2m", "semantic_seg", "queries", "encoder_hidden_states"]
AUX_KEYS = ["pred_logits", "pred_boxes", "pred_boxes_xyxy", "presence_logit_dec", "pred_logits_o2m", "pred_boxes_o2m",
            "pred_boxes_xyxy_o2m"]


def dump_outputs(res, tag, out):
    for k in OUT_KEYS:
        if k in out and out[k] is not None:
            res[f"{tag}/{k}"] = np_(out[k])
    for i, aux in enumerate(out.get("aux_outputs", [])):
        for k in AUX_KEYS:
            if k in aux:
                res[f"{tag}/aux{i}/{k}"] = np_(aux[k])
        if "indices" in aux:
            res[f"{tag}/aux{i}/indices"] = np.stack([np_(aux["indices"][0]), np_(aux["indices"][1])])
    if "indices" in out:
        res[f"{tag}/indices"] = np.stack([np_(out["indices"][0]), np_(out["indices"][1])])


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "tiny"
    CFG, RESOLUTION, LORA_CFG, LEARNING_RATE = D.CONFIGS[which]
    sam3_manifest._install_stubs()
    sam3_manifest._patch_cuda_literals()
    sys.modules["timm.layers"].trunc_normal_ = torch.nn.init.trunc_normal_
    import types
    tm = types.ModuleType("torchmetrics.functional")
    tm.f1_score = lambda *a, **k: torch.tensor(0.0)
    sys.modules["torchmetrics.functional"] = tm
    import torchmetrics
    torchmetrics.functional = tm
    tv_ops = sys.modules["torchvision.ops"]
    tv_ops.roi_align = lambda feats, boxes, size: feats.new_zeros((sum(len(b) for b in boxes), feats.shape[1], size, size))
    import torchvision
    torchvision.ops = tv_ops
    from sam3.model.model_misc import SAM3Output
    from sam3.train.loss import loss_fns as LF
    from sam3.train.loss.sam3_loss import Sam3LossWrapper
    from sam3.train.matcher import BinaryHungarianMatcherV2, BinaryOneToManyMatcher
    import lora_layers as ref_root
    assert ref_root.__file__.startswith(REF)
    LF.sigmoid_focal_loss = functools.partial(LF.sigmoid_focal_loss, triton=False)

    torch.manual_seed(0)
    model = build_reference_tiny(CFG)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        if which != "tiny":     # weights by name-seeded draws (reproduced by the test): nothing of them is stored
            for n, p in model.named_parameters():
                p.copy_(D.seeded_parameter(n, p.shape))
        else:
            te = model.backbone.language_backbone.encoder
            te.positional_embedding.copy_(torch.randn(te.positional_embedding.shape, generator=g) * 0.01)
            te.text_projection.copy_(torch.randn(te.text_projection.shape, generator=g) * te.width ** -0.5)
            for n, p in model.named_parameters():      # non-trivial norms / biases; non-zero final box-head layer
                if p.ndim == 1:
                    p.add_(torch.randn(p.shape, generator=g) * 0.1)
            model.transformer.decoder.bbox_embed.layers[-1].weight.copy_(
                torch.randn(model.transformer.decoder.bbox_embed.layers[-1].weight.shape, generator=g) * 0.05)
    res = {}
    param_names = {n for n, _ in model.named_parameters()}
    for k, v in model.state_dict().items():
        if which != "tiny" and k in param_names:
            continue                                    # reproduced from its name (seeded_parameter)
        if v.is_complex():
            res[f"sd/{k}.re"], res[f"sd/{k}.im"] = np_(v.real), np_(v.imag)
        else:
            res[f"sd/{k}"] = np_(v)
    res["sd_keys"] = np.array(list(model.state_dict().keys()))
    res["param_names"] = np.array(sorted(param_names))

    full = which == "full"
    batch = reference_batch(RESOLUTION, D.FULL_SAMPLES if full else None)
    fi, ft = batch.find_inputs[0], batch.find_targets[0]
    res["batch/img_batch"] = np_(batch.img_batch)
    res["batch/texts"] = np.array(batch.find_text_batch)
    for k in ("img_ids", "text_ids", "input_boxes", "input_boxes_mask", "input_boxes_label", "input_points",
              "input_points_mask"):
        res[f"batch/find_input/{k}"] = np_(getattr(fi, k))
    for k in ("num_boxes", "boxes", "boxes_padded", "repeated_boxes", "segments", "semantic_segments",
              "is_valid_segment", "is_exhaustive", "object_ids", "object_ids_padded"):
        res[f"batch/find_target/{k}"] = np_(getattr(ft, k))

    if not full:        # (the full-size fixture is one adapted training step: ~10 minutes of CPU as it is)
        # eval-mode forward (no DAC, no aux bookkeeping)
        model.eval()
        with torch.no_grad():
            out = model(batch)[0]
        dump_outputs(res, "eval", out)

        # training-mode forward of the un-adapted model (matching inside forward)
        model.train()
        out = model(batch)[0]
        dump_outputs(res, "train", out)
    model.train()

    # LoRA + the native CLI's loss stack and loop
    with contextlib.redirect_stdout(io.StringIO()):
        ref_root.apply_lora_to_model(model, ref_root.LoRAConfig(**LORA_CFG))
    names = [n for n, m in model.named_modules() if isinstance(m, ref_root.LoRALinear)]
    gb = torch.Generator().manual_seed(D.LORA_B_SEED)
    with torch.no_grad():
        for n, m in model.named_modules():
            if isinstance(m, ref_root.LoRALayer):
                if which != "tiny":     # reproduced from the name by the test, not stored
                    A, B = D.seeded_adapter(n, m.lora_A.shape, m.lora_B.shape)
                    m.lora_A.copy_(A), m.lora_B.copy_(B)
                    continue
                m.lora_B.copy_(torch.randn(m.lora_B.shape, generator=gb) * D.LORA_B_STD)
                res[f"lora/{n}.lora_A"], res[f"lora/{n}.lora_B"] = np_(m.lora_A), np_(m.lora_B)
    res["lora_module_names"] = np.array(names)
    cfg = CLI_LOSS_CFG
    matcher = BinaryHungarianMatcherV2(**cfg["matcher"])
    wrapper = Sam3LossWrapper(loss_fns_find=[LF.Boxes(**cfg["boxes"]), LF.IABCEMdetr(**cfg["ce"]), LF.Masks(**cfg["masks"])],
                              matcher=matcher, o2m_matcher=BinaryOneToManyMatcher(**cfg["o2m"]), **cfg["wrapper"])
    # The reference's OWN mixed-precision mode as a yardstick: the hydra trainer runs its model under
    # torch.autocast(bf16) (sam3_lora/train/native_trainer.py:956-1021).  The same adapted model, same batch, forward + loss
    # under CPU autocast(bf16) against its fp32 forward: how far bf16 moves the reference's own logits / boxes / loss.
    def fwd_loss():
        outputs = model(batch)
        targets = [model.back_convert(t) for t in batch.find_targets]
        with SAM3Output.iteration_mode(outputs, iter_mode=SAM3Output.IterMode.ALL_STEPS_PER_STAGE) as it:
            for stage_out, tg in zip(it, targets):
                for o in stage_out:
                    o["indices"] = matcher(o, tg)
                    for a in o.get("aux_outputs", []):
                        a["indices"] = matcher(a, tg)
        return outputs[0][0] if isinstance(outputs[0], list) else outputs[0], float(wrapper(outputs, targets)["core_loss"])
    yardstick_only = "--yardstick" in sys.argv      # tiny: measured in its own invocation (e2e_tiny.npz predates it and must
    if (which == "tiny" or full) and not yardstick_only:      # not move: extra forwards shift CPU reduction order by an ulp)
        raise_skip = True
    else:
        raise_skip = False
    try:
        if raise_skip:
            raise RuntimeError("skipped for the tiny fixture (run with --yardstick)")
        with torch.no_grad():
            o32, l32 = fwd_loss()
            with torch.autocast("cpu", dtype=torch.bfloat16):
                o16, l16 = fwd_loss()
        relmax = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-12))
        res["ref_autocast_bf16/pred_logits"] = np.float64(max([relmax(o16["pred_logits"], o32["pred_logits"])] + [
            relmax(a["pred_logits"], b["pred_logits"]) for a, b in zip(o16["aux_outputs"], o32["aux_outputs"])]))
        res["ref_autocast_bf16/pred_boxes"] = np.float64(max([relmax(o16["pred_boxes"], o32["pred_boxes"])] + [
            relmax(a["pred_boxes"], b["pred_boxes"]) for a, b in zip(o16["aux_outputs"], o32["aux_outputs"])]))
        res["ref_autocast_bf16/core_loss"] = np.float64(abs(l16 - l32) / abs(l32))
        # does the reference's own mixed-precision forward keep the assignment of its fp32 forward?  (final + auxiliary outputs;
        # the number of outputs whose matched (query, target) pairs differ)
        pairs = [(o16, o32)] + list(zip(o16["aux_outputs"], o32["aux_outputs"]))
        flips = sum(0 if all(torch.equal(x, y) for x, y in zip(a["indices"][:2], b["indices"][:2])) else 1 for a, b in pairs)
        res["ref_autocast_bf16/outputs_with_different_matching"] = np.float64(flips)
        res["ref_autocast_bf16/outputs_matched"] = np.float64(len(pairs))
        for k in ("presence_logit_dec", "pred_masks"):
            if k in o16 and k in o32:
                res[f"ref_autocast_bf16/{k}"] = np.float64(relmax(o16[k], o32[k]))
        if yardstick_only:
            # A/B gradients of the first step under autocast(bf16) against fp32 (the same max|d| / max|ref| per tensor the e2e
            # tests use): how far the reference's own mixed precision moves the quantity the optimizer consumes
            def grads(ctx):
                for p_ in model.parameters():
                    p_.grad = None
                with ctx:
                    outputs = model(batch)
                    targets = [model.back_convert(t) for t in batch.find_targets]
                    with SAM3Output.iteration_mode(outputs, iter_mode=SAM3Output.IterMode.ALL_STEPS_PER_STAGE) as it:
                        for stage_out, tg in zip(it, targets):
                            for o in stage_out:
                                o["indices"] = matcher(o, tg)
                                for a in o.get("aux_outputs", []):
                                    a["indices"] = matcher(a, tg)
                    total = wrapper(outputs, targets)["core_loss"]
                total.backward()
                return {n: (m.lora_A.grad.float().clone(), m.lora_B.grad.float().clone()) for n, m in model.named_modules()
                        if isinstance(m, ref_root.LoRALayer) and m.lora_A.grad is not None
                        and (which == "tiny" or full or any(w in n for w in D.WIDE_GRAD_MODULES))}
            g32 = grads(contextlib.nullcontext())
            g16 = grads(torch.autocast("cpu", dtype=torch.bfloat16))
            per = {n: max(relmax(g16[n][0], g32[n][0]), relmax(g16[n][1], g32[n][1])) for n in g32 if n in g16}
            res["ref_autocast_bf16/worst_AB_grad"] = np.float64(max(per.values()))
            res["ref_autocast_bf16/median_AB_grad"] = np.float64(float(np.median(list(per.values()))))
            print("A/B gradients under autocast(bf16) vs fp32: worst %.3e median %.3e over %d adapters" % (
                res["ref_autocast_bf16/worst_AB_grad"], res["ref_autocast_bf16/median_AB_grad"], len(per)))
        print("reference under autocast(bf16) vs its fp32: logits %.3e boxes %.3e loss %.3e" % (
            res["ref_autocast_bf16/pred_logits"], res["ref_autocast_bf16/pred_boxes"], res["ref_autocast_bf16/core_loss"]))
        import json
        yp = os.path.join(HERE, "ref_autocast_bf16.json")
        yd = json.load(open(yp)) if os.path.exists(yp) else {}
        yd.setdefault(which, {}).update({k.split("/")[1]: float(v) for k, v in res.items() if k.startswith("ref_autocast_bf16/")})
        json.dump(yd, open(yp, "w"), indent=1, sort_keys=True)
        if yardstick_only:
            return
    except Exception as e:      # CPU autocast coverage is torch's business; the fixture works without the yardstick
        print("autocast yardstick unavailable:", type(e).__name__, str(e)[:200])
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=LEARNING_RATE, weight_decay=D.WD)
    losses = []
    import time as _time
    for step in range(1 if full else D.STEPS):
        _t0 = _time.time()
        outputs = model(batch)
        targets = [model.back_convert(t) for t in batch.find_targets]
        with SAM3Output.iteration_mode(outputs, iter_mode=SAM3Output.IterMode.ALL_STEPS_PER_STAGE) as it:
            for stage_out, tg in zip(it, targets):
                for o in stage_out:
                    o["indices"] = matcher(o, tg)
                    for a in o.get("aux_outputs", []):
                        a["indices"] = matcher(a, tg)
        loss_dict = wrapper(outputs, targets)
        total = loss_dict["core_loss"]
        opt.zero_grad()
        total.backward()
        if step == 0:
            dump_outputs(res, "lora", outputs[0][0] if isinstance(outputs[0], list) else outputs[0])
            for k, v in loss_dict.items():
                res[f"loss/{k}"] = np.float64(float(v))
            for n, m in model.named_modules():
                if isinstance(m, ref_root.LoRALayer) and full:
                    if any(w in n for w in D.FULL_GRAD_MODULES):
                        res[f"gA/{n}"], res[f"gB/{n}"] = np_(m.lora_A.grad), np_(m.lora_B.grad)
                    else:       # a strided sample of every other adapter's gradients + their maxima (the test's error scale)
                        ga, gb = m.lora_A.grad.flatten(), m.lora_B.grad.flatten()
                        res[f"gAs/{n}"], res[f"gBs/{n}"] = np_(ga[::D.FULL_GRAD_SAMPLE]), np_(gb[::D.FULL_GRAD_SAMPLE])
                        res[f"gAmax/{n}"], res[f"gBmax/{n}"] = np.float64(ga.abs().max()), np.float64(gb.abs().max())
                elif isinstance(m, ref_root.LoRALayer) and (which == "tiny" or any(w in n for w in D.WIDE_GRAD_MODULES)):
                    res[f"gA/{n}"], res[f"gB/{n}"] = np_(m.lora_A.grad), np_(m.lora_B.grad)
            print("step %d: %.1f s, core_loss %.6f" % (step, _time.time() - _t0, float(total)), flush=True)
        opt.step()
        if step == 0:
            for n, m in model.named_modules():
                if isinstance(m, ref_root.LoRALayer) and which == "tiny":
                    res[f"A1/{n}"], res[f"B1/{n}"] = np_(m.lora_A), np_(m.lora_B)
        losses.append(total.item())
    res["losses"] = np.array(losses, np.float64)
    if which != "tiny":         # the big per-query mask tensors are pinned by the tiny fixture; keep this one small
        for k in [k for k in res if k.endswith(("pred_masks", "pred_masks_o2m", "encoder_hidden_states")) and not k.startswith("ref_autocast_bf16/")]:
            res[k] = res[k][:, :4] if res[k].ndim == 4 else res[k][::8]
    if full:                    # 1008^2 image and masks: the test rebuilds the image from its seed; masks at 4 queries, every 4th pixel
        del res["batch/img_batch"]
        for k in [k for k in res if k.endswith(("pred_masks", "pred_masks_o2m", "semantic_seg"))]:
            res[k] = res[k][..., ::4, ::4]
        for k in ("batch/find_target/segments", "batch/find_target/semantic_segments"):
            if k in res:
                res[k] = np.packbits(res[k].astype(bool), axis=-1)
    out_path = os.path.join(HERE, f"e2e_{which}.npz")
    np.savez_compressed(out_path, **res)
    print("adapted:", len(names), "modules; losses:", " ".join(f"{l:.6f}" for l in losses))
    print("arrays:", len(res), "; bytes:", os.path.getsize(out_path))


if __name__ == "__main__":
    main()
