"""Inputs shared by make_e2e_golden.py (runs the REFERENCE classes) and tests/test_sam3_e2e.py (runs this library):
the tiny model configuration, a toy tokenizer (the reference's VETextEncoder takes any callable), the raw samples of
one batch, and the LoRA / optimiser settings.  Data and plain numbers only."""
import torch

# same structure as SAM3, every width small -- mirrors sam3_lora_amd.sam3_image.TINY_CONFIG
TINY = dict(
    vit=dict(img_size=112, pretrain_img_size=56, patch_size=14, embed_dim=64, depth=4, num_heads=2, mlp_ratio=4.625,
             drop_path_rate=0.0, window_size=4, global_att_blocks=(1, 3)),
    d_model=32, heads=2, ffn=64, dropout=0.0, enc_layers=2, dec_layers=3, num_queries=10, geo_layers=2,
    text=dict(width=48, heads=2, layers=2, context_length=8, vocab_size=64), scoring_hidden=64, roi_size=3)

WORDS = {"crack": [5, 9], "pothole": [7, 11, 13], "object": [3]}
SOT_ID, EOT_ID = 62, 63


def toy_tokenizer(texts, context_length=8):
    """<sot> ids <eot>, right-padded with 0 (the layout of the CLIP tokenizer at a 64-entry vocabulary)."""
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        ids = [SOT_ID] + WORDS[t] + [EOT_ID]
        out[i, :len(ids)] = torch.tensor(ids)
    return out


RES = 112
# (text, [(cx, cy, w, h) normalised ...]) per image; image 2 has no object
SAMPLES = [("crack", [(0.30, 0.35, 0.30, 0.20), (0.70, 0.60, 0.25, 0.40)]),
           ("pothole", [(0.50, 0.50, 0.40, 0.30)]),
           ("crack", [])]


def make_images(seed=11):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(3, RES, RES, generator=g) for _ in SAMPLES]


def box_mask(box):
    cx, cy, w, h = box
    x0, x1 = int(round((cx - w / 2) * RES)), int(round((cx + w / 2) * RES))
    y0, y1 = int(round((cy - h / 2) * RES)), int(round((cy + h / 2) * RES))
    m = torch.zeros(RES, RES, dtype=torch.bool)
    m[y0:y1, x0:x1] = True
    return m


# LoRA as the root injector applies it (vision + text + DETR targets: BASELINE configs[3]'s shape at tiny widths)
LORA = dict(rank=4, alpha=8, dropout=0.0, target_modules=["fc1", "fc2", "linear1", "linear2", "c_fc", "c_proj"],
            apply_to_vision_encoder=True, apply_to_text_encoder=True, apply_to_geometry_encoder=False,
            apply_to_detr_encoder=True, apply_to_detr_decoder=True, apply_to_mask_decoder=False)
LORA_B_SEED, LORA_B_STD = 5, 0.05
LR, WD, STEPS = 1e-3, 0.01, 4
