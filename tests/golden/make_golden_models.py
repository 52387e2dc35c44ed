"""Toy model + injector configurations shared by make_golden.py (which runs the REFERENCE
injectors on them) and the tests (which run ours).  Our own definitions; module names mimic the
reference model's component prefixes."""
import torch.nn as nn


# ---- toy models (our own definitions; only the reference INJECTORS are exercised) -------
class ToyAttn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.q_proj = nn.Linear(d, d)
        self.k_proj = nn.Linear(d, d)
        self.v_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)


class ToyBlock(nn.Module):
    def __init__(self, d, h):
        super().__init__()
        self.self_attn = ToyAttn(d)
        self.cross_attn_image = nn.MultiheadAttention(d, 4)
        self.linear1 = nn.Linear(d, h)
        self.linear2 = nn.Linear(h, d)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(d, h)
        self.mlp.fc2 = nn.Linear(h, d)


class ToySam(nn.Module):
    """Names mimic the reference model's component prefixes."""

    def __init__(self, d=32, h=64):
        super().__init__()
        self.backbone = nn.Module()
        self.backbone.vision_backbone = nn.Module()
        self.backbone.vision_backbone.trunk = nn.Module()
        self.backbone.vision_backbone.trunk.blocks = nn.ModuleList([ToyBlock(d, h) for _ in range(2)])
        self.backbone.language_backbone = nn.Module()
        self.backbone.language_backbone.encoder = nn.ModuleList([ToyBlock(d, h)])
        self.geometry_encoder = ToyBlock(d, h)
        self.transformer = nn.Module()
        self.transformer.encoder = nn.ModuleList([ToyBlock(d, h)])
        self.transformer.decoder = nn.ModuleList([ToyBlock(d, h)])
        self.segmentation_head = nn.Module()
        self.segmentation_head.mask_decoder = ToyBlock(d, h)
        self.hs_proj = nn.Linear(d, d)
        self.prompt_project = nn.Linear(d, d)



ROOT_CONFIGS = {
    "default": dict(),
    "fc_only_vision": dict(target_modules=["fc1", "fc2"], apply_to_text_encoder=False,
                           apply_to_detr_encoder=False, apply_to_detr_decoder=False),
    "qkv_all": dict(target_modules=["q_proj", "k_proj", "v_proj", "out_proj"],
                    apply_to_geometry_encoder=True, apply_to_mask_decoder=True),
    "full_yaml_like": dict(target_modules=["q_proj", "k_proj", "v_proj", "out_proj", "fc1", "fc2"],
                           apply_to_geometry_encoder=True, apply_to_mask_decoder=True),
    "decoder_only": dict(target_modules=["q_proj", "k_proj", "v_proj"], apply_to_vision_encoder=False,
                         apply_to_text_encoder=False, apply_to_detr_encoder=False),
    "linear12": dict(target_modules=["linear1", "linear2"]),
}
PKG_CONFIGS = {
    "default": None,
    "all": ["all"],
    "fc": ["fc1", "fc2"],
    "proj": ["proj"],
    "qkv": ["q_proj", "k_proj", "v_proj"],
    "linear1": ["linear1"],
}


