"""fp8 frozen-W mode (BASELINE configs[4]; row f-1): the HIP quantiser against torch's own fp8 casts (bit-exact), and the
fp8 base GEMMs inside the adapted modules against the bf16 build (the parity BASELINE asks for: fp8 vs bf16)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("fmt", ["e4m3", "e5m2"])
@pytest.mark.parametrize("src", [torch.bfloat16, torch.float32])
def test_quantiser_matches_torch_cast_bitwise_and_tracks_amax(fmt, src):
    from sam3_lora_amd import _ffi
    from sam3_lora_amd.fp8 import Fp8Quantizer
    tdt, code, fmax = (torch.float8_e4m3fn, _ffi.FP8_E4M3, 448.0) if fmt == "e4m3" else (torch.float8_e5m2, _ffi.FP8_E5M2, 57344.0)
    g = torch.Generator(device=DEV).manual_seed(0)
    x = (torch.randn(777, 1024, device=DEV, generator=g) * 3).to(src)
    x[0, 0] = 50.0
    q = Fp8Quantizer(code)
    out, scale = q(x)                                   # first call calibrates on x itself
    amax = x.abs().max().float()
    assert torch.allclose(scale, (amax / fmax).reshape(1))
    want = (x.float() * (fmax / amax)).clamp(-fmax, fmax).to(tdt)
    assert out.dtype == tdt and torch.equal(out.view(torch.uint8), want.view(torch.uint8))
    # delayed scaling: the second call uses the amax the first one observed (= the same), the third the second's
    y = x * 0.5
    out2, scale2 = q(y)
    assert torch.allclose(scale2, (amax / fmax).reshape(1))
    out3, scale3 = q(y)
    assert torch.allclose(scale3, (amax * 0.5 / fmax).reshape(1))
    # values beyond the delayed amax saturate instead of overflowing
    big = x * 4
    out4, _ = q(big)
    assert torch.isfinite(out4.float()).all() and out4.float().abs().max().item() == fmax


def test_fp8_frozen_linears_track_the_bf16_build():
    """LoRA-adapted MLP (fused node) and a plain frozen Linear: outputs and input gradients of the fp8 route within 8 % of
    the bf16 build's max magnitude (two e4m3 operands: 2^-4 relative per element, averaged over K), adapter gradients
    within 10 %; switching the mode off restores the bf16 result bit for bit."""
    import lora_layers as L
    from sam3_lora_amd import fp8
    from sam3_lora_amd.functional import frozen_linear, TransposedCopy
    from sam3_lora_amd.vit import Mlp
    torch.manual_seed(0)
    mlp = Mlp(256, 1024)
    mlp.fc1, mlp.fc2 = L.LoRALinear(mlp.fc1, rank=8, alpha=16), L.LoRALinear(mlp.fc2, rank=8, alpha=16)
    with torch.no_grad():
        mlp.fc1.lora.lora_B.normal_(0, 0.02), mlp.fc2.lora.lora_B.normal_(0, 0.02)
    mlp.to(DEV)
    lin = torch.nn.Linear(256, 512).to(DEV).bfloat16().requires_grad_(False)
    for m in mlp.modules():
        if isinstance(m, torch.nn.Linear):
            m.to(torch.bfloat16).requires_grad_(False)
    x = torch.randn(4, 300, 256, device=DEV).bfloat16()

    def run():
        for p in mlp.parameters():
            p.grad = None
        a = x.clone().requires_grad_(True)
        y = mlp(a)
        z = frozen_linear(y, lin, TransposedCopy())
        (z.float() ** 2).mean().backward()
        return y.detach().float(), z.detach().float(), a.grad.float(), mlp.fc1.lora.lora_A.grad.clone(), mlp.fc2.lora.lora_B.grad.clone()

    ref = run()
    fp8.enable_fp8_frozen(True)
    try:
        got = run()
        assert len(fp8._WEIGHTS) == 3                              # fc1, fc2, lin took the fp8 route
    finally:
        fp8.enable_fp8_frozen(False)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    assert rel(got[0], ref[0]) < 0.08 and rel(got[1], ref[1]) < 0.08 and rel(got[2], ref[2]) < 0.08
    assert rel(got[3], ref[3]) < 0.10 and rel(got[4], ref[4]) < 0.10
    assert not torch.equal(got[0], ref[0])
    again = run()
    assert all(torch.equal(a, b) for a, b in zip(again, ref))
