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


def test_the_scale_shrinks_at_most_one_octave_per_call_and_grows_at_once():
    """Round 6, the protocol's short memory: a call scales with max(observed by its predecessor, half of what the predecessor scaled
    with).  A role whose magnitude drops 8x and comes back saturates by one octave at most instead of by the whole drop."""
    from sam3_lora_amd import _ffi
    from sam3_lora_amd.fp8 import Fp8Quantizer
    q = Fp8Quantizer(_ffi.FP8_E4M3)
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(64, 256, device=DEV, generator=g).bfloat16()
    A = float(x.abs().max())
    scales = []
    for t in (x, x, x * 0.125, x * 0.125, x * 0.125, x * 0.125, x, x):
        scales.append(float(q(t)[1]) * 448.0 / A)
    # calibrate, observed A, observed A, then the drop: the predecessor saw A/8 but scaled with A -> A/2, A/4, A/8; the return: the
    # predecessor saw A/8 (one call late, as always) -> A/8, then A at once
    want = [1.0, 1.0, 1.0, 0.5, 0.25, 0.125, 0.125, 1.0]
    assert all(abs(a - b) <= 1e-6 * max(b, 1e-3) for a, b in zip(scales, want)), (scales, want)


def test_an_empty_observation_keeps_the_previous_scale():
    """Delayed scaling after a call that observed NOTHING (an all-zero tensor: a gradient that vanished for a step, a fully masked
    batch): the amax it gathered is 0.  The next call must not scale with it -- "amax = tiny" multiplies a real tensor by 7.5e9 and
    saturates every element to +-max, finite garbage that turns non-finite a few layers on (the suspected path of the one non-finite
    fp8 run of round 4) -- it keeps the scale of the role's last real observation.  Fails on the round-4 library: every element
    of the third image is +-448 there."""
    from sam3_lora_amd import _ffi
    from sam3_lora_amd.fp8 import Fp8Quantizer
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(512, 1024, device=DEV, generator=g).bfloat16()
    q = Fp8Quantizer(_ffi.FP8_E4M3)
    img0, s0 = q(x)                                     # calibrates on x; gathers amax(x) for the next call
    s0 = s0.clone()
    img1, s1 = q(torch.zeros_like(x))                   # scales with amax(x); observes nothing
    assert torch.allclose(s1, s0) and float(img1.float().abs().max()) == 0.0
    img2, s2 = q(x)                                     # previous amax = 0: keeps s0
    assert torch.allclose(s2, s0), (float(s2), float(s0))
    deq = img2.float() * s2
    assert float((deq - x.float()).abs().max()) <= 2.0 ** -4 * float(x.float().abs().max())
    assert float((img2.float().abs() == 448.0).float().mean()) < 1e-3          # not saturated wholesale
    img3, s3 = q(x)                                     # and the call after that scales with the amax img2's call gathered
    assert torch.allclose(s3, s0) and torch.equal(img3.view(torch.uint8), img0.view(torch.uint8))
    # the very first call on an all-zero tensor (nothing to calibrate on): scale 1, a zero image, no NaN
    q2 = Fp8Quantizer(_ffi.FP8_E5M2)
    z, sz = q2(torch.zeros_like(x))
    assert float(sz) == 1.0 and float(z.float().abs().max()) == 0.0
    img4, s4 = q2(x)                                    # still nothing observed before it: scale 1
    assert float(s4) == 1.0 and torch.isfinite(img4.float()).all()


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


def _slot_state(amax_value):
    from sam3_lora_amd import _ffi
    amax = torch.zeros(2, _ffi.FP8_AMAX_FLOATS, device=DEV)
    amax[0, 5 * _ffi.FP8_AMAX_STRIDE] = amax_value     # any slot: the reader takes the maximum
    return amax, torch.empty(1, device=DEV)


@pytest.mark.parametrize("fmt", ["e4m3", "e5m2"])
def test_producers_write_the_image_the_separate_quantiser_would(fmt):
    """The fp8 image a producing kernel writes beside its bf16 output (LayerNorm forward; the GELU- and GELU'-fused adapter
    passes) == torch's cast of that bf16 output at the delayed scale, bit for bit; the gathered amax == max |output|."""
    import ctypes
    from sam3_lora_amd import _ffi
    from sam3_lora_amd import functional as Fn
    from sam3_lora_amd import vit as V
    tdt, code, fmax = (torch.float8_e4m3fn, _ffi.FP8_E4M3, 448.0) if fmt == "e4m3" else (torch.float8_e5m2, _ffi.FP8_E5M2, 57344.0)
    g = torch.Generator(device=DEV).manual_seed(1)

    def expect(t, amax_prev):
        return (t.float() * (fmax / amax_prev)).clamp(-fmax, fmax).to(tdt).view(torch.uint8)

    # LayerNorm
    M, C = 1000, 1024
    x = torch.randn(M, C, device=DEV, generator=g).bfloat16()
    w, b = (1 + 0.1 * torch.randn(C, device=DEV, generator=g)).bfloat16(), (0.1 * torch.randn(C, device=DEV, generator=g)).bfloat16()
    amax, scale = _slot_state(3.0)
    img = torch.empty(M, C, dtype=tdt, device=DEV)
    y = V._FrozenLayerNorm.apply(x, w, b, 1e-5, (img, code, amax[0], amax[1], scale))
    y_plain = V._FrozenLayerNorm.apply(x, w, b, 1e-5)
    assert torch.equal(y, y_plain)
    assert torch.equal(img.view(torch.uint8), expect(y, 3.0)) and torch.allclose(scale, torch.tensor([3.0 / fmax], device=DEV))
    assert float(amax[1][::_ffi.FP8_AMAX_STRIDE].max()) == float(y.float().abs().max())
    # adapter forward with GELU: the image is GELU(y)'s
    M, fin, fout, r = 777, 264, 520, 16
    xa = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
    base = torch.randn(M, fout, device=DEV, generator=g).bfloat16()
    A, B = torch.randn(fin, r, device=DEV, generator=g) / 16, torch.randn(r, fout, device=DEV, generator=g) / 4
    yq, act = base.clone(), torch.empty_like(base)
    amax, scale = _slot_state(2.0)
    img = torch.empty(M, fout, dtype=tdt, device=DEV)
    Fn.lora_fwd_(xa, A, B, yq, 2.0, 0, gelu_out=act, q8=(img, code, amax[0], amax[1], scale))
    y0, act0 = base.clone(), torch.empty_like(base)
    Fn.lora_fwd_(xa, A, B, y0, 2.0, 0, gelu_out=act0)
    assert torch.equal(yq, y0) and torch.equal(act, act0)
    assert torch.equal(img.view(torch.uint8), expect(act, 2.0)) and float(amax[1][::_ffi.FP8_AMAX_STRIDE].max()) == float(act.float().abs().max())
    # adapter backward with GELU': the image is the pre-activation gradient's
    gy = torch.randn(M, fout, device=DEV, generator=g).bfloat16()
    h = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
    gbase = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
    res = []
    for q8 in (None, "q8"):
        gx = gbase.clone()
        gA, gB = torch.zeros_like(A), torch.zeros_like(B)
        amax, scale = _slot_state(5.0)
        img = torch.empty(M, fin, dtype=tdt, device=DEV)
        Fn.lora_bwd_(gy, xa, None, A, B, gx, gA, gB, 2.0, 0, gelu_pre=h, q8=(img, code, amax[0], amax[1], scale) if q8 else None)
        res.append((gx, gA, gB))
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    assert torch.equal(img.view(torch.uint8), expect(res[1][0], 5.0)) and float(amax[1][::_ffi.FP8_AMAX_STRIDE].max()) == float(res[1][0].float().abs().max())
    # refused where the image cannot ride: rank > 16, dropout on the input gradient
    A32, B32 = torch.randn(fin, 32, device=DEV) / 16, torch.randn(32, fout, device=DEV) / 4
    with pytest.raises(Fn.LoRAKernelError):
        Fn.lora_fwd_(xa, A32, B32, base.clone(), 2.0, 0, gelu_out=torch.empty_like(base), q8=(img, code, amax[0], amax[1], scale))


def test_fused_and_separate_quantisation_give_the_same_training_steps(monkeypatch):
    """A trunk block (LayerNorm -> attention -> LayerNorm -> LoRA MLP) stepped three times in the fp8 frozen-W mode with the
    producers writing the fp8 images themselves (from the second step on: the first calibrates) and with every image made by
    the separate quantiser: outputs and gradients bit-identical -- same values, same delayed scales, fewer passes."""
    import lora_layers as L
    from sam3_lora_amd import fp8
    from sam3_lora_amd import vit as V

    def run(fused):
        torch.manual_seed(0)
        blk = V.Block(256, 4, 4.0, True, 0.0, window_size=0, input_size=(8, 8), rope_pt_size=(8, 8), rope_interp=False)
        blk.mlp.fc1, blk.mlp.fc2 = L.LoRALinear(blk.mlp.fc1, rank=8, alpha=16), L.LoRALinear(blk.mlp.fc2, rank=8, alpha=16)
        with torch.no_grad():
            blk.mlp.fc1.lora.lora_B.normal_(0, 0.02), blk.mlp.fc2.lora.lora_B.normal_(0, 0.02)
        blk.to(DEV)
        for n, p in blk.named_parameters():
            if "lora_" not in n:
                p.requires_grad_(False)
                p.data = p.data.to(torch.bfloat16)
        if not fused:
            monkeypatch.setattr(fp8, "producer_slots", lambda *a, **k: None)
        fp8.enable_fp8_frozen(True)
        outs = []
        try:
            x = torch.randn(4, 8, 8, 256, device=DEV).bfloat16()
            for step in range(3):
                for p in blk.parameters():
                    p.grad = None
                xi = (x * (1 + 0.1 * step)).requires_grad_(True)
                y = blk(xi)
                (y.float() ** 2).mean().backward()
                outs.append((y.detach().clone(), xi.grad.clone(), blk.mlp.fc1.lora.lora_A.grad.clone(), blk.mlp.fc2.lora.lora_B.grad.clone()))
        finally:
            fp8.enable_fp8_frozen(False)
            monkeypatch.undo()
        return outs

    a, b = run(True), run(False)
    for sa, sb in zip(a, b):
        for u, v in zip(sa, sb):
            assert torch.equal(u, v)
    assert not torch.equal(a[0][0], a[1][0])


def test_non_finite_elements_keep_their_encoding_and_stay_out_of_the_amax():
    """A NaN element leaves as NaN and an Inf as the format's non-finite encoding (e4m3fn has no Inf: NaN) -- a diverged step
    still shows -- but neither enters the running amax; a FINITE value beyond the delayed range saturates: tensors carry
    non-finite values in positions nobody reads (fully masked softmax rows, padded keys), and a NaN amax would turn the next
    call's scale -- hence the whole tensor -- into NaN."""
    from sam3_lora_amd import _ffi
    from sam3_lora_amd.fp8 import Fp8Quantizer
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(64, 1024, device=DEV, generator=g).bfloat16()
    x[3, 5], x[7, 9], x[9, 1] = float("nan"), float("inf"), 30.0
    q = Fp8Quantizer(_ffi.FP8_E4M3)
    out, scale = q(x)
    assert torch.allclose(scale, torch.tensor([30.0 / 448.0], device=DEV))
    f = out.float()
    assert torch.isnan(f[3, 5]) and torch.isnan(f[7, 9]) and int(torch.isnan(f).sum()) == 2
    assert f[9, 1] == 448.0
    out2, scale2 = q(x * 0.5)                       # scaled with the amax gathered above: finite
    assert torch.allclose(scale2, torch.tensor([30.0 / 448.0], device=DEV)) and torch.isfinite(scale2).all()
    out3, scale3 = q(x)
    assert torch.allclose(scale3, torch.tensor([15.0 / 448.0], device=DEV))
    x[11, 2] = -1.0e6                               # finite, far beyond the delayed range: saturates
    out4, scale4 = q(x)
    assert torch.allclose(scale4, torch.tensor([30.0 / 448.0], device=DEV)) and out4.float()[11, 2] == -448.0


def test_fp8_image_and_recomputed_input_ride_on_the_same_pass():
    """sam3_lora_bwd_act_q8 with x == NULL: the GELU' pass writes the e5m2 image AND recomputes GELU(h) for gA.  Image, gx and gB
    bit-identical to the x-given call, gA equal up to summation order."""
    from sam3_lora_amd import _ffi
    from sam3_lora_amd import functional as Fn
    g = torch.Generator(device=DEV).manual_seed(11)
    M, fin, fout, r = 1500, 520, 264, 16
    h = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
    a, hh = torch.empty_like(h), h.clone()
    Fn.lora_fwd_(torch.zeros(M, 64, device=DEV, dtype=torch.bfloat16), torch.zeros(64, r, device=DEV), torch.zeros(r, fin, device=DEV),
                 hh, 2.0, 0, gelu_out=a)
    gy = torch.randn(M, fout, device=DEV, generator=g).bfloat16()
    gbase = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
    A, B = torch.randn(fin, r, device=DEV, generator=g) / 16, torch.randn(r, fout, device=DEV, generator=g) / 4
    tT = Fn.lora_fwd_(a, A, B, torch.zeros(M, fout, device=DEV, dtype=torch.bfloat16), 2.0, 0, save_t=True)
    res = []
    for x in (a, None):
        gx = gbase.clone()
        gA, gB = torch.zeros_like(A), torch.zeros_like(B)
        amax, scale = _slot_state(4.0)
        img = torch.empty(M, fin, dtype=torch.float8_e5m2, device=DEV)
        Fn.lora_bwd_(gy, x, tT, A, B, gx, gA, gB, 2.0, 0, gelu_pre=h, q8=(img, _ffi.FP8_E5M2, amax[0], amax[1], scale))
        res.append((gx, gB, img.view(torch.uint8), amax[1][::_ffi.FP8_AMAX_STRIDE].max().clone(), gA))
    for u, v in zip(res[0][:4], res[1][:4]):
        assert torch.equal(u, v)
    assert ((res[0][4] - res[1][4]).abs().max() / res[0][4].abs().max()).item() < 2e-6


@pytest.mark.parametrize("M,fin,fout,rank,hl", [(776, 256, 520, 16, True), (1000, 1024, 776, 8, True), (300, 384, 264, 24, True),
                                                (41472, 1024, 4736, 16, True), (776, 256, 520, 16, False)])
def test_fused_fp8_linear_is_one_rounding_from_the_dequantised_oracle(M, fin, fout, rank, hl, monkeypatch):
    """sam3_lora_linear_fwd_q8 (the fp8-W variant of the fused GEMM): e4m3 x and W on the scaled fp8 MFMA, the LoRA
    branch as a bf16 hi + lo K step in the same accumulator, bias, GELU -- against fp64 arithmetic on the DEQUANTISED operands
    (sx xq) (sw wq)^T + b + s (x A) B: every element within one bf16 rounding (the quantisation itself is the mode's, pinned by
    test_quantiser_*); a = GELU(bf16(h)) to one rounding; the saved t^T is the plain forward's, bit for bit; and with the
    delayed-scaling slots of the next GEMM handed in, GELU(h) also leaves as the e4m3 image a separate quantisation pass over it
    would write, with the amax gathered for the next call."""
    from sam3_lora_amd import _ffi, functional as Fn
    from sam3_lora_amd.fp8 import Fp8Quantizer, Fp8Weight
    if not hl:
        monkeypatch.setenv("SAM3_LORA_SINGLE_ROUND", "1")
    _ffi.load().sam3_lora_debug_reload_knobs()
    g = torch.Generator(device=DEV).manual_seed(M + rank)
    x = torch.randn(M, fin, device=DEV, generator=g).bfloat16()
    W = (torch.randn(fout, fin, device=DEV, generator=g) / fin ** 0.5).bfloat16()
    b = torch.randn(fout, device=DEV, generator=g).bfloat16()
    A = torch.randn(fin, rank, device=DEV, generator=g) / fin ** 0.5
    B = torch.randn(rank, fout, device=DEV, generator=g) * 0.1
    s = 2.0
    st = Fp8Weight(W)
    xq, sx = st.qx(x)
    y, a, tT = Fn.lora_linear_fwd_q8_(x, xq, sx, st.wq, st.scale, b, A, B, s, 0, save_t=True, gelu=True)
    rows = torch.arange(M, device=DEV) if M <= 2048 else torch.randint(0, M, (512,), generator=torch.Generator().manual_seed(1)).to(DEV)
    xd = xq[rows].double() * sx.double()
    want = xd @ (st.wq.double() * st.scale.double()).t() + b.double() + s * ((x[rows].double() @ A.double()) @ B.double())
    err = (y[rows].double() - want).abs()
    bound = 2.0 ** -8 * want.abs() + (3e-5 if hl else 1e-2) * want.abs().max()
    assert (err <= bound).all(), (int((err > bound).sum()), float((err / want.abs().max()).max()))
    ga = torch.nn.functional.gelu(y[rows].double())
    assert ((a[rows].double() - ga).abs() <= 2.0 ** -8 * ga.abs() + 1e-6 * ga.abs().max() + 1e-6).all()
    # the saved t^T: what the stand-alone forward saves
    y2 = torch.zeros(M, fout, device=DEV, dtype=torch.bfloat16)
    tT2 = Fn.lora_fwd_(x, A, B, y2, s, 0, save_t=True)
    assert torch.equal(tT.view(torch.uint8), tT2.view(torch.uint8))
    # repeated: the same bits (a DMA / read race would show here)
    y3, a3, _ = Fn.lora_linear_fwd_q8_(x, xq, sx, st.wq, st.scale, b, A, B, s, 0, gelu=True)
    assert torch.equal(y3, y) and torch.equal(a3, a)
    # the fp8 image of GELU(h) for the next GEMM: delayed scaling with the amax of a previous tensor of the same role
    q = Fp8Quantizer(_ffi.FP8_E4M3)
    q(a * 0.5)                                   # calibrates: amax = max |a| / 2 -> some of a's values will saturate
    slots = q.begin(a.device)
    image = torch.empty(M, fout, dtype=torch.float8_e4m3fn, device=DEV)
    y4, a4, _ = Fn.lora_linear_fwd_q8_(x, xq, sx, st.wq, st.scale, b, A, B, s, 0, gelu=True, q8=(image, _ffi.FP8_E4M3) + slots)
    assert torch.equal(y4, y) and torch.equal(a4, a)
    amax_prev = (a * 0.5).abs().max().float()
    assert torch.allclose(slots[2], (amax_prev / 448.0).reshape(1))
    want_img = (a.float() * (448.0 / amax_prev)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    assert torch.equal(image.view(torch.uint8), want_img.view(torch.uint8))
    assert float(slots[1].max()) == float(a.abs().max().float())       # gathered for the next call
    monkeypatch.delenv("SAM3_LORA_SINGLE_ROUND", raising=False)
    _ffi.load().sam3_lora_debug_reload_knobs()


def test_a_step_with_non_finite_gradients_is_skipped_on_the_device():
    """trainer.NonFiniteStepGuard on the GPU with torch's fused AdamW (round 6): a trunk block in the fp8 frozen-W mode is stepped five
    times; in the third step one incoming gradient element is made Inf.  That step must leave A / B, the moments and the step counts
    untouched and be counted; the steps before and after it are ordinary updates."""
    import lora_layers as L
    from sam3_lora_amd import fp8
    from sam3_lora_amd import vit as V
    from sam3_lora_amd.trainer import NonFiniteStepGuard, make_adamw
    torch.manual_seed(0)
    blk = V.Block(256, 4, 4.0, True, 0.0, window_size=0, input_size=(8, 8), rope_pt_size=(8, 8), rope_interp=False)
    blk.mlp.fc1, blk.mlp.fc2 = L.LoRALinear(blk.mlp.fc1, rank=8, alpha=16), L.LoRALinear(blk.mlp.fc2, rank=8, alpha=16)
    with torch.no_grad():
        blk.mlp.fc1.lora.lora_B.normal_(0, 0.02), blk.mlp.fc2.lora.lora_B.normal_(0, 0.02)
    blk.to(DEV)
    params = []
    for n, p in blk.named_parameters():
        if "lora_" in n:
            params.append(p)
        else:
            p.requires_grad_(False)
            p.data = p.data.to(torch.bfloat16)
    opt = make_adamw(params, lr=1e-2, weight_decay=0.01)
    guard = NonFiniteStepGuard(opt, torch.device(DEV))
    assert guard.fused
    fp8.enable_fp8_frozen(True)
    try:
        x = torch.randn(4, 8, 8, 256, device=DEV).bfloat16()
        moved = []
        for step in range(5):
            opt.zero_grad(set_to_none=True)
            y = blk(x)
            w = torch.ones_like(y, dtype=torch.float32)
            if step == 2:
                w[0, 0, 0, 0] = float("inf")
            (y.float() * w).mean().backward()
            before = [p.detach().clone() for p in params]
            guard.step()
            moved.append(any(not torch.equal(a, b) for a, b in zip(before, params)))
            assert all(bool(torch.isfinite(p).all()) for p in params)
    finally:
        fp8.enable_fp8_frozen(False)
    assert moved == [True, True, False, True, True] and float(guard.skipped) == 1.0
    assert all(float(st["step"]) == 4.0 for st in opt.state.values())
