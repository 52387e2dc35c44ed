"""
The trunk's attention forward kernel (csrc/attn_kernels.hip, C-ABI sam3_attn_fwd) against a plain fp32 evaluation of
softmax(q k^T / sqrt(d)) v on the same bf16 inputs (the op's PyTorch fp32 reference), at the trunk's shapes (576-token
windows, the 5184-token grid, 16 heads of 64) and at ragged ones; and its autograd pairing with PyTorch-ROCm's attention
backward against autograd through the fp32 reference.  Tolerances: o is bf16 -> one rounding (2^-8 relative) + 1e-3 of max
(the probabilities enter the second product rounded to bf16, as in every flash kernel); lse fp32 1e-3 absolute;
gradients 2e-2 of max |reference| (bf16 backward kernel).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(q, k, v):
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))         # [B, H, L, D]
    s = qf @ kf.transpose(-1, -2) * q.shape[-1] ** -0.5
    lse = torch.logsumexp(s, dim=-1)
    return (torch.softmax(s, dim=-1) @ vf).transpose(1, 2), lse


@pytest.mark.parametrize("B,L,H", [(3, 576, 16), (1, 5184, 2), (2, 64, 4), (2, 160, 3), (5, 96, 1), (1, 256, 4)])
def test_forward_matches_fp32_reference(B, L, H):
    from sam3_lora_amd.vit import _HipAttention
    g = torch.Generator(device=DEV).manual_seed(L + H)
    q, k, v = (torch.randn(B, L, H, 64, device=DEV, generator=g).bfloat16() for _ in range(3))
    q = q * 2.0                                             # sharper rows: the running max really moves between tiles
    assert _HipAttention.usable(q, k, v), "the kernel did not calibrate against PyTorch's forward on this box"
    o, lse = _HipAttention._launch(q, k, v)
    ro, rlse = _ref(q, k, v)
    assert o.shape == q.shape and o.dtype == torch.bfloat16 and lse.shape == (B, H, L)
    err = (o.float() - ro).abs()
    assert bool((err <= 2.0 ** -8 * ro.abs() + 1e-3 * ro.abs().max()).all()), float(err.max() / ro.abs().max())
    assert float((lse - rlse).abs().max()) < 1e-3
    assert "libsam3_lora_amd.so" in open("/proc/self/maps").read()


def test_strided_heads_inside_a_wider_tensor_and_spiked_rows():
    """q / k / v as the [..., H, 64] slices of a packed buffer (row stride 3 * H * 64); one key aligned with one query by a
    large factor so that its row's maximum jumps by ~60 in the middle of the key sequence (the online rescale path)."""
    from sam3_lora_amd.vit import _HipAttention
    g = torch.Generator(device=DEV).manual_seed(1)
    B, L, H = 2, 192, 4
    packed = torch.randn(B, L, 3, H, 64, device=DEV, generator=g).bfloat16()
    packed[0, 130, 1, 2] = packed[0, 7, 0, 2] * 8.0            # key 130 of head 2 ~ 8 x query 7
    q, k, v = (packed[:, :, i].contiguous() for i in range(3))
    o, lse = _HipAttention._launch(q, k, v)
    ro, rlse = _ref(q, k, v)
    assert float((o.float() - ro).abs().max() / ro.abs().max()) < 1e-2 and float((lse - rlse).abs().max()) < 2e-3
    assert float(rlse[0, 2, 7]) > float(rlse[0, 2].median()) + 20.0          # the spike is really there


@pytest.mark.parametrize("B,L,H", [(2, 576, 4), (1, 256, 2)])
def test_autograd_pairing_with_pytorch_backward(B, L, H):
    from sam3_lora_amd.vit import _attention, _HipAttention
    g = torch.Generator(device=DEV).manual_seed(3)
    q, k, v = (torch.randn(B, L, H, 64, device=DEV, generator=g).bfloat16().requires_grad_(True) for _ in range(3))
    go = torch.randn(B, L, H, 64, device=DEV, generator=g).bfloat16()
    o = _attention(q, k, v)
    assert type(o.grad_fn).__name__ == "_HipAttentionBackward"
    o.backward(go)
    got = [t.grad.float().clone() for t in (q, k, v)]
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ro, _ = _ref(qr, kr, vr)
    ro.backward(go.float())
    for a, b in zip(got, (qr.grad, kr.grad, vr.grad)):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-2


def test_other_shapes_keep_pytorch(monkeypatch):
    from sam3_lora_amd.vit import _attention, _HipAttention
    q = torch.randn(2, 64, 2, 32, device=DEV).bfloat16()            # head dimension 32 (the tiny fixture): PyTorch's kernel
    assert not _HipAttention.usable(q, q, q)
    o = _attention(q, q, q)
    assert o.shape == q.shape
    qf = torch.randn(2, 64, 2, 64, device=DEV)                       # fp32: PyTorch's kernel
    assert not _HipAttention.usable(qf, qf, qf)
    monkeypatch.setenv("SAM3_HIP_ATTENTION", "0")
    qb = torch.randn(2, 64, 2, 64, device=DEV).bfloat16()
    assert not _HipAttention.usable(qb, qb, qb)
