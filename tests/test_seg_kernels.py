"""
Channels-last GroupNorm (+ ReLU) kernels of the mask head (include/sam3_seg_amd.h, csrc/seg_kernels.hip) and the
channels-last mask dot product, against the PyTorch operators they stand in for
(sam3/model/maskformer_segmentation.py:205-222 ``relu(GroupNorm(8, C)(conv(x)))``, :48-52 the mask einsum).
"""
import pytest
import torch
import torch.nn.functional as F

from sam3_lora_amd.sam3_seghead import MaskPredictor, PixelDecoder, _MaskDot, gn_kernels_support, group_norm_relu


def _gn(C, G, device, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    gn = torch.nn.GroupNorm(G, C)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        gn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    return gn.to(device=device, dtype=dtype).requires_grad_(False)


def test_cpu_and_trainable_affine_use_the_pytorch_operators():
    gn = _gn(32, 8, "cpu", torch.float32)
    x = torch.randn(2, 32, 6, 5)
    assert not gn_kernels_support(x, gn)
    assert torch.equal(group_norm_relu(x, gn), F.relu(gn(x)))
    assert torch.equal(group_norm_relu(x, gn, relu=False), gn(x))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(8, 256, 144, 144), (2, 256, 37, 23), (1, 64, 9, 7), (3, 32, 16, 16)])
@pytest.mark.parametrize("relu", [True, False])
def test_group_norm_relu_kernels_match_torch(dtype, shape, relu):
    N, C, H, W = shape
    G = 8
    dev = "cuda"
    gn = _gn(C, G, dev, dtype)
    if C // G % (8 if dtype == torch.bfloat16 else 4):
        assert not gn_kernels_support(torch.empty(shape, device=dev, dtype=dtype), gn)
        return
    g = torch.Generator().manual_seed(1)
    x0 = (torch.randn(shape, generator=g) * 1.5 + 0.4).to(dev, dtype)
    gy0 = torch.randn(shape, generator=g).to(dev, dtype)
    for fmt in (torch.channels_last, torch.contiguous_format):
        x = x0.detach().clone(memory_format=fmt).requires_grad_(True)
        assert gn_kernels_support(x, gn)
        y = group_norm_relu(x, gn, relu=relu)
        assert y.shape == x.shape and y.dtype == dtype
        assert y.is_contiguous(memory_format=torch.channels_last)
        y.backward(gy0.contiguous(memory_format=fmt))
        # reference: the same expression in fp32 on the same (rounded) input
        xr = x0.detach().float().clone().requires_grad_(True)
        yr = F.group_norm(xr, G, gn.weight.float(), gn.bias.float(), gn.eps)
        yr = F.relu(yr) if relu else yr
        yr.backward(gy0.float())
        tol = 1e-2 if dtype == torch.bfloat16 else 2e-5          # bf16: one output rounding (2^-8) of O(4) values
        assert (y.float() - yr).abs().max().item() <= tol * max(yr.abs().max().item(), 1.0)
        assert (x.grad.float() - xr.grad).abs().max().item() <= tol * max(xr.grad.abs().max().item(), 1e-3)
    # bit-reproducible: fixed-order reductions
    xa = x0.contiguous(memory_format=torch.channels_last)
    assert torch.equal(group_norm_relu(xa, gn, relu=relu), group_norm_relu(xa, gn, relu=relu))


@pytest.mark.gpu
def test_pixel_decoder_on_channels_last_levels_matches_the_pytorch_form():
    dev, dt = "cuda", torch.bfloat16
    torch.manual_seed(0)
    dec = PixelDecoder(64, 2).to(dev, dt).requires_grad_(False)
    feats = [torch.randn(2, 64, s, s, device=dev, dtype=dt).clone(memory_format=torch.channels_last).requires_grad_(True)
             for s in (32, 16, 8)]
    out = dec(feats)
    out.float().square().mean().backward()
    got = [f.grad.clone() for f in feats]
    for f in feats:
        f.grad = None
    x = feats[-1]
    for i, finer in enumerate(reversed(feats[:-1])):          # the reference expression, PyTorch operators only
        x = finer + F.interpolate(x, size=finer.shape[-2:], mode="nearest")
        x = F.relu(dec.norms[i](dec.conv_layers[i](x)))
    x.float().square().mean().backward()
    assert (out.float() - x.float()).abs().max().item() <= 2e-2 * x.float().abs().max().item()
    for a, f in zip(got, feats):
        assert (a.float() - f.grad.float()).abs().max().item() <= 3e-2 * f.grad.float().abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("lead", [False, True])
def test_channels_last_mask_dot_equals_einsum(lead):
    dev, dt = "cuda", torch.bfloat16
    g = torch.Generator().manual_seed(2)
    B, Q, C, H, W = 2, 5, 32, 12, 10
    pix0 = torch.randn(B, C, H, W, generator=g).to(dev, dt)
    q0 = torch.randn(*((3,) if lead else ()), B, Q, C, generator=g).to(dev, dt)
    go = torch.randn(*((3,) if lead else ()), B, Q, H, W, generator=g).to(dev, dt)
    pix = pix0.clone(memory_format=torch.channels_last).requires_grad_(True)
    q = q0.clone().requires_grad_(True)
    out = _MaskDot.apply(q, pix)
    out.backward(go)
    assert pix.grad.is_contiguous(memory_format=torch.channels_last)
    pr, qr = pix0.float().requires_grad_(True), q0.float().requires_grad_(True)
    ref = torch.einsum(("l" if lead else "") + "bqc,bchw->" + ("l" if lead else "") + "bqhw", qr, pr)
    ref.backward(go.float())
    rel = lambda a, b: ((a.float() - b).abs().max() / b.abs().max()).item()
    assert out.shape == ref.shape and rel(out, ref) < 1e-2
    assert rel(q.grad, qr.grad) < 1e-2 and rel(pix.grad, pr.grad) < 1e-2
    # the module picks this path for a channels-last embedding and the einsum otherwise: same numbers
    mp = MaskPredictor(C, C).to(dev, dt)
    a = mp(q0, pix0.contiguous(memory_format=torch.channels_last))
    b = mp(q0, pix0)
    assert rel(a, b.float()) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("mode", ["log", "linear"])
@pytest.mark.parametrize("presence_row", [True, False])
def test_rpb_bias_kernel_matches_the_operator_formulation(dtype, mode, presence_row):
    """sam3_rpb_bias_fwd (box-relative position bias of the decoder's image cross-attention) against the operator chain
    of TransformerDecoder._get_rpb_matrix on the same weights: fp32 to rounding noise, bf16 to one bf16 step of the
    output (both round at the same places; the GEMMs accumulate in a different order)."""
    from sam3_lora_amd.sam3_image import TINY_CONFIG, build_sam3_image_model
    model = build_sam3_image_model(device="cpu", eval_mode=False, config=dict(TINY_CONFIG, d_model=64, heads=8, ffn=64),
                                   match_in_forward=False, seed=3)
    dec = model.transformer.decoder
    dec.boxRPB = mode
    dec.requires_grad_(False).to("cuda", dtype)
    g = torch.Generator().manual_seed(5)
    Q, B, H, W = 23, 3, 18, 24
    cxcy = torch.rand(Q, B, 2, generator=g)
    wh = torch.rand(Q, B, 2, generator=g) * 0.5 + 0.01
    boxes = torch.cat([cxcy, wh], -1).cuda()
    assert dec._rpb_kernel_applies(boxes)
    got = dec._get_rpb_matrix(boxes, (H, W), presence_row=presence_row)
    assert got.shape == (B, 8, Q + int(presence_row), H * W) and got.dtype == dtype and got.is_contiguous()
    dec._rpb_kernel_applies = lambda rb: False                       # the operator formulation of the same method
    ref = dec._get_rpb_matrix(boxes, (H, W), presence_row=presence_row)
    assert ref.shape == got.shape
    if presence_row:
        assert torch.equal(got[:, :, 0], torch.zeros_like(got[:, :, 0]))
    err = (got.float() - ref.float()).abs().max().item()
    scale = ref.float().abs().max().item()
    assert err <= (2e-2 if dtype == torch.bfloat16 else 2e-6) * max(scale, 1e-3), (err, scale)
    # trainable MLP weights or boxes with a gradient keep the differentiable operator chain
    del dec._rpb_kernel_applies
    dec.boxRPB_embed_x.layers[0].weight.requires_grad_(True)
    assert not dec._rpb_kernel_applies(boxes)
    dec.boxRPB_embed_x.layers[0].weight.requires_grad_(False)
    assert not dec._rpb_kernel_applies(boxes.clone().requires_grad_(True))
