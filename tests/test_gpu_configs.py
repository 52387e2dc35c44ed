"""
GPU parity on the configurations BASELINE.json names, at their real sizes, against the ORACLE (numpy, fp64
accumulation) -- not a torch expression on the GPU:

  configs[1]  full_lora_config @ r=16 (alpha 32), batch 8 -> M = 41,472 rows: fc1 (1024 -> 4736) and fc2 (4736 -> 1024),
              forward and backward, bf16 and exact fp32;
  configs[0]  minimal r=4 (alpha 8), batch 2 -> M = 10,368, both widths;
  configs[3]  r=32 (alpha 64) at both widths; and the LITERAL full_lora_config.yaml (r=32, alpha 64, dropout 0.1) with the
              oracle's specification of the dropout stream;
  ranks above 32 (the reference has no limit): 40, 64, 80 against the oracle; rank groups with caller-held operand images.

Forward and input-gradient values are checked on a sample of rows (every row is independent of the others: the oracle
evaluates exactly those rows); the weight gradients are sums over all M rows and are checked in full.
Tolerances: fp32 activations 2e-5 (exact products, fp32 accumulation over up to 82,944 rows against fp64); bf16
activations with r <= 32 (one rank group on the hi + lo operands, include/sam3_lora_amd.h -- since round 4 also the reference's
default rank 32 and its literal dropout 0.1): weight gradients 3e-5, every sampled bf16 output element within ONE rounding of the
fp64 value (2^-8 relative); ranks above 32 (several groups, y / gx rounded once per group): 1e-2 of max |reference|.

  configs[4]  r=8 (alpha 16), batch 16 -> M = 82,944 rows, both widths, bf16 (the adapter side of the fp8 frozen-W mode is
              bf16 / fp32 exactly as in the other configurations; the fp8 base GEMMs are covered by test_fp8.py and the
              whole-model fp8 step of test_sam3_e2e.py);
  configs[3]  r=32 on the widths its text / DETR targets have: text tower c_fc / c_proj 1024 <-> 4096 at M = 8 prompts x 32
              tokens, DETR FFN linear1 / linear2 256 <-> 2048 at M = 4 x 5184 tokens.
"""
import numpy as np
import pytest
import torch

from oracle import lora_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from sam3_lora_amd import functional as Fn

DEV = "cuda:0"
TOK = 5184


def _relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _inputs(M, fin, fout, r, seed, layout=0):
    rng = np.random.default_rng(seed)
    x = O.bf16_round(rng.standard_normal((M, fin), dtype=np.float32))
    gy = O.bf16_round(rng.standard_normal((M, fout), dtype=np.float32))
    base = O.bf16_round(rng.standard_normal((M, fout), dtype=np.float32))
    gxb = O.bf16_round(rng.standard_normal((M, fin), dtype=np.float32))
    A = (rng.uniform(-1, 1, (fin, r) if layout == 0 else (r, fin)) / np.sqrt(r)).astype(np.float32)
    B = (rng.standard_normal((r, fout) if layout == 0 else (fout, r)) * 0.02).astype(np.float32)
    return x, gy, base, gxb, A, B


def _oracle_full_grads(gy, x, A, B, s, layout, mask=None, chunk=4096):
    """gA, gB in fp64, accumulated over row chunks (keeps the fp64 temporaries small)."""
    gA = gB = None
    for i in range(0, x.shape[0], chunk):
        m = None if mask is None else mask[i:i + chunk]
        _, a, b = O.adapter_backward(gy[i:i + chunk], x[i:i + chunk], A, B, s, layout, drop_scale_mask=m,
                                     acc_dtype=np.float64)
        gA = a if gA is None else gA + a
        gB = b if gB is None else gB + b
    return gA, gB


def _one_rounding(got, ref, slack=3e-5, intermediate=None):
    """|got - ref| <= half an ulp of ref (one rounding to bf16) + slack; `intermediate`: the value that was ALSO rounded on the way
    (the in-place result after the first of two rank groups) adds half an ulp of ITS magnitude."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    bound = 2.0 ** -8 * np.abs(ref) + slack * np.abs(ref).max()
    if intermediate is not None:
        bound = bound + 2.0 ** -8 * np.abs(np.asarray(intermediate, np.float64))
    bad = np.abs(got - ref) > bound
    assert not bad.any(), (int(bad.sum()), float((np.abs(got - ref) / (np.abs(ref).max() + 1e-30)).max()))


CASES = [  # (tag, batch, fin, fout, rank, alpha, drop)
    ("c1-fc1-r16", 8, 1024, 4736, 16, 32, 0.0), ("c1-fc2-r16", 8, 4736, 1024, 16, 32, 0.0),
    ("c4-fc1-r8-b16", 16, 1024, 4736, 8, 16, 0.0), ("c4-fc2-r8-b16", 16, 4736, 1024, 8, 16, 0.0),
    ("c0-fc1-r4", 2, 1024, 4736, 4, 8, 0.0), ("c0-fc2-r4", 2, 4736, 1024, 4, 8, 0.0),
    ("c3-fc1-r32", 8, 1024, 4736, 32, 64, 0.0), ("c3-fc2-r32", 8, 4736, 1024, 32, 64, 0.0),
    ("literal-full-fc1", 8, 1024, 4736, 32, 64, 0.1), ("literal-full-fc2", 8, 4736, 1024, 32, 64, 0.1),
]


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("tag,batch,fin,fout,r,alpha,drop", CASES, ids=[c[0] for c in CASES])
def test_named_configs_at_full_size_against_oracle(tag, batch, fin, fout, r, alpha, drop, dtype):
    if dtype == "f32" and tag.startswith(("c0", "c3-fc2", "c4")):
        pytest.skip("fp32 covered by the r=16 and literal-config cases at this size")
    M, s = batch * TOK, alpha / r
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    exact = dtype == "bf16" and r <= 32                 # hi + lo operands (one rank group): fp32 arithmetic on bf16 data
    tol = (3e-5 if exact else 1e-2) if dtype == "bf16" else 2e-5
    x, gy, base, gxb, A, B = _inputs(M, fin, fout, r, seed=len(tag) + r)
    seed, p = 1234567, drop
    mask = O.dropout_scale_mask(M, fin, p, seed) if p > 0 else None
    dx, dgy = torch.from_numpy(x).to(DEV).to(td), torch.from_numpy(gy).to(DEV).to(td)
    dA, dB = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    y = torch.from_numpy(base).to(DEV).to(td)
    tT = Fn.lora_fwd_(dx, dA, dB, y, s, 0, save_t=True, drop_p=p, seed=seed)
    gx = torch.from_numpy(gxb).to(DEV).to(td)
    gA, gB = torch.zeros_like(dA), torch.zeros_like(dB)
    Fn.lora_bwd_(dgy, dx, tT, dA, dB, gx, gA, gB, s, 0, drop_p=p, seed=seed)
    torch.cuda.synchronize()
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(M - 40, M),
                                     np.random.default_rng(1).integers(0, M, 400)]))
    mrows = None if mask is None else mask[rows]
    want_y = base[rows] + O.adapter_delta(x[rows], A, B, s, 0, drop_scale_mask=mrows, acc_dtype=np.float64)
    gx_l, _, _ = O.adapter_backward(gy[rows], x[rows], A, B, s, 0, drop_scale_mask=mrows, acc_dtype=np.float64)
    if exact:
        _one_rounding(y[rows].float().cpu().numpy(), want_y)
        _one_rounding(gx[rows].float().cpu().numpy(), gxb[rows] + gx_l)
    else:
        assert _relmax(y[rows].float().cpu().numpy(), want_y) < tol, "forward"
        assert _relmax(gx[rows].float().cpu().numpy(), gxb[rows] + gx_l) < tol, "input gradient"
    gA_w, gB_w = _oracle_full_grads(gy, x, A, B, s, 0, mask)
    assert _relmax(gA.cpu().numpy(), gA_w) < tol, "gA"
    assert _relmax(gB.cpu().numpy(), gB_w) < tol, "gB"
    if dtype == "f32":     # element-wise relative error of every sampled output that is not small against the tensor
        got, ref = y[rows].float().cpu().numpy(), want_y
        big = np.abs(ref) >= 0.05 * np.abs(ref).max()
        assert (np.abs(got - ref)[big] / np.abs(ref)[big]).max() < 5e-4


C3_SHAPES = [("text-c_fc", 8 * 32, 1024, 4096), ("text-c_proj", 8 * 32, 4096, 1024),
             ("detr-linear1", 4 * TOK, 256, 2048), ("detr-linear2", 4 * TOK, 2048, 256)]


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("tag,M,fin,fout", C3_SHAPES, ids=[c[0] for c in C3_SHAPES])
def test_config3_text_and_detr_widths_r32(tag, M, fin, fout, layout, dtype):
    """BASELINE configs[3]: r = 32, alpha = 64 on the Linears its text / DETR targets reach (c_fc / c_proj of the
    24-layer text tower, linear1 / linear2 of the fusion encoder and decoder), both adapter layouts, whole tensors against
    the fp64 oracle."""
    r, s = 32, 2.0
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    tol = 1e-2 if dtype == "bf16" else 2e-5
    x, gy, base, gxb, A, B = _inputs(M, fin, fout, r, seed=M + fin, layout=layout)
    dx, dgy = torch.from_numpy(x).to(DEV).to(td), torch.from_numpy(gy).to(DEV).to(td)
    dA, dB = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    y = torch.from_numpy(base).to(DEV).to(td)
    tT = Fn.lora_fwd_(dx, dA, dB, y, s, layout, save_t=True)
    gx = torch.from_numpy(gxb).to(DEV).to(td)
    gA, gB = torch.zeros_like(dA), torch.zeros_like(dB)
    Fn.lora_bwd_(dgy, dx, tT, dA, dB, gx, gA, gB, s, layout)
    rows = np.unique(np.concatenate([np.arange(0, min(M, 64)), np.random.default_rng(2).integers(0, M, 300)]))
    want_y = base[rows] + O.adapter_delta(x[rows], A, B, s, layout, acc_dtype=np.float64)
    gx_l, _, _ = O.adapter_backward(gy[rows], x[rows], A, B, s, layout, acc_dtype=np.float64)
    assert _relmax(y[rows].float().cpu().numpy(), want_y) < tol
    assert _relmax(gx[rows].float().cpu().numpy(), gxb[rows] + gx_l) < tol
    gA_w, gB_w = _oracle_full_grads(gy, x, A, B, s, layout)
    assert _relmax(gA.cpu().numpy(), gA_w) < tol and _relmax(gB.cpu().numpy(), gB_w) < tol


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("rank", [40, 64, 80])
def test_ranks_above_32_run_as_groups(rank, layout, dtype):
    M, fin, fout, s = 300, 256, 384, 64 / rank
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    tol = 1e-2 if dtype == "bf16" else 1e-5
    x, gy, base, gxb, A, B = _inputs(M, fin, fout, rank, seed=rank + layout, layout=layout)
    dx, dgy = torch.from_numpy(x).to(DEV).to(td), torch.from_numpy(gy).to(DEV).to(td)
    dA, dB = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    results = []
    for packed in (None, Fn.pack_operands(dA, dB, layout, dtype=td)):
        for saved in (True, False):
            y = torch.from_numpy(base).to(DEV).to(td)
            tT = Fn.lora_fwd_(dx, dA, dB, y, s, layout, save_t=saved, packed=packed)
            gx = torch.from_numpy(gxb).to(DEV).to(td)
            gA, gB = torch.full_like(dA, 3.0), torch.full_like(dB, 3.0)
            Fn.lora_bwd_(dgy, dx, tT, dA, dB, gx, gA, gB, s, layout, accumulate=False, packed=packed)
            results.append((y, gx, gA, gB))
    y, gx, gA, gB = results[0]
    want_y = base + O.adapter_delta(x, A, B, s, layout, acc_dtype=np.float64)
    gx_l, gA_w, gB_w = O.adapter_backward(gy, x, A, B, s, layout, acc_dtype=np.float64)
    assert _relmax(y.float().cpu().numpy(), want_y) < tol
    assert _relmax(gx.float().cpu().numpy(), gxb + gx_l) < tol
    assert _relmax(gA.cpu().numpy(), gA_w) < tol and _relmax(gB.cpu().numpy(), gB_w) < tol
    for other in results[1:]:       # operand images held by the caller / t recomputed: bit-identical
        for a, b in zip(results[0], other):
            assert torch.equal(a, b)


def test_modules_accept_rank_64_and_validate_rank():
    import lora_layers as L
    from sam3_lora_amd import _ffi
    lin = torch.nn.Linear(128, 256)
    mod = L.LoRALinear(lin, rank=64, alpha=128).to(DEV)
    with torch.no_grad():
        mod.lora.lora_B.normal_(0, 0.02)
    x = torch.randn(50, 128, device=DEV, requires_grad=True)
    y = mod(x)
    y.sum().backward()
    ref = torch.nn.functional.linear(x, lin.weight.to(DEV), lin.bias.to(DEV)) + (x @ mod.lora.lora_A @ mod.lora.lora_B) * 2.0
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4)
    assert mod.lora.lora_A.grad.shape == (128, 64) and torch.isfinite(mod.lora.lora_A.grad).all()
    with pytest.raises(ValueError, match="rank"):
        L.LoRALayer(16, 16, rank=0)
    with pytest.raises(ValueError, match="rank"):
        L.LoRALayer(16, 16, rank=_ffi.MAX_RANK + 1)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_direct_grad_accumulation_equals_autograd_accumulation(dtype):
    """enable_direct_grad_accumulation: the kernel adds into param.grad (accumulate = 1) -- same values as the gradients
    autograd would have accumulated, two steps in a row, single Linears (one of them used twice in the graph) and the
    fused MLP node; post-accumulate hooks still fire exactly once per parameter and backward."""
    import lora_layers as L
    from sam3_lora_amd.vit import Mlp
    td = torch.bfloat16 if dtype == "bf16" else torch.float32

    def build():
        torch.manual_seed(3)
        mlp = Mlp(64, 128)
        mlp.fc1, mlp.fc2 = L.LoRALinear(mlp.fc1, rank=8, alpha=16), L.LoRALinear(mlp.fc2, rank=8, alpha=16)
        head = L.LoRALinear(torch.nn.Linear(64, 64), rank=4, alpha=8)
        net = torch.nn.Sequential(mlp, head, head)          # `head` is used twice: final only after its second backward
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, L.LoRALayer):
                    m.lora_B.normal_(0, 0.05)
        net.to(DEV)
        for m in net.modules():
            if isinstance(m, torch.nn.Linear):
                m.to(td)
                m.weight.requires_grad_(False)
                m.bias.requires_grad_(False)
        return net

    x = torch.randn(3, 50, 64, device=DEV).to(td)
    grads = []
    for direct in (False, True):
        net = build()
        params = [p for p in net.parameters() if p.requires_grad]
        seen = []
        Fn.enable_direct_grad_accumulation(direct)
        try:
            for p in params:
                p.grad = torch.zeros_like(p)
                p.register_post_accumulate_grad_hook(seen.append)
            for _ in range(2):                      # two accumulating backward passes
                net(x.clone().requires_grad_(True)).float().pow(2).mean().backward()
        finally:
            Fn.enable_direct_grad_accumulation(False)
        grads.append([p.grad.clone() for p in params])
        assert len(seen) == 2 * len(params) and {id(p) for p in seen} == {id(p) for p in params}
    for a, b in zip(*grads):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (a - b).abs().max()


def test_batched_pack_equals_per_adapter_pack_and_repack_adapters():
    """sam3_lora_pack_many: byte-identical to sam3_lora_pack per adapter (mixed shapes and ranks, both dtypes); and
    functional.repack_adapters refreshes every module cache after an optimizer-like update."""
    import lora_layers as L
    g = torch.Generator(device=DEV).manual_seed(0)
    shapes = [(1024, 4736, 16), (4736, 1024, 16), (256, 256, 4), (64, 128, 40), (128, 64, 32)] * 5      # 25 adapters: 2 launches
    pairs = [(torch.randn(fi, r, device=DEV, generator=g), torch.randn(r, fo, device=DEV, generator=g)) for fi, fo, r in shapes]
    for td in (torch.bfloat16, torch.float32):
        # blobs carry alignment padding that no kernel writes: compare on zero-initialised buffers
        sizes = [Fn.pack_operands(A, B, 0, dtype=td).numel() for A, B in pairs]
        many = Fn.pack_operands_many(pairs, 0, dtype=td, outs=[torch.zeros(n, dtype=torch.uint8, device=DEV) for n in sizes])
        for (A, B), blob, n in zip(pairs, many, sizes):
            assert torch.equal(blob, Fn.pack_operands(A, B, 0, dtype=td, out=torch.zeros(n, dtype=torch.uint8, device=DEV)))
    net = torch.nn.Sequential(*[L.LoRALinear(torch.nn.Linear(64, 64), rank=8, alpha=16) for _ in range(3)]).to(DEV)
    with torch.no_grad():
        for m in net:
            m.lora.lora_B.normal_(0, 0.1)
    x = torch.randn(10, 64, device=DEV)
    y0 = net(x)
    blobs0 = [m.lora._packed._held[Fn.DT_F32][1] for m in net]
    with torch.no_grad():
        for m in net:
            m.lora.lora_B.mul_(0.5)
    assert Fn.repack_adapters(net) == 3
    for m, b0 in zip(net, blobs0):
        stamp, blob = m.lora._packed._held[Fn.DT_F32]
        assert blob is b0 and stamp[1] == m.lora.lora_A._version and stamp[3] == m.lora.lora_B._version    # refreshed in place
    y1 = net(x)
    for m, b0 in zip(net, blobs0):
        assert m.lora._packed._held[Fn.DT_F32][1] is b0            # the forward found fresh images: no lazy re-pack
    fresh = torch.nn.Sequential(*[L.LoRALinear(m.original_layer, rank=8, alpha=16) for m in net]).to(DEV)
    with torch.no_grad():
        for a, b in zip(fresh, net):
            a.lora.lora_A.copy_(b.lora.lora_A), a.lora.lora_B.copy_(b.lora.lora_B)
    assert torch.equal(fresh(x), y1) and not torch.equal(y0, y1)
