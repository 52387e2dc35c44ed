"""
CPU oracle for the SAM3-LoRA adapter hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``sam3_lora_amd``) never does: it calls the HIP
kernels through the C-ABI in ``include/sam3_lora_amd.h`` and raises if the shared library
is missing.

What is restated (numpy, fp32 unless ``dtype`` says otherwise), and where it comes from in
the reference (paths relative to ``/root/reference``):

* root adapter     ``lora_layers.py:25-55``   A[in,r] ~ U(+-1/sqrt(r)), B[r,out]=0,
                                              ``(drop(x) @ A @ B) * (alpha/rank)``
* root wrapper     ``lora_layers.py:87-91``   ``original_layer(x) + lora(x)``
* package adapter  ``sam3_lora/lora/lora_layer.py:47-79``  A[r,in] ~ U(+-1/sqrt(in)), B[out,r]=0,
                                              ``F.linear(drop(x), B @ A) * (alpha/rank)``
* package wrapper  ``sam3_lora/lora/lora_layer.py:142-158`` ``linear(x) + lora(x)``
* merge            ``sam3_lora/lora/lora_layer.py:81-88,160-178``
* injection rules  ``lora_layers.py:174-198`` (root, basename match + component gates) and
                   ``sam3_lora/lora/lora_utils.py:59-92`` (package, substring match)
* backward         the reference has no hand-written backward (autograd); the formulas here are
                   the analytic gradients of the two forward expressions above and are pinned
                   against ``torch.autograd`` run on the reference's own modules by
                   ``tests/golden/make_golden.py`` (fixtures in ``tests/golden/*.npz``).

Parity status: PINNED.  The reference's own tests hold no known-answer vector for this path
(SURVEY.md section 4); the oracle is instead checked against outputs of the reference itself,
imported in the build container, on seeded inputs (see ``tests/test_oracle_golden.py``).

Layout codes (shared with the C-ABI):
    LAYOUT_ROOT    = 0 : A[in, r],  B[r, out]    (``lora_layers``)
    LAYOUT_PACKAGE = 1 : A[r, in],  B[out, r]    (``sam3_lora.lora``)
"""
from __future__ import annotations

import math
import re
from typing import Dict, Iterable, List, Optional, Sequence, Set, Tuple

import numpy as np

LAYOUT_ROOT = 0
LAYOUT_PACKAGE = 1


# --------------------------------------------------------------------------------------
# bf16 helpers (round-to-nearest-even, the rounding the HIP kernels and torch use)
# --------------------------------------------------------------------------------------
def bf16_round(a: np.ndarray) -> np.ndarray:
    """fp32 -> nearest-even bf16 -> fp32 (NaN preserved)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    r = ((u + 0x7FFF + lsb) >> 16) << 16
    out = r.astype(np.uint32).view(np.float32).reshape(a.shape)
    nan = np.isnan(a)
    if nan.any():
        out = out.copy()
        out[nan] = np.nan
    return out


# --------------------------------------------------------------------------------------
# dropout keep-mask of the HIP kernels (integer arithmetic -> bit-exact target).
# The reference uses torch's nn.Dropout stream (lora_layers.py:42,54), which cannot be matched
# across devices; the kernels therefore define their own counter-based stream and
# THIS function is its specification.  What is pinned to the reference is the SEMANTICS
# (mask applied to x before A only, kept values scaled by 1/(1-p), gradient masked the same way):
# see adapter_delta/adapter_backward with drop_scale_mask and tests/test_oracle_golden.py.
# --------------------------------------------------------------------------------------
def _fmix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h.astype(np.uint64) * np.uint64(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h.astype(np.uint64) * np.uint64(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def dropout_threshold(p: float) -> int:
    """thr = round-half-even(p * 65536) clamped to [1, 65536]; an element is kept iff u16 >= thr."""
    return int(min(max(int(np.rint(np.float32(p) * np.float32(65536.0))), 1), 65536))


def dropout_keep(M: int, width: int, p: float, seed: int, offset: int = 0) -> np.ndarray:
    """bool[M, width]: keep mask of element e = row*width + col.

    counter c = e >> 1 (two 16-bit draws per 32-bit hash);  h = fmix32(lo32(c) ^ k0 ^ hi32(c)*0x85EBCA6B)
    with k0 = lo32(seed ^ seed>>32) ^ lo32(offset)*0x9E3779B9 ^ hi32(offset); even e takes the low
    half of h, odd e the high half.
    """
    seed, offset = int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1)
    k0 = ((seed ^ (seed >> 32)) & 0xFFFFFFFF) ^ (((offset & 0xFFFFFFFF) * 0x9E3779B9) & 0xFFFFFFFF) ^ (offset >> 32)
    e = np.arange(M * width, dtype=np.uint64)
    c = e >> np.uint64(1)
    lo = (c & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = ((c >> np.uint64(32)).astype(np.uint64) * np.uint64(0x85EBCA6B)).astype(np.uint32)
    h = _fmix32(lo ^ np.uint32(k0) ^ hi)
    u16 = np.where((e & np.uint64(1)) == 0, h & np.uint32(0xFFFF), h >> np.uint32(16))
    return (u16 >= np.uint32(dropout_threshold(p))).reshape(M, width)


def dropout_scale_mask(M: int, width: int, p: float, seed: int, offset: int = 0) -> np.ndarray:
    """The multiplier nn.Dropout(p) applies in training: 0 or 1/(1-p) (0 everywhere for p == 1)."""
    keep = dropout_keep(M, width, p, seed, offset)
    inv = np.float32(1.0 / (1.0 - p)) if p < 1.0 else np.float32(0.0)
    return keep.astype(np.float32) * inv


# --------------------------------------------------------------------------------------
# canonical views: every formula below works on A_c[in, r], B_c[r, out]
# --------------------------------------------------------------------------------------
def _canon(A: np.ndarray, B: np.ndarray, layout: int) -> Tuple[np.ndarray, np.ndarray]:
    if layout == LAYOUT_ROOT:
        return A, B
    if layout == LAYOUT_PACKAGE:
        return A.T, B.T
    raise ValueError(f"unknown layout {layout}")


def _uncanon_grads(gA_c: np.ndarray, gB_c: np.ndarray, layout: int):
    if layout == LAYOUT_ROOT:
        return gA_c, gB_c
    return np.ascontiguousarray(gA_c.T), np.ascontiguousarray(gB_c.T)


def scaling_of(alpha: float, rank: int) -> float:
    """``self.scaling = alpha / rank`` -- lora_layers.py:35, lora_layer.py:44."""
    return alpha / rank


def init_bound(layout: int, in_features: int, rank: int) -> float:
    """Bound of kaiming_uniform_(a=sqrt(5)) on lora_A.

    gain = sqrt(2/(1+5)) = sqrt(1/3); bound = gain*sqrt(3/fan_in) = 1/sqrt(fan_in).
    torch's fan_in is ``tensor.size(1)``: root A is [in, r] -> fan_in = r
    (lora_layers.py:38,45); package A is [r, in] -> fan_in = in (lora_layer.py:47,58).
    """
    fan_in = rank if layout == LAYOUT_ROOT else in_features
    return 1.0 / math.sqrt(fan_in)


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------
def adapter_delta(x, A, B, scaling: float, layout: int, drop_scale_mask=None,
                  acc_dtype=np.float32) -> np.ndarray:
    """LoRA branch only.

    root    : ``self.dropout(x) @ self.lora_A @ self.lora_B * scaling``  lora_layers.py:54-55
    package : ``F.linear(drop(x), lora_B @ lora_A) * scaling``           lora_layer.py:73-79
    Mathematically identical; the association order differs (package materialises B@A).
    ``drop_scale_mask`` is the elementwise multiplier nn.Dropout applies in training
    (0 or 1/(1-p)); None means Identity (p == 0 or eval mode).
    """
    x2 = np.asarray(x, dtype=acc_dtype).reshape(-1, x.shape[-1])
    if drop_scale_mask is not None:
        x2 = x2 * np.asarray(drop_scale_mask, dtype=acc_dtype).reshape(x2.shape)
    A = np.asarray(A, dtype=acc_dtype)
    B = np.asarray(B, dtype=acc_dtype)
    if layout == LAYOUT_ROOT:
        d = (x2 @ A) @ B
    else:
        d = x2 @ (B @ A).T
    d = d * acc_dtype(scaling)
    return d.reshape(*x.shape[:-1], d.shape[-1])


def base_linear(x, W, b, acc_dtype=np.float32) -> np.ndarray:
    """``nn.Linear``: x @ W.T + b with W[out, in]."""
    x2 = np.asarray(x, dtype=acc_dtype).reshape(-1, x.shape[-1])
    y = x2 @ np.asarray(W, dtype=acc_dtype).T
    if b is not None:
        y = y + np.asarray(b, dtype=acc_dtype)
    return y.reshape(*x.shape[:-1], y.shape[-1])


def lora_linear_forward(x, W, b, A, B, scaling: float, layout: int, drop_scale_mask=None,
                        acc_dtype=np.float32) -> np.ndarray:
    """``original_layer(x) + lora(x)`` -- lora_layers.py:91 / lora_layer.py:153-156."""
    return base_linear(x, W, b, acc_dtype) + adapter_delta(
        x, A, B, scaling, layout, drop_scale_mask, acc_dtype)


# --------------------------------------------------------------------------------------
# backward (analytic gradients of the expressions above; pinned against torch.autograd)
# --------------------------------------------------------------------------------------
def adapter_backward(gy, x, A, B, scaling: float, layout: int, drop_scale_mask=None,
                     acc_dtype=np.float32):
    """Gradients of ``adapter_delta`` wrt x, A, B given upstream ``gy``.

    With xd = x*mask, t = xd @ A_c, g = scaling*gy:
        gB_c = t.T @ g            [r, out]
        gt   = g @ B_c.T          [M, r]
        gA_c = xd.T @ gt          [in, r]
        gx   = (gt @ A_c.T)*mask  [M, in]
    Returned gA/gB are in the caller's ``layout``.
    """
    A_c, B_c = _canon(np.asarray(A, dtype=acc_dtype), np.asarray(B, dtype=acc_dtype), layout)
    x2 = np.asarray(x, dtype=acc_dtype).reshape(-1, x.shape[-1])
    g = np.asarray(gy, dtype=acc_dtype).reshape(-1, gy.shape[-1]) * acc_dtype(scaling)
    if drop_scale_mask is not None:
        m = np.asarray(drop_scale_mask, dtype=acc_dtype).reshape(x2.shape)
        xd = x2 * m
    else:
        m = None
        xd = x2
    t = xd @ A_c
    gB_c = t.T @ g
    gt = g @ B_c.T
    gA_c = xd.T @ gt
    gx = gt @ A_c.T
    if m is not None:
        gx = gx * m
    gA, gB = _uncanon_grads(gA_c, gB_c, layout)
    return gx.reshape(x.shape), gA, gB


def lora_linear_backward(gy, x, W, A, B, scaling: float, layout: int, drop_scale_mask=None,
                         acc_dtype=np.float32):
    """Gradient wrt x of the wrapped layer (base + adapter) and wrt A, B (W, b are frozen:
    lora_layers.py:73-76, lora_layer.py:118-121)."""
    gx_l, gA, gB = adapter_backward(gy, x, A, B, scaling, layout, drop_scale_mask, acc_dtype)
    g2 = np.asarray(gy, dtype=acc_dtype).reshape(-1, gy.shape[-1])
    gx_b = (g2 @ np.asarray(W, dtype=acc_dtype)).reshape(x.shape)
    return gx_b + gx_l, gA, gB


def merged_weight(W, A, B, scaling: float, layout: int) -> np.ndarray:
    """``linear.weight + (lora_B @ lora_A) * scaling`` -- lora_layer.py:81-88,167."""
    A_c, B_c = _canon(np.asarray(A, np.float32), np.asarray(B, np.float32), layout)
    return np.asarray(W, np.float32) + (A_c @ B_c).T * np.float32(scaling)


# --------------------------------------------------------------------------------------
# what the HIP kernels compute, restated with their intermediate roundings
# (bf16 operands, fp32 accumulate, t and gt rounded to bf16 between the two contractions).
# Used for tight GPU parity; the plain fp32 functions above are the reference semantics.
# --------------------------------------------------------------------------------------
def adapter_delta_bf16_model(x, A, B, scaling: float, layout: int) -> np.ndarray:
    A_c, B_c = _canon(np.asarray(A, np.float32), np.asarray(B, np.float32), layout)
    xb = bf16_round(np.asarray(x, np.float32).reshape(-1, x.shape[-1]))
    t = bf16_round(xb.astype(np.float64) @ bf16_round(A_c).astype(np.float64))
    d = t.astype(np.float64) @ bf16_round(B_c).astype(np.float64)
    return (d * scaling).astype(np.float32).reshape(*x.shape[:-1], B_c.shape[1])


# --------------------------------------------------------------------------------------
# injection rules (host logic; the product re-implements these, tests compare)
# --------------------------------------------------------------------------------------
def root_should_apply(module_name: str, target_modules: Iterable[str], *,
                      apply_to_vision_encoder=True, apply_to_text_encoder=True,
                      apply_to_geometry_encoder=False, apply_to_detr_encoder=True,
                      apply_to_detr_decoder=True, apply_to_mask_decoder=False) -> bool:
    """lora_layers.py:174-198: component gates by substring, ``out_proj`` never, then the
    LAST dotted component must be in ``target_modules``."""
    n = module_name
    gates = (
        (("vision_encoder", "vision_backbone"), apply_to_vision_encoder),
        (("text_encoder", "language_backbone"), apply_to_text_encoder),
        (("geometry_encoder",), apply_to_geometry_encoder),
        (("detr_encoder", "transformer.encoder"), apply_to_detr_encoder),
        (("detr_decoder", "transformer.decoder"), apply_to_detr_decoder),
        (("mask_decoder",), apply_to_mask_decoder),
    )
    for needles, enabled in gates:
        if not enabled and any(k in n for k in needles):
            return False
    base = n.split(".")[-1]
    if base == "out_proj":
        return False
    return base in set(target_modules)


_PKG_PATTERNS = (
    r".*\.self_attn\.",
    r".*\.cross_attn\.",
    r".*\.cross_attn_image\.",
    r".*\.ca_text\.",
    r".*\.linear[12]$",
    r".*\.(q|k|v|out)_proj$",
)

PKG_DEFAULT_TARGETS = ("q_proj", "k_proj", "v_proj", "out_proj", "linear1", "linear2")
PKG_ALL_TARGETS = ("q_proj", "k_proj", "v_proj", "out_proj", "linear1", "linear2",
                   "in_proj", "cross_attn", "self_attn")


def package_targets(target_modules: Optional[Sequence[str]]) -> Set[str]:
    """lora_utils.py:38-56: default set; the literal "all" replaces the set."""
    t = set(PKG_DEFAULT_TARGETS if target_modules is None else target_modules)
    if "all" in t:
        t = set(PKG_ALL_TARGETS)
    return t


def package_should_inject(name: str, targets: Set[str]) -> bool:
    """lora_utils.py:59-92: substring of the FULL name; then a regex pass whose inner test
    is ``target in pattern`` (the pattern text, not the name)."""
    for t in targets:
        if t in name:
            return True
    for pat in _PKG_PATTERNS:
        if re.match(pat, name):
            for t in targets:
                if t in pat:
                    return True
    return False


def select_linears(linear_names: Sequence[str], rule) -> List[str]:
    return [n for n in linear_names if rule(n)]
