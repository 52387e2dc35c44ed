/*
 * sam3_vit_amd -- helper kernels for the ViT trunk that HOSTS the LoRA adapters (not part of the LoRA
 * boundary in sam3_lora_amd.h; same library).  Device pointers, row-major, `stream` = hipStream_t.
 * Return 0 on success, -22 on bad arguments, -5 on a launch error.
 */
#ifndef SAM3_VIT_AMD_H
#define SAM3_VIT_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/*
 * qkv split + 2-D axial RoPE in one pass.  Replaces sam3/model/vitdet.py:466-471 (reshape/permute of the
 * fused qkv) and :68-90 (apply_rotary_enc).  qkv [B, L, 3, H, D] -> q, k (rotated), v: [B, L, H, D] contiguous.
 * cos_t / sin_t: fp32 [L, D/2].  dtype 0 = bf16, 1 = fp32.  D % 8 == 0.
 */
int sam3_vit_qkv_rope_fwd(const void* qkv, const float* cos_t, const float* sin_t, void* q, void* k, void* v,
                          int64_t B, int L, int H, int D, int dtype, void* stream);

/* gradient of the above: gq/gk/gv are [B, H, L, D]-shaped views with element strides (sb, sh, sl, 1). */
int sam3_vit_qkv_rope_bwd(const void* gq, const void* gk, const void* gv, int64_t sb, int64_t sh, int64_t sl,
                          const float* cos_t, const float* sin_t, void* gqkv, int64_t B, int L, int H, int D,
                          int dtype, void* stream);

/*
 * Window-attention forms (vitdet.py:93-141 window_partition / window_unpartition, :597-613 Block.forward): the token
 * permutation is folded into the kernels that touch the rows anyway, so no partition / unpartition copy runs.
 *   qkv_rope_win_*: qkv / gqkv rows are in IMAGE order [B_img, Hh, Ww, 3*H*D]; q, k, v (and their gradients) are in
 *   WINDOW order: B = B_img * (Hh/ws) * (Ww/ws) windows of L = ws*ws tokens.  ws == 0: no windows (identity).
 *   win_residual: y = x + scale[img] * unpartition(h) in one pass (scale = stochastic-depth mask / keep, or NULL);
 *   backward != 0 computes gh = scale[img] * partition(gy) from x := gy (the gradient of x is gy itself).
 */
int sam3_vit_qkv_rope_win_fwd(const void* qkv, const float* cos_t, const float* sin_t, void* q, void* k, void* v,
                              int64_t B, int L, int H, int D, int ws, int Hh, int Ww, int dtype, void* stream);
int sam3_vit_qkv_rope_win_bwd(const void* gq, const void* gk, const void* gv, int64_t sb, int64_t sh, int64_t sl,
                              const float* cos_t, const float* sin_t, void* gqkv, int64_t B, int L, int H, int D,
                              int ws, int Hh, int Ww, int dtype, void* stream);
int sam3_vit_win_residual(const void* x, const void* h, const float* scale, void* y, int64_t B_img, int Hh, int Ww,
                          int C, int ws, int backward, int dtype, void* stream);

/*
 * LayerNorm over the last dimension of x[M, C] with frozen affine parameters (vitdet.py:563-571, nn.LayerNorm eps 1e-5):
 * one pass each way, fp32 statistics.  gamma / beta have the activation dtype; C % 8 == 0, C <= 4096.  The backward
 * produces the input gradient only (the parameters are frozen under LoRA).
 */
int sam3_vit_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                           int64_t M, int C, float eps, int dtype, void* stream);
/* fp8 frozen-W mode: the forward also writes y's fp8 image for the frozen GEMM that consumes it (the qkv / fc1 inputs), with
 * the delayed-scaling protocol of sam3_fp8_amd.h (amax_in / amax_out: SAM3_FP8_AMAX_FLOATS floats each: SAM3_FP8_AMAX_SLOTS slots, one 128-byte line per slot).  bf16 only (-95 else). */
int sam3_vit_layernorm_fwd_q8(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                              int64_t M, int C, float eps, int dtype, void* q8_out, int64_t ldq, int fmt, const float* amax_in,
                              float* amax_out, float* scale_out, void* stream);
int sam3_vit_layernorm_bwd(const void* gy, const void* x, const void* gamma, const float* mean, const float* rstd,
                           void* gx, int64_t M, int C, int dtype, void* stream);
/* the same plus the gradient that arrives on the skip path around the norm (x feeds both the norm and the residual):
 * gx = add + LN'(gy) in one pass; add may be NULL. */
int sam3_vit_layernorm_bwd_add(const void* gy, const void* x, const void* gamma, const float* mean, const float* rstd,
                               const void* add, void* gx, int64_t M, int C, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
