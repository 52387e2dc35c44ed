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

#ifdef __cplusplus
}
#endif
#endif
