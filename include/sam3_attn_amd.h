/*
 * sam3_attn_amd -- attention forward of the ViT trunk that hosts the adapters (same library as sam3_lora_amd.h; not part of
 * the LoRA boundary).  Replaces the scaled_dot_product_attention call of sam3/model/vitdet.py:339-515 (Attention.forward)
 * for the trunk's shapes: bf16, head dimension 64, 576-token windows and the 5184-token grid.
 *
 *   o[b, l, h, :] = sum_j softmax_j(q[b, l, h, :] . k[b, j, h, :] * scale) v[b, j, h, :]
 *   lse[b, h, l]  = log sum_j exp(q . k_j * scale)                    (fp32, natural logarithm: what PyTorch's attention
 *                                                                      backward takes together with o)
 *
 * q, k, v, o: device pointers to [B, L, H, head_dim] tensors sharing the element strides (stride_b, stride_l, stride_h, 1),
 * 16-byte aligned, strides multiples of 8.  lse: [B, H, L] contiguous.  `stream` = hipStream_t.
 * Returns 0, -22 (bad argument), -95 (shape / dtype this build has no kernel for: head_dim != 64 or dtype != 0 (bf16) --
 * the caller keeps PyTorch's kernel), -5 (launch error).
 */
#ifndef SAM3_ATTN_AMD_H
#define SAM3_ATTN_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int sam3_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int L, int H, int head_dim,
                  int64_t stride_b, int64_t stride_l, int64_t stride_h, float scale, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
