/*
 * sam3_lora_amd -- C-ABI of two helpers of the model around the adapter path (gfx950): GroupNorm (+ ReLU) on
 * channels-last maps for the mask head, and the decoder's box-relative position bias.
 *
 * Host helper of the model around the adapter path (the pixel decoder that produces
 * the mask logits the mask loss reads).  Replaces, for frozen affine parameters,
 *     sam3/model/maskformer_segmentation.py:205-222   PixelDecoder.forward: relu(GroupNorm(8, 256)(conv3x3(x)))
 * The trunk hands the neck a channels-last map, MIOpen's kernels keep that layout through every convolution of neck and
 * pixel decoder, and ATen's GroupNorm is the one operator in between that only works on NCHW: it costs a 340 MB layout
 * copy before, a 64-workgroup moments kernel (2.6 ms at [8, 256, 288, 288]) and leaves an NCHW tensor that the next
 * convolution converts back.  These kernels read and write the channels-last memory directly.
 *
 * Conventions as in sam3_lora_amd.h: device pointers owned by the caller, caller-provided workspace, `stream` is a
 * hipStream_t, nothing synchronises or allocates, 0 on success / negative code + thread-local message otherwise.
 * dtype codes: 0 = bf16, 1 = fp32.  Layout: x, y, gy, gx are [N, HW, C] row-major (= torch channels_last of
 * [N, C, H, W]); gamma, beta fp32 [C]; stats fp32 [N, G, 2] = (mean, rstd) per image and group.
 * Supported: C / G a multiple of 16 bytes' worth of elements (8 bf16 / 4 fp32), C / that vector width a power of two
 * <= 256.  All reductions are fixed-order: bit-reproducible.
 */
#ifndef SAM3_SEG_AMD_H
#define SAM3_SEG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* sam3_seg_last_error(void);

/* 0 when (C, G, dtype) is supported by the kernels, negative otherwise (callers then keep their own formulation) */
int sam3_gn_nhwc_supported(int C, int G, int dtype);

size_t sam3_gn_nhwc_workspace_bytes(int N, int64_t HW, int C, int G);

/* y = act( (x - mean) * rstd * gamma + beta ),  act = ReLU when relu != 0;  stats receives (mean, rstd) */
int sam3_gn_nhwc_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int N, int64_t HW,
                     int C, int G, float eps, int relu, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Input gradient (gamma, beta frozen).  With xhat = (x - mean) rstd, g' = gy * [act'] * gamma and, per (image, group)
 * over its m = HW * C / G elements, a = sum g', b = sum g' xhat:
 *     gx = rstd * ( g' - a / m - xhat * b / m )
 * The ReLU mask is recomputed from x (y is not needed).
 */
int sam3_gn_nhwc_bwd(const void* x, const void* gy, const float* gamma, const float* beta, const float* stats, void* gx,
                     int N, int64_t HW, int C, int G, int relu, int dtype, void* workspace, size_t workspace_bytes,
                     void* stream);

/*
 * Decoder: box-relative position bias of the image cross-attention (sam3/model/decoder.py:357-407, boxRPB "log" or the
 * plain offsets).  Forward only: the reference boxes are detached and the two MLPs frozen in LoRA training.
 *   boxes  fp32 [Q, B, 4] cxcywh in [0, 1];  mlp_x / mlp_y: {W1 [hidden, 2], b1 [hidden], W2 [heads, hidden], b2 [heads]}
 *   in the layer dtype (bf16 / fp32);  out [B, heads, Q + presence_row, H * W] in the layer dtype, written once:
 *   out[b, h, q, y * W + x] = mlp_y(offsets of row y / H to the box's y-edges)[h] + mlp_x(offsets of column x / W)[h],
 *   an all-zero row first when presence_row != 0.  Offsets are log-scaled (sign(8 d) log2(|8 d| + 1) / 3) when log_scale.
 */
int sam3_rpb_bias_fwd(const float* boxes, const void* const* mlp_x, const void* const* mlp_y, void* out, int B, int Q, int H,
                      int W, int hidden, int heads, int presence_row, int log_scale, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAM3_SEG_AMD_H */
