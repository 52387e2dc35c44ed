/*
 * sam3_lora_amd -- C-ABI of the mask-loss kernels (gfx950): the step right after the LoRA adapter path in the SAM3
 * training step.
 *
 * Replaces, for the matched instances of one decoder output,
 *     sam3/train/loss/loss_fns.py:679-707   Masks.get_loss: bilinear upsample of the mask logits to the target size
 *     sam3/train/loss/loss_fns.py:159-176   sigmoid_focal_loss (the pure-PyTorch formula; the reference's default is a
 *                                           Triton kernel, sam3/train/loss/sigmoid_focal_loss.py:75-208)
 *     sam3/train/loss/loss_fns.py:79-123    dice_loss
 * by one pass over the target resolution that never materialises the upsampled tensor.
 *
 * Conventions as in sam3_lora_amd.h: device pointers owned by the caller, caller-provided workspace, `stream` is a
 * hipStream_t, nothing synchronises or allocates, 0 on success / negative code + thread-local message otherwise.
 * dtype codes: 0 = bf16, 1 = fp32 (logits `src` and, separately, the gradient `gsrc`).
 *
 *   src   [N, h, w]   mask logits of the matched queries (contiguous)
 *   tgt   [N, H, W]   ground-truth masks, one byte per pixel (torch.bool storage), non-zero = foreground
 *   x(n, Y, X) = bilinear(src[n], align_corners = False) at the target resolution, fp32
 *   p = sigmoid(x)
 */
#ifndef SAM3_LOSS_AMD_H
#define SAM3_LOSS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* sam3_loss_last_error(void);

size_t sam3_mask_loss_workspace_bytes(int N, int H, int W);

/*
 * sums[n] = ( sum_px focal(x, t),  sum_px p t,  sum_px p,  sum_px t )      fp32 [N, 4]
 *   focal(x, t) = a_t * BCEwithLogits(x, t) * (1 - p_t)^gamma,  p_t = p t + (1 - p)(1 - t),
 *   a_t = alpha t + (1 - alpha)(1 - t) (alpha < 0: no class weighting).
 * The losses of the reference follow on the host side of the boundary:
 *   loss_mask = sum_n sums[n][0] / (H W) / num_boxes
 *   loss_dice = sum_n (1 - (2 sums[n][1] + 1) / (sums[n][2] + sums[n][3] + 1)) / num_boxes
 * Fixed-order reductions: bit-reproducible.
 */
int sam3_mask_loss_fwd(const void* src, const void* tgt, float* sums, int N, int h, int w, int H, int W,
                       float alpha, float gamma, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Gradient of any scalar L(sums) with respect to the logits:  coef[n] = dL/dsums[n]  (fp32 [N, 4]; the 4th entry is
 * unused -- the target is constant),
 *   gsrc[n, i, j] = sum_{Y, X} wy(Y, i) wx(X, j) * ( coef[n][0] dfocal/dx + (coef[n][1] t + coef[n][2]) p (1 - p) )(n, Y, X)
 * written (not accumulated) as a gather over the target pixels whose bilinear stencil touches (i, j): no atomics.
 */
int sam3_mask_loss_bwd(const void* src, const void* tgt, const float* coef, void* gsrc, int N, int h, int w, int H, int W,
                       float alpha, float gamma, int dtype, int grad_dtype, void* stream);

/*
 * Matched box pairs (sam3/train/loss/loss_fns.py:345-400 "Boxes", :410-470 the IoU-aware targets of "IABCEMdetr"; the
 * arithmetic of sam3/model/box_ops.py:generalized_box_iou on the diagonal): a[i], b[i] are xyxy fp32 boxes, [T, 4].
 *   out[i]  = ( IoU(a[i], b[i]),  GIoU(a[i], b[i]) )                                   fp32 [T, 2]
 *   ga[i]   = coef[i][0] * dIoU/da[i] + coef[i][1] * dGIoU/da[i]                       fp32 [T, 4]  (b is a target)
 * One launch each instead of ~35 small elementwise operators (and as many autograd nodes) per decoder output.
 */
int sam3_box_pair_fwd(const float* a, const float* b, float* out, int T, void* stream);
int sam3_box_pair_bwd(const float* a, const float* b, const float* coef, float* ga, int T, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAM3_LOSS_AMD_H */
