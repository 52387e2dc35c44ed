/*
 * sam3_lora_amd -- C-ABI of the fp8 activation quantiser (gfx950) used by the "fp8 frozen-W" mode of the frozen
 * Linears (BASELINE.json configs[4]): the frozen weight is stored once as OCP e4m3 with a
 * per-tensor scale, the base GEMMs run on hipBLASLt's fp8 MFMA kernels (through torch._scaled_mm), the LoRA branch stays
 * bf16 on the adapter kernels.  What is hand-written here is the step that would otherwise dominate: turning the bf16
 * activation into fp8 -- one read, one half-size write, the running amax for the next step gathered in the same pass
 * ("delayed scaling": this call scales with the amax the PREVIOUS call on this tensor role observed).
 *
 * The reference has no fp8 path; this mode is a build-side extension with parity measured against the bf16 build.
 * Conventions as in sam3_lora_amd.h (device pointers, hipStream_t as void*, 0 / negative code + thread-local text).
 */
#ifndef SAM3_FP8_AMD_H
#define SAM3_FP8_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAM3_FP8_E4M3 0   /* OCP e4m3fn, max 448   (activations, weights)   */
#define SAM3_FP8_E5M2 1   /* OCP e5m2,   max 57344 (gradients)              */

const char* sam3_fp8_last_error(void);

/*
 *   scale      = max(*amax_in, 2^-24) / fmt_max                      (written to *scale_out: the dequantisation factor)
 *   out[i]     = fp8( clamp(x[i] / scale, -fmt_max, fmt_max) )       round-to-nearest-even, saturating
 *   *amax_out  = max(*amax_out, max_i |x[i]|)                        (caller zeroes it beforehand)
 * x: n elements, bf16 (src_dtype 0) or fp32 (1), 16-byte aligned, n % 16 == 0.  out: n bytes, 16-byte aligned.
 */
int sam3_fp8_quantize(const void* x, void* out, const float* amax_in, float* amax_out, float* scale_out, int64_t n,
                      int src_dtype, int fmt, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAM3_FP8_AMD_H */
