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
/* amax state of one tensor role: SAM3_FP8_AMAX_SLOTS slots, slot s at float index s * SAM3_FP8_AMAX_STRIDE (one 128-byte line
 * per slot: atomics on the same line serialise like atomics on the same address) -- an array of SAM3_FP8_AMAX_FLOATS floats.
 * A writer raises ONE slot (at most one atomic per workgroup / wave, spread over the slots), a reader takes the maximum of
 * all of them.  Float 1 of the array (inside slot 0's line) carries the EFFECTIVE amax the call scaled with -- the protocol's short
 * memory, below; the other floats between the slots are never read or written by the kernels. */
#define SAM3_FP8_AMAX_SLOTS 64
#define SAM3_FP8_AMAX_STRIDE 32
#define SAM3_FP8_AMAX_FLOATS (SAM3_FP8_AMAX_SLOTS * SAM3_FP8_AMAX_STRIDE)

const char* sam3_fp8_last_error(void);

/*
 *   observed   = max_s amax_in[s]                                    (what the PREVIOUS call on this role gathered)
 *   eff        = observed > 0 ? max(observed, 0.5 * amax_in[1]) : amax_in[1]     (amax_in[1]: the previous call's eff; the range shrinks at
 *                                          most 2x per call and grows at once; an empty observation carries the range over)
 *   scale      = max(eff, 2^-24) / fmt_max                           (written to *scale_out: the dequantisation factor; eff to amax_out[1]).
 *                eff == 0 (nothing observed yet at all): *scale_out is left as the caller initialised it (1) and used
 *   out[i]     = fp8( x[i] / scale )       round-to-nearest-even; a finite value beyond the format's range saturates to
 *                                          +-fmt_max; NaN stays NaN; +-Inf leaves as the format's non-finite encoding
 *                                          (e5m2: Inf; e4m3fn has none: NaN)
 *   amax_out[slot s] = max(amax_out[slot s], max_i |x[i]| over the FINITE elements slot s's workgroups saw)   (caller zeroes it)
 * amax_in / amax_out: SAM3_FP8_AMAX_FLOATS floats each (layout above).  x: n elements, bf16 (src_dtype 0) or fp32 (1), 16-byte aligned,
 * n % 16 == 0.  out: n bytes, 16-byte aligned.
 *
 * The same protocol is carried by the PRODUCING kernels of this library, which then write the fp8 image beside their bf16
 * output and this pass disappears: sam3_lora_fwd_act_q8 / sam3_lora_bwd_act_q8 (sam3_lora_amd.h: GELU(h) for fc2's GEMM,
 * the pre-activation gradient for fc1's input-gradient GEMM) and sam3_vit_layernorm_fwd_q8 (sam3_vit_amd.h: the inputs of
 * the qkv and fc1 GEMMs).
 */
int sam3_fp8_quantize(const void* x, void* out, const float* amax_in, float* amax_out, float* scale_out, int64_t n,
                      int src_dtype, int fmt, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAM3_FP8_AMD_H */
