/*
 * sam3_lora_amd -- C-ABI of the MI355X (gfx950) LoRA adapter hot path.
 *
 * This is the drop-in boundary for the ONE path this library accelerates: the LoRA branch
 * of an adapted nn.Linear inside the SAM3 image-model training step,
 *
 *     y = W x + b  +  (alpha/r) * B A drop(x)          and its backward into gx, gA, gB.
 *
 * The reference (Sompote/sam3_lora) is 100 % Python and has no FFI seam for this path; each
 * entry point below names the reference code whose arithmetic it replaces (paths relative to
 * the reference checkout).  The host-side mirrors of the reference modules
 * (sam3_lora_amd/lora_layers.py, sam3_lora_amd/lora/) bind these with ctypes; INTEGRATION.md
 * shows the binding a maintainer would add to the reference itself.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; matrices are row-major, leading
 *     dimensions (ld*) are in ELEMENTS; activation base pointers and ld*sizeof(elem) must be
 *     16-byte aligned; in_features and out_features must be multiples of 8.
 *   - `stream` is a hipStream_t (passed as void* so that this header needs no HIP include).
 *     Nothing here synchronises the stream or the device, allocates, or frees.
 *   - scratch memory comes from the caller (`workspace`); size it with the *_workspace_bytes
 *     functions.  It may be reused by the next call on the same stream.
 *   - return value: 0 on success, a negative SAM3_LORA_E* code otherwise; the calling thread
 *     can read a description from sam3_lora_last_error() (thread-local).  No C++ exception
 *     crosses this boundary.  The compute entry points are thread-safe and keep no thread-affine state
 *     (PyTorch runs backward, and the activation-checkpoint recompute, on other threads); the
 *     only process-wide state belongs to the profiling / debugging aids at the end of this file
 *     (stage mask, in-situ timer, tuning knobs read once from the environment), which are not
 *     meant to be used concurrently with training.
 *   - LoRA parameters A and B are the fp32 master tensors in the layout of the reference
 *     module they come from:
 *         SAM3_LORA_LAYOUT_ROOT    (0): A[in, r],  B[r, out]   lora_layers.py:38-39
 *         SAM3_LORA_LAYOUT_PACKAGE (1): A[r, in],  B[out, r]   sam3_lora/lora/lora_layer.py:47-48
 *     1 <= r <= SAM3_LORA_MAX_RANK; ranks above 32 run as consecutive groups of 32 rank indices (one more pass over
 *     the activations per group -- the reference has no rank limit, configs/full_lora_config.yaml:12).
 *   - activation dtype (x, y, gy, gx):
 *         SAM3_LORA_BF16  bf16 data on the bf16 MFMAs, fp32 accumulation.  The operand images of A / B and the rank-r intermediates
 *                         t / gt of every rank group (<= 32 rank indices) are carried as hi + lo bf16 pairs (16 mantissa bits),
 *                         so the branch and its gradients are fp32 arithmetic on the caller's bf16 tensors -- the only bf16
 *                         roundings are those of x / gy (the caller's) and of y / gx on the way out (once per rank group).
 *                         SAM3_LORA_SINGLE_ROUND=1 in the environment: A, B, t, gt rounded to bf16 once each instead;
 *                         SAM3_LORA_HL_MAX_RANK=16: that only for groups of 17..32 rank indices (round 3's behaviour);
 *         SAM3_LORA_F32   exact fp32: fp32 operands on v_mfma_f32_16x16x4_f32, fp32 intermediates -- the arithmetic
 *                         of the reference's un-autocast training (train_sam3_lora_native.py).
 *     gA/gB are accumulated and returned in fp32 either way.
 */
#ifndef SAM3_LORA_AMD_H
#define SAM3_LORA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAM3_LORA_ABI_VERSION 6
#define SAM3_LORA_MAX_RANK 1024

#define SAM3_LORA_LAYOUT_ROOT 0
#define SAM3_LORA_LAYOUT_PACKAGE 1
/* OR-ed into `layout` of sam3_lora_fwd / sam3_lora_bwd: `A` is the blob written by sam3_lora_pack for the current
 * values of A and B (and `B` is ignored).  The low bits still name the layout of gA_accum / gB_accum. */
#define SAM3_LORA_PREPACKED 0x100

#define SAM3_LORA_BF16 0
#define SAM3_LORA_F32 1

#define SAM3_LORA_OK 0
#define SAM3_LORA_EINVAL (-22)     /* bad argument (shape, alignment, layout, dtype, rank) */
#define SAM3_LORA_ENOMEM (-12)     /* workspace too small */
#define SAM3_LORA_ENOTSUP (-95)    /* valid request this build does not implement */
#define SAM3_LORA_ELAUNCH (-5)     /* HIP reported a launch error */

/* ABI version of the loaded library (== SAM3_LORA_ABI_VERSION of the header it was built from). */
int sam3_lora_abi_version(void);

/* Thread-local description of the last error returned to the calling thread ("" if none). */
const char* sam3_lora_last_error(void);

/* Bytes of the `tT` tensor that sam3_lora_fwd can emit for the backward: per rank group t = drop(x) A_c as bf16
 * [r_pad, M_pad] in MFMA fragment order (SAM3_LORA_BF16) or fp32 [M_pad, r_pad] row-major (SAM3_LORA_F32). */
size_t sam3_lora_saved_t_bytes(int64_t M, int rank, int dtype);

size_t sam3_lora_fwd_workspace_bytes(int64_t M, int in_features, int out_features, int rank, int dtype);
size_t sam3_lora_bwd_workspace_bytes(int64_t M, int in_features, int out_features, int rank, int dtype);

/*
 * Operand images.  Every fwd/bwd call first converts the fp32 masters into the MFMA operand images it needs (bf16 for
 * SAM3_LORA_BF16 activations, zero-padded fp32 for SAM3_LORA_F32; one small launch).  A and B only change at the optimizer step, while a training step calls fwd, the
 * activation-checkpoint recompute fwd and bwd on the same values: sam3_lora_pack writes all images once into a
 * caller-held blob (sam3_lora_packed_bytes, 256-byte aligned) that the three calls then take through
 * `layout | SAM3_LORA_PREPACKED`.  The caller re-packs after A or B changed; results are bit-identical either way.
 */
size_t sam3_lora_packed_bytes(int in_features, int out_features, int rank, int dtype);
int sam3_lora_pack(const void* A, const void* B, void* packed, int in_features, int out_features, int rank,
                   int layout, int dtype, void* stream);
/* The same for `count` adapters at once (HOST tables of device pointers and of shapes; one layout and dtype): the
 * images of up to 16 adapters are written per launch.  After an optimizer step every adapter of the model is stale;
 * this re-packs the 64 adapters of the SAM3 trunk with 4 launches instead of 64. */
int sam3_lora_pack_many(int count, const void* const* A, const void* const* B, void* const* packed,
                        const int* in_features, const int* out_features, const int* rank, int layout, int dtype,
                        void* stream);

/*
 * Forward of the LoRA branch, fused with the residual add into the base output:
 *
 *     y_inout[M, out] += scaling * (drop(x)[M, in] @ A_c[in, r]) @ B_c[r, out]
 *
 * Replaces   lora_layers.py:49-55 (LoRALayer.forward) + the add in :87-91 (LoRALinear.forward)
 * and        sam3_lora/lora/lora_layer.py:62-79 (LoRALayer.forward, which materialises B@A)
 *            + the add in :142-158 (LinearWithLoRA.forward).
 * On entry y_inout holds the frozen layer's output W x + b (computed by the caller, e.g.
 * hipBLASLt through PyTorch-ROCm).  If `tT_out` is non-NULL it receives t = drop(x) @ A_c
 * (sam3_lora_saved_t_bytes) for sam3_lora_bwd; pass NULL in inference.
 *
 * drop_p > 0 selects training-mode dropout on x with a counter-based generator keyed by
 * (seed, offset, element index); the same triple must be passed to sam3_lora_bwd.
 */
int sam3_lora_fwd(const void* x, const void* A, const void* B, void* y_inout, void* tT_out,
                  int64_t M, int in_features, int out_features, int rank,
                  int64_t ldx, int64_t ldy, int layout, float scaling,
                  float drop_p, uint64_t seed, uint64_t offset, int dtype,
                  void* workspace, size_t workspace_bytes, void* stream);

/*
 * Backward of the LoRA branch (the reference relies on torch.autograd for it):
 *
 *     g  = scaling * gy                       t  = drop(x) @ A_c   (from tT_saved, or recomputed)
 *     gB_c += t^T @ g         [r, out]        gt = g @ B_c^T       [M, r]
 *     gA_c += drop(x)^T @ gt  [in, r]         gx_inout += (gt @ A_c^T) * dropmask   [M, in]
 *
 * gx_inout holds the frozen layer's input gradient gy @ W on entry (caller-computed) and the
 * full gradient on exit; pass NULL to skip it (input does not require grad).
 * gA_accum / gB_accum are fp32 tensors in the caller's `layout`; accumulate != 0 adds into
 * them (torch's `.grad +=`), accumulate == 0 overwrites.  The M-reduction is a fixed-order
 * two-stage sum: results are bit-reproducible run to run for identical inputs.
 */
int sam3_lora_bwd(const void* gy, const void* x, const void* tT_saved, const void* A, const void* B,
                  void* gx_inout, float* gA_accum, float* gB_accum,
                  int64_t M, int in_features, int out_features, int rank,
                  int64_t ldgy, int64_t ldx, int64_t ldgx, int layout, float scaling,
                  float drop_p, uint64_t seed, uint64_t offset, int dtype, int accumulate,
                  void* workspace, size_t workspace_bytes, void* stream);

/*
 * Activation fused into the rank-r update (the MLP of a transformer block: fc1 -> GELU -> fc2; SURVEY section 8f-1
 * "fuse GELU too").  The in-place update already streams the [M, out] tensor once; these variants make that pass also do
 * the elementwise work that would otherwise be its own read+write of the same tensor:
 *   sam3_lora_fwd_act: as sam3_lora_fwd, and act_out[M, out] (row pitch ldact) = act(y_inout after the update).
 *   sam3_lora_bwd_act: as sam3_lora_bwd, then gx_inout *= act'(pre_act[M, in]) -- for the layer that CONSUMES the
 *                      activation (fc2): gx_inout leaves as the gradient of the producing layer's pre-activation.
 * act: SAM3_LORA_ACT_GELU = exact (erf) GELU, torch.nn.GELU()'s default, evaluated in fp32 on the rounded tensor.
 *
 * sam3_lora_bwd_act with x == NULL: "this layer's input IS act(pre_act)" (what sam3_lora_fwd_act wrote as act_out).  The pass
 * that applies act'(pre_act) then also recomputes act(pre_act) tile by tile and contracts it with gt for gA -- the second full
 * read of the stored activation (e * M * in bytes) disappears and the caller need not keep the activation for the backward.
 * Needs tT_saved, gx_inout, and sam3_lora_bwd_act_recomputes_input(rank, dtype, drop_p) != 0 (bf16 hi + lo kernels, one rank group:
 * rank <= 32, with or without dropout -- the recomputed tile is masked with the keep bits the pass draws for gx anyway; in
 * sam3_lora_bwd_act_q8: rank <= 16 without dropout); gA then differs from the x-given form only in summation order (fp32, <= 1e-6 relative).
 */
int sam3_lora_bwd_act_recomputes_input(int rank, int dtype, float drop_p);
#define SAM3_LORA_ACT_NONE 0
#define SAM3_LORA_ACT_GELU 1
int sam3_lora_fwd_act(const void* x, const void* A, const void* B, void* y_inout, void* tT_out,
                      int64_t M, int in_features, int out_features, int rank,
                      int64_t ldx, int64_t ldy, int layout, float scaling,
                      float drop_p, uint64_t seed, uint64_t offset, int dtype,
                      void* workspace, size_t workspace_bytes, void* stream,
                      int act, void* act_out, int64_t ldact);
int sam3_lora_bwd_act(const void* gy, const void* x, const void* tT_saved, const void* A, const void* B,
                      void* gx_inout, float* gA_accum, float* gB_accum,
                      int64_t M, int in_features, int out_features, int rank,
                      int64_t ldgy, int64_t ldx, int64_t ldgx, int layout, float scaling,
                      float drop_p, uint64_t seed, uint64_t offset, int dtype, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream,
                      int act, const void* pre_act, int64_t ldpre);

/*
 * fp8 frozen-W mode (include/sam3_fp8_amd.h): the same two calls, and the tensor the NEXT frozen GEMM consumes -- act_out
 * (forward: fc2's input) resp. gx_inout after the activation derivative (backward: the input of fc1's input-gradient GEMM) --
 * also leaves as an fp8 image q8_out[M, width] (row pitch ldq bytes), quantised from the bf16-ROUNDED values with the delayed
 * scale of sam3_fp8_quantize's protocol (amax_in / amax_out: SAM3_FP8_AMAX_FLOATS floats each -- SAM3_FP8_AMAX_SLOTS slots, one 128-byte line per slot, sam3_fp8_amd.h; scale_out: 1 float; fmt:
 * SAM3_FP8_E4M3 / SAM3_FP8_E5M2): bit-identical to running sam3_fp8_quantize over that tensor afterwards, without the extra
 * read + write pass.  bf16 activations, rank <= 16, act == SAM3_LORA_ACT_GELU, and (backward) drop_p == 0; SAM3_LORA_ENOTSUP
 * otherwise -- the caller then quantises separately.
 */
int sam3_lora_fwd_act_q8(const void* x, const void* A, const void* B, void* y_inout, void* tT_out,
                         int64_t M, int in_features, int out_features, int rank,
                         int64_t ldx, int64_t ldy, int layout, float scaling,
                         float drop_p, uint64_t seed, uint64_t offset, int dtype,
                         void* workspace, size_t workspace_bytes, void* stream,
                         int act, void* act_out, int64_t ldact,
                         void* q8_out, int64_t ldq, int fmt, const float* amax_in, float* amax_out, float* scale_out);
int sam3_lora_bwd_act_q8(const void* gy, const void* x, const void* tT_saved, const void* A, const void* B,
                         void* gx_inout, float* gA_accum, float* gB_accum,
                         int64_t M, int in_features, int out_features, int rank,
                         int64_t ldgy, int64_t ldx, int64_t ldgx, int layout, float scaling,
                         float drop_p, uint64_t seed, uint64_t offset, int dtype, int accumulate,
                         void* workspace, size_t workspace_bytes, void* stream,
                         int act, const void* pre_act, int64_t ldpre,
                         void* q8_out, int64_t ldq, int fmt, const float* amax_in, float* amax_out, float* scale_out);

/*
 * The adapter INSIDE the frozen GEMM: one MFMA kernel for the whole adapted Linear,
 *
 *     y_out[M, out] = x[M, in] @ W[out, in]^T + bias[out] + scaling * (drop(x) @ A_c) @ B_c         act_out = act(y_out)
 *
 * with fp32 accumulation of all three terms and ONE rounding to bf16 -- lora_layers.py:87-91 (LoRALinear.forward:
 * original_layer(x) + lora(x)) and, with act = SAM3_LORA_ACT_GELU, the fc1 -> GELU site of sam3/model/vitdet.py:585-590.
 * Against the frozen-GEMM-then-sam3_lora_fwd_act pair, [M, out] is written once per tensor and never re-read in the forward.
 * W (row pitch ldw elements) and bias (or NULL) are the frozen layer's parameters in the activation dtype; A / B / layout /
 * tT_out / dropout arguments as sam3_lora_fwd (the rank-r intermediate t = drop(x) A_c still comes from one pass over x, and
 * rides into the GEMM as one more K step against the hi + lo image of scaling * B_c).  y_out need not be initialised.
 * Supported (sam3_lora_linear_fwd_supported != 0): SAM3_LORA_BF16, rank <= 32 (one rank group), in_features a multiple of 64, out_features a
 * multiple of 8; SAM3_LORA_ENOTSUP otherwise -- the caller then runs its GEMM and sam3_lora_fwd / _fwd_act.
 */
int sam3_lora_linear_fwd_supported(int in_features, int out_features, int rank, int dtype);
size_t sam3_lora_linear_fwd_workspace_bytes(int64_t M, int in_features, int out_features, int rank, int dtype);
int sam3_lora_linear_fwd(const void* x, const void* W, const void* bias, const void* A, const void* B, void* y_out, void* tT_out,
                         int64_t M, int in_features, int out_features, int rank, int64_t ldx, int64_t ldw, int64_t ldy,
                         int layout, float scaling, float drop_p, uint64_t seed, uint64_t offset, int dtype,
                         void* workspace, size_t workspace_bytes, void* stream, int act, void* act_out, int64_t ldact);
/*
 * The same kernel in the fp8 frozen-W mode (BASELINE.json configs[4]; include/sam3_fp8_amd.h): the frozen GEMM runs on the e4m3
 * images of its operands -- x_q8[M, in] (row pitch ldxq BYTES; e.g. what the LayerNorm before the MLP wrote beside its bf16 output)
 * and w_q8[out, in] (row pitch ldwq bytes; made once, the weight is frozen) with one dequantisation scale each (DEVICE scalars:
 * x ~ *scale_x * x_q8, delayed scaling) -- on v_mfma_scale_f32_16x16x128_f8f6f4, half the operand bytes and half the matrix-pipe time
 * of the bf16 form; the LoRA branch stays bf16 / hi + lo (it reads the bf16 `x`), joins the same fp32 accumulator, and y_out / act_out
 * are rounded once.  With act = SAM3_LORA_ACT_GELU and q8_out != NULL, GELU(y_out) ALSO leaves as the fp8 image the next frozen GEMM
 * (fc2) consumes, by the protocol of sam3_lora_fwd_act_q8 (bit-identical to quantising act_out afterwards).  in_features % 128 == 0;
 * otherwise as sam3_lora_linear_fwd (same workspace size).  Replaces, in that mode, hipBLASLt's fp8 GEMM + sam3_lora_fwd_act_q8.
 */
int sam3_lora_linear_fwd_q8(const void* x, const void* x_q8, int64_t ldxq, const float* scale_x, const void* w_q8, int64_t ldwq,
                            const float* scale_w, const void* bias, const void* A, const void* B, void* y_out, void* tT_out,
                            int64_t M, int in_features, int out_features, int rank, int64_t ldx, int64_t ldy, int layout, float scaling,
                            float drop_p, uint64_t seed, uint64_t offset, int dtype, void* workspace, size_t workspace_bytes,
                            void* stream, int act, void* act_out, int64_t ldact,
                            void* q8_out, int64_t ldq, int fmt, const float* amax_in, float* amax_out, float* scale_out);

/*
 * The mirror of sam3_lora_linear_fwd for the input gradient of an adapted Linear that follows an activation (fc2 of timm's Mlp,
 * sam3/model/vitdet.py:585-590: a = GELU(h), y = fc2(a)): ONE kernel for
 *
 *     gx_out[M, in] = ( gy[M, out] @ Wt[in, out]^T  +  scaling * (gy @ B_c^T) @ A_c^T ) * act'(pre_act[M, in])
 *
 * i.e. autograd's `gy @ W` of the frozen layer plus the adapter's input gradient, times GELU'(h), fp32 accumulation and one
 * rounding.  `Wt` is the TRANSPOSED frozen weight ([in, out] row-major: the caller keeps that copy once, the weight is frozen); A / B
 * are the fp32 masters in the caller's layout (no SAM3_LORA_PREPACKED here); no dropout mask on the branch (SAM3_LORA_ENOTSUP with the
 * adapter's dropout active: use the two-pass form).  The weight gradients still come from sam3_lora_bwd with gx_inout = NULL.
 * Workspace: sam3_lora_linear_fwd_workspace_bytes(M, out_features, in_features, rank, dtype).
 * Built to decide SURVEY 8f-1's "backward mirror" with a measurement: it LOSES to hipBLASLt + sam3_lora_bwd_act (whose pass also forms
 * the layer's gA from the recomputed activation) at the benchmark's fc2 site -- DESIGN.md section 4a -- and is off by default
 * (SAM3_LORA_MIRROR=1 in the fused MLP node).
 */
int sam3_lora_linear_dgrad_act(const void* gy, const void* Wt, const void* A, const void* B, void* gx_out,
                               int64_t M, int in_features, int out_features, int rank, int64_t ldgy, int64_t ldwt, int64_t ldgx,
                               int layout, float scaling, int dtype, void* workspace, size_t workspace_bytes, void* stream,
                               int act, const void* pre_act, int64_t ldpre);

/*
 * Merge for adapter-free inference: Wm[out, in] = W[out, in] + scaling * (A_c @ B_c)^T, fp32.
 * Replaces sam3_lora/lora/lora_layer.py:81-88 (merge_weights) and :160-178.
 */
int sam3_lora_merge(const float* W, const float* A, const float* B, float* Wm,
                    int in_features, int out_features, int rank, int layout, float scaling,
                    void* stream);

/*
 * Profiling aid, not part of the training path: restrict which internal stages the NEXT calls of
 * sam3_lora_fwd / sam3_lora_bwd launch (process-wide; returns the previous mask; default all).
 * bench.py uses it to time one kernel at a time with HIP events on the caller's stream.  With a
 * partial mask the outputs of the call are meaningless.
 */
#define SAM3_LORA_STAGE_PACK 1u     /* k_pack   : fp32 A/B -> bf16 operand images              */
#define SAM3_LORA_STAGE_T1 2u       /* k_t1     : t = x.A_c (fwd) / gt = gy.B_c^T (bwd)         */
#define SAM3_LORA_STAGE_T2 4u       /* k_t2     : y += s.t.B_c (fwd) / gx += s.gt.A_c^T (bwd)   */
#define SAM3_LORA_STAGE_T3_GB 8u    /* k_t3     : gB partials = t^T.gy (+ gt partials, r <= 16) */
#define SAM3_LORA_STAGE_T3_GA 16u   /* k_t3c / k_t3 : gA partials = gt^T.x                      */
#define SAM3_LORA_STAGE_REDUCE 32u  /* fixed-order sum of the partials into gA/gB: rides on the backward's k_t2 launch (bf16, gx wanted), k_reduce otherwise */
#define SAM3_LORA_STAGE_GT_REDUCE 64u /* k_gt_reduce : chunk sum of the gt partials k_t3 emitted (r <= 16)  */
#define SAM3_LORA_STAGE_FUSED 128u  /* k_fused_linear : frozen GEMM + rank-r K step + bias + activation (sam3_lora_linear_fwd) */
#define SAM3_LORA_STAGE_T3W 256u    /* k_t3w    : backward version 2 over gy -- gB partials + gt (partials, or its images when out <= 1024) */
#define SAM3_LORA_STAGE_XGX 512u    /* retired (round 5's k_xgx, measured slower and removed in round 6): selects nothing; the value stays reserved */
#define SAM3_LORA_STAGE_ALL 0xffffffffu
unsigned sam3_lora_debug_set_stages(unsigned mask);

/* Tuning / validation knobs (SAM3_LORA_T3_GATHER, SAM3_LORA_TWO_PASS_GY, SAM3_LORA_T1_NO_SPLIT, SAM3_LORA_T1_LDS_PAD,
 * SAM3_LORA_T2_TPW, SAM3_LORA_T3_WGS, SAM3_LORA_T3E_WGS, SAM3_LORA_SINGLE_ROUND, SAM3_LORA_NO_RIDE, SAM3_LORA_FUSED_WGS,
 * SAM3_LORA_FUSED_HALF, SAM3_LORA_FUSED_PROBE, SAM3_LORA_HL_MAX_RANK, SAM3_LORA_BWD_V2,
 * SAM3_LORA_T3W_WGS, SAM3_LORA_XCD_ORDER, SAM3_LORA_GA_IN_T2, SAM3_LORA_T1_BK, SAM3_LORA_T3_COOP; INTEGRATION.md section D says what each selects) are read from the
 * environment once, at the first launch; this re-reads them (tests that flip a knob between calls).  SAM3_LORA_SINGLE_ROUND
 * changes the layout of packed blobs and saved t: blobs made before a flip must be re-packed. */
void sam3_lora_debug_reload_knobs(void);

/*
 * In-situ kernel timer (profiling aid): between prof_start and prof_stop every launch of a stage in
 * `stage_mask` is bracketed by a pair of HIP events recorded on the caller's stream, inside the real
 * fwd/bwd call sequence (up to `capacity` launches).  prof_stop synchronises those events, writes per
 * launch the elapsed microseconds, the stage bit and the streamed dimension (K for T1, N for T2/T3) and
 * returns the number of samples (or a negative error).  Not thread-safe; one profiler at a time.
 */
int sam3_lora_prof_start(unsigned stage_mask, int capacity);
int sam3_lora_prof_stop(float* us_out, int* stage_out, int* dim_out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* SAM3_LORA_AMD_H */
