// sam3_lora_amd -- host-model helper kernels for the ViT trunk that hosts the adapters (gfx950).
//
// qkv split + 2-D axial RoPE in ONE pass (reference: sam3/model/vitdet.py:68-90 apply_rotary_enc via complex
// fp32 views, :466-471 the qkv reshape/permute).  On PyTorch-ROCm the unfused form costs ~10 elementwise
// passes over q and k per attention call (fp32 up-casts, complex multiply, stack/flatten, permute copies) and
// was 35 % of the trunk's training step; here q, k are rotated and q, k, v are split out of the fused qkv
// activation with one read and one write, 16 bytes per lane both ways.
//
//   qkv   [B, L, 3, H, D]  (the qkv Linear's output, row = token)       bf16 or fp32
//   q,k,v [B, L, H, D]     contiguous (callers hand .transpose(1, 2) views to SDPA -- no permute copy)
//   cos/sin [L, D/2] fp32  (adjacent element pairs (2i, 2i+1) are one complex number)
//
// backward: gqkv[b,l,0,h,:] = R(-theta) gq[b,l,h,:], same for k, copy for v; gq/gk/gv may be arbitrary
// [B, H, L, D]-shaped strided views with unit last stride (whatever SDPA's backward returns).
#include <hip/hip_runtime.h>

#include <cstdint>
#include "fp8_common.inc"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;

namespace {

__device__ __forceinline__ unsigned vpack2(float a, float b) {
    bf16x2v v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float vlo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float vhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

struct F8 {
    float v[8];
};
__device__ __forceinline__ F8 ld8(const bf16_t* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    return F8{{vlo(u.x), vhi(u.x), vlo(u.y), vhi(u.y), vlo(u.z), vhi(u.z), vlo(u.w), vhi(u.w)}};
}
__device__ __forceinline__ F8 ld8(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    return F8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
// read-once streams (the fused qkv activation, SDPA's gradients, the window-order branch output) bypass the caches
typedef __attribute__((ext_vector_type(4))) unsigned vu32x4;
typedef __attribute__((ext_vector_type(4))) float vf32x4;
#ifndef SAM3_NT_LOADS
#define SAM3_NT_LOADS 1
#endif
__device__ __forceinline__ F8 ld8_nt(const bf16_t* p) {
#if !SAM3_NT_LOADS
    return ld8(p);
#endif
    const vu32x4 u = __builtin_nontemporal_load(reinterpret_cast<const vu32x4*>(p));
    return F8{{vlo(u[0]), vhi(u[0]), vlo(u[1]), vhi(u[1]), vlo(u[2]), vhi(u[2]), vlo(u[3]), vhi(u[3])}};
}
__device__ __forceinline__ F8 ld8_nt(const float* p) {
#if !SAM3_NT_LOADS
    return ld8(p);
#endif
    const vf32x4 a = __builtin_nontemporal_load(reinterpret_cast<const vf32x4*>(p));
    const vf32x4 b = __builtin_nontemporal_load(reinterpret_cast<const vf32x4*>(p + 4));
    return F8{{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}};
}
__device__ __forceinline__ void st8(bf16_t* p, const F8& f) {
    *reinterpret_cast<uint4*>(p) =
        make_uint4(vpack2(f.v[0], f.v[1]), vpack2(f.v[2], f.v[3]), vpack2(f.v[4], f.v[5]), vpack2(f.v[6], f.v[7]));
}
__device__ __forceinline__ void st8(float* p, const F8& f) {
    *reinterpret_cast<float4*>(p) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(f.v[4], f.v[5], f.v[6], f.v[7]);
}
// rotate the 4 complex pairs of f by +theta (sign = +1) or -theta (sign = -1)
__device__ __forceinline__ F8 rot(const F8& f, const float4 c, const float4 s, float sign) {
    const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {s.x * sign, s.y * sign, s.z * sign, s.w * sign};
    F8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = f.v[2 * i], b = f.v[2 * i + 1];
        o.v[2 * i] = a * cc[i] - b * ss[i];
        o.v[2 * i + 1] = a * ss[i] + b * cc[i];
    }
    return o;
}

// Window attention (vitdet.py:93-141 window_partition / window_unpartition) is a permutation of token ROWS, and
// LayerNorm / Linear act per row, so the permutation never needs its own pass: the kernels below enumerate tokens
// in window order (window b, position l) and address the image-order tensor through img_token(); ws == 0 is the
// identity (global-attention blocks).
struct WinMap {
    int ws, Hh, Ww;      // window edge, token grid of one image (Hh % ws == 0, Ww % ws == 0)
};
__device__ __forceinline__ long long img_token(long long tok, int L, const WinMap w) {
    if (w.ws == 0) return tok;
    const int nWw = w.Ww / w.ws, nW = nWw * (w.Hh / w.ws);
    const long long bw = tok / L;
    const int l = (int)(tok % L);
    const long long img = bw / nW;
    const int wi = (int)(bw % nW);
    const int wy = wi / nWw, wx = wi % nWw, iy = l / w.ws, ix = l % w.ws;
    return img * ((long long)w.Hh * w.Ww) + (long long)(wy * w.ws + iy) * w.Ww + (wx * w.ws + ix);
}

// one thread = 8 consecutive elements of one (token, which in {q,k,v}, head); consecutive threads walk the
// 3*H*D row of a token, so both the read and the three writes are fully coalesced
template <typename T>
__global__ __launch_bounds__(256) void k_qkv_rope_fwd(const T* __restrict__ qkv, const float* __restrict__ cs,
                                                      const float* __restrict__ sn, T* __restrict__ q,
                                                      T* __restrict__ k, T* __restrict__ v, long long ntok, int L,
                                                      int H, int D, WinMap wm) {
    const int cpr = 3 * H * D / 8;   // 16-byte chunks per token row
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= ntok * cpr) return;
    const long long tok = idx / cpr;
    const int c = (int)(idx % cpr);
    const int hd8 = H * D / 8;
    const int which = c / hd8, ch = c % hd8;     // chunk inside the [H, D] slab
    const int d0 = (ch * 8) % D;                 // first element index inside the head
    F8 f = ld8_nt(qkv + img_token(tok, L, wm) * (3LL * H * D) + (long long)c * 8);
    T* dst = (which == 0 ? q : which == 1 ? k : v) + tok * ((long long)H * D) + (long long)ch * 8;
    if (which < 2) {
        const int l = (int)(tok % L);
        const float4 cc = *reinterpret_cast<const float4*>(cs + (long long)l * (D / 2) + d0 / 2);
        const float4 ss = *reinterpret_cast<const float4*>(sn + (long long)l * (D / 2) + d0 / 2);
        f = rot(f, cc, ss, 1.f);
    }
    st8(dst, f);
}

template <typename T>
__global__ __launch_bounds__(256) void k_qkv_rope_bwd(const T* __restrict__ gq, const T* __restrict__ gk,
                                                      const T* __restrict__ gv, long long sb, long long sh,
                                                      long long sl, const float* __restrict__ cs,
                                                      const float* __restrict__ sn, T* __restrict__ gqkv,
                                                      long long ntok, int L, int H, int D, WinMap wm) {
    const int cpr = 3 * H * D / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= ntok * cpr) return;
    const long long tok = idx / cpr;
    const int c = (int)(idx % cpr);
    const int hd8 = H * D / 8;
    const int which = c / hd8, ch = c % hd8;
    const int h = (ch * 8) / D, d0 = (ch * 8) % D;
    const long long b = tok / L;
    const int l = (int)(tok % L);
    const T* src = (which == 0 ? gq : which == 1 ? gk : gv) + b * sb + (long long)h * sh + (long long)l * sl + d0;
    F8 f = ld8_nt(src);
    if (which < 2) {
        const float4 cc = *reinterpret_cast<const float4*>(cs + (long long)l * (D / 2) + d0 / 2);
        const float4 ss = *reinterpret_cast<const float4*>(sn + (long long)l * (D / 2) + d0 / 2);
        f = rot(f, cc, ss, -1.f);
    }
    st8(gqkv + img_token(tok, L, wm) * (3LL * H * D) + (long long)c * 8, f);
}

// y[img token] = x[img token] + scale[img] * h[window token]   (residual add fused with window_unpartition and the
// stochastic-depth mask; scale == nullptr means 1).  backward: gh[window token] = scale[img] * gy[img token].
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void k_win_residual(const T* __restrict__ x, const T* __restrict__ h,
                                                      const float* __restrict__ scale, T* __restrict__ out,
                                                      long long ntok, int L, int C, WinMap wm) {
    const int cpr = C / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= ntok * cpr) return;
    const long long tok = idx / cpr;
    const int c = (int)(idx % cpr);
    const long long it = img_token(tok, L, wm);
    const float sc = scale ? scale[it / ((long long)wm.Hh * wm.Ww)] : 1.f;
    if (BWD) {      // x = gy (image order), out = gh (window order)
        F8 f = ld8(x + it * C + (long long)c * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) f.v[i] *= sc;
        st8(out + tok * C + (long long)c * 8, f);
    } else {
        const F8 a = ld8(x + it * C + (long long)c * 8), b = ld8_nt(h + tok * C + (long long)c * 8);
        F8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = a.v[i] + sc * b.v[i];
        st8(out + it * C + (long long)c * 8, o);
    }
}


// ------------------------------------------------------------------------------------------------------------
// LayerNorm over the channel dimension with FROZEN affine parameters (the trunk's norm1 / norm2 / ln_pre; reference
// vitdet.py:563-571 nn.LayerNorm(eps=1e-5)).  One wave per token row, the row lives in registers (CPL chunks of 8
// elements per lane), statistics in fp32 with a two-pass variance.  HBM-bound: forward reads x and writes y once,
// backward reads gy and x and writes gx once (no weight / bias gradients: they are frozen under LoRA).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Q8 (bf16 only): y also leaves as an fp8 image for the frozen GEMM that consumes it (fp8_common.inc)
template <typename T, int CPL, bool Q8 = false, int QF = SAM3_FP8_E4M3>
__global__ __launch_bounds__(256) void k_ln_fwd(const T* __restrict__ x, const T* __restrict__ gamma,
                                                const T* __restrict__ beta, T* __restrict__ y, float* __restrict__ mean,
                                                float* __restrict__ rstd, long long M, int C, float eps,
                                                Q8Out q8 = Q8Out{nullptr, 0, nullptr, nullptr, nullptr, 0}) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    Q8Scale qs{0.f, 0.f, 0u};
    unsigned seen = 0u;         // packed running amax (q8_pack8_bf16)
    if (Q8) qs = q8_begin(q8, blockIdx.x == 0 && threadIdx.x == 0, blockIdx.x);
    if (row >= M) return;
    F8 v[CPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            v[i] = ld8(x + row * C + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i].v[j];
        }
    }
    const float mu = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
        if ((i * 64 + lane) * 8 < C) {
#pragma unroll
            for (int j = 0; j < 8; ++j) q += (v[i].v[j] - mu) * (v[i].v[j] - mu);
        }
    const float rs = rsqrtf(wave_sum(q) / C + eps);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            const F8 g = ld8(gamma + c), b = ld8(beta + c);
            F8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = (v[i].v[j] - mu) * rs * g.v[j] + b.v[j];
            st8(y + row * C + c, o);
            if (Q8) {       // the bf16-ROUNDED values, as a separate quantisation pass over y would see them
                const uint4 p = make_uint4(vpack2(o.v[0], o.v[1]), vpack2(o.v[2], o.v[3]), vpack2(o.v[4], o.v[5]), vpack2(o.v[6], o.v[7]));
                *reinterpret_cast<uint2*>(q8.q + row * q8.ld + c) = q8_pack8_bf16<QF>(p, qs, seen);
            }
        }
    }
    if (Q8) q8_end_wave(q8, q8_seen16_to_float(seen), blockIdx.x, qs.have);     // the 4 waves of a workgroup share a slot
}

// gx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = gy * gamma,  xhat = (x - mean) * rstd
// `add` (nullable): the gradient arriving on the skip path around the norm -- gx = add + LN'(gy); saves the separate
// accumulation pass autograd would run for a tensor that feeds both the norm and the residual
template <typename T, int CPL>
__global__ __launch_bounds__(256) void k_ln_bwd(const T* __restrict__ gy, const T* __restrict__ x,
                                                const T* __restrict__ gamma, const float* __restrict__ mean,
                                                const float* __restrict__ rstd, T* __restrict__ gx, long long M, int C,
                                                const T* __restrict__ add) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float mu = mean[row], rs = rstd[row];
    F8 g[CPL], xh[CPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            const F8 a = ld8(gy + row * C + c), w = ld8(gamma + c), xv = ld8(x + row * C + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                g[i].v[j] = a.v[j] * w.v[j];
                xh[i].v[j] = (xv.v[j] - mu) * rs;
                s1 += g[i].v[j];
                s2 += g[i].v[j] * xh[i].v[j];
            }
        }
    }
    const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            F8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = rs * (g[i].v[j] - m1 - xh[i].v[j] * m2);
            if (add) {
                const F8 a = ld8_nt(add + row * C + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) o.v[j] += a.v[j];
            }
            st8(gx + row * C + c, o);
        }
    }
}

template <typename T>
int launch_ln(bool bwd, const void* a, const void* b, const void* gamma, const void* beta, void* out, float* mean,
              float* rstd, long long M, int C, float eps, hipStream_t st, const void* add = nullptr, const Q8Out* q8 = nullptr) {
    const dim3 grid((unsigned)((M + 3) / 4));
    const int cpl = (C + 511) / 512;
    if (q8) {       // forward with the fp8 image (bf16 only; checked by the caller)
#define LNQ_LAUNCH(N, F) hipLaunchKernelGGL((k_ln_fwd<bf16_t, N, true, F>), grid, dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)gamma, \
                                            (const bf16_t*)beta, (bf16_t*)out, mean, rstd, M, C, eps, *q8)
#define LNQ_CASE(N) case N: if (q8->fmt == SAM3_FP8_E4M3) LNQ_LAUNCH(N, SAM3_FP8_E4M3); else LNQ_LAUNCH(N, SAM3_FP8_E5M2); break;
        switch (cpl) {
            LNQ_CASE(1) LNQ_CASE(2) LNQ_CASE(3) LNQ_CASE(4) LNQ_CASE(5) LNQ_CASE(6) LNQ_CASE(7) LNQ_CASE(8)
            default: return -22;
        }
#undef LNQ_CASE
#undef LNQ_LAUNCH
        return hipGetLastError() == hipSuccess ? 0 : -5;
    }
#define LN_CASE(N)                                                                                                     \
    case N:                                                                                                            \
        if (bwd) hipLaunchKernelGGL((k_ln_bwd<T, N>), grid, dim3(256), 0, st, (const T*)a, (const T*)b, (const T*)gamma, \
                                    (const float*)mean, (const float*)rstd, (T*)out, M, C, (const T*)add);             \
        else hipLaunchKernelGGL((k_ln_fwd<T, N>), grid, dim3(256), 0, st, (const T*)a, (const T*)gamma, (const T*)beta, \
                                (T*)out, mean, rstd, M, C, eps);                                                       \
        break;
    switch (cpl) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
        default: return -22;
    }
#undef LN_CASE
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // namespace

extern "C" {

// returns 0 on success, -22 on bad arguments, -5 on a launch error (same codes as sam3_lora_amd.h)
int sam3_vit_qkv_rope_win_fwd(const void* qkv, const float* cos_t, const float* sin_t, void* q, void* k, void* v,
                              int64_t B, int L, int H, int D, int ws, int Hh, int Ww, int dtype, void* stream) {
    if (!qkv || !cos_t || !sin_t || !q || !k || !v || B <= 0 || L <= 0 || H <= 0 || D <= 0 || (D % 8)) return -22;
    if (ws < 0 || (ws > 0 && (Hh <= 0 || Ww <= 0 || Hh % ws || Ww % ws || L != ws * ws || (B * L) % ((int64_t)Hh * Ww))))
        return -22;
    const WinMap wm{ws, Hh, Ww};
    const long long ntok = B * L, total = ntok * (3LL * H * D / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == 0)
        hipLaunchKernelGGL(k_qkv_rope_fwd<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, cos_t,
                           sin_t, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v, ntok, L, H, D, wm);
    else if (dtype == 1)
        hipLaunchKernelGGL(k_qkv_rope_fwd<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)qkv, cos_t,
                           sin_t, (float*)q, (float*)k, (float*)v, ntok, L, H, D, wm);
    else
        return -22;
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int sam3_vit_qkv_rope_fwd(const void* qkv, const float* cos_t, const float* sin_t, void* q, void* k, void* v,
                          int64_t B, int L, int H, int D, int dtype, void* stream) {
    return sam3_vit_qkv_rope_win_fwd(qkv, cos_t, sin_t, q, k, v, B, L, H, D, 0, 0, 0, dtype, stream);
}

// gq/gk/gv: [B, H, L, D]-shaped views with element strides (sb, sh, sl, 1), identical for the three
int sam3_vit_qkv_rope_win_bwd(const void* gq, const void* gk, const void* gv, int64_t sb, int64_t sh, int64_t sl,
                              const float* cos_t, const float* sin_t, void* gqkv, int64_t B, int L, int H, int D,
                              int ws, int Hh, int Ww, int dtype, void* stream) {
    if (!gq || !gk || !gv || !cos_t || !sin_t || !gqkv || B <= 0 || L <= 0 || H <= 0 || D <= 0 || (D % 8)) return -22;
    if (ws < 0 || (ws > 0 && (Hh <= 0 || Ww <= 0 || Hh % ws || Ww % ws || L != ws * ws || (B * L) % ((int64_t)Hh * Ww))))
        return -22;
    const WinMap wm{ws, Hh, Ww};
    const long long ntok = B * L, total = ntok * (3LL * H * D / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == 0)
        hipLaunchKernelGGL(k_qkv_rope_bwd<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gq,
                           (const bf16_t*)gk, (const bf16_t*)gv, sb, sh, sl, cos_t, sin_t, (bf16_t*)gqkv, ntok, L, H, D, wm);
    else if (dtype == 1)
        hipLaunchKernelGGL(k_qkv_rope_bwd<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)gq,
                           (const float*)gk, (const float*)gv, sb, sh, sl, cos_t, sin_t, (float*)gqkv, ntok, L, H, D, wm);
    else
        return -22;
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int sam3_vit_qkv_rope_bwd(const void* gq, const void* gk, const void* gv, int64_t sb, int64_t sh, int64_t sl,
                          const float* cos_t, const float* sin_t, void* gqkv, int64_t B, int L, int H, int D,
                          int dtype, void* stream) {
    return sam3_vit_qkv_rope_win_bwd(gq, gk, gv, sb, sh, sl, cos_t, sin_t, gqkv, B, L, H, D, 0, 0, 0, dtype, stream);
}

// backward == 0: y = x + scale[img] * unpartition(h)   (x, y: [B_img, Hh, Ww, C] image order; h: window order)
// backward != 0: x is gy (image order), y receives gh = scale[img] * partition(gy) (window order); h is unused
int sam3_vit_win_residual(const void* x, const void* h, const float* scale, void* y, int64_t B_img, int Hh, int Ww,
                          int C, int ws, int backward, int dtype, void* stream) {
    if (!x || !y || (!backward && !h) || B_img <= 0 || Hh <= 0 || Ww <= 0 || C <= 0 || (C % 8) || ws <= 0 || Hh % ws ||
        Ww % ws)
        return -22;
    const WinMap wm{ws, Hh, Ww};
    const long long ntok = B_img * Hh * Ww, total = ntok * (C / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    const int L = ws * ws;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        if (backward) hipLaunchKernelGGL((k_win_residual<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)h, scale, (bf16_t*)y, ntok, L, C, wm);
        else hipLaunchKernelGGL((k_win_residual<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)h, scale, (bf16_t*)y, ntok, L, C, wm);
    } else if (dtype == 1) {
        if (backward) hipLaunchKernelGGL((k_win_residual<float, true>), grid, dim3(256), 0, st, (const float*)x, (const float*)h, scale, (float*)y, ntok, L, C, wm);
        else hipLaunchKernelGGL((k_win_residual<float, false>), grid, dim3(256), 0, st, (const float*)x, (const float*)h, scale, (float*)y, ntok, L, C, wm);
    } else {
        return -22;
    }
    return hipGetLastError() == hipSuccess ? 0 : -5;
}


// y = LayerNorm(x) * gamma + beta over the last dimension of x[M, C]; also writes the fp32 row statistics the
// backward needs.  gamma / beta have the activation dtype.  C % 8 == 0, C <= 4096.
int sam3_vit_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                           int64_t M, int C, float eps, int dtype, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || M <= 0 || C <= 0 || (C % 8) || C > 4096) return -22;
    if (dtype == 0) return launch_ln<bf16_t>(false, x, nullptr, gamma, beta, y, mean, rstd, M, C, eps, (hipStream_t)stream);
    if (dtype == 1) return launch_ln<float>(false, x, nullptr, gamma, beta, y, mean, rstd, M, C, eps, (hipStream_t)stream);
    return -22;
}

int sam3_vit_layernorm_fwd_q8(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                              int64_t M, int C, float eps, int dtype, void* q8_out, int64_t ldq, int fmt, const float* amax_in,
                              float* amax_out, float* scale_out, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || M <= 0 || C <= 0 || (C % 8) || C > 4096) return -22;
    if (dtype != 0) return -95;
    if (!q8_out || !amax_in || !amax_out || !scale_out || ldq < C || (ldq & 7) || ((uintptr_t)q8_out & 7) ||
        (fmt != SAM3_FP8_E4M3 && fmt != SAM3_FP8_E5M2))
        return -22;
    const Q8Out q8{(unsigned char*)q8_out, (long long)ldq, amax_in, amax_out, scale_out, fmt};
    return launch_ln<bf16_t>(false, x, nullptr, gamma, beta, y, mean, rstd, M, C, eps, (hipStream_t)stream, nullptr, &q8);
}

// input gradient only (gamma, beta frozen): gx = (add ? add : 0) + LN'(gy) from gy, x and the saved statistics;
// `add` [M, C] is the gradient of the skip path around the norm (NULL: none)
int sam3_vit_layernorm_bwd_add(const void* gy, const void* x, const void* gamma, const float* mean, const float* rstd,
                               const void* add, void* gx, int64_t M, int C, int dtype, void* stream) {
    if (!gy || !x || !gamma || !mean || !rstd || !gx || M <= 0 || C <= 0 || (C % 8) || C > 4096) return -22;
    if (dtype == 0) return launch_ln<bf16_t>(true, gy, x, gamma, nullptr, gx, (float*)mean, (float*)rstd, M, C, 0.f, (hipStream_t)stream, add);
    if (dtype == 1) return launch_ln<float>(true, gy, x, gamma, nullptr, gx, (float*)mean, (float*)rstd, M, C, 0.f, (hipStream_t)stream, add);
    return -22;
}

int sam3_vit_layernorm_bwd(const void* gy, const void* x, const void* gamma, const float* mean, const float* rstd,
                           void* gx, int64_t M, int C, int dtype, void* stream) {
    return sam3_vit_layernorm_bwd_add(gy, x, gamma, mean, rstd, nullptr, gx, M, C, dtype, stream);
}

}  // extern "C"
