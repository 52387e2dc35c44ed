// sam3_lora_amd -- attention forward of the ViT trunk that hosts the adapters (C-ABI of include/sam3_attn_amd.h), gfx950.
//
//   O = softmax(Q K^T * scale) V      Q, K, V, O: [B, L, H, 64] bf16 (token-major: what the qkv / RoPE kernel writes and the
//                                     output projection reads -- no head-major copies);  LSE[B, H, L] fp32 for the backward
//
// The trunk calls it on 576-token windows (28 blocks) and on the whole 5184-token grid (4 blocks), 16 heads of 64
// (sam3/model/vitdet.py:339-515).  PyTorch-ROCm's kernels run these shapes at ~270 TFLOP/s (profiles/r02q_fullstep_kernel_stats:
// attn_fwd 361 us per call), i.e. ~11 % of the bf16 MFMA peak.  Flash-style single pass, written for 64-wide wavefronts:
//
//   * workgroup = 3 waves x 32 query rows (576 = 6 x 96, 5184 = 54 x 96: no ragged tile at either trunk shape); K / V stream
//     through LDS in 64-row tiles, double-buffered, one barrier per tile;
//   * BOTH products are computed TRANSPOSED on v_mfma_f32_32x32x16_bf16 so that a lane owns ONE query row throughout:
//       S^T[kv, q] = K[kv, :] . Q[q, :]      A = K rows (ds_read_b128 from the XOR-swizzled tile), B = Q (registers)
//       O^T[d, q] += V^T[d, kv] . P^T[kv, q]  A = V^T (ds_read_b64_tr_b16: the hardware transpose read), B = P^T
//     The C layout of the first (lane = q, registers = kv) IS the B-operand layout of the second after an in-lane
//     bf16 pack -- no cross-lane movement of P at all -- and the accumulator of the second has lane = q again, so the online
//     softmax's running max / sum / rescale are per-lane scalars.  The one cross-lane step per 64 keys is the max exchange
//     between the two half-waves that share a row (lane ^ 32).
//   * contraction index order inside an MFMA is free as long as both operands agree: the k-slots of the second product are
//     {4h .. 4h+3} u {8+4h .. 8+4h+3} per 16 keys (h = lane >> 5), exactly the rows a lane holds after the first product.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sam3_attn_amd.h"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int D = 64;            // head dimension
constexpr int QW = 32;           // query rows per wave
constexpr int NW = 3;            // waves per workgroup
constexpr int QB = QW * NW;      // query rows per workgroup
constexpr int KB = 64;           // keys per tile

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// K tile [64 keys][64 d] bf16, 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ (r & 7)
__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// V tile [64 keys][64 d] bf16, 128-byte rows, 32-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 3): the four rows
// one transpose read touches land in four different bank groups
__device__ __forceinline__ int v_off(int row, int chunk32) { return row * 128 + ((chunk32 ^ ((row >> 1) & 3)) << 5); }

__global__ __launch_bounds__(NW * 64, 3) void k_attn_fwd(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                      const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                      float* __restrict__ LSE, int L, int H, long long sb, long long sl,
                                                      long long sh, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2 * KB * 128];      // [buffer][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const long long base = (long long)b * sb + (long long)h * sh;
    const int q0 = blockIdx.x * QB + wave * QW;
    const bool active = q0 < L;                 // a wave past the end still loads tiles and joins the barriers
    const int q = q0 + ql;
    const bool qok = q < L;

    // Q fragments: B-operand of S^T, lane (q, hi) holds Q[q][16 s + 8 hi .. + 7]
    uint4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qf[s] = qok ? *reinterpret_cast<const uint4*>(Q + base + (long long)q * sl + 16 * s + 8 * hi) : make_uint4(0u, 0u, 0u, 0u);

    // tile loads: 2 x 64 rows x 8 chunks of 16 B = 1024 chunks over 192 threads
    constexpr int NCH = 2 * KB * 8, NLD = (NCH + NW * 64 - 1) / (NW * 64);
    uint4 stage[NLD];
    auto gload = [&](int kb) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NW * 64;
            const int which = c >> 9, row = (c >> 3) & 63, ch = c & 7;      // which: 0 = K, 1 = V
            const int key = kb * KB + row;
            const bf16_t* src = (which ? V : K) + base + (long long)(key < L ? key : L - 1) * sl + ch * 8;
            stage[i] = c < NCH ? *reinterpret_cast<const uint4*>(src) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NW * 64;
            if (c < NCH) {
                const int which = c >> 9, row = (c >> 3) & 63, ch = c & 7;
                unsigned char* dst = lds[buf] + which * (KB * 128);
                // K: 16-byte chunk swizzle; V: the 16-byte chunk keeps its place inside its swizzled 32-byte chunk
                const int off = which ? v_off(row, ch >> 1) + ((ch & 1) << 4) : k_off(row, ch);
                *reinterpret_cast<uint4*>(dst + off) = stage[i];
            }
        }
    };

    f32x16 ot[2];                   // O^T accumulators: [d tile][rows d = (r&3) + 8 (r>>2) + 4 hi, column q = lane & 31]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;          // running max (in the exp2 domain) and this lane's part of the row sum

    const int nkb = (L + KB - 1) / KB;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) gload(kb + 1);
        if (active) {
            const unsigned char* kt = lds[buf];
            const unsigned char* vt = lds[buf] + KB * 128;
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                // ---- S^T[32 keys, 32 q] ----
                f32x16 st;
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = 0.f;
                const int krow = half * 32 + ql;        // A-operand row of this lane
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint4 kf = *reinterpret_cast<const uint4*>(kt + k_off(krow, 2 * s + hi));
                    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]), st, 0, 0, 0);
                }
                // keys beyond L (ragged last tile) never win the max and add nothing to the sum
                const int key0 = kb * KB + half * 32 + 4 * hi;
                float mx = -1e30f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2);
                    st[r] = key < L ? st[r] * scale_log2e : -1e30f;
                    mx = fmaxf(mx, st[r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));         // the other half-wave holds the other 16 keys of this row
                const float m_new = fmaxf(m_run, mx);
                const float alpha = exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[t][r] *= alpha;
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[r] = exp2f(st[r] - m_new);
                    psum += st[r];
                }
                l_run += psum;
                // P^T as the B-operand: k-step s covers keys 16 s .. 16 s + 15 of this half; this lane's slots e = 0..7 are the
                // keys 16 s + 4 hi + {0..3} and 16 s + 8 + 4 hi + {0..3} = registers 8 s .. 8 s + 7
                uint4 pb[2];
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    pb[s] = make_uint4(pack2(st[8 * s], st[8 * s + 1]), pack2(st[8 * s + 2], st[8 * s + 3]),
                                       pack2(st[8 * s + 4], st[8 * s + 5]), pack2(st[8 * s + 6], st[8 * s + 7]));
                // ---- O^T[d, q] += V^T[d, keys] . P^T ----
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                const int grp = lane >> 4, nn = lane & 15;       // 16-lane group of the transpose read
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        // the lane's row in the A-operand is d = 32 t + 16 (grp & 1) + nn; it needs V[key][d] for the keys of
                        // its k-slots.  A transpose read hands a 16-lane group the 4 x 16 block (4 keys x 16 d): lane nn
                        // supplies the address of key (nn >> 2), d columns 4 (nn & 3) .. + 3 and receives column nn.
                        const int keyA = half * 32 + 16 * s + 4 * hi + (nn >> 2), keyB = keyA + 8;
                        const int dcol = 32 * t + 16 * (grp & 1) + 4 * (nn & 3);            // element column of the 8-byte piece
                        const unsigned char* pa = vt + v_off(keyA, dcol >> 4) + ((dcol & 15) << 1);
                        const unsigned char* pbp = vt + v_off(keyB, dcol >> 4) + ((dcol & 15) << 1);
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
                        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pbp);
                        const s16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                        ot[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pb[s]), ot[t], 0, 0, 0);
                    }
                }
            }
        }
        if (kb + 1 < nkb) sstore(buf ^ 1);
        __syncthreads();
    }
    if (!active) return;
    // row sum: the two half-waves hold disjoint keys of the same row
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qok) {
        // lane (q, hi) holds O[q][32 t + 8 j + 4 hi + {0..3}] in registers 4 j .. 4 j + 3 of tile t: 8-byte stores
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint2 w = make_uint2(pack2(ot[t][4 * j] * inv, ot[t][4 * j + 1] * inv),
                                           pack2(ot[t][4 * j + 2] * inv, ot[t][4 * j + 3] * inv));
                *reinterpret_cast<uint2*>(O + base + (long long)q * sl + 32 * t + 8 * j + 4 * hi) = w;
            }
        if (hi == 0) LSE[(long long)bh * L + q] = (m_run + log2f(l_tot)) * 0.69314718055994531f;    // natural log
    }
}

}  // namespace

extern "C" {

int sam3_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int L, int H, int head_dim,
                  int64_t stride_b, int64_t stride_l, int64_t stride_h, float scale, int dtype, void* stream) {
    if (!q || !k || !v || !o || !lse || B <= 0 || L <= 0 || H <= 0) return -22;
    if (head_dim != D || dtype != 0) return -95;                // bf16, head dimension 64 only: callers fall back to PyTorch
    if ((stride_l % 8) || (stride_h % 8) || (stride_b % 8) || stride_h < D) return -22;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)o & 15)) return -22;
    if (B * H > 65535 * 32LL) return -22;
    dim3 grid((unsigned)((L + QB - 1) / QB), (unsigned)(B * H));
    hipLaunchKernelGGL(k_attn_fwd, grid, dim3(NW * 64), 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (bf16_t*)o, lse, L, H, (long long)stride_b, (long long)stride_l,
                       (long long)stride_h, scale * 1.4426950408889634f);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

}  // extern "C"
