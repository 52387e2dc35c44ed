// sam3_lora_amd -- hand-written CDNA4 (gfx950) kernels for the LoRA adapter hot path and the
// C-ABI of include/sam3_lora_amd.h.  gfx950 only: 64-wide wavefronts, bf16 MFMA
// (v_mfma_f32_16x16x32_bf16 / 16x16x16_bf16), LDS transpose reads (ds_read_b64_tr_b16).
//
// The path is HBM-bound (arithmetic intensity ~9 FLOP/B against a machine balance of ~300), so
// the design is about touching each activation byte once, in full 128-B lines:
//
//   T1  k_t1   T[M,r]    = X[M,K] . W1^T           row reduction  (t = x.A ; gt = gy.B^T)
//   T2  k_t2   Y[M,N]   += s * T[M,r] . W2[r,N]    rank-r update  (y += s.t.B ; gx += s.gt.A^T)
//   T3  k_t3   G[r,N]    = T^T[r,M] . X[M,N]       column reduction over M (gB ; gA), split over
//                                                  row ranges, fixed-order second-stage sum
//   T3E k_t3e  T3 over gy AND gt = gy.B^T from the same pass (r <= 16): the backward reads gy once;
//              k_gt_reduce sums the per-column-pair partials of gt in fixed order
//   T2 also carries the MLP's activation: ACT=1 writes GELU(y) beside y, ACT=2 multiplies by GELU'(h)
//
// Read-once activation streams are loaded non-temporal; the read half of T2's in-place update is not.
//
// HL ("hi + lo", bf16 activations, r <= 16): every bf16 operand that is NOT caller data -- the images of A and B, and the
// rank-r intermediates t / gt -- is carried as a PAIR of bf16 values v = hi + lo (hi = bf16(v), lo = bf16(v - hi): 16
// mantissa bits), laid out exactly like a rank-32 operand (rank indices 16..31 = the lo parts).  The contractions then run
// hi.hi + hi.lo + lo.hi on the same MFMAs -- free on kernels that sit at < 1 % MFMA utilisation -- and the branch
// s.(x A) B and its four gradient products are fp32 arithmetic on the caller's bf16 data: the only bf16 roundings left
// are the ones of the caller's own tensors (x, gy in; y, gx out).  SAM3_LORA_SINGLE_ROUND=1 restores the single-rounded
// operands (one bf16 rounding of A, B, t, gt each).
//
// MFMA operand roles are chosen so that no result ever needs a cross-lane shuffle:
//   * T1 computes t^T (A-operand = LoRA weight, B-operand = activation rows): the C/D layout
//     (lane&15 = activation row, (lane>>4)*4+reg = rank index) is exactly the B-operand layout
//     of T2's K=r MFMA and gives 8-byte row-major stores of t.
//   * T2 computes the update transposed (A-operand = W2^T) so every lane ends with 4 consecutive
//     output columns of one row; the fp32 tile goes through a wave-private LDS slab once and is
//     re-read as 16-byte row segments matching the coalesced global load/store of Y.
//   * T3 contracts over rows, the strided dimension of a row-major activation tile: the tile is
//     staged row-major in LDS (XOR-swizzled 16-B chunks) and the B-operand is fetched with the
//     hardware transpose read.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "sam3_lora_amd.h"
#include "fp8_common.inc"

typedef unsigned short bf16_t;  // storage type of a bf16 element
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};  // RNE; lowers to v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// XCD-aware tile order.  The dispatcher is observed to place workgroup b (dispatch order: blockIdx.x fastest) on XCD b % 8,
// and each XCD has its own L2: workgroups that share an operand slice (the column chunks of one row group share its t / gt
// rows) should sit on ONE XCD, or every XCD fetches the slice again (PMC, profiles/r03_traffic.json: +19-22 MB per launch,
// 1.1-1.3x the algorithmic bytes on the 1024-wide launches).  This maps dispatch index `orig` of `nwg` workgroups to a tile
// index such that each XCD owns a CONTIGUOUS range of tile indices (bijective for any nwg); with tiles numbered row-group-major
// the column chunks of a row group land on the same XCD.  Placement is a speed matter only: results do not depend on it.
__device__ __forceinline__ unsigned xcd_tile_index(unsigned orig, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = orig & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// Streamed-once activation reads (k_t1, k_t3, k_t3e, the pre-activation tile of k_t2<ACT=2>, the gt partials) are
// issued NON-TEMPORAL (`global_load ... nt`): they are consumed exactly once, and keeping them out of L2 / Infinity
// Cache measured -10 % on the whole adapter step on MI355X (same-box A/B, profiles/r01e_nt_loads.txt: k_t3 96.7 -> 70.9
// us, k_t1 85.5 -> 77.6 us, k_t3e 107.9 -> 87.3 us at 4736 columns).  The read half of k_t2's in-place update must NOT
// be non-temporal (122 -> 145 us), and non-temporal stores change nothing -- both stay compile-time switches.
#ifndef SAM3_NT_LOADS
#define SAM3_NT_LOADS 1
#endif
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ uint4 ldg16(const void* p) {
#if SAM3_NT_LOADS
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(t[0], t[1], t[2], t[3]);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}
__device__ __forceinline__ uint4 ldg16_rmw(const void* p) {     // the read half of an in-place update
#if defined(SAM3_NT_RMW) && SAM3_NT_RMW
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(t[0], t[1], t[2], t[3]);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}
__device__ __forceinline__ void stg16(void* p, uint4 v) {       // streamed-out result
#if defined(SAM3_NT_STORES) && SAM3_NT_STORES
    const u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}
// 8 consecutive activation elements -> packed bf16x8 (as uint4)
__device__ __forceinline__ uint4 load8(const bf16_t* p) { return ldg16(p); }
__device__ __forceinline__ uint4 load8(const float* p) {
    const float4 a = __builtin_bit_cast(float4, ldg16(p));
    const float4 b = __builtin_bit_cast(float4, ldg16(p + 4));
    return make_uint4(pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w));
}
__device__ __forceinline__ uint4 zero4() { return make_uint4(0u, 0u, 0u, 0u); }

// ------------------------------------------------------------------------------------------
// dropout on the branch input x (nn.Dropout semantics, lora_layers.py:54): counter-based, so the
// forward, the checkpoint recompute and the three backward consumers regenerate the same mask.
//   element e = row*width + col ; 4 x murmur3-fmix32 per aligned group of 8 elements, 16 bits each;
//   keep  <=>  u16 >= thr  (thr = round(p * 65536)).  Restated in oracle/lora_oracle.py:dropout_keep.
// ------------------------------------------------------------------------------------------
struct DropKey {
    unsigned k0;   // seed/offset mix
    unsigned thr;  // 0 => dropout off
    int width;     // logical row width of x (in_features)
};

__device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

// keep-bits (bit j <=> element e8 + j is kept) of the aligned 8-element group starting at e8.
// e8 % 8 == 0, so the four counters c0 .. c0+3 share their high word and differ only in the low two bits.
__device__ __forceinline__ unsigned keep8(unsigned long long e8, const DropKey& dk) {
    const unsigned long long c0 = e8 >> 1;
    const unsigned lo = (unsigned)c0, key = dk.k0 ^ ((unsigned)(c0 >> 32) * 0x85ebca6bu);
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned h = fmix32((lo + j) ^ key);
        bits |= ((h & 0xffffu) >= dk.thr ? 1u : 0u) << (2 * j);
        bits |= ((h >> 16) >= dk.thr ? 1u : 0u) << (2 * j + 1);
    }
    return bits;
}

__device__ __forceinline__ uint4 drop8(uint4 v, unsigned long long e8, const DropKey& dk) {
    const unsigned b = keep8(e8, dk);
    auto m = [&](int j) { return ((b >> (2 * j)) & 1u ? 0xffffu : 0u) | ((b >> (2 * j + 1)) & 1u ? 0xffff0000u : 0u); };
    return make_uint4(v.x & m(0), v.y & m(1), v.z & m(2), v.w & m(3));
}

// ------------------------------------------------------------------------------------------
// pack: fp32 LoRA master weights (either reference layout) -> bf16 operand images, zero padded
//   dst[i][j] (row-major I x J) = (i < Iv && j < Jv) ? src[i*si + j*sj] : 0
// ------------------------------------------------------------------------------------------
struct PackJob {
    const float* src;
    void* dst;            // bf16 image, or fp32 image when f32 != 0 (exact-fp32 path)
    int I, J, Iv, Jv;     // dst is I x J, valid region Iv x Jv (the rest is zero-filled)
    long long si, sj;
    int ldd;              // row pitch of dst in elements (0 => J); > J when packing into a slice of a wider buffer
    int f32;
    int hl;               // bf16 hi + lo image: 1 = the second half of the I rows holds the lo parts of the first half (I == 32 / 64);
                          // 2 = columns (J == 32 / 64), interleaved per 4 rank indices: [hi 4g..4g+3 | lo 4g..4g+3] at columns
                          // 8g..8g+7; Iv / Jv then bound the rank index
};

constexpr int PACK_JOBS_MAX = 64;    // 64 x 56 B of kernel arguments per launch
struct PackJobs {
    PackJob j[PACK_JOBS_MAX];
};

__global__ __launch_bounds__(256) void k_pack(PackJobs jobs) {
    const PackJob jb = jobs.j[blockIdx.y];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)jb.I * jb.J) return;
    const int i = (int)(idx / jb.J), j = (int)(idx % jb.J);
    // source index (both halves read the same master).  hl == 2 (k_t2's operand rows) interleaves the halves per group of 4
    // rank indices -- [hi 0..3 | lo 0..3 | hi 4..7 | lo 4..7 | ...] -- so that a lane's hi and lo fragments are ONE 16-byte load
    const int half = jb.I >> 1;
    const int is = jb.hl == 1 ? (i >= half ? i - half : i) : i, js = jb.hl == 2 ? ((j >> 3) * 4 + (j & 3)) : j;
    const bool lo = (jb.hl == 1 && i >= half) || (jb.hl == 2 && ((j >> 2) & 1));
    float v = (is < jb.Iv && js < jb.Jv) ? jb.src[is * jb.si + js * jb.sj] : 0.f;
    const long long o = (long long)i * (jb.ldd ? jb.ldd : jb.J) + j;
    if (jb.f32) {
        reinterpret_cast<float*>(jb.dst)[o] = v;
        return;
    }
    if (lo) v -= bf_lo(pack2(v, 0.f));          // the residual of the hi half's rounding
    reinterpret_cast<bf16_t*>(jb.dst)[o] = (bf16_t)(pack2(v, 0.f) & 0xffffu);
}

// ------------------------------------------------------------------------------------------
// Streaming discipline shared by T1 / T2 / T3 (measured on MI355X, profiles/r01c_t1_variants.txt):
//   * every global load in a main loop is UNCONDITIONAL.  Tails are handled by clamping the address
//     into the tensor and zeroing the value with a bit mask when it is consumed.  A predicated load
//     becomes an exec-masked branch, and hipcc then falls back to `s_waitcnt vmcnt(0)` at the first
//     consumer -- which silently drains the prefetch and leaves the kernel latency-bound (~5.0 TB/s).
//   * a load whose value is needed "now" (LoRA fragments, t fragments) is issued one stage EARLY,
//     in front of the big prefetch: vmcnt is an in-order counter, so waiting for the newest load
//     would wait for everything older too.
//   * the software pipeline is written straight-line (two named register sets, loop unrolled by 2,
//     iteration count padded to even with zero-masked stages) so the waits are counted, not zero.
//   * no workgroup barrier inside a streaming loop: each wave streams its own rows through its own LDS
//     slab (a wave's LDS operations execute in order, so only the compiler needs pinning).
// ------------------------------------------------------------------------------------------
template <typename XT>
struct Raw8;  // 8 consecutive activation elements exactly as loaded (conversion happens at use)
template <>
struct Raw8<bf16_t> {
    uint4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = ldg16(p); }
    __device__ __forceinline__ uint4 packed() const { return v; }
};
template <>
struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = __builtin_bit_cast(float4, ldg16(p));
        b = __builtin_bit_cast(float4, ldg16(p + 4));
    }
    __device__ __forceinline__ uint4 packed() const {
        return make_uint4(pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w));
    }
};
__device__ __forceinline__ uint4 and4(uint4 v, unsigned m) { return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m); }

// ------------------------------------------------------------------------------------------
// T1: T[Mp, RP] (row-major bf16) and TTf (fragment-major bf16) = X[M, K] . W1[RP, K]^T
//   workgroup = 64 rows x full K, 4 waves x 16 rows; K streamed in 128-column chunks through a
//   double-buffered, XOR-swizzled LDS tile shared by the 4 waves (W1 chunk re-used 4x).
//   This plain structure measured FASTEST of four variants on MI355X at K=4736, M=41472 (same box A/B or
//   back-to-back runs, profiles/r01c_t1_variants.txt): 77 us (this) | 91 us (same tile, branch-free
//   counted-vmcnt distance-2 prefetch) | 99 us (256-column chunks) | 103 us (wave-private slabs with the
//   LoRA operand read fragment-major from L2 -- operand traffic = activation traffic).
//   TTf  fragment-major image of T^T:  block (m/32, rt) = 64 lanes x 8 elements, lane = g*16 + n holds
//        T[(m/32)*32 + g*8 .. +8][rt*16 + n]   -- exactly the A-operand T3 needs, 1 KB per load
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {   // LDS ops of one wave execute in order; this pins the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One lane's 4 consecutive rank entries of row m (rank tile rt) -> the row-major image T[Mp][RP] and the fragment-major
// image TTf (block (m/32, rt) = 64 lanes x 8 elements: exactly the A-operand k_t3 needs, 1 KB per load).
template <int RT>
__device__ __forceinline__ void store_t4(bf16_t* __restrict__ T, bf16_t* __restrict__ TTf, long long m, int rt, int r0,
                                         unsigned p0, unsigned p1) {
    constexpr int RP = RT * 16;
    *reinterpret_cast<uint2*>(T + m * RP + rt * 16 + r0) = make_uint2(p0, p1);
    const long long blk = m >> 5;
    const int gq = (int)(m & 31) >> 3, jq = (int)(m & 7);
    bf16_t* tb = TTf + (((blk * RT + rt) * 4 + gq) * 16 + r0) * 8 + jq;   // [blk][rt][gq][n' = r0 + j][jq]
    tb[0] = (bf16_t)(p0 & 0xffffu);
    tb[8] = (bf16_t)(p0 >> 16);
    tb[16] = (bf16_t)(p1 & 0xffffu);
    tb[24] = (bf16_t)(p1 >> 16);
}
// HL: the fp32 values v -> hi = bf16(v) as rank tile 0, lo = bf16(v - hi) as rank tile 1 of a rank-32 image
// The row-major image (k_t2's operand) interleaves the halves per 4 rank indices: T[m][8 g .. 8 g + 3] = hi, [8 g + 4 .. + 7] =
// lo (g = r0 / 4) -- one 16-byte store here, one 16-byte load per tile in k_t2.
// RH rank tiles per half (RH = 1: r <= 16, RH = 2: r <= 32), `tt` = the rank tile of v: row image T[m][RH * 32] = tile after tile of
// 32 entries; fragment-major image: 2 RH blocks per 32-row step, [hi tiles 0..RH-1 | lo tiles 0..RH-1].
template <int RH>
__device__ __forceinline__ void store_t4_hl(bf16_t* __restrict__ T, bf16_t* __restrict__ TTf, long long m, int tt, int r0, f32x4 v) {
    const unsigned h0 = pack2(v[0], v[1]), h1 = pack2(v[2], v[3]);
    const unsigned l0 = pack2(v[0] - bf_lo(h0), v[1] - bf_hi(h0)), l1 = pack2(v[2] - bf_lo(h1), v[3] - bf_hi(h1));
    *reinterpret_cast<uint4*>(T + m * (RH * 32) + tt * 32 + 2 * r0) = make_uint4(h0, h1, l0, l1);
    const long long blk = m >> 5;
    const int gq = (int)(m & 31) >> 3, jq = (int)(m & 7);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {        // fragment-major image: hi block, lo block of this rank tile (k_t3's A-operands)
        const unsigned p0 = hh ? l0 : h0, p1 = hh ? l1 : h1;
        bf16_t* tb = TTf + (((blk * (2 * RH) + hh * RH + tt) * 4 + gq) * 16 + r0) * 8 + jq;
        tb[0] = (bf16_t)(p0 & 0xffffu);
        tb[8] = (bf16_t)(p0 >> 16);
        tb[16] = (bf16_t)(p1 & 0xffffu);
        tb[24] = (bf16_t)(p1 >> 16);
    }
}

// store_t4_hl for a whole 256-thread workgroup that owns 64 CONSECUTIVE rows starting at a multiple of 64 (r <= 16: one rank tile): the row-major
// image goes out as before (16 bytes per lane); the fragment-major image -- eight 2-byte stores per lane, 16 bytes apart, in store_t4_hl --
// is assembled in 4 KB of LDS and leaves as ONE 16-byte store per thread (the 64 rows' image is 4 KB contiguous: two 32-row steps x
// [hi | lo] x 4 x 16 x 8).  Every thread of the workgroup must call it (one barrier inside); `stage` = 4 KB of LDS nobody else is using.
__device__ __forceinline__ void store_t4_hl_wg64(bf16_t* __restrict__ T, bf16_t* __restrict__ TTf, long long m0, int row, int r0, f32x4 v,
                                                 bf16_t* stage) {
    const unsigned h0 = pack2(v[0], v[1]), h1 = pack2(v[2], v[3]);
    const unsigned l0 = pack2(v[0] - bf_lo(h0), v[1] - bf_hi(h0)), l1 = pack2(v[2] - bf_lo(h1), v[3] - bf_hi(h1));
    *reinterpret_cast<uint4*>(T + (m0 + row) * 32 + 2 * r0) = make_uint4(h0, h1, l0, l1);
    const int blk = row >> 5, gq = (row & 31) >> 3, jq = row & 7;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const unsigned p0 = hh ? l0 : h0, p1 = hh ? l1 : h1;
        bf16_t* tb = stage + (((blk * 2 + hh) * 4 + gq) * 16 + r0) * 8 + jq;
        tb[0] = (bf16_t)(p0 & 0xffffu);
        tb[8] = (bf16_t)(p0 >> 16);
        tb[16] = (bf16_t)(p1 & 0xffffu);
        tb[24] = (bf16_t)(p1 >> 16);
    }
    __syncthreads();
    reinterpret_cast<uint4*>(TTf + (m0 >> 5) * 1024)[threadIdx.x] = reinterpret_cast<const uint4*>(stage)[threadIdx.x];
}

// PART (small M, r <= 16): blockIdx.y selects a range of K chunks and the workgroup writes its fp32 partial to
// P[split][row][16]; k_gt_reduce then adds the splits in fixed order and emits T / TTf.  With M/64 workgroups alone a
// small batch leaves most CUs idle (M = 5184: 81 workgroups on 256 CUs).
// HL (RT == 2): W1 rows 16..31 are the lo parts of A_c^T / B_c; the two accumulators are added (t = x.hi + x.lo in fp32)
// and t leaves as a hi + lo pair.
template <typename XT, int RT, int BK, bool PART = false, bool HL = false>
__global__ __launch_bounds__(256) void k_t1(const XT* __restrict__ X, long long ldx,
                                            const bf16_t* __restrict__ W1, bf16_t* __restrict__ T,
                                            bf16_t* __restrict__ TTf, long long M, long long Mp, int K,
                                            DropKey dk, float* __restrict__ P = nullptr, int kc_per = 0) {
    static_assert(!PART || RT == 1 || HL, "split-K partials are laid out for one rank tile");
    static_assert(!HL || RT == 2 || RT == 4, "hi + lo operands: rank tiles [hi .. | lo ..]");
    static_assert(!(PART && HL) || RT == 2, "split-K partials are laid out for one rank tile");
    constexpr int RP = RT * 16, BM = 64, CPR = BK / 8;          // CPR 16-byte chunks per tile row
    constexpr int RH = HL ? RT / 2 : RT;                        // rank tiles of the result
    constexpr int RPP = 256 / CPR;                               // tile rows covered per pass of 256 threads
    constexpr int XP = BM / RPP, WP = RP / RPP;                  // passes for the x tile / the W1 tile
    static_assert(RP % RPP == 0, "tile geometry");
    __shared__ uint4 xs[2][BM * CPR];
    __shared__ uint4 ws[2][RP * CPR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const long long m0 = (long long)blockIdx.x * BM;
    const int nk = (K + BK - 1) / BK;
    const int lrow = tid / CPR, lc = tid % CPR;

    uint4 xr[XP], wr[WP];
    auto gload = [&](int kc) {
        const int k = kc * BK + lc * 8;
        // the W1 chunk first: it comes from the L2 and has landed when the x chunk (HBM) arrives; behind the x loads it was the last
        // thing the chunk's s_waitcnt saw (20.6 -> 19.3 us at K = 1024, equal at 4736)
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int r = lrow + RPP * j;
            wr[j] = (k < K) ? *reinterpret_cast<const uint4*>(W1 + (long long)r * K + k) : zero4();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const long long m = m0 + lrow + RPP * i;
            xr[i] = (m < M && k < K) ? load8(X + m * ldx + k) : zero4();
        }
    };
    // dropout is applied here, when the chunk moves to LDS -- not at load time, which would expose the
    // load latency of every chunk (measured: 141 vs 77 us at K=4736)
    auto sstore = [&](int buf, int kc) {
        const int k = kc * BK + lc * 8;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int row = lrow + RPP * i;
            uint4 v = xr[i];
            if (dk.thr) v = drop8(v, (unsigned long long)(m0 + row) * dk.width + k, dk);
            xs[buf][row * CPR + (lc ^ (row & (CPR - 1)))] = v;
        }
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int r = lrow + RPP * j;
            ws[buf][r * CPR + (lc ^ (r & (CPR - 1)))] = wr[j];
        }
    };

    f32x4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int kc0 = PART ? (int)blockIdx.y * kc_per : 0;
    const int kc1 = PART ? (kc0 + kc_per < nk ? kc0 + kc_per : nk) : nk;
    gload(kc0);
    sstore(0, kc0);
    __syncthreads();
    for (int kc = kc0; kc < kc1; ++kc) {
        const int buf = (kc - kc0) & 1;
        if (kc + 1 < kc1) gload(kc + 1);
        const int row = wave * 16 + n;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int c = kk * 4 + g;
            const bf16x8 xf = __builtin_bit_cast(bf16x8, xs[buf][row * CPR + (c ^ (n & (CPR - 1)))]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8 wf = __builtin_bit_cast(bf16x8, ws[buf][(rt * 16 + n) * CPR + (c ^ (n & (CPR - 1)))]);
                // D[i = rank idx][n = activation row] += sum_k W1[i][k] * X[row][k]
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[rt], 0, 0, 0);
            }
        }
        if (kc + 1 < kc1) sstore(buf ^ 1, kc + 1);
        __syncthreads();
    }
    // lane (n, g) holds t[row = m0 + wave*16 + n][rank idx = rt*16 + g*4 + j]
    const long long m = m0 + wave * 16 + n;
    if (HL) {                               // x.A_hi + x.A_lo, rank tile by rank tile
#pragma unroll
        for (int tt = 0; tt < RH; ++tt) acc[tt] += acc[RH + tt];
    }
    if (PART) {
        *reinterpret_cast<f32x4*>(P + ((long long)blockIdx.y * Mp + m) * 16 + g * 4) = acc[0];
        return;
    }
    if (HL) {
        if (RH == 1) {      // (the chunk loop ended on a barrier: xs is free)
            store_t4_hl_wg64(T, TTf, m0, wave * 16 + n, g * 4, acc[0], reinterpret_cast<bf16_t*>(&xs[0][0]));
            return;
        }
#pragma unroll
        for (int tt = 0; tt < RH; ++tt) store_t4_hl<RH>(T, TTf, m, tt, g * 4, acc[tt]);
        return;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
        store_t4<RT>(T, TTf, m, rt, g * 4, pack2(acc[rt][0], acc[rt][1]), pack2(acc[rt][2], acc[rt][3]));
}

// ------------------------------------------------------------------------------------------
// T2: Y[M, N] += scale * T[M, RP] . W2[RP, N]      (W2 given transposed: W2t[N, RP])
//   one wave = 128 output columns (W2 fragments live in registers) x a strided set of 16-row
//   tiles; no workgroup barrier anywhere -- each wave streams on its own, one tile ahead.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void mask8(f32x4& a, f32x4& b, unsigned keep) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!((keep >> j) & 1u)) a[j] = 0.f;
        if (!((keep >> (4 + j)) & 1u)) b[j] = 0.f;
    }
}

// exact (erf) GELU, nn.GELU()'s default, and its derivative -- evaluated in fp32 on the ROUNDED pre-activation, as the
// unfused elementwise kernels would see it
// cdf and pdf share one exponential: E = exp(-x^2/2) = exp(-u^2) with u = x/sqrt(2), and
// erf(|u|) = 1 - (a1 t + ... + a5 t^5) E, t = 1/(1 + p|u|)  (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 -- far below one
// bf16 or even fp32-output rounding of the activation).  ~16 VALU ops per element instead of ~50 for erff + expf: with
// the library erff the GELU' pass was VALU-bound (238 us for 1.18 GB).
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& E) {
    const float au = fabsf(x) * 0.70710678118654752f;
    E = __expf(-au * au);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * au);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.f - poly * E;
    cdf = 0.5f * (1.f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, E;
    gelu_parts(x, cdf, E);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf, E;
    gelu_parts(x, cdf, E);
    return cdf + x * 0.39894228040143268f * E;
}
// ACT: 0 = none; 1 = forward, also write act(y_new) to AUX; 2 = backward, y_new *= act'(AUX) (AUX = pre-activation)
__device__ __forceinline__ uint4 act8(uint4 y, uint4 h, int act) {
    float v[8] = {bf_lo(y.x), bf_hi(y.x), bf_lo(y.y), bf_hi(y.y), bf_lo(y.z), bf_hi(y.z), bf_lo(y.w), bf_hi(y.w)};
    if (act == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = gelu_f(v[i]);
    } else {
        const float hv[8] = {bf_lo(h.x), bf_hi(h.x), bf_lo(h.y), bf_hi(h.y), bf_lo(h.z), bf_hi(h.z), bf_lo(h.w), bf_hi(h.w)};
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= gelu_grad_f(hv[i]);
    }
    return make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}

// ------------------------------------------------------------------------------------------
// second-stage, fixed-order reduction of the T3 partials into the fp32 gradient tensors
//   dst[r*sr + n*sn] (+)= scale * sum_rs part[rs][r][n]
// block = 64 consecutive (r, n) elements; wave w sums partials w, w+4, w+8, ... (independent loads in
// flight), then the four wave sums are combined in the fixed order 0,1,2,3 -> bit-reproducible.
// The bf16 backward does not launch it on its own: the blocks RIDE on the last kernel of the call, k_t2 over gx (the
// partials are complete by then -- stream order), as its leading blockIdx.y rows: 16 MB of partial reads hidden inside
// a 35-120 us streaming kernel instead of a dependent 7-9 us launch at the end of every backward call.
// ------------------------------------------------------------------------------------------
struct ReduceJob {
    const float* part;
    float* dst;
    int NR, RP, N, rank;
    long long sr, sn;
};
struct ReduceRide {
    ReduceJob j0, j1;
    float scale;
    int accumulate;
    int rows;       // leading blockIdx.y rows of the host kernel that run reduction blocks (0 = none riding)
    int nblk;       // reduction blocks per job
};

__device__ __forceinline__ void reduce_block(const ReduceJob& jb, long long blk, float scale, int accumulate, float* sm /* [4][64] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long idx = blk * 64 + lane;
    const bool ok = jb.dst != nullptr && idx < (long long)jb.rank * jb.N;
    int r = 0, n = 0;
    float s = 0.f;
    if (ok) {
        r = (int)(idx / jb.N);
        n = (int)(idx % jb.N);
        const long long stride = (long long)jb.RP * jb.N;
        const float* p = jb.part + (long long)r * jb.N + n;
        int rs = wave;
        for (; rs + 12 < jb.NR; rs += 16) {
            const float a0 = p[rs * stride], a1 = p[(rs + 4) * stride], a2 = p[(rs + 8) * stride],
                        a3 = p[(rs + 12) * stride];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; rs < jb.NR; rs += 4) s += p[rs * stride];
    }
    sm[wave * 64 + lane] = s;
    __syncthreads();
    if (ok && wave == 0) {
        const float tot = ((sm[lane] + sm[64 + lane]) + sm[128 + lane]) + sm[192 + lane];
        float* d = jb.dst + r * jb.sr + n * jb.sn;
        *d = accumulate ? (*d + scale * tot) : scale * tot;
    }
}

__global__ __launch_bounds__(256) void k_reduce(ReduceJob j0, ReduceJob j1, float scale, int accumulate) {
    __shared__ float sm[4 * 64];
    reduce_block(blockIdx.y == 0 ? j0 : j1, blockIdx.x, scale, accumulate, sm);
}

// LDS swizzle of the row-major bf16 slabs read with ds_read_b64_tr_b16 (k_t3, k_t3e, k_t2's GA tile)
__device__ __forceinline__ int t3_h(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename YT>
struct YTile;  // 16 rows x 128 cols, lane L owns rows p*4 + (L>>4), cols (L&15)*8 .. +8

template <>
struct YTile<bf16_t> {
    uint4 v[4];
    template <bool FAST, bool STREAM = false>     // STREAM: read once and not written back (the pre-activation tile)
    __device__ __forceinline__ void load(const bf16_t* Y, long long ldy, long long m0, int col, int lane,
                                         long long M, int N) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long long m = m0 + p * 4 + (lane >> 4);
            const bf16_t* src = Y + m * ldy + col;
            if (FAST) v[p] = STREAM ? ldg16(src) : ldg16_rmw(src);
            else v[p] = (m < M && col < N) ? (STREAM ? ldg16(src) : ldg16_rmw(src)) : zero4();
        }
    }
    // Q8: the tensor the NEXT frozen GEMM consumes (ACT == 1: act(y); ACT == 2: the updated gradient) also leaves as fp8
    // GA (ACT == 2): act(h) of the tile -- the input the GELU'-fused backward's LoRA layer saw in the forward, bit for bit what
    // k_t2<ACT = 1> stored -- is left in `atile` (bf16 [16 rows][16 chunks of 8], t3_h swizzle) for the gA contraction
    template <bool FAST, bool DROP, int ACT, bool Q8 = false, int QF = 0, bool GA = false>
    __device__ __forceinline__ void add_store(bf16_t* Y, long long ldy, long long m0, int col, int lane,
                                              long long M, int N, const float* slab, int ldw, float scale,
                                              const DropKey& dk, bf16_t* AUX, long long ldaux, const YTile<bf16_t>& haux,
                                              const Q8Out* q8 = nullptr, const Q8Scale* qs = nullptr, unsigned* seen = nullptr,
                                              uint4* atile = nullptr) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int rl = p * 4 + (lane >> 4);
            const long long m = m0 + rl;
            f32x4 a = *reinterpret_cast<const f32x4*>(slab + rl * ldw + (lane & 15) * 8);
            f32x4 b = *reinterpret_cast<const f32x4*>(slab + rl * ldw + (lane & 15) * 8 + 4);
            if (DROP) mask8(a, b, keep8((unsigned long long)m * dk.width + col, dk));
            uint4 o;
            o.x = pack2(bf_lo(v[p].x) + scale * a[0], bf_hi(v[p].x) + scale * a[1]);
            o.y = pack2(bf_lo(v[p].y) + scale * a[2], bf_hi(v[p].y) + scale * a[3]);
            o.z = pack2(bf_lo(v[p].z) + scale * b[0], bf_hi(v[p].z) + scale * b[1]);
            o.w = pack2(bf_lo(v[p].w) + scale * b[2], bf_hi(v[p].w) + scale * b[3]);
            const bool ok = FAST || (m < M && col < N);
            if (ACT == 2) o = act8(o, haux.v[p], 2);
            if (ACT == 2 && GA) {       // (zeros beyond M / N: h loads as 0); under dropout the layer's input was drop(act(h)): the same keep bits
                uint4 av = act8(haux.v[p], haux.v[p], 1);
                if (DROP) {
                    const unsigned kb = keep8((unsigned long long)m * dk.width + col, dk);
                    auto km = [&](int j) { return ((kb >> (2 * j)) & 1u ? 0xffffu : 0u) | ((kb >> (2 * j + 1)) & 1u ? 0xffff0000u : 0u); };
                    av = make_uint4(av.x & km(0), av.y & km(1), av.z & km(2), av.w & km(3));
                }
                atile[rl * 16 + ((lane & 15) ^ (t3_h(rl) << 1))] = av;
            }
            if (ok) stg16(Y + m * ldy + col, o);
            uint4 a8 = o;
            if (ACT == 1) {
                a8 = act8(o, o, 1);
                if (ok) stg16(AUX + m * ldaux + col, a8);
            }
            if (Q8 && ok) {         // the bf16-ROUNDED values, as a separate quantisation pass over the stored tensor would see them
                *reinterpret_cast<uint2*>(q8->q + m * q8->ld + col) = q8_pack8_bf16<QF>(a8, *qs, *seen);
            }
        }
    }
};

template <>
struct YTile<float> {
    f32x4 v[4][2];
    template <bool FAST, bool STREAM = false>
    __device__ __forceinline__ void load(const float* Y, long long ldy, long long m0, int col, int lane,
                                         long long M, int N) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long long m = m0 + p * 4 + (lane >> 4);
            const bool ok = FAST || (m < M && col < N);
            v[p][0] = ok ? *reinterpret_cast<const f32x4*>(Y + m * ldy + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
            v[p][1] = ok ? *reinterpret_cast<const f32x4*>(Y + m * ldy + col + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    template <bool FAST, bool DROP, int ACT, bool Q8 = false, int QF = 0, bool GA = false>
    __device__ __forceinline__ void add_store(float* Y, long long ldy, long long m0, int col, int lane,
                                              long long M, int N, const float* slab, int ldw, float scale,
                                              const DropKey& dk, float* AUX, long long ldaux, const YTile<float>& haux,
                                              const Q8Out* = nullptr, const Q8Scale* = nullptr, unsigned* = nullptr, uint4* = nullptr) {
        static_assert(!Q8 && !GA, "fp8 outputs and the in-pass gA contraction ride on bf16 tensors only");
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int rl = p * 4 + (lane >> 4);
            const long long m = m0 + rl;
            f32x4 a = *reinterpret_cast<const f32x4*>(slab + rl * ldw + (lane & 15) * 8);
            f32x4 b = *reinterpret_cast<const f32x4*>(slab + rl * ldw + (lane & 15) * 8 + 4);
            if (DROP) mask8(a, b, keep8((unsigned long long)m * dk.width + col, dk));
            if (FAST || (m < M && col < N)) {
                f32x4 o0 = v[p][0] + scale * a, o1 = v[p][1] + scale * b;
                if (ACT == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o0[j] *= gelu_grad_f(haux.v[p][0][j]);
                        o1[j] *= gelu_grad_f(haux.v[p][1][j]);
                    }
                }
                *reinterpret_cast<f32x4*>(Y + m * ldy + col) = o0;
                *reinterpret_cast<f32x4*>(Y + m * ldy + col + 4) = o1;
                if (ACT == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o0[j] = gelu_f(o0[j]);
                        o1[j] = gelu_f(o1[j]);
                    }
                    *reinterpret_cast<f32x4*>(AUX + m * ldaux + col) = o0;
                    *reinterpret_cast<f32x4*>(AUX + m * ldaux + col + 4) = o1;
                }
            }
        }
    }
};

// HL (RT == 2): W2t = [hi | lo] of the LoRA operand, T = [hi | lo] of t: delta = hi.t_hi + hi.t_lo + lo.t_hi.
// GA (ACT == 2, hi + lo, no dropout): the pass ALSO contracts act(h) -- recomputed from the pre-activation tile it holds anyway --
// with the gt fragments: GApart[row block][16][N] = sum over the block's rows of gt^T (x) act(h), i.e. the layer's gA partials,
// which k_t3 otherwise reads the whole stored activation a second time for (394 MB at N = 4736).
struct GaEmit {
    const bf16_t* GTTf;     // gt, fragment-major (k_gt_reduce<HL> / store_t4_hl): [32-row step][hi | lo][64 lanes][8]
    float* part;            // [row blocks][16][N]
};
template <typename YT, int RT, bool DROP, int ACT = 0, bool HL = false, bool Q8 = false, int QF = 0, bool GA = false>
__global__ __launch_bounds__(256, GA ? 2 : (HL && RT == 4) ? (ACT == 1 ? 2 : 3) : (HL && !DROP) ? (ACT == 2 ? 3 : 4) : 1) void k_t2(YT* __restrict__ Y, long long ldy, const bf16_t* __restrict__ T,
                                            const bf16_t* __restrict__ W2t, long long M, int N, float scale,
                                            int tiles_per_wg, DropKey dk, YT* __restrict__ AUX, long long ldaux,
                                            ReduceRide ride, Q8Out q8, int xcd_order, GaEmit ga) {
    static_assert(!GA || (ACT == 2 && HL && (RT == 2 || RT == 4) && sizeof(YT) == 2), "the in-pass gA contraction: GELU' pass of the hi + lo kernels, r <= 32");
    static_assert(!(GA && Q8) || (RT == 2 && !DROP), "the fp8 image rides on the r <= 16 form without a mask");
    static_assert(!Q8 || ACT != 0, "the fp8 image is the one of the activation-fused passes");
    static_assert(!HL || RT == 2 || RT == 4, "hi + lo operands: RT / 2 rank tiles, each as [hi 4 | lo 4] per 4 rank indices");
    constexpr int RH = HL ? RT / 2 : 1;         // hi + lo: rank tiles (r <= 16: 1, r <= 32: 2)
    constexpr int RP = RT * 16, CW = 128, LDW = CW + 4;
    __shared__ __attribute__((aligned(16))) float slab_all[4][16 * LDW];
    __shared__ uint4 atile_all[GA ? 4 : 1][GA ? 16 * 16 : 1];       // GA: act(h) of the wave's tile, bf16
    // r <= 32 (two rank tiles): the W fragments -- the same for the four waves -- live in LDS (16 KB), not in 64 registers per lane:
    // three workgroups per CU instead of two
    constexpr bool WSH = HL && RT == 4;
    __shared__ uint4 wsh[WSH ? 2 : 1][WSH ? 8 : 1][WSH ? 64 : 1];
    if (blockIdx.y < (unsigned)ride.rows) {     // riding reduction blocks (scheduled first; see reduce_block)
        const long long e = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        if (e < 2LL * ride.nblk)
            reduce_block(e < ride.nblk ? ride.j0 : ride.j1, e % ride.nblk, ride.scale, ride.accumulate, &slab_all[0][0]);
        return;
    }
    const unsigned lead = (unsigned)ride.rows;  // grid rows in front of the body's
    // body tile (bx = column block, by = row block): all column blocks of a row block on one XCD (they share its T rows)
    unsigned pbody = (blockIdx.y - lead) * gridDim.x + blockIdx.x;
    if (xcd_order) pbody = xcd_tile_index(pbody, gridDim.x * (gridDim.y - lead));
    const unsigned bx = pbody % gridDim.x, by = pbody / gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    float* slab = slab_all[wave];
    const int c0 = bx * CW;
    Q8Scale qs{0.f, 0.f, 0u};
    unsigned seen = 0u;         // packed running amax (q8_pack8_bf16)
    const unsigned q8_id = (bx + by * gridDim.x) * 4 + wave;
    if (Q8) qs = q8_begin(q8, bx == 0 && by == 0 && tid == 0, q8_id);

    // W2^T fragments (MFMA A-operand: i = output column, k = rank index), kept for the whole kernel
    uint2 wlo[8], whi[8];
    if (WSH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * 256 + tid, tt = e >> 9, ct = (e >> 6) & 7, l = e & 63;
            const int col = c0 + ct * 16 + (l & 15);
            wsh[tt][ct][l] = col < N ? *reinterpret_cast<const uint4*>(W2t + (long long)col * RP + tt * 32 + (l >> 4) * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
    }
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
        const int col = c0 + ct * 16 + n;
        const bool ok = col < N;
        wlo[ct] = whi[ct] = make_uint2(0u, 0u);
        if (WSH) {
        } else if (HL) {       // interleaved image: this lane's hi and lo quads are adjacent; rank tile tt at entries tt * 32 ..
            const uint4 w4 = ok ? *reinterpret_cast<const uint4*>(W2t + (long long)col * RP + g * 8) : make_uint4(0u, 0u, 0u, 0u);
            wlo[ct] = make_uint2(w4.x, w4.y);
            whi[ct] = make_uint2(w4.z, w4.w);
        } else {
            wlo[ct] = ok ? *reinterpret_cast<const uint2*>(W2t + (long long)col * RP + g * 4) : make_uint2(0u, 0u);
            if (RT == 2)
                whi[ct] = ok ? *reinterpret_cast<const uint2*>(W2t + (long long)col * RP + 16 + g * 4) : make_uint2(0u, 0u);
        }
    }

    const long long ntiles = (M + 15) / 16, nfull = M / 16;
    const long long t_begin = (long long)by * tiles_per_wg + wave;
    const long long t_end = min((long long)(by + 1) * tiles_per_wg, ntiles);
    const int col = c0 + (lane & 15) * 8;

    // T fragment of a tile (MFMA B-operand: k = rank index, n = activation row); T has Mp >= 16*ntiles rows
    uint4 tq2n = make_uint4(0u, 0u, 0u, 0u), tq2 = tq2n;        // r <= 32: the second rank tile's (t_hi | t_lo) quad: next / current tile
    auto load_t = [&](long long t, uint2& lo, uint2& hi) {
        if (HL) {       // (t_hi, t_lo) of this lane's 4 rank indices: one 16-byte load per rank tile (store_t4_hl's layout)
            const uint4 t4 = *reinterpret_cast<const uint4*>(T + (t * 16 + n) * RP + g * 8);
            lo = make_uint2(t4.x, t4.y);
            hi = make_uint2(t4.z, t4.w);
            if (RT == 4) tq2n = *reinterpret_cast<const uint4*>(T + (t * 16 + n) * RP + 32 + g * 8);
            return;
        }
        lo = *reinterpret_cast<const uint2*>(T + (t * 16 + n) * RP + g * 4);
        hi = make_uint2(0u, 0u);
        if (RT == 2) hi = *reinterpret_cast<const uint2*>(T + (t * 16 + n) * RP + 16 + g * 4);
    };
    auto delta_to_slab = [&](const uint2& tlo, const uint2& thi) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            if (RT == 1) {
                d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, wlo[ct]),
                                                             __builtin_bit_cast(s16x4, tlo), d, 0, 0, 0);
            } else if (HL) {
                // (w_hi, w_lo) . (t_hi, t_hi)  +  (w_hi, w_lo) . (t_lo, 0)  =  w_hi t_hi + w_lo t_hi + w_hi t_lo   (lo.lo is below
                // fp32 resolution).  Both are the K = 32 form on the SAME A-operand quad (the rank-32 kernel's registers: no extra
                // operand registers).  NOT a K = 16 MFMA for the third product: hipcc 7.2 emits v_mfma_f32_16x16x16_bf16 directly behind
                // the v_mfma_f32_16x16x32_bf16 whose result it accumulates onto, without wait states, and MI355X then returns wrong,
                // run-to-run varying sums (reproduced stand-alone: tools/probes/mfma_chain_probe.hip, profiles/r03b_mfma_k16_after_k32.txt;
                // 16 s_nop states by hand, or an unrelated MFMA in between, make it exact).  One MFMA shape per accumulator chain.
                const uint4 wa = WSH ? wsh[0][ct][lane] : make_uint4(wlo[ct].x, wlo[ct].y, whi[ct].x, whi[ct].y);
                const uint4 tb = make_uint4(tlo.x, tlo.y, tlo.x, tlo.y);
                const uint4 tc = make_uint4(thi.x, thi.y, 0u, 0u);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa), __builtin_bit_cast(bf16x8, tb), d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa), __builtin_bit_cast(bf16x8, tc), d, 0, 0, 0);
                if (RT == 4) {      // rank indices 16..31: the same two products on the second tile's quads
                    const uint4 tb2 = make_uint4(tq2.x, tq2.y, tq2.x, tq2.y);
                    const uint4 tc2 = make_uint4(tq2.z, tq2.w, 0u, 0u);
                    const uint4 w2 = wsh[WSH ? 1 : 0][ct][lane];
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2), __builtin_bit_cast(bf16x8, tb2), d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2), __builtin_bit_cast(bf16x8, tc2), d, 0, 0, 0);
                }
            } else {
                const uint4 wa = make_uint4(wlo[ct].x, wlo[ct].y, whi[ct].x, whi[ct].y);
                const uint4 tb = make_uint4(tlo.x, tlo.y, thi.x, thi.y);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa),
                                                           __builtin_bit_cast(bf16x8, tb), d, 0, 0, 0);
            }
            // lane (n, g): delta[row n][cols ct*16 + g*4 .. +4]
            *reinterpret_cast<f32x4*>(slab + n * LDW + ct * 16 + g * 4) = d;
        }
    };
    // GA: gt fragment of a 16-row tile as the K = 32 A-operand [hi rows 0..15 | lo rows 0..15] (lane (n, g): K slots 8 g .. 8 g + 7),
    // gathered from the 32-row-step image; the B-operand [act(h) rows 0..15 twice] comes from `atile` by transpose reads
    uint4* atile = atile_all[GA ? wave : 0];
    f32x4 gacc[GA ? RH : 1][GA ? 8 : 1];
    if (GA) {
#pragma unroll
        for (int tt = 0; tt < RH; ++tt)
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) gacc[tt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    struct GtFrag { uint4 v[GA ? RH : 1]; };
    // fragment-major image of gt (store_t4_hl<RH>): per 32-row step 2 RH blocks [hi tiles 0..RH-1 | lo tiles 0..RH-1] of 4 x 16 x 8
    auto load_gt = [&](long long t) -> GtFrag {
        GtFrag f;
#pragma unroll
        for (int tt = 0; tt < (GA ? RH : 1); ++tt)
            f.v[tt] = *reinterpret_cast<const uint4*>(ga.GTTf + ((((t >> 1) * (2 * RH) + (g >> 1) * RH + tt) * 4 + (int)(t & 1) * 2 + (g & 1)) * 16 + n) * 8);
        return f;
    };
    auto ga_accumulate = [&](const GtFrag& gf) {
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const char* base = reinterpret_cast<const char*>(atile);
        const int rowA = (g & 1) * 8 + (n >> 2), rowB = rowA + 4;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const int c = ct * 2 + ((n & 3) >> 1), half = n & 1;
            const char* pa = base + ((rowA * 16 + (c ^ (t3_h(rowA) << 1))) * 16 + half * 8);
            const char* pb = base + ((rowB * 16 + (c ^ (t3_h(rowB) << 1))) * 16 + half * 8);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
            const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            // D[i = rank idx][n = column] += sum_row (gt_hi + gt_lo)[row][i] * act(h)[row][col]
#pragma unroll
            for (int tt = 0; tt < (GA ? RH : 1); ++tt)
                gacc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gf.v[tt]), __builtin_bit_cast(bf16x8, both),
                                                                      gacc[tt][ct], 0, 0, 0);
        }
    };
    const bool colfull = c0 + CW <= N;
    const long long t_fast_end = colfull ? min(t_end, nfull) : t_begin;   // tiles with no tail at all
    long long t = t_begin;
    if (HL && !GA && tiles_per_wg <= 12 && t < t_fast_end) {
        // At most three tiles per wave (launch_t2's geometry at every size it was swept on): their T fragments are fetched UP FRONT, with
        // the W fragments, and the loop below is unrolled over them.  Inside the loop a T load sits in the middle of the queue of the tile
        // loads and the s_waitcnt that hands it to the next iteration also waits for the three tile loads issued before it; hoisted
        // (measured with the loads dropped altogether: 132.8 -> 126.5 us at N = 4736, 37.1 -> 36.0 at N = 1024).  The unrolled form also
        // stops re-reading the wave's own last tile as its "next" one.
        uint4 tp[3], tp2[RT == 4 ? 3 : 1];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const long long ti = t + 4 * i < t_end ? t + 4 * i : t;
            tp[i] = *reinterpret_cast<const uint4*>(T + (ti * 16 + n) * RP + g * 8);
            if (RT == 4) tp2[i] = *reinterpret_cast<const uint4*>(T + (ti * 16 + n) * RP + 32 + g * 8);
        }
        YTile<YT> cur, nxt, hcur, hnxt;
        nxt.template load<true>(Y, ldy, t * 16, col, lane, M, N);
        if (ACT == 2) hnxt.template load<true, true>(AUX, ldaux, t * 16, col, lane, M, N);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (t < t_fast_end) {       // wave-uniform
                cur = nxt;
                if (ACT == 2) hcur = hnxt;
                if (i < 2 && t + 4 < t_fast_end) {
                    nxt.template load<true>(Y, ldy, (t + 4) * 16, col, lane, M, N);
                    if (ACT == 2) hnxt.template load<true, true>(AUX, ldaux, (t + 4) * 16, col, lane, M, N);
                }
                if (RT == 4) tq2 = tp2[i];
                delta_to_slab(make_uint2(tp[i].x, tp[i].y), make_uint2(tp[i].z, tp[i].w));
                wave_sync();
                cur.template add_store<true, DROP, ACT, Q8, QF, GA>(Y, ldy, t * 16, col, lane, M, N, slab, LDW, scale, dk, AUX, ldaux, hcur, &q8, &qs, &seen, atile);
                wave_sync();
                t += 4;
            }
        }
    } else if (t < t_fast_end) {
        // branch-free stream: tile t+4 (and its t fragment, issued first) is in flight while tile t is updated.  (TWO tiles
        // ahead -- 124 VGPRs, the same 4 waves per SIMD, twice the bytes in flight -- measured SLOWER on MI355X: 159 vs 132 us
        // at N = 4736, profiles/r03j_adapter_sweep.json: this stream is not short of outstanding loads.)
        YTile<YT> cur, nxt, hcur, hnxt;      // hcur/hnxt: the pre-activation tile (ACT == 2 only)
        uint2 tlo, thi, nlo, nhi;
        GtFrag gfc{}, gfn{};
        load_t(t, nlo, nhi);
        if (GA) gfn = load_gt(t);
        nxt.template load<true>(Y, ldy, t * 16, col, lane, M, N);
        if (ACT == 2) hnxt.template load<true, true>(AUX, ldaux, t * 16, col, lane, M, N);
        for (; t < t_fast_end; t += 4) {
            cur = nxt;
            if (ACT == 2) hcur = hnxt;
            tlo = nlo;
            thi = nhi;
            tq2 = tq2n;
            gfc = gfn;
            const long long tn = t + 4 < t_fast_end ? t + 4 : t;   // last iteration re-reads its own tile (L2 hit)
            load_t(tn, nlo, nhi);
            if (GA) gfn = load_gt(tn);
            nxt.template load<true>(Y, ldy, tn * 16, col, lane, M, N);
            if (ACT == 2) hnxt.template load<true, true>(AUX, ldaux, tn * 16, col, lane, M, N);
            delta_to_slab(tlo, thi);
            wave_sync();
            cur.template add_store<true, DROP, ACT, Q8, QF, GA>(Y, ldy, t * 16, col, lane, M, N, slab, LDW, scale, dk, AUX, ldaux, hcur, &q8, &qs, &seen, atile);
            wave_sync();
            if (GA) ga_accumulate(gfc);
        }
    }
    for (; t < t_end; t += 4) {   // ragged tiles (last rows / last column chunk): predicated path
        YTile<YT> cur, hcur;
        uint2 tlo, thi;
        load_t(t, tlo, thi);
        tq2 = tq2n;
        cur.template load<false>(Y, ldy, t * 16, col, lane, M, N);
        if (ACT == 2) hcur.template load<false, true>(AUX, ldaux, t * 16, col, lane, M, N);
        delta_to_slab(tlo, thi);
        wave_sync();
        cur.template add_store<false, DROP, ACT, Q8, QF, GA>(Y, ldy, t * 16, col, lane, M, N, slab, LDW, scale, dk, AUX, ldaux, hcur, &q8, &qs, &seen, atile);
        wave_sync();
        if (GA) ga_accumulate(load_gt(t));
    }
    if (Q8) q8_end_wave(q8, q8_seen16_to_float(seen), q8_id, qs.have);
    if (GA) {   // fixed-order cross-wave sum ((w0 + w1) + w2) + w3 through LDS, as k_t3; wave w writes column tiles 2w, 2w + 1
        float* red = &slab_all[0][0];           // [4 waves][8 col tiles][64 lanes][4] floats = 32 KB of the 33 KB slab area
        float* out = ga.part + (long long)by * (16 * RH) * N;       // one rank tile at a time through the same area
#pragma unroll
        for (int tt = 0; tt < RH; ++tt) {
            __syncthreads();
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) *reinterpret_cast<f32x4*>(red + ((wave * 8 + ct) * 64 + lane) * 4) = gacc[tt][ct];
            __syncthreads();
#pragma unroll
            for (int jc = 0; jc < 2; ++jc) {
                const int ct = wave * 2 + jc;
                f32x4 s4 = *reinterpret_cast<const f32x4*>(red + ((0 * 8 + ct) * 64 + lane) * 4);
#pragma unroll
                for (int w = 1; w < 4; ++w) s4 += *reinterpret_cast<const f32x4*>(red + ((w * 8 + ct) * 64 + lane) * 4);
                const int ocol = c0 + ct * 16 + n;
                if (ocol < N) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) out[(long long)(tt * 16 + g * 4 + jj) * N + ocol] = s4[jj];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// T3: Gpart[rg][RP][N] = sum_{m in row group rg} T[m, :]^T (x) X[m, :]
//   WAVE-PRIVATE streaming like T1: workgroup = 128 columns x a row group; each of its 4 waves walks its
//   own quarter of the rows in 32-row steps (8 coalesced 16-B loads per lane: 4 rows x 256 B per
//   instruction) through a private 8 KB LDS slab; the B-operand is fetched with ds_read_b64_tr_b16
//   (hardware transpose); the A-operand (t^T) arrives fragment-major, 1 KB per step, from T1.
//   No barrier until the end, where the 4 waves' accumulators are added through LDS in fixed order.
// ------------------------------------------------------------------------------------------

// HL (RT == 2): the two fragment blocks of a step are the hi and lo parts of the same 16 rank indices and accumulate
// into ONE rank tile (RTA = 1): G = t_hi^T X + t_lo^T X.
// The kernel's body as a function of its tile number.
// `ptile`: tile number (column chunk fastest), `nchunks`: column chunks of N, `xs`: 32 KB of the workgroup's LDS.
template <typename XT, int RT, bool GATHER, bool DROP, bool HL>
__device__ __forceinline__ void t3_body(const XT* __restrict__ X, long long ldx, const bf16_t* __restrict__ TTf, float* __restrict__ Gpart,
                                        long long M, long long Mp, int N, int rows_per_wg, const DropKey& dk, unsigned ptile, unsigned nchunks,
                                        uint4 (*xs)[32 * 16]) {
    static_assert(!HL || RT == 2 || RT == 4, "hi + lo operands: fragment blocks [hi tiles | lo tiles] per step");
    constexpr int RTA = HL ? RT / 2 : RT;       // rank tiles of the result
    constexpr int RP = RTA * 16, CW = 128, CPR = 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int c0 = (int)(ptile % nchunks) * CW;
    const int rg = (int)(ptile / nchunks);
    // rows of this wave: quarter `wave` of [rg*rows_per_wg, +rows_per_wg), rows_per_wg % 128 == 0
    const long long w_begin = (long long)rg * rows_per_wg + (long long)wave * (rows_per_wg / 4);
    long long w_end = w_begin + rows_per_wg / 4;
    if (w_end > Mp) w_end = Mp;
    const int nst = w_end > w_begin ? (int)((w_end - w_begin) / 32) : 0, nst2 = (nst + 1) & ~1;
    const int lr = lane >> 4, lc = lane & 15;
    const int col = c0 + lc * 8;
    const int colc = col < N ? col : N - 8;
    const unsigned cmask = col < N ? 0xffffffffu : 0u;

    struct Regs {
        uint4 t[RT];        // t^T fragment of the step (issued BEFORE the x loads: it must land first)
        Raw8<XT> x[8];
    };
    // row indices and the row pitch as 32-bit values (check_common / check_act bound M and ld* below 2^31): one v_min_u32 + one
    // v_mad_u64_u32 per load instead of a 64-bit compare / select / multiply chain (120 VALU instructions per step before)
    const unsigned m_last = (unsigned)(M - 1), ldx32 = (unsigned)ldx, wb32 = (unsigned)w_begin;
    const XT* const Xc = X + colc;
    const bf16_t* const TTl = TTf + lane * 8;
    auto gload = [&](int s0, Regs& r_) {
        const unsigned mb = wb32 + (unsigned)(s0 < nst ? s0 : nst - 1) * 32u;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            r_.t[rt] = *reinterpret_cast<const uint4*>(TTl + (unsigned long long)((mb >> 5) * RT + rt) * 512u);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned m = min(mb + (unsigned)(lr + 4 * q), m_last);
            r_.x[q].load(Xc + (unsigned long long)m * ldx32);
        }
    };
    f32x4 acc[RTA][8];
#pragma unroll
    for (int rt = 0; rt < RTA; ++rt)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[rt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4* slab = xs[wave];
    auto put = [&](int s0, const Regs& r_) {          // the step's 32 x 128 tile into the wave's slab
        const long long mb = w_begin + (long long)s0 * 32;
        const unsigned smask = s0 < nst ? cmask : 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = lr + 4 * q;
            const long long m = mb + row;
            uint4 v = and4(r_.x[q].packed(), m < M ? smask : 0u);
            if (DROP) v = drop8(v, (unsigned long long)m * dk.width + col, dk);
            slab[row * CPR + (lc ^ (t3_h(row) << 1))] = v;
        }
    };
    auto eat = [&](const uint4* tf) {                  // acc += t^T (fragments tf) . slab
        wave_sync();
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            bf16x8 xf;
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            if (!GATHER) {
                // 16-lane group g reads the [4 rows x 16 cols] blocks at rows g*8+{0..3} and g*8+{4..7};
                // lane q supplies the address of row (q>>2), cols (q&3)*4..+3 and receives column q.
                const int rowA = g * 8 + (n >> 2), rowB = rowA + 4;
                const int c = ct * 2 + ((n & 3) >> 1), half = n & 1;
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                const char* base = reinterpret_cast<const char*>(slab);
                const char* pa = base + ((rowA * CPR + (c ^ (t3_h(rowA) << 1))) * 16 + half * 8);
                const char* pb = base + ((rowB * CPR + (c ^ (t3_h(rowB) << 1))) * 16 + half * 8);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
                const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                xf = __builtin_bit_cast(bf16x8, both);
            } else {
                // validation path: explicit 2-byte gathers, lane (n, g) <- X[g*8+jj][ct*16+n]
                const bf16_t* b16 = reinterpret_cast<const bf16_t*>(slab);
                s16x8 both;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int row = g * 8 + jj;
                    const int e = ct * 16 + n, c = e >> 3;
                    both[jj] = (short)b16[(row * CPR + (c ^ (t3_h(row) << 1))) * 8 + (e & 7)];
                }
                xf = __builtin_bit_cast(bf16x8, both);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                // D[i = rank idx][n = column] += sum_m T[m][i] * X[m][col]
                acc[rt % RTA][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, tf[rt]), xf,
                                                                           acc[rt % RTA][ct], 0, 0, 0);    // HL: hi and lo block of a tile -> one accumulator
        }
        wave_sync();
    };
    auto stage = [&](int s0, const Regs& r_) {
        put(s0, r_);
        eat(r_.t);
    };

    if (nst > 0) {
        Regs rA, rB;               // distance-1 prefetch, two named register sets (see k_t1)
        gload(0, rA);
        for (int s = 0; s < nst2; s += 2) {
            gload(s + 1, rB);
            __builtin_amdgcn_sched_barrier(0);     // the next step's loads go out BEFORE this step is consumed (hipcc otherwise hoists
            stage(s, rA);                          // the first consumer above them and waits for most of the queue first)
            gload(s + 2, rA);
            __builtin_amdgcn_sched_barrier(0);
            stage(s + 1, rB);      // zeros if it is the padding step
        }
    }
    // fixed-order cross-wave sum ((w0 + w1) + w2) + w3 through LDS, one rank tile at a time; wave w then
    // writes column tiles 2w and 2w+1 of the partial
    float* red = reinterpret_cast<float*>(&xs[0][0]);     // [4 waves][8 col tiles][64 lanes][4] floats = 32 KB
    float* out = Gpart + (long long)rg * RP * N;
#pragma unroll
    for (int rt = 0; rt < RTA; ++rt) {
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
            *reinterpret_cast<f32x4*>(red + ((wave * 8 + ct) * 64 + lane) * 4) = acc[rt][ct];
        __syncthreads();
#pragma unroll
        for (int jc = 0; jc < 2; ++jc) {
            const int ct = wave * 2 + jc;
            f32x4 s4 = *reinterpret_cast<const f32x4*>(red + ((0 * 8 + ct) * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < 4; ++w) s4 += *reinterpret_cast<const f32x4*>(red + ((w * 8 + ct) * 64 + lane) * 4);
            const int ocol = c0 + ct * 16 + n;
            if (ocol < N) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) out[(long long)(rt * 16 + g * 4 + jj) * N + ocol] = s4[jj];
            }
        }
    }
}

template <typename XT, int RT, bool GATHER, bool DROP, bool HL = false>
__global__ __launch_bounds__(256, (HL && RT == 4 && !DROP) ? 2 : 1) void k_t3(const XT* __restrict__ X, long long ldx,
                                            const bf16_t* __restrict__ TTf, float* __restrict__ Gpart,
                                            long long M, long long Mp, int N, int rows_per_wg, DropKey dk, int xcd_order) {
    __shared__ uint4 xs[4][32 * 16];       // 32 rows x 256 B per wave; reused as the reduction buffer
    // tile (column chunk, row group): the chunks of a row group on one XCD (they share its t^T fragments)
    unsigned ptile = blockIdx.y * gridDim.x + blockIdx.x;
    if (xcd_order) ptile = xcd_tile_index(ptile, gridDim.x * gridDim.y);
    t3_body<XT, RT, GATHER, DROP, HL>(X, ldx, TTf, Gpart, M, Mp, N, rows_per_wg, dk, ptile, gridDim.x, xs);
}

// ------------------------------------------------------------------------------------------
// T3C (round 6; bf16, hi + lo, r <= 16): k_t3's contraction with the tile handed over the way k_t1 hands its chunks over -- all 256 threads
// load ONE [64 rows x 128 columns] tile (4 x 16 bytes per thread), write it to the LDS, barrier; one tile is in flight while the previous
// one is contracted.  Same ownership as k_t3 (workgroup = column chunk x row group), same partial layout, but the four waves walk the SAME
// rows in lockstep instead of four quarters of the group on their own.  Why: the bare access patterns, measured
// (tools/probes/read_patterns.hip, profiles/r06u_read_patterns.txt, 393 MB, 481 workgroups): wave-private 32 x 256-byte pieces 5.2 TB/s
// (k_t3 itself: 5.4-5.5); the cooperative 64-row tile walking down the rows 6.7 TB/s -- as fast as a linear read.  The HBM system rewards
// a workgroup whose loads of one instant cover neighbouring rows, with FEWER bytes under way (16 KB per workgroup against 32).
// Wave w owns column tiles 2w, 2w + 1 of the chunk over ALL rows of the group: 8 accumulator registers instead of 32, no cross-wave sum at
// the end; the t^T fragments of the 64 rows (4 KB, contiguous in the fragment-major image) are fetched ONCE per workgroup (one 16-byte load
// per thread) and shared through the LDS instead of once per wave.
// ------------------------------------------------------------------------------------------
// CT: 16-column tiles per wave -- 2: a 128-column chunk, [64 rows x 256 B] tiles (what is launched); 4: a 256-column chunk, [32 rows x 512 B]
// tiles: the same 16 KB per tile and the same bare pattern speed (profiles/r06w_read_patterns.txt) with HALF the t^T bytes per activation
// byte -- measured EQUAL at N = 4736 and at 1024 (71-74 against 69-71 us, 21.9 against 21.7: profiles/r06y_sweep_t3c_cw.json), so the
// fragment traffic is not what is left here; kept as a template parameter, not dispatched.
template <bool DROP, int CT>
__global__ __launch_bounds__(256) void k_t3c(const bf16_t* __restrict__ X, long long ldx, const bf16_t* __restrict__ TTf,
                                             float* __restrict__ Gpart, long long M, long long Mp, int N, int rows_per_wg, DropKey dk,
                                             int xcd_order) {
    static_assert(CT == 2 || CT == 4, "column tiles per wave");
    constexpr int CW = CT * 64, SUB = CW / 128, TR = 128 / CT;         // tile = TR rows x CW columns = 16 KB
    constexpr int LPR = CW / 8, RPP = 256 / LPR, NH = TR / 32;        // lanes per tile row, rows per pass, 32-row halves per tile
    __shared__ uint4 xs[2][NH * SUB][32 * 16];  // [buffer][32-row half x 128-column sub-chunk][row][16-byte chunk]: k_t3's slab layout each
    __shared__ uint4 ts[2][NH * 128];           // [buffer][(half * 2 + hi / lo) * 64 + lane]
    unsigned ptile = blockIdx.y * gridDim.x + blockIdx.x;
    if (xcd_order) ptile = xcd_tile_index(ptile, gridDim.x * gridDim.y);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int c0 = (int)(ptile % gridDim.x) * CW;
    const int rg = (int)(ptile / gridDim.x);
    const long long w_begin = (long long)rg * rows_per_wg;      // rows_per_wg % 128 == 0, Mp % 64 == 0: whole tiles
    long long w_end = w_begin + rows_per_wg;
    if (w_end > Mp) w_end = Mp;
    const int nst = w_end > w_begin ? (int)((w_end - w_begin) / TR) : 0;
    const int lr = tid / LPR, lc = tid % LPR;
    const int col = c0 + lc * 8;
    const unsigned cmask = col < N ? 0xffffffffu : 0u;
    const unsigned m_last = (unsigned)(M - 1), ldx32 = (unsigned)ldx, wb32 = (unsigned)w_begin;
    const bf16_t* const Xc = X + (col < N ? col : N - 8);
    uint4 xr[4], tr_;
    auto gload = [&](int s0) {
        const unsigned mb = wb32 + (unsigned)s0 * TR;
        // the tile's t^T fragments: [hi | lo] x 64 lanes x 8 per 32-row step, contiguous in the fragment-major image
        // (CT == 4: 2 KB per tile -- the upper half of the workgroup fetches and stores duplicates: a load under `tid < 128` is a divergent
        // branch, and hipcc then drains the whole load queue behind it before the tile loads are issued: 90 us instead of 70)
        tr_ = *reinterpret_cast<const uint4*>(TTf + (unsigned long long)(mb >> 5) * 1024u + (unsigned)(tid & (NH * 128 - 1)) * 8u);
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[i] = ldg16(Xc + (unsigned long long)min(mb + (unsigned)(lr + RPP * i), m_last) * ldx32);
    };
    auto sstore = [&](int buf, int s0) {
        const long long mb = w_begin + (long long)s0 * TR;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lr + RPP * i, r32 = row & 31;
            const long long m = mb + row;
            uint4 v = and4(xr[i], m < M ? cmask : 0u);
            if (DROP) v = drop8(v, (unsigned long long)m * dk.width + col, dk);
            xs[buf][(row >> 5) * SUB + (lc >> 4)][r32 * 16 + ((lc & 15) ^ (t3_h(r32) << 1))] = v;
        }
        ts[buf][tid & (NH * 128 - 1)] = tr_;
    };
    f32x4 acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (nst > 0) {
        gload(0);
        sstore(0, 0);
        __syncthreads();
        for (int s = 0; s < nst; ++s) {
            const int buf = s & 1;
            if (s + 1 < nst) gload(s + 1);
            __builtin_amdgcn_sched_barrier(0);      // the next tile's loads go out before this one is contracted
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const bf16x8 thi = __builtin_bit_cast(bf16x8, ts[buf][(h * 2 + 0) * 64 + lane]);
                const bf16x8 tlo = __builtin_bit_cast(bf16x8, ts[buf][(h * 2 + 1) * 64 + lane]);
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    // as k_t3: 16-lane group g reads the [4 rows x 16 cols] blocks at rows g*8 + {0..3} and g*8 + {4..7} of the half
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                    const int ctw = wave * CT + j, ct = ctw & 7;
                    const int rowA = g * 8 + (n >> 2), rowB = rowA + 4;
                    const int c = ct * 2 + ((n & 3) >> 1), half = n & 1;
                    const char* base = reinterpret_cast<const char*>(&xs[buf][h * SUB + (ctw >> 3)][0]);
                    const char* pa = base + ((rowA * 16 + (c ^ (t3_h(rowA) << 1))) * 16 + half * 8);
                    const char* pb = base + ((rowB * 16 + (c ^ (t3_h(rowB) << 1))) * 16 + half * 8);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
                    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x8 xf = __builtin_bit_cast(bf16x8, both);
                    // D[i = rank idx][n = column] += sum_m (t_hi + t_lo)[m][i] * X[m][col]
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(thi, xf, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tlo, xf, acc[j], 0, 0, 0);
                }
            }
            if (s + 1 < nst) sstore(buf ^ 1, s + 1);
            __syncthreads();
        }
    }
    float* out = Gpart + (long long)rg * 16 * N;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int ocol = c0 + (wave * CT + j) * 16 + n;
        if (ocol < N) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) out[(long long)(g * 4 + jj) * N + ocol] = acc[j][jj];
        }
    }
}

// ------------------------------------------------------------------------------------------
// T3E (r <= 16, backward over gy): ONE pass over gy produces both the column reduction gB = t^T . gy (as k_t3) and
// the row reduction gt = gy . B_c^T that k_t1 would otherwise re-read gy for.  A workgroup owns TWO 128-column chunks
// of a row group; each wave streams its rows as sub-steps (rows of step s, chunk 0), (rows of step s, chunk 1) through
// its private slab: the distance-1 prefetch of k_t3 with the two named register sets bound to the two chunks.
// Per sub-step: 8 tr-read MFMAs into that chunk's gB accumulators + 8 row-major MFMAs contracting the slab with the
// chunk's 128 columns of B_c (W1b image, in registers) into the step's gt partial, written after the second chunk to
// GTP[column pair][row][16] (fp32).  k_gt_reduce sums the pairs in fixed order.  Extra traffic: 2 x ceil(N/256) x M
// x 64 B (partials out and back, 100 MB at N = 4736) instead of a second M x N x 2 B read (393 MB).
// ------------------------------------------------------------------------------------------
// HL: TTf holds (t_hi, t_lo) blocks per step (rank-32 layout) and W1b rows 16..31 the lo parts of B_c: both products
// run hi + lo into the same fp32 accumulators.
template <typename XT, bool HL>
__global__ __launch_bounds__(256) void k_t3e(const XT* __restrict__ X, long long ldx, const bf16_t* __restrict__ TTf,
                                             float* __restrict__ Gpart, long long M, long long Mp, int N,
                                             int rows_per_wg, const bf16_t* __restrict__ W1b,
                                             float* __restrict__ GTP, int xcd_order) {
    constexpr int RP = 16, CPR = 16, NH = HL ? 2 : 1;
    __shared__ uint4 xs[4][32 * CPR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    unsigned ptile = blockIdx.y * gridDim.x + blockIdx.x;       // tile order as in k_t3
    if (xcd_order) ptile = xcd_tile_index(ptile, gridDim.x * gridDim.y);
    const unsigned cpair = ptile % gridDim.x;
    const int c0 = (int)cpair * 256;
    const int rg = (int)(ptile / gridDim.x);
    const long long w_begin = (long long)rg * rows_per_wg + (long long)wave * (rows_per_wg / 4);
    long long w_end = w_begin + rows_per_wg / 4;
    if (w_end > Mp) w_end = Mp;
    const int nst = w_end > w_begin ? (int)((w_end - w_begin) / 32) : 0;
    const int lr = lane >> 4, lc = lane & 15;
    int colc[2];
    unsigned cmask[2];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int col = c0 + cc * 128 + lc * 8;
        colc[cc] = col < N ? col : N - 8;
        cmask[cc] = col < N ? 0xffffffffu : 0u;
    }
    __shared__ uint4 wbs[NH * 8][64];        // [half][chunk][ks]: B operand of the gt contraction, staged below
    struct Regs {
        uint4 t[NH];
        Raw8<XT> x[8];
    };
    auto gload = [&](int s0, int cc, Regs& r_) {
        const int s = s0 < nst ? s0 : nst - 1;
        const long long mb = w_begin + (long long)s * 32;
#pragma unroll
        for (int h = 0; h < NH; ++h)
            r_.t[h] = *reinterpret_cast<const uint4*>(TTf + (((mb >> 5) * NH + h) * 64 + lane) * 8);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const long long m = mb + lr + 4 * q;
            r_.x[q].load(X + (m < M ? m : M - 1) * ldx + colc[cc]);
        }
    };
    f32x4 acc[2][8];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[cc][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ga[2];
    uint4* slab = xs[wave];
    auto stage = [&](int s0, int cc, const Regs& r_) {
        const long long mb = w_begin + (long long)s0 * 32;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = lr + 4 * q;
            slab[row * CPR + (lc ^ (t3_h(row) << 1))] = and4(r_.x[q].packed(), mb + row < M ? cmask[cc] : 0u);
        }
        wave_sync();
#pragma unroll
        for (int rtile = 0; rtile < 2; ++rtile) {
            if (cc == 0) ga[rtile] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = rtile * 16 + n;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 xa = slab[row * CPR + ((ks * 4 + g) ^ (t3_h(row) << 1))];
                // D[i = rank idx][n = row] += sum_col B_c[rank idx][col] * gy[row][col]: the transposed product, so that a lane
                // ends with 4 consecutive rank entries of one row -- one 16-byte store, 1 KB contiguous per wave (the other
                // operand order left 4-byte stores 64 B apart: 780 K partial-line writes per launch at N = 4736)
#pragma unroll
                for (int h = 0; h < NH; ++h)
                    ga[rtile] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wbs[h * 8 + cc * 4 + ks][lane]),
                                                                       __builtin_bit_cast(bf16x8, xa), ga[rtile], 0, 0, 0);
            }
            if (cc == 1)
                *reinterpret_cast<f32x4*>(GTP + ((long long)cpair * Mp + mb + rtile * 16 + n) * 16 + g * 4) = ga[rtile];
        }
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            const int rowA = g * 8 + (n >> 2), rowB = rowA + 4;
            const int c = ct * 2 + ((n & 3) >> 1), half = n & 1;
            const char* base = reinterpret_cast<const char*>(slab);
            const char* pa = base + ((rowA * CPR + (c ^ (t3_h(rowA) << 1))) * 16 + half * 8);
            const char* pb = base + ((rowB * CPR + (c ^ (t3_h(rowB) << 1))) * 16 + half * 8);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
            const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int h = 0; h < NH; ++h)
                acc[cc][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, r_.t[h]),
                                                                     __builtin_bit_cast(bf16x8, both), acc[cc][ct], 0, 0, 0);
        }
        wave_sync();
    };
    Regs rA, rB;                   // rA <-> chunk 0, rB <-> chunk 1; one sub-step of prefetch distance
    if (nst > 0) gload(0, 0, rA);  // in flight while the B_c fragments are staged
    // B operand of the gt contraction, per lane: B_c[r = n][c0 + cc*128 + ks*32 + g*8 .. +8]; the same for all four
    // waves and all row steps -> staged once in LDS in fragment order (32 registers per lane otherwise)
    if (wave < 2 * NH) {
        const int cc = wave & 1, h = wave >> 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cb = c0 + cc * 128 + ks * 32 + g * 8;
            wbs[h * 8 + cc * 4 + ks][lane] =
                and4(*reinterpret_cast<const uint4*>(W1b + (long long)(h * 16 + n) * N + (cb < N ? cb : N - 8)), cb < N ? 0xffffffffu : 0u);
        }
    }
    __syncthreads();
    if (nst > 0) {
        for (int s = 0; s < nst; ++s) {
            gload(s, 1, rB);
            stage(s, 0, rA);
            gload(s + 1, 0, rA);   // clamped to the last step at the end (harmless re-read)
            stage(s, 1, rB);
        }
    }
    // fixed-order cross-wave sum of the gB accumulators, one chunk at a time (as k_t3 does per rank tile)
    float* red = reinterpret_cast<float*>(&xs[0][0]);
    float* out = Gpart + (long long)rg * RP * N;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
            *reinterpret_cast<f32x4*>(red + ((wave * 8 + ct) * 64 + lane) * 4) = acc[cc][ct];
        __syncthreads();
#pragma unroll
        for (int jc = 0; jc < 2; ++jc) {
            const int ct = wave * 2 + jc;
            f32x4 s4 = *reinterpret_cast<const f32x4*>(red + ((0 * 8 + ct) * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < 4; ++w) s4 += *reinterpret_cast<const f32x4*>(red + ((w * 8 + ct) * 64 + lane) * 4);
            const int ocol = c0 + cc * 128 + ct * 16 + n;
            if (ocol < N) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) out[(long long)(g * 4 + jj) * N + ocol] = s4[jj];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// gt = sum over column pairs (fixed order) of the fp32 partials k_t3e wrote -> bf16 T[Mp, 16] row-major
// (k_t2's operand) and TTf fragment-major (k_t3's operand), exactly the two images k_t1 would have produced.
// ------------------------------------------------------------------------------------------
template <bool HL, int RH = 1>
__global__ __launch_bounds__(256) void k_gt_reduce(const float* __restrict__ GTP, int nchunks, bf16_t* __restrict__ T,
                                                   bf16_t* __restrict__ TTf, long long Mp) {
    static_assert(RH == 1 || HL, "two rank tiles: the hi + lo images of r <= 32 (k_t3w<RH = 2>)");
    __shared__ __attribute__((aligned(16))) bf16_t stage[(HL && RH == 1) ? 2048 : 8];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;     // one thread = 4 consecutive rank entries
    if (idx >= Mp * 4 * RH) return;     // (Mp is a multiple of 64: at RH == 1 a workgroup is 64 whole rows, never cut)
    const long long m = idx / (4 * RH);
    const int r0 = (int)(idx % (4 * RH)) * 4;
    // the partials of a row in groups of eight, every load of a group issued before the first is consumed (the chunk count is a run-time
    // value: the plain loop waited for each load in turn -- five dependent round trips to the L2 / Infinity Cache at N = 4736); the sum
    // keeps its fixed order 0, 1, 2, ...
    f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cb = 0; cb < nchunks; cb += 8) {
        f32x4 part[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int cc = cb + c < nchunks ? cb + c : nchunks - 1;
            part[c] = __builtin_bit_cast(f32x4, ldg16(GTP + ((long long)cc * Mp + m) * (16 * RH) + r0));
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (cb + c < nchunks) s4 += part[c];
    }
    if (HL && RH == 1) store_t4_hl_wg64(T, TTf, (long long)blockIdx.x * 64, (int)(threadIdx.x >> 2), r0, s4, stage);
    else if (HL) store_t4_hl<RH>(T, TTf, m, r0 >> 4, r0 & 15, s4);
    else store_t4<1>(T, TTf, m, 0, r0, pack2(s4[0], s4[1]), pack2(s4[2], s4[3]));
}

// ------------------------------------------------------------------------------------------
// Backward, version 2 (hi + lo operands, r <= 16): the pass over gy.
//
// T3W  k_t3w  the pass over gy.  Like k_t3e it yields the gB partials AND gt = gy . B_c^T from one read of gy, but the
//             workgroup's 8 waves split its COLUMNS (one 128-column chunk each: 1024 columns per workgroup) and walk the same
//             32-row steps; the eight chunk contributions to gt meet in LDS once per step (one raw barrier, double-buffered
//             exchange area; the prefetch of the next step is in flight across it).  gt therefore leaves as ceil(N / 1024)
//             partials (5 at N = 4736) instead of ceil(N / 256) (19): 13 MB out and back instead of 50, and with
//             N <= 1024 (FINAL) the sum is complete in the kernel and the two bf16 images of gt are written directly --
//             no k_gt_reduce launch.  A wave owns its columns for the whole row group, so the gB partial goes from its
//             accumulators to memory without the cross-wave pass of k_t3 / k_t3e.  The spill balance:
//             gt partials 64 M ceil(N / 1024) bytes + gB partials 64 N x row groups, 28 MB at M = 41,472, N = 4736 with one
//             workgroup per CU, against 58 MB for k_t3e's 256-column pairs.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier() {     // workgroup barrier that orders LDS traffic only: global loads in flight stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// RH: rank tiles of the group (1: r <= 16, 2: r <= 32 -- round 6: the reference's default rank took two passes over gy before, k_t1<RT = 4>
// for gt and k_t3<RT = 4> for gB).  Operand images as everywhere: W1b = [hi rows 0 .. 16 RH - 1 | lo rows] x N; TTf = per 32-row step
// 2 RH blocks [hi tiles | lo tiles]; gt leaves with 16 RH entries per row.  RH = 2 needs 128 KB of LDS and ~250 registers: one workgroup
// per CU, which is what the plan launches anyway.
template <typename XT, bool FINAL, int RH = 1>
__global__ __launch_bounds__(512, RH == 1 ? 2 : 1) void k_t3w(const XT* __restrict__ X, long long ldx, const bf16_t* __restrict__ TTf,
                                                float* __restrict__ Gpart, long long M, long long Mp, int N, int steps_per_wg,
                                                const bf16_t* __restrict__ W1b, float* __restrict__ GTP,
                                                bf16_t* __restrict__ T_out, bf16_t* __restrict__ TTf_out, int xcd_order) {
    constexpr int CPR = 16, RW = 16 * RH;                                        // RW: rank entries per row of gt
    __shared__ uint4 xs[8][32 * CPR];                                           // 8 KB per wave: its 32 x 128 tile
    __shared__ __attribute__((aligned(16))) float gtx[2][8][32 * RW];           // [step parity][wave][row][rank]: gt contributions
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    unsigned ptile = blockIdx.y * gridDim.x + blockIdx.x;       // (column group, row group): the groups of a row range on one XCD
    if (xcd_order) ptile = xcd_tile_index(ptile, gridDim.x * gridDim.y);
    const int cg = (int)(ptile % gridDim.x), rg = (int)(ptile / gridDim.x);
    const int nchunks = (N + 127) / 128;
    const int chunk = cg * 8 + wave;
    const bool active = chunk < nchunks;                        // wave-uniform
    const int c0 = chunk * 128;
    const int s_begin = rg * steps_per_wg;
    const int s_total = (int)(Mp >> 5);
    const int nst = (s_begin + steps_per_wg < s_total ? s_begin + steps_per_wg : s_total) - s_begin;
    const int lr = lane >> 4, lc = lane & 15;
    const int col = c0 + lc * 8;
    const int colc = (active && col < N) ? col : (N - 8);
    const unsigned cmask = (active && col < N) ? 0xffffffffu : 0u;

    // B operand of the gt contraction: B_c[r = tt*16 + n][c0 + ks*32 + g*8 .. +8], hi and lo rows of the image
    uint4 bw[RH][2][4];
#pragma unroll
    for (int tt = 0; tt < RH; ++tt)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cb = c0 + ks * 32 + g * 8;
                const bool okc = active && cb < N;
                bw[tt][h][ks] = and4(*reinterpret_cast<const uint4*>(W1b + (long long)(h * RW + tt * 16 + n) * N + (okc ? cb : N - 8)), okc ? 0xffffffffu : 0u);
            }
    struct Regs {
        uint4 t[2 * RH];
        Raw8<XT> x[8];
    };
    // row indices and the row pitch as 32-bit values (the launcher checks M and ldx < 2^31): one v_min_u32 + one v_mad_u64_u32 per
    // load instead of a 64-bit compare / select / multiply chain
    const unsigned m_last = (unsigned)(M - 1), ldx32 = (unsigned)ldx;
    const XT* const Xc = X + colc;
    const bf16_t* const TTl = TTf + lane * 8;
    auto gload = [&](int s0, Regs& r_) {
        const unsigned s = (unsigned)(s_begin + (s0 < nst ? s0 : nst - 1));
#pragma unroll
        for (int b = 0; b < 2 * RH; ++b) r_.t[b] = *reinterpret_cast<const uint4*>(TTl + (unsigned long long)(s * (2u * RH) + b) * 512u);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned m = min(s * 32u + (unsigned)(lr + 4 * q), m_last);
            r_.x[q].load(Xc + (unsigned long long)m * ldx32);
        }
    };
    f32x4 acc[RH][8];
#pragma unroll
    for (int tt = 0; tt < RH; ++tt)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[tt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4* slab = xs[wave];
    struct TFrag { uint4 t[2 * RH]; };
    auto put = [&](int s0, const Regs& r_) {        // the step's tile into the wave's slab
        const long long mb = (long long)(s_begin + s0) * 32;
        const unsigned smask = s0 < nst ? cmask : 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = lr + 4 * q;
            slab[row * CPR + (lc ^ (t3_h(row) << 1))] = and4(r_.x[q].packed(), mb + row < M ? smask : 0u);
        }
        wave_sync();
    };
    auto eat = [&](int s0, const TFrag& tf) {       // both contractions on the slab
        float* gout = &gtx[s0 & 1][wave][0];
#pragma unroll
        for (int rtile = 0; rtile < 2; ++rtile) {
            f32x4 ga[RH];
#pragma unroll
            for (int tt = 0; tt < RH; ++tt) ga[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = rtile * 16 + n;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 xa = slab[row * CPR + ((ks * 4 + g) ^ (t3_h(row) << 1))];
                // D[i = rank idx][n = row] += sum_col B_c[rank idx][col] * gy[row][col]   (k_t3e's transposed product)
#pragma unroll
                for (int tt = 0; tt < RH; ++tt)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        ga[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bw[tt][h][ks]), __builtin_bit_cast(bf16x8, xa), ga[tt], 0, 0, 0);
            }
#pragma unroll
            for (int tt = 0; tt < RH; ++tt)      // lane (n, g): gt[row n][ranks tt*16 + 4g .. + 3]
                *reinterpret_cast<f32x4*>(gout + (rtile * 16 + n) * RW + tt * 16 + g * 4) = ga[tt];
        }
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            const int rowA = g * 8 + (n >> 2), rowB = rowA + 4;
            const int c = ct * 2 + ((n & 3) >> 1), half = n & 1;
            const char* base = reinterpret_cast<const char*>(slab);
            const char* pa = base + ((rowA * CPR + (c ^ (t3_h(rowA) << 1))) * 16 + half * 8);
            const char* pb = base + ((rowB * CPR + (c ^ (t3_h(rowB) << 1))) * 16 + half * 8);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
            const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int tt = 0; tt < RH; ++tt)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[tt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, tf.t[h * RH + tt]), __builtin_bit_cast(bf16x8, both), acc[tt][ct], 0, 0, 0);
        }
        wave_sync();
    };
    auto stage = [&](int s0, const Regs& r_) {
        TFrag tf;
#pragma unroll
        for (int b = 0; b < 2 * RH; ++b) tf.t[b] = r_.t[b];
        put(s0, r_);
        eat(s0, tf);
    };
    // after the step's barrier: 2 RH waves (rotating with the step) add the eight contributions in the fixed order 0 .. 7 and emit
    // 4 consecutive rank entries per lane
    auto emit = [&](int s0) {
        constexpr int EW = 2 * RH;                      // emitting waves: 64 EW lanes x 4 entries = the step's 32 rows x RW ranks
        const int rw = (s0 % (8 / EW)) * EW;
        if (wave < rw || wave >= rw + EW) return;
        const int l2 = (wave - rw) * 64 + lane;
        const int row = l2 / (4 * RH), r0 = (l2 % (4 * RH)) * 4;
        const float* gin = &gtx[s0 & 1][0][row * RW + r0];
        f32x4 s4 = *reinterpret_cast<const f32x4*>(gin);
#pragma unroll
        for (int w = 1; w < 8; ++w) s4 += *reinterpret_cast<const f32x4*>(gin + w * (32 * RW));
        const long long m = (long long)(s_begin + s0) * 32 + row;
        if (FINAL) store_t4_hl<RH>(T_out, TTf_out, m, r0 >> 4, r0 & 15, s4);
        else *reinterpret_cast<f32x4*>(GTP + ((long long)cg * Mp + m) * RW + r0) = s4;
    };
    // One code path for every wave (an idle wave of the last column group streams masked duplicates of the group's last 16
    // bytes per row -- one line per row, L2 hits): no divergent control flow around the loads, so that the waits stay counted.
    // The next step's loads are issued BEFORE the current step is consumed (sched_barrier pins that order: hipcc otherwise hoists
    // the first consumer above them and waits for the whole queue first).
    // (Two steps ahead -- three register sets at one workgroup per CU, 128 KB per CU under way instead of 64 -- measured equal on MI355X,
    // round 6: 73.2-73.4 us against 73.2-73.6 at N = 4736, 27.6-29.1 against 26.7-29.3 at N = 1024 (profiles/r06c_sweep_t3w_depth.json):
    // this pass is not short of outstanding loads.)
    if constexpr (RH == 2) {
        // ONE register set (two would not fit beside 64 accumulators and 64 operand registers: 68-104 bytes of scratch per lane): the
        // slab is the second buffer -- the step's tile goes to the LDS, the NEXT step's loads are issued into the same registers and are
        // in flight while both contractions run on the slab
        Regs r;
        gload(0, r);
        for (int s = 0; s < nst; ++s) {
            TFrag tf;
#pragma unroll
            for (int b = 0; b < 2 * RH; ++b) tf.t[b] = r.t[b];
            put(s, r);
            gload(s + 1, r);            // clamped to the last step at the end (harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            eat(s, tf);
            lds_barrier();
            emit(s);
        }
    } else {
        Regs rA, rB;
        gload(0, rA);
        for (int s = 0; s < nst; s += 2) {
            gload(s + 1, rB);
            __builtin_amdgcn_sched_barrier(0);
            stage(s, rA);
            lds_barrier();
            emit(s);
            gload(s + 2, rA);
            __builtin_amdgcn_sched_barrier(0);
            stage(s + 1, rB);           // the padding step of an odd count stages zeros
            lds_barrier();
            if (s + 1 < nst) emit(s + 1);
        }
    }
    if (!active) return;
    // the wave's accumulators ARE the row group's partial for its columns: lane (n, g) holds G[rank tt*16 + 4g + jj][col ct*16 + n]
    float* out = Gpart + (long long)rg * RW * N;
#pragma unroll
    for (int tt = 0; tt < RH; ++tt)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const int ocol = c0 + ct * 16 + n;
            if (ocol < N) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) out[(long long)(tt * 16 + g * 4 + jj) * N + ocol] = acc[tt][ct][jj];
            }
        }
}

// ------------------------------------------------------------------------------------------
// merge: Wm[o][i] = W[o][i] + scaling * sum_r A_c[i][r] * B_c[r][o]     (fp32, one-off)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_merge(const float* __restrict__ W, const float* __restrict__ A,
                                               const float* __restrict__ B, float* __restrict__ Wm, int in_f,
                                               int out_f, int rank, long long a_si, long long a_sr,
                                               long long b_sr, long long b_so, float scaling) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)in_f * out_f) return;
    const int o = (int)(idx / in_f), i = (int)(idx % in_f);
    float s = 0.f;
    for (int r = 0; r < rank; ++r) s += A[i * a_si + r * a_sr] * B[r * b_sr + o * b_so];
    Wm[idx] = W[idx] + scaling * s;
}

#include "lora_f32_kernels.inc"

// ==========================================================================================
// host side: C-ABI
// ==========================================================================================
#include "fused_linear.inc"

namespace {

thread_local char g_err[512] = "";

// profiling aid (sam3_lora_debug_set_stages): which stages fwd/bwd launch.  Default: all.
unsigned g_stages = 0xffffffffu;
inline bool stage_on(unsigned bit) { return (g_stages & bit) != 0; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

inline long long round_up(long long a, long long b) { return (a + b - 1) / b * b; }
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int rpad(int rank) { return rank <= 16 ? 16 : 32; }
inline size_t esize(int dtype) { return dtype == SAM3_LORA_F32 ? 4 : 2; }

struct T3Plan {
    int nchunks, NR, rows_per_wg, br;
};
bool env_flag(const char* name);
long long env_int(const char* name, long long dflt);

// Geometry of ONE rank group (<= 32 rank indices).  bf16 activations and r <= 16: the hi + lo form (see the file header) --
// operand images, t and gt are laid out as for rank 32 (RP = 32, two 16-wide tiles = hi | lo) while the gradients keep
// one 16-wide rank tile (RG = 16).
struct Geo {
    int RP;     // width of the operand images and of t / gt in elements (16 or 32)
    int RT;     // RP / 16: rank tiles the kernels load
    int RG;     // width of the gradient partials' rank tile (16 or 32)
    bool hl;
};
inline Geo geo_of(int rank, int dtype) {
    Geo g;
    // hi + lo for every rank group (<= 32 rank indices); SAM3_LORA_HL_MAX_RANK=16 restores round 3's single-rounded 17..32
    g.hl = dtype != SAM3_LORA_F32 && rank <= (int)env_int("SAM3_LORA_HL_MAX_RANK", 32) && !env_flag("SAM3_LORA_SINGLE_ROUND");
    g.RG = rpad(rank);
    g.RP = g.hl ? 2 * g.RG : g.RG;
    g.RT = g.RP / 16;
    return g;
}

// k_t3c (cooperative tiles) serves the bf16 hi + lo kernels at r <= 16 unless the validation gather path or the wave-private form is asked for
bool t3_coop_enabled() { return env_int("SAM3_LORA_T3_COOP", 1) != 0 && !env_flag("SAM3_LORA_T3_GATHER"); }
T3Plan plan_t3(long long Mp, int N, int RT) {
    // workgroup = 128 columns x a row group (4 waves x a quarter each, 32-row steps).  ~190 VGPRs allow
    // 2 workgroups per CU: keep all of them co-resident (<= 512) so there is no tail round, and keep the
    // number of row groups (= number of fp32 partials to reduce) small.
    T3Plan p;
    p.br = 32;
    p.nchunks = (N + 127) / 128;
    const long long units = (Mp + 127) / 128;          // 128-row units (4 waves x 32 rows)
    long long nrg = env_int("SAM3_LORA_T3_WGS", RT == 2 ? 256 : 512) / p.nchunks;   // r > 16: ~230 VGPRs, one workgroup per CU
    if (nrg > units) nrg = units;
    if (nrg < 1) nrg = 1;
    const long long upg = (units + nrg - 1) / nrg;     // units per row group
    p.rows_per_wg = (int)(upg * 128);
    p.NR = (int)((units + upg - 1) / upg);
    return p;
}

T3Plan plan_t3e(long long Mp, int N) {
    // k_t3e: workgroup = 256 columns (two chunks) x a row group; `nchunks` counts column PAIRS here
    T3Plan p;
    p.br = 32;
    p.nchunks = (N + 255) / 256;
    const long long units = (Mp + 127) / 128;
    long long nrg = env_int("SAM3_LORA_T3E_WGS", 512) / p.nchunks;
    if (nrg > units) nrg = units;
    if (nrg < 1) nrg = 1;
    const long long upg = (units + nrg - 1) / nrg;
    p.rows_per_wg = (int)(upg * 128);
    p.NR = (int)((units + upg - 1) / upg);
    return p;
}

T3Plan plan_t3w(long long Mp, int N) {
    // k_t3w: workgroup = 8 waves x one 128-column chunk each (`nchunks` counts column GROUPS of 1024) x a range of 32-row steps.
    // One workgroup per CU (96 KB of LDS): ~256 of them, so that the gB partials (one per row group) stay small.
    T3Plan p;
    p.br = 32;
    p.nchunks = ((N + 127) / 128 + 7) / 8;
    const long long steps = Mp / 32;
    long long nrg = env_int("SAM3_LORA_T3W_WGS", 256) / p.nchunks;
    if (nrg > steps) nrg = steps;
    if (nrg < 1) nrg = 1;
    const long long spw = (steps + nrg - 1) / nrg;
    p.rows_per_wg = (int)(spw * 32);
    p.NR = (int)((steps + spw - 1) / spw);
    return p;
}
// version 2 of the bf16 backward (k_t3w): hi + lo kernels, one rank group of <= 16; SAM3_LORA_BWD_V2=0 restores k_t3e
bool bwd_v2_enabled() { return env_int("SAM3_LORA_BWD_V2", 1) != 0; }

int check_common(long long M, int in_f, int out_f, int rank, int layout, int dtype) {
    if (M <= 0 || M >= (1LL << 31)) return fail(SAM3_LORA_EINVAL, "M must be in [1, 2^31) (got %lld)", M);
    if (in_f <= 0 || out_f <= 0 || (in_f % 8) || (out_f % 8))
        return fail(SAM3_LORA_EINVAL, "in_features/out_features must be positive multiples of 8 (got %d, %d)", in_f, out_f);
    if (rank < 1 || rank > SAM3_LORA_MAX_RANK)
        return fail(SAM3_LORA_EINVAL, "rank must be in [1, %d] (got %d)", SAM3_LORA_MAX_RANK, rank);
    if (layout != SAM3_LORA_LAYOUT_ROOT && layout != SAM3_LORA_LAYOUT_PACKAGE)
        return fail(SAM3_LORA_EINVAL, "unknown layout %d", layout);
    if (dtype != SAM3_LORA_BF16 && dtype != SAM3_LORA_F32) return fail(SAM3_LORA_EINVAL, "unknown dtype %d", dtype);
    return 0;
}

int check_act(const void* p, long long ld, int width, int dtype, const char* what) {
    if (!p) return fail(SAM3_LORA_EINVAL, "%s is NULL", what);
    if (ld < width || ld >= (1LL << 31)) return fail(SAM3_LORA_EINVAL, "ld of %s (%lld) < row width (%d), or beyond 2^31 elements", what, ld, width);
    if (((uintptr_t)p & 15) || ((ld * (long long)esize(dtype)) & 15))
        return fail(SAM3_LORA_EINVAL, "%s: base pointer and row pitch must be 16-byte aligned", what);
    return 0;
}

// fp8 image riding on an activation-fused pass (sam3_lora_fwd_act_q8 / _bwd_act_q8): the hi + lo kernels only (one rank
// group of <= 16, bf16), and the pass must carry an activation
int check_q8(const Q8Out* q8, int width, int rank, int dtype, int act) {
    if (!q8->q || !q8->amax_in || !q8->amax_out || !q8->scale_out) return fail(SAM3_LORA_EINVAL, "fp8 output: NULL pointer");
    if (q8->fmt != SAM3_FP8_E4M3 && q8->fmt != SAM3_FP8_E5M2) return fail(SAM3_LORA_EINVAL, "fp8 output: unknown format %d", q8->fmt);
    if (q8->ld < width || (q8->ld & 7) || ((uintptr_t)q8->q & 7)) return fail(SAM3_LORA_EINVAL, "fp8 output: row pitch / base must be 8-byte aligned");
    if (act != SAM3_LORA_ACT_GELU) return fail(SAM3_LORA_EINVAL, "fp8 output rides on the activation-fused passes only");
    if (!geo_of(rank, dtype).hl || rank > 16)
        return fail(SAM3_LORA_ENOTSUP, "fp8 output needs bf16 activations and rank <= 16 (hi + lo kernels)");
    return 0;
}

int launch_ok(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SAM3_LORA_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

// dropout request -> kernel key (thr == 0 means off) and the 1/(1-p) factor folded into the fp32 scales
DropKey make_dropkey(float p, uint64_t seed, uint64_t offset, int width, float* inv_keep) {
    DropKey dk{0u, 0u, width};
    *inv_keep = 1.f;
    if (p > 0.f) {
        long thr = lrintf(p * 65536.f);
        if (thr < 1) thr = 1;
        if (thr > 65536) thr = 65536;
        dk.thr = (unsigned)thr;
        dk.k0 = (unsigned)(seed ^ (seed >> 32)) ^ ((unsigned)offset * 0x9e3779b9u) ^ (unsigned)(offset >> 32);
        *inv_keep = p < 1.f ? 1.f / (1.f - p) : 0.f;
    }
    return dk;
}

// Tuning / validation knobs come from the environment ONCE (first use, or sam3_lora_debug_reload_knobs): no getenv on
// the launch path.  A dozen names, looked up by pointer-stable literals through a tiny table.
struct Knob {
    const char* name;
    bool set;
    long long value;
};
Knob g_knobs[] = {{"SAM3_LORA_T3_WGS", false, 0},       {"SAM3_LORA_T3E_WGS", false, 0},   {"SAM3_LORA_T1_NO_SPLIT", false, 0},
                  {"SAM3_LORA_T1_LDS_PAD", false, 0},   {"SAM3_LORA_T2_TPW", false, 0},    {"SAM3_LORA_T3_GATHER", false, 0},
                  {"SAM3_LORA_TWO_PASS_GY", false, 0},  {"SAM3_LORA_SINGLE_ROUND", false, 0}, {"SAM3_LORA_NO_RIDE", false, 0},
                  {"SAM3_LORA_XCD_ORDER", false, 0},       {"SAM3_LORA_GA_IN_T2", false, 0},  {"SAM3_LORA_FUSED_WGS", false, 0},
                  {"SAM3_LORA_FUSED_HALF", false, 0},  {"SAM3_LORA_FUSED_PROBE", false, 0},
                  {"SAM3_LORA_HL_MAX_RANK", false, 0},
                  {"SAM3_LORA_T1_BK", false, 0},       {"SAM3_LORA_T3_COOP", false, 0},
                  {"SAM3_LORA_BWD_V2", false, 0},       {"SAM3_LORA_T3W_WGS", false, 0}};
std::atomic<bool> g_knobs_loaded{false};
void load_knobs() {
    for (Knob& k : g_knobs) {
        const char* v = getenv(k.name);
        k.set = v && v[0];
        k.value = k.set ? atoll(v) : 0;
        if (k.set && k.value == 0 && v[0] != '0') k.value = 1;      // non-numeric text counts as "on"
    }
    g_knobs_loaded.store(true, std::memory_order_release);
}
const Knob* knob(const char* name) {
    if (!g_knobs_loaded.load(std::memory_order_acquire)) load_knobs();
    for (const Knob& k : g_knobs)
        if (!strcmp(k.name, name)) return &k;
    return nullptr;
}
bool env_flag(const char* name) {
    const Knob* k = knob(name);
    return k && k->set && k->value != 0;
}
long long env_int(const char* name, long long dflt) {
    const Knob* k = knob(name);
    return (k && k->set) ? k->value : dflt;
}

// XCD-aware tile order (xcd_tile_index) of k_t2 / k_t3 / k_t3e: on for NARROW launches only.  Measured on MI355X, same box,
// interleaved rounds (profiles/r03p_xcd_order_sweep.json): at N = 1024 the shared t / gt rows are a fifth of a launch's traffic
// when every XCD fetches them again, and keeping a row group's column chunks on one XCD gains 3-5 % (k_t2 38.2 -> 36.9 us,
// k_t3 24.4 -> 23.2, k_t3e 28.2 -> 27.3); at N = 4736 they are 2-4 % of the traffic and eight XCDs streaming eight separate row
// ranges LOSES more than that (k_t2 132 -> 137 us, and the k_t1 that reads its output next 75 -> 86 us: the rows written last are
// no longer the ones read first).  SAM3_LORA_XCD_ORDER=0 / 1 forces it off / on everywhere.
int xcd_order_for(int N) {
    const long long forced = env_int("SAM3_LORA_XCD_ORDER", -1);
    return forced >= 0 ? (forced != 0) : (N <= 1024);
}

// strides of the canonical views A_c[in, r], B_c[r, out] inside the caller's tensors
struct Strides {
    long long a_si, a_sr, b_sr, b_so;
};
Strides strides_of(int layout, int in_f, int out_f, int rank) {
    Strides s;
    if (layout == SAM3_LORA_LAYOUT_ROOT) {  // A[in, r], B[r, out]
        s.a_si = rank; s.a_sr = 1; s.b_sr = out_f; s.b_so = 1;
    } else {  // A[r, in], B[out, r]
        s.a_si = 1; s.a_sr = in_f; s.b_sr = 1; s.b_so = rank;
    }
    return s;
}

// ---- in-situ kernel timer (sam3_lora_prof_start/stop): HIP events recorded on the caller's stream
// immediately before and after each launch of the selected stages, inside the real call sequence.
struct ProfState {
    unsigned mask = 0;
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;  // 2*cap events
    int* stage = nullptr;
    int* dim = nullptr;
} g_prof;

struct ProfScope {
    int slot = -1;
    hipStream_t st;
    ProfScope(unsigned stage_bit, int dim, hipStream_t s) : st(s) {
        if ((g_prof.mask & stage_bit) && g_prof.n < g_prof.cap) {
            slot = g_prof.n++;
            g_prof.stage[slot] = (int)stage_bit;
            g_prof.dim[slot] = dim;
            hipEventRecord(g_prof.ev[2 * slot], st);
        }
    }
    ~ProfScope() {
        if (slot >= 0) hipEventRecord(g_prof.ev[2 * slot + 1], st);
    }
};

void launch_pack(const PackJob* jobs, int n, hipStream_t st) {
    for (int base = 0; base < n; base += PACK_JOBS_MAX) {       // one launch per 64 jobs (16 layers x 4 images)
        const int cnt = n - base < PACK_JOBS_MAX ? n - base : PACK_JOBS_MAX;
        PackJobs pj;
        long long nmax = 0;
        int dim = 0;
        for (int i = 0; i < PACK_JOBS_MAX; ++i) {
            pj.j[i] = i < cnt ? jobs[base + i] : PackJob{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const long long ne = (long long)pj.j[i].I * pj.j[i].J;
            nmax = ne > nmax ? ne : nmax;
            dim = pj.j[i].J > dim ? pj.j[i].J : dim;
        }
        dim3 grid((unsigned)((nmax + 255) / 256), (unsigned)cnt);
        ProfScope ps(SAM3_LORA_STAGE_PACK, dim, st);
        hipLaunchKernelGGL(k_pack, grid, dim3(256), 0, st, pj);
    }
}
void launch_pack(const PackJob& a, const PackJob& b, hipStream_t st) {
    const PackJob jobs[2] = {a, b};
    launch_pack(jobs, b.dst ? 2 : 1, st);
}

// caller-held operand images (sam3_lora_pack): [ W1 = A_c^T | W2t = B_c^T | W1b = B_c | W2tb = A_c ], bf16
// Ranks above 32 run as consecutive groups of <= 32 rank indices (the kernels hold one or two 16-wide rank tiles):
// y += s (x A_g) B_g per group, and the four gradient products per group -- A_c / B_c slices are addressed through
// the strides of the full tensors, so no copy is made.
inline int group_size(int /*rank*/, int /*dtype*/) { return 32; }
inline int n_groups(int rank, int dtype) { const int gs = group_size(rank, dtype); return (rank + gs - 1) / gs; }
inline int group_rank(int rank, int g, int dtype) {
    const int gs = group_size(rank, dtype);
    return rank - gs * g < gs ? rank - gs * g : gs;
}

// caller-held operand images (sam3_lora_pack) of ONE group: [ W1 = A_c^T | W2t = B_c^T | W1b = B_c | W2tb = A_c ],
// bf16 for SAM3_LORA_BF16 activations, fp32 for SAM3_LORA_F32 (exact path); groups follow each other in the blob
struct PackedLayout {
    size_t w1, w2t, w1b, w2tb, total;
};
PackedLayout packed_layout(int in_f, int out_f, int rank, int dtype) {
    const int RP = geo_of(rank, dtype).RP;
    const size_t e = dtype == SAM3_LORA_F32 ? 4 : 2;
    PackedLayout p;
    size_t off = 0;
    p.w1 = off; off += al256((size_t)RP * round_up(in_f, 128) * e);
    p.w2t = off; off += al256((size_t)out_f * RP * e);
    p.w1b = off; off += al256((size_t)RP * round_up(out_f, 128) * e);
    p.w2tb = off; off += al256((size_t)in_f * RP * e);
    p.total = off;
    return p;
}
size_t packed_total(int in_f, int out_f, int rank, int dtype) {
    size_t n = 0;
    for (int g = 0; g < n_groups(rank, dtype); ++g) n += packed_layout(in_f, out_f, group_rank(rank, g, dtype), dtype).total;
    return n;
}
// bytes of the saved t of one group: bf16 fragment-major [r_pad, M_pad], or fp32 row-major [M_pad, r_pad]
inline size_t saved_t_group_bytes(long long M, int rank, int dtype) {
    return (size_t)geo_of(rank, dtype).RP * (size_t)round_up(M, 64) * (dtype == SAM3_LORA_F32 ? 4 : 2);
}

template <typename XT>
void launch_t1(const void* X, long long ldx, const bf16_t* W1, bf16_t* T, bf16_t* TT, long long M, long long Mp, int K,
               int RT, bool hl, hipStream_t st, DropKey dk = DropKey{0u, 0u, 0}, float* part = nullptr) {
    dim3 grid((unsigned)(Mp / 64));
    const int nk = (K + 127) / 128;
    // small M: split K over workgroups so that ~640 of them exist (fp32 partials in `part`, fixed-order sum after)
    if ((RT == 1 || (hl && RT == 2)) && part && K >= 2048 && grid.x < 512 && !env_flag("SAM3_LORA_T1_NO_SPLIT")) {
        int ks = (int)((640 + grid.x - 1) / grid.x);
        ks = ks > 8 ? 8 : ks;
        const int kc_per = (nk + ks - 1) / ks;
        ks = (nk + kc_per - 1) / kc_per;
        if (ks > 1) {
            {
                ProfScope ps(SAM3_LORA_STAGE_T1, K, st);
                if (hl)
                    hipLaunchKernelGGL((k_t1<XT, 2, 128, true, true>), dim3(grid.x, (unsigned)ks), dim3(256), 0, st, (const XT*)X, ldx,
                                       W1, T, TT, M, Mp, K, dk, part, kc_per);
                else
                    hipLaunchKernelGGL((k_t1<XT, 1, 128, true>), dim3(grid.x, (unsigned)ks), dim3(256), 0, st, (const XT*)X, ldx, W1, T,
                                       TT, M, Mp, K, dk, part, kc_per);
            }
            ProfScope ps(SAM3_LORA_STAGE_GT_REDUCE, K, st);
            const dim3 rg((unsigned)((Mp * 4 + 255) / 256));
            if (hl) hipLaunchKernelGGL(k_gt_reduce<true>, rg, dim3(256), 0, st, (const float*)part, ks, T, TT, Mp);
            else hipLaunchKernelGGL(k_gt_reduce<false>, rg, dim3(256), 0, st, (const float*)part, ks, T, TT, Mp);
            return;
        }
    }
    ProfScope ps(SAM3_LORA_STAGE_T1, K, st);
    // Workgroups hold 40-48 KB of LDS, so up to 4 (r <= 16) fit a CU.  When the grid needs more than one residency
    // round, a nearly empty last round costs a whole round: cap the residency (dynamic-LDS padding) at the value whose
    // last round is fullest.  Measured at M = 82,944 (1296 workgroups): 163 us uncapped, 138-141 us capped.
    const int lds_static = 32768 + 2 * RT * 8192 / 2;       // xs 32 KB + ws 2 x RT x 4 KB
    const int omax = 163840 / lds_static > 4 ? 4 : 163840 / lds_static;
    unsigned pad = 0;
    if ((long long)grid.x > 256LL * omax && K >= 2048) {      // short sweeps (K = 1024) measured better uncapped
        double best = -1.0;
        for (int o = omax; o >= 2; --o) {
            const double r = (double)grid.x / (256.0 * o), full = r / (double)((long long)(r + 0.999999));
            const double score = full * (o >= 4 ? 1.0 : o == 3 ? 0.97 : 0.82);
            if (score > best + 1e-9) {
                best = score;
                pad = o == omax ? 0u : (unsigned)((163840 / o - lds_static - 256) & ~255);
            }
        }
    }
    pad = (unsigned)env_int("SAM3_LORA_T1_LDS_PAD", pad);
    if (RT == 1)
        hipLaunchKernelGGL((k_t1<XT, 1, 128>), grid, dim3(256), pad, st, (const XT*)X, ldx, W1, T, TT, M, Mp, K, dk);
    else if (hl && RT == 4) {   // r <= 32 as hi + lo: four rank tiles [hi 0, hi 1 | lo 0, lo 1]
        // 128-column chunks need 64 KB of LDS here (the W1 tile is as large as the x tile): two workgroups per CU, too few bytes in
        // flight.  64-column chunks (32 KB, five workgroups): 94.9 / 27.7 us at K = 4736 / 1024 against 119.4 / 32.8 us
        // (M = 41,472, same box, profiles/r04t_adapter_rank32_bk*.json; single-rounded rank 32: 83.3 / 23.7).  SAM3_LORA_T1_BK=128
        // selects the wide chunks.
        if (env_int("SAM3_LORA_T1_BK", 64) == 64)
            hipLaunchKernelGGL((k_t1<XT, 4, 64, false, true>), grid, dim3(256), 0, st, (const XT*)X, ldx, W1, T, TT, M, Mp, K, dk);
        else
            hipLaunchKernelGGL((k_t1<XT, 4, 128, false, true>), grid, dim3(256), pad, st, (const XT*)X, ldx, W1, T, TT, M, Mp, K, dk);
    }
    else if (hl)
        hipLaunchKernelGGL((k_t1<XT, 2, 128, false, true>), grid, dim3(256), pad, st, (const XT*)X, ldx, W1, T, TT, M, Mp, K, dk);
    else
        hipLaunchKernelGGL((k_t1<XT, 2, 128>), grid, dim3(256), pad, st, (const XT*)X, ldx, W1, T, TT, M, Mp, K, dk);
}

// `ride`: the reduction blocks of this backward call (see reduce_block); rows == 0 when nothing rides
// GELU' pass that also contracts act(h) with gt (k_t2<GA>): 48 tiles per workgroup keep the gA partials small (54 row blocks at
// M = 41,472: 16 MB at N = 4736 -- the plain pass runs 12, whose 216 blocks would write 65 MB)
constexpr int GA_TILES_PER_WG = 48;
int ga_row_blocks(long long M) { return (int)(((M + 15) / 16 + GA_TILES_PER_WG - 1) / GA_TILES_PER_WG); }
// x == NULL in sam3_lora_bwd_act: "the layer's input is act(pre_act)" -- recomputed inside the GELU' pass (k_t2<GA>).  Needs the
// hi + lo kernels (bf16, one rank group of <= 16), no dropout mask on the branch input; combines with the fp8 image.  SAM3_LORA_GA_IN_T2=1 takes the same
// route when x IS given (A/B of the two forms; the caller then vouches that x == act(pre_act)).
bool ga_in_pass_supported(int rank, int dtype, float drop_p) {
    // one rank group of the hi + lo kernels (r <= 32), with or without the dropout mask (round 6: r <= 16 without a mask before -- the
    // reference's default configuration, r = 32 / dropout 0.1, kept fc2's input for a k_t3 pass over [M, 4736] of its own)
    (void)drop_p;
    return dtype == SAM3_LORA_BF16 && rank <= 32 && geo_of(rank, dtype).hl;
}
bool ga_in_t2_enabled() { return env_flag("SAM3_LORA_GA_IN_T2"); }

template <typename YT>
void launch_t2(void* Y, long long ldy, const bf16_t* T, const bf16_t* W2t, long long M, int N, float scale, int RT, bool hl,
               hipStream_t st, DropKey dk = DropKey{0u, 0u, 0}, int act = 0, void* aux = nullptr, long long ldaux = 0,
               const ReduceRide* ride_in = nullptr, const Q8Out* q8_in = nullptr, const GaEmit* ga_in = nullptr) {
    const long long ntiles = (M + 15) / 16;
    const int nchunks = (N + 127) / 128;
    // 3 tiles per wave measured best on MI355X for both N = 4736 and N = 1024 at M = 41472 (sweep 4..48:
    // 121 / 36 us at 12 vs 125 / 37 us at 32 / 8); fewer per workgroup only when that leaves too few workgroups.
    long long tiles_per_wg = 12;
    while (tiles_per_wg > 4 && nchunks * ((ntiles + tiles_per_wg - 1) / tiles_per_wg) < 1024) tiles_per_wg -= 4;
    tiles_per_wg = env_int("SAM3_LORA_T2_TPW", tiles_per_wg);
    if (ga_in) tiles_per_wg = GA_TILES_PER_WG;
    const GaEmit ga = ga_in ? *ga_in : GaEmit{nullptr, nullptr};
    ReduceRide ride{};
    if (ride_in) {
        ride = *ride_in;
        ride.rows = (int)((2LL * ride.nblk + nchunks - 1) / nchunks);
    }
    dim3 grid((unsigned)nchunks, (unsigned)((ntiles + tiles_per_wg - 1) / tiles_per_wg) + (unsigned)ride.rows);
    ProfScope ps(SAM3_LORA_STAGE_T2, N, st);
    const Q8Out q8 = q8_in ? *q8_in : Q8Out{nullptr, 0, nullptr, nullptr, nullptr, 0};
    const int xcd = xcd_order_for(N);
    if (q8.q && !ga_in) {     // fp8 image beside the bf16 output: activation-fused passes of the hi + lo kernels, no dropout mask (checked by the caller)
#define T2_Q8(AV, FV) hipLaunchKernelGGL((k_t2<bf16_t, 2, false, AV, true, true, FV>), grid, dim3(256), 0, st, (bf16_t*)Y, ldy, T, W2t, M, N, \
                                         scale, (int)tiles_per_wg, dk, (bf16_t*)aux, ldaux, ride, q8, xcd, ga)
        if (act == 1) { if (q8.fmt == SAM3_FP8_E4M3) T2_Q8(1, SAM3_FP8_E4M3); else T2_Q8(1, SAM3_FP8_E5M2); }
        else { if (q8.fmt == SAM3_FP8_E4M3) T2_Q8(2, SAM3_FP8_E4M3); else T2_Q8(2, SAM3_FP8_E5M2); }
#undef T2_Q8
        return;
    }
    if (ga_in) {    // checked by the caller: bf16, hi + lo, GELU' pass; the fp8 image only at r <= 16 without a mask
#define T2_GA(RTV, DV, QV, FV) hipLaunchKernelGGL((k_t2<bf16_t, RTV, DV, 2, true, QV, FV, true>), grid, dim3(256), 0, st, (bf16_t*)Y, ldy, T, W2t, M, N, \
                                                  scale, (int)tiles_per_wg, dk, (bf16_t*)aux, ldaux, ride, q8, xcd, ga)
        if (RT == 4) { if (dk.thr) T2_GA(4, true, false, 0); else T2_GA(4, false, false, 0); }
        else if (dk.thr) T2_GA(2, true, false, 0);
        else if (!q8.q) T2_GA(2, false, false, 0);
        else if (q8.fmt == SAM3_FP8_E4M3) T2_GA(2, false, true, SAM3_FP8_E4M3);
        else T2_GA(2, false, true, SAM3_FP8_E5M2);
#undef T2_GA
        return;
    }
#define T2_LAUNCH(RTV, DV, AV, HV) \
    hipLaunchKernelGGL((k_t2<YT, RTV, DV, AV, HV>), grid, dim3(256), 0, st, (YT*)Y, ldy, T, W2t, M, N, scale, (int)tiles_per_wg, dk, \
                       (YT*)aux, ldaux, ride, q8, xcd, ga)
#define T2_RT(RTV, HV)                                                                     \
    do {                                                                               \
        if (act == 1) T2_LAUNCH(RTV, false, 1, HV);            /* forward: no mask on y */   \
        else if (act == 2) { if (dk.thr) T2_LAUNCH(RTV, true, 2, HV); else T2_LAUNCH(RTV, false, 2, HV); } \
        else { if (dk.thr) T2_LAUNCH(RTV, true, 0, HV); else T2_LAUNCH(RTV, false, 0, HV); }    \
    } while (0)
    if (RT == 1) T2_RT(1, false); else if (hl && RT == 4) T2_RT(4, true); else if (hl) T2_RT(2, true); else T2_RT(2, false);
#undef T2_RT
#undef T2_LAUNCH
}

template <typename XT>
void launch_t3(const void* X, long long ldx, const bf16_t* TT, float* part, long long M, long long Mp, int N,
               const T3Plan& p, int RT, bool hl, unsigned stage_bit, hipStream_t st, DropKey dk = DropKey{0u, 0u, 0}) {
    dim3 grid((unsigned)p.nchunks, (unsigned)p.NR);
    const bool gather = env_flag("SAM3_LORA_T3_GATHER");
    const int xcd = xcd_order_for(N);
    ProfScope ps(stage_bit, N, st);
#define T3_LAUNCH(RTV, GV, HV) \
    do { if (dk.thr) hipLaunchKernelGGL((k_t3<XT, RTV, GV, true, HV>), grid, dim3(256), 0, st, (const XT*)X, ldx, TT, part, M, Mp, N, p.rows_per_wg, dk, xcd); \
         else hipLaunchKernelGGL((k_t3<XT, RTV, GV, false, HV>), grid, dim3(256), 0, st, (const XT*)X, ldx, TT, part, M, Mp, N, p.rows_per_wg, dk, xcd); } while (0)
    if (RT == 1) {
        if (gather) T3_LAUNCH(1, true, false); else T3_LAUNCH(1, false, false);
    } else if (hl && RT == 4) {
        if (gather) T3_LAUNCH(4, true, true); else T3_LAUNCH(4, false, true);
    } else if (hl) {
        if (gather) T3_LAUNCH(2, true, true);
        else if (sizeof(XT) == 2 && t3_coop_enabled()) {    // the cooperative-tile form (k_t3c)
            if (dk.thr) hipLaunchKernelGGL((k_t3c<true, 2>), grid, dim3(256), 0, st, (const bf16_t*)X, ldx, TT, part, M, Mp, N, p.rows_per_wg, dk, xcd);
            else hipLaunchKernelGGL((k_t3c<false, 2>), grid, dim3(256), 0, st, (const bf16_t*)X, ldx, TT, part, M, Mp, N, p.rows_per_wg, dk, xcd);
        } else T3_LAUNCH(2, false, true);
    } else {
        if (gather) T3_LAUNCH(2, true, false); else T3_LAUNCH(2, false, false);
    }
#undef T3_LAUNCH
}

// gB partials AND gt partials from one pass over gy (r <= 16); then the fixed-order chunk sum -> GT / GTT images
template <typename XT>
void launch_t3_emit(const void* X, long long ldx, const bf16_t* TT, float* part, long long M, long long Mp, int N,
                    const T3Plan& p, bool hl, const bf16_t* W1b, float* GTP, bf16_t* GT, bf16_t* GTT, hipStream_t st) {
    dim3 grid((unsigned)p.nchunks, (unsigned)p.NR);
    const int xcd = xcd_order_for(N);
    {
        ProfScope ps(SAM3_LORA_STAGE_T3_GB, N, st);
        if (hl) hipLaunchKernelGGL((k_t3e<XT, true>), grid, dim3(256), 0, st, (const XT*)X, ldx, TT, part, M, Mp, N, p.rows_per_wg, W1b, GTP, xcd);
        else hipLaunchKernelGGL((k_t3e<XT, false>), grid, dim3(256), 0, st, (const XT*)X, ldx, TT, part, M, Mp, N, p.rows_per_wg, W1b, GTP, xcd);
    }
    ProfScope ps(SAM3_LORA_STAGE_GT_REDUCE, N, st);
    const dim3 rg((unsigned)((Mp * 4 + 255) / 256));
    if (hl) hipLaunchKernelGGL(k_gt_reduce<true>, rg, dim3(256), 0, st, (const float*)GTP, p.nchunks, GT, GTT, Mp);
    else hipLaunchKernelGGL(k_gt_reduce<false>, rg, dim3(256), 0, st, (const float*)GTP, p.nchunks, GT, GTT, Mp);
}

// version 2: the pass over gy (k_t3w).  `final_images`: N <= 1024 and the caller wants the bf16 images of gt (GT, GTT) -- the sum is
// complete inside the kernel; otherwise p.nchunks fp32 partials go to GTP.
template <typename XT>
void launch_t3w(const void* X, long long ldx, const bf16_t* TT, float* part, long long M, long long Mp, int N, const T3Plan& p,
                const bf16_t* W1b, float* GTP, bf16_t* GT, bf16_t* GTT, bool final_images, hipStream_t st, int RH = 1) {
    dim3 grid((unsigned)p.nchunks, (unsigned)p.NR);
    const int xcd = p.nchunks > 1 ? 1 : 0;      // the column groups of a row range share its t^T fragments and write neighbouring gt partial rows
    ProfScope ps(SAM3_LORA_STAGE_T3W, N, st);
#define T3W_LAUNCH(FV, RV) hipLaunchKernelGGL((k_t3w<XT, FV, RV>), grid, dim3(512), 0, st, (const XT*)X, ldx, TT, part, M, Mp, N, p.rows_per_wg / 32, W1b, GTP, GT, GTT, xcd)
    if (RH == 2) { if (final_images) T3W_LAUNCH(true, 2); else T3W_LAUNCH(false, 2); }
    else if (final_images) T3W_LAUNCH(true, 1);
    else T3W_LAUNCH(false, 1);
#undef T3W_LAUNCH
}

// ---- exact-fp32 launchers (lora_f32_kernels.inc) ----------------------------------------------------
void launch32_t1(const void* X, long long ldx, const float* W1, float* T, long long M, long long Mp, int K, int RT,
                 hipStream_t st, DropKey dk = DropKey{0u, 0u, 0}) {
    dim3 grid((unsigned)(Mp / 64));
    ProfScope ps(SAM3_LORA_STAGE_T1, K, st);
#define L(RTV, DV) hipLaunchKernelGGL((k32_t1<RTV, DV>), grid, dim3(256), 0, st, (const float*)X, ldx, W1, T, M, Mp, K, dk)
    if (RT == 1) { if (dk.thr) L(1, true); else L(1, false); }
    else { if (dk.thr) L(2, true); else L(2, false); }
#undef L
}
void launch32_t2(void* Y, long long ldy, const float* T, const float* W2t, long long M, int N, float scale, int RT,
                 hipStream_t st, DropKey dk = DropKey{0u, 0u, 0}, int act = 0, void* aux = nullptr, long long ldaux = 0) {
    const long long ntiles = (M + 15) / 16;
    const int nchunks = (N + 127) / 128;
    long long tiles_per_wg = 16;
    while (tiles_per_wg > 4 && nchunks * ((ntiles + tiles_per_wg - 1) / tiles_per_wg) < 1024) tiles_per_wg -= 4;
    dim3 grid((unsigned)nchunks, (unsigned)((ntiles + tiles_per_wg - 1) / tiles_per_wg));
    ProfScope ps(SAM3_LORA_STAGE_T2, N, st);
#define L(RTV, DV, AV) \
    hipLaunchKernelGGL((k32_t2<RTV, DV, AV>), grid, dim3(256), 0, st, (float*)Y, ldy, T, W2t, M, N, scale, (int)tiles_per_wg, dk, \
                       (float*)aux, ldaux)
#define LR(RTV)                                                                  \
    do {                                                                         \
        if (act == 1) L(RTV, false, 1);                                          \
        else if (act == 2) { if (dk.thr) L(RTV, true, 2); else L(RTV, false, 2); } \
        else { if (dk.thr) L(RTV, true, 0); else L(RTV, false, 0); }             \
    } while (0)
    if (RT == 1) LR(1); else LR(2);
#undef LR
#undef L
}
void launch32_t3(const void* X, long long ldx, const float* T, float* part, long long M, long long Mp, int N,
                 const T3Plan& p, int RT, unsigned stage_bit, hipStream_t st, DropKey dk = DropKey{0u, 0u, 0}) {
    dim3 grid((unsigned)p.nchunks, (unsigned)p.NR);
    ProfScope ps(stage_bit, N, st);
#define L(RTV, DV) hipLaunchKernelGGL((k32_t3<RTV, DV>), grid, dim3(256), 0, st, (const float*)X, ldx, T, part, M, Mp, N, p.rows_per_wg, dk)
    if (RT == 1) { if (dk.thr) L(1, true); else L(1, false); }
    else { if (dk.thr) L(2, true); else L(2, false); }
#undef L
}

// ---- workspace plans of ONE rank group ---------------------------------------------------------------
struct FwdWs {
    size_t w1, w2t, t, tt, t1p, total;
};
// fp32 split-K partials of k_t1 (bf16 path, r <= 16 and fewer than 512 row tiles): up to 8 splits x Mp x 16
inline size_t t1_part_bytes(long long Mp, int RG, int dtype) {
    return (dtype != SAM3_LORA_F32 && RG == 16 && Mp / 64 < 512) ? (size_t)8 * Mp * 64 : 0;
}
FwdWs fwd_ws(long long M, int in_f, int out_f, int rank, int dtype) {
    const Geo gq = geo_of(rank, dtype);
    const int RP = gq.RP;
    const long long Mp = round_up(M, 64);
    const size_t e = dtype == SAM3_LORA_F32 ? 4 : 2;
    FwdWs w;
    size_t off = 0;
    w.w1 = off; off += al256((size_t)RP * round_up(in_f, 128) * e);
    w.w2t = off; off += al256((size_t)out_f * RP * e);
    w.t = off; off += al256((size_t)Mp * RP * e);
    w.tt = off; off += al256((size_t)RP * Mp * e);
    w.t1p = off; off += al256(t1_part_bytes(Mp, gq.RG, dtype));
    w.total = off;
    return w;
}

struct BwdWs {
    size_t w1b, w2tb, w1a, gt, gtt, t, tt, pb, pa, gtp, total;
    T3Plan pB, pA, pE, pW;  // pE: the one-pass (k_t3e) plan over gy, bf16 path with r <= 16; pW: version 2's (k_t3w)
};
BwdWs bwd_ws(long long M, int in_f, int out_f, int rank, int dtype) {
    const Geo gq = geo_of(rank, dtype);
    const int RP = gq.RP, RG = gq.RG;
    const long long Mp = round_up(M, 64);
    const size_t e = dtype == SAM3_LORA_F32 ? 4 : 2;
    BwdWs w;
    w.pB = plan_t3(Mp, out_f, RG / 16);
    w.pA = plan_t3(Mp, in_f, RG / 16);
    w.pE = plan_t3e(Mp, out_f);
    w.pW = plan_t3w(Mp, out_f);
    size_t off = 0;
    w.w1b = off; off += al256((size_t)RP * round_up(out_f, 128) * e);
    w.w2tb = off; off += al256((size_t)in_f * RP * e);
    w.w1a = off; off += al256((size_t)RP * round_up(in_f, 128) * e);
    w.gt = off; off += al256((size_t)Mp * RP * e);
    w.gtt = off; off += al256((size_t)RP * Mp * e);
    w.t = off; off += al256((size_t)Mp * RP * e);
    w.tt = off; off += al256((size_t)RP * Mp * e);
    {
        int nrb = w.pB.NR > w.pE.NR ? w.pB.NR : w.pE.NR;
        if (dtype != SAM3_LORA_F32 && (RG == 16 || RG == 32) && w.pW.NR > nrb) nrb = w.pW.NR;
        w.pb = off; off += al256((size_t)nrb * RG * out_f * 4);
    }
    {
        const int nra = ga_row_blocks(M);       // k_t2<GA> writes one gA partial per row block of ITS grid
        w.pa = off; off += al256((size_t)(w.pA.NR > nra ? w.pA.NR : nra) * RG * in_f * 4);
    }
    {   // gt partials of k_t3e (bf16, r <= 16); the same region serves k_t1's split-K partials at small M
        const size_t a = dtype == SAM3_LORA_F32 ? 0 : RG == 16 ? (size_t)w.pE.nchunks * Mp * 16 * 4        // k_t3e's (19 at N = 4736); k_t3w's fit inside
                         : RG == 32 ? (size_t)w.pW.nchunks * Mp * 32 * 4 : 0;                                 // k_t3w<RH = 2>'s gt partials
        const size_t b = t1_part_bytes(Mp, RG, dtype);
        w.gtp = off; off += al256(a > b ? a : b);
    }
    w.total = off;
    return w;
}

}  // namespace

extern "C" {

int sam3_lora_abi_version(void) { return SAM3_LORA_ABI_VERSION; }

void sam3_lora_debug_reload_knobs(void) { load_knobs(); }

unsigned sam3_lora_debug_set_stages(unsigned mask) {
    const unsigned old = g_stages;
    g_stages = mask;
    return old;
}

int sam3_lora_prof_start(unsigned stage_mask, int capacity) {
    g_err[0] = 0;
    if (g_prof.ev) return fail(SAM3_LORA_EINVAL, "profiler already started");
    if (capacity <= 0 || capacity > (1 << 20)) return fail(SAM3_LORA_EINVAL, "bad capacity %d", capacity);
    g_prof.ev = new hipEvent_t[2 * (size_t)capacity];
    g_prof.stage = new int[capacity];
    g_prof.dim = new int[capacity];
    for (int i = 0; i < 2 * capacity; ++i)
        if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return fail(SAM3_LORA_ELAUNCH, "hipEventCreate failed");
    g_prof.cap = capacity;
    g_prof.n = 0;
    g_prof.mask = stage_mask;
    return 0;
}

int sam3_lora_prof_stop(float* us_out, int* stage_out, int* dim_out, int capacity) {
    g_err[0] = 0;
    if (!g_prof.ev) return fail(SAM3_LORA_EINVAL, "profiler not started");
    g_prof.mask = 0;
    const int n = g_prof.n < capacity ? g_prof.n : capacity;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        hipEventSynchronize(g_prof.ev[2 * i + 1]);
        hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        us_out[i] = ms * 1e3f;
        stage_out[i] = g_prof.stage[i];
        dim_out[i] = g_prof.dim[i];
    }
    for (int i = 0; i < 2 * g_prof.cap; ++i) hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev; delete[] g_prof.stage; delete[] g_prof.dim;
    g_prof.ev = nullptr; g_prof.stage = nullptr; g_prof.dim = nullptr;
    g_prof.cap = 0; g_prof.n = 0;
    return n;
}

const char* sam3_lora_last_error(void) { return g_err; }

size_t sam3_lora_saved_t_bytes(int64_t M, int rank, int dtype) {
    if (M <= 0 || rank < 1 || rank > SAM3_LORA_MAX_RANK) return 0;
    size_t n = 0;
    for (int g = 0; g < n_groups(rank, dtype); ++g) n += saved_t_group_bytes(M, group_rank(rank, g, dtype), dtype);
    return n;
}

size_t sam3_lora_fwd_workspace_bytes(int64_t M, int in_features, int out_features, int rank, int dtype) {
    if (check_common(M, in_features, out_features, rank, 0, dtype)) return 0;
    return fwd_ws(M, in_features, out_features, group_rank(rank, 0, dtype), dtype).total;
}

size_t sam3_lora_bwd_workspace_bytes(int64_t M, int in_features, int out_features, int rank, int dtype) {
    if (check_common(M, in_features, out_features, rank, 0, dtype)) return 0;
    return bwd_ws(M, in_features, out_features, group_rank(rank, 0, dtype), dtype).total;
}

size_t sam3_lora_packed_bytes(int in_features, int out_features, int rank, int dtype) {
    if (check_common(1, in_features, out_features, rank, 0, dtype)) return 0;
    return packed_total(in_features, out_features, rank, dtype);
}

// the pack jobs of one adapter (all rank groups) appended to `jobs`; returns the number appended
static int pack_jobs_of(const void* A, const void* B, void* packed, int in_features, int out_features, int rank, int layout,
                        int dtype, PackJob* jobs) {
    const Strides s = strides_of(layout, in_features, out_features, rank);
    const int f32 = dtype == SAM3_LORA_F32;
    char* p = (char*)packed;
    int n = 0;
    const int gs = group_size(rank, dtype);
    for (int g = 0; g < n_groups(rank, dtype); ++g) {
        const int rg = group_rank(rank, g, dtype);
        const Geo gq = geo_of(rg, dtype);
        const int RP = gq.RP, hr = gq.hl ? 1 : 0, hc = gq.hl ? 2 : 0;       // hi + lo along the image's rows / columns
        const PackedLayout pl = packed_layout(in_features, out_features, rg, dtype);
        const float* Ag = (const float*)A + (long long)gs * g * s.a_sr;
        const float* Bg = (const float*)B + (long long)gs * g * s.b_sr;
        jobs[n++] = PackJob{Ag, p + pl.w1, RP, in_features, rg, in_features, s.a_sr, s.a_si, 0, f32, hr};
        jobs[n++] = PackJob{Bg, p + pl.w2t, out_features, RP, out_features, rg, s.b_so, s.b_sr, 0, f32, hc};
        jobs[n++] = PackJob{Bg, p + pl.w1b, RP, out_features, rg, out_features, s.b_sr, s.b_so, 0, f32, hr};
        jobs[n++] = PackJob{Ag, p + pl.w2tb, in_features, RP, in_features, rg, s.a_si, s.a_sr, 0, f32, hc};
        p += pl.total;
    }
    return n;
}

int sam3_lora_pack(const void* A, const void* B, void* packed, int in_features, int out_features, int rank, int layout,
                   int dtype, void* stream) {
    g_err[0] = 0;
    int rc;
    if ((rc = check_common(1, in_features, out_features, rank, layout, dtype))) return rc;
    if (!A || !B || !packed) return fail(SAM3_LORA_EINVAL, "NULL pointer");
    if (((uintptr_t)packed & 255)) return fail(SAM3_LORA_EINVAL, "packed operands must be 256-byte aligned");
    PackJob* jobs = new PackJob[4 * (size_t)n_groups(rank, dtype)];
    const int n = pack_jobs_of(A, B, packed, in_features, out_features, rank, layout, dtype, jobs);
    launch_pack(jobs, n, (hipStream_t)stream);
    delete[] jobs;
    return launch_ok("sam3_lora_pack");
}

int sam3_lora_pack_many(int count, const void* const* A, const void* const* B, void* const* packed,
                        const int* in_features, const int* out_features, const int* rank, int layout, int dtype,
                        void* stream) {
    g_err[0] = 0;
    if (count < 0) return fail(SAM3_LORA_EINVAL, "count must be >= 0");
    if (count == 0) return 0;
    if (!A || !B || !packed || !in_features || !out_features || !rank) return fail(SAM3_LORA_EINVAL, "NULL table");
    size_t total = 0;
    for (int i = 0; i < count; ++i) {
        int rc;
        if ((rc = check_common(1, in_features[i], out_features[i], rank[i], layout, dtype))) return rc;
        if (!A[i] || !B[i] || !packed[i]) return fail(SAM3_LORA_EINVAL, "NULL pointer in entry %d", i);
        if (((uintptr_t)packed[i] & 255)) return fail(SAM3_LORA_EINVAL, "packed operands must be 256-byte aligned (entry %d)", i);
        total += 4 * (size_t)n_groups(rank[i], dtype);
    }
    PackJob* jobs = new PackJob[total];
    int n = 0;
    for (int i = 0; i < count; ++i)
        n += pack_jobs_of(A[i], B[i], packed[i], in_features[i], out_features[i], rank[i], layout, dtype, jobs + n);
    launch_pack(jobs, n, (hipStream_t)stream);
    delete[] jobs;
    return launch_ok("sam3_lora_pack_many");
}

// forward of one rank group.  A_g / B_g: the group's slice of the fp32 masters (strides `s` of the full tensors), or the
// group's operand blob when `pre`.
static void fwd_group(const void* x, const void* A_g, const void* B_g, bool pre, void* y_inout, void* tT_out, long long M,
                      int in_features, int out_features, int rank, long long ldx, long long ldy, const Strides& s,
                      float scale, const DropKey& dk, int dtype, char* ws, hipStream_t st, int act, void* act_out,
                      long long ldact, const Q8Out* q8 = nullptr) {
    const Geo gq = geo_of(rank, dtype);
    const int RP = gq.RP, RT = gq.RT, hr = gq.hl ? 1 : 0, hc = gq.hl ? 2 : 0;
    const long long Mp = round_up(M, 64);
    const bool f32 = dtype == SAM3_LORA_F32;
    const FwdWs w = fwd_ws(M, in_features, out_features, rank, dtype);
    const PackedLayout pl = packed_layout(in_features, out_features, rank, dtype);
    void* W1 = pre ? (void*)((char*)A_g + pl.w1) : (void*)(ws + w.w1);
    void* W2t = pre ? (void*)((char*)A_g + pl.w2t) : (void*)(ws + w.w2t);
    // W1[RP][in] = A_c^T ; W2t[out][RP] = B_c^T
    if (!pre && stage_on(SAM3_LORA_STAGE_PACK)) {
        PackJob ja{(const float*)A_g, W1, RP, in_features, rank, in_features, s.a_sr, s.a_si, 0, f32, hr};
        PackJob jb{(const float*)B_g, W2t, out_features, RP, out_features, rank, s.b_so, s.b_sr, 0, f32, hc};
        launch_pack(ja, jb, st);
    }
    if (f32) {
        // exact-fp32 path: t kept in fp32, row-major; the saved tensor for the backward IS that array
        float* T = tT_out ? (float*)tT_out : (float*)(ws + w.t);
        if (stage_on(SAM3_LORA_STAGE_T1)) launch32_t1(x, ldx, (const float*)W1, T, M, Mp, in_features, RT, st, dk);
        if (stage_on(SAM3_LORA_STAGE_T2))
            launch32_t2(y_inout, ldy, T, (const float*)W2t, M, out_features, scale, RT, st, DropKey{0u, 0u, 0}, act ? 1 : 0,
                        act_out, ldact);
        return;
    }
    bf16_t* T = (bf16_t*)(ws + w.t);
    bf16_t* TT = tT_out ? (bf16_t*)tT_out : (bf16_t*)(ws + w.tt);
    if (stage_on(SAM3_LORA_STAGE_T1))
        launch_t1<bf16_t>(x, ldx, (const bf16_t*)W1, T, TT, M, Mp, in_features, RT, gq.hl, st, dk,
                          gq.RG == 16 ? (float*)(ws + w.t1p) : nullptr);
    if (stage_on(SAM3_LORA_STAGE_T2))
        launch_t2<bf16_t>(y_inout, ldy, T, (const bf16_t*)W2t, M, out_features, scale, RT, gq.hl, st, DropKey{0u, 0u, 0},
                          act ? 1 : 0, act_out, ldact, nullptr, act ? q8 : nullptr);
}

static int fwd_impl(const void* x, const void* A, const void* B, void* y_inout, void* tT_out, int64_t M,
                    int in_features, int out_features, int rank, int64_t ldx, int64_t ldy, int layout, float scaling,
                    float drop_p, uint64_t seed, uint64_t offset, int dtype, void* workspace, size_t workspace_bytes,
                    void* stream, int act, void* act_out, int64_t ldact, const Q8Out* q8 = nullptr) {
    if (drop_p < 0.f || drop_p > 1.f) { g_err[0] = 0; return fail(SAM3_LORA_EINVAL, "drop_p must be in [0, 1] (got %g)", drop_p); }
    g_err[0] = 0;
    int rc;
    const bool pre = (layout & SAM3_LORA_PREPACKED) != 0;
    layout &= ~SAM3_LORA_PREPACKED;
    if ((rc = check_common(M, in_features, out_features, rank, layout, dtype))) return rc;
    if (q8 && (rc = check_q8(q8, out_features, rank, dtype, act))) return rc;
    if ((rc = check_act(x, ldx, in_features, dtype, "x"))) return rc;
    if ((rc = check_act(y_inout, ldy, out_features, dtype, "y_inout"))) return rc;
    if (!A || (!B && !pre)) return fail(SAM3_LORA_EINVAL, "A or B is NULL");
    if (pre && ((uintptr_t)A & 255)) return fail(SAM3_LORA_EINVAL, "packed operands must be 256-byte aligned");
    float inv_keep; const DropKey dk = make_dropkey(drop_p, seed, offset, in_features, &inv_keep);
    const size_t need = fwd_ws(M, in_features, out_features, group_rank(rank, 0, dtype), dtype).total;
    if (!workspace || workspace_bytes < need)
        return fail(SAM3_LORA_ENOMEM, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    if (((uintptr_t)workspace & 255)) return fail(SAM3_LORA_EINVAL, "workspace must be 256-byte aligned");
    if (tT_out && ((uintptr_t)tT_out & 15)) return fail(SAM3_LORA_EINVAL, "tT_out must be 16-byte aligned");
    if (act != SAM3_LORA_ACT_NONE && act != SAM3_LORA_ACT_GELU) return fail(SAM3_LORA_EINVAL, "unknown activation %d", act);
    if (act && (rc = check_act(act_out, ldact, out_features, dtype, "act_out"))) return rc;
    if (dtype != SAM3_LORA_BF16 && dtype != SAM3_LORA_F32) return fail(SAM3_LORA_EINVAL, "unknown dtype %d", dtype);

    const Strides s = strides_of(layout, in_features, out_features, rank);
    const int ng = n_groups(rank, dtype), gs = group_size(rank, dtype);
    const char* blob = (const char*)A;
    char* tT = (char*)tT_out;
    for (int g = 0; g < ng; ++g) {
        const int rg = group_rank(rank, g, dtype);
        const void* Ag = pre ? (const void*)blob : (const void*)((const float*)A + (long long)gs * g * s.a_sr);
        const void* Bg = pre ? nullptr : (const void*)((const float*)B + (long long)gs * g * s.b_sr);
        // the activation rides on the LAST group's update: by then y holds the complete sum
        const int act_g = (g == ng - 1) ? act : 0;
        fwd_group(x, Ag, Bg, pre, y_inout, tT, M, in_features, out_features, rg, ldx, ldy, s, scaling * inv_keep, dk, dtype,
                  (char*)workspace, (hipStream_t)stream, act_g, act_out, ldact, q8);
        if (pre) blob += packed_layout(in_features, out_features, rg, dtype).total;
        if (tT) tT += saved_t_group_bytes(M, rg, dtype);
    }
    return launch_ok("sam3_lora_fwd");
}

int sam3_lora_fwd(const void* x, const void* A, const void* B, void* y_inout, void* tT_out, int64_t M,
                  int in_features, int out_features, int rank, int64_t ldx, int64_t ldy, int layout, float scaling,
                  float drop_p, uint64_t seed, uint64_t offset, int dtype, void* workspace, size_t workspace_bytes,
                  void* stream) {
    return fwd_impl(x, A, B, y_inout, tT_out, M, in_features, out_features, rank, ldx, ldy, layout, scaling, drop_p, seed,
                    offset, dtype, workspace, workspace_bytes, stream, SAM3_LORA_ACT_NONE, nullptr, 0);
}

int sam3_lora_fwd_act(const void* x, const void* A, const void* B, void* y_inout, void* tT_out, int64_t M,
                      int in_features, int out_features, int rank, int64_t ldx, int64_t ldy, int layout, float scaling,
                      float drop_p, uint64_t seed, uint64_t offset, int dtype, void* workspace, size_t workspace_bytes,
                      void* stream, int act, void* act_out, int64_t ldact) {
    return fwd_impl(x, A, B, y_inout, tT_out, M, in_features, out_features, rank, ldx, ldy, layout, scaling, drop_p, seed,
                    offset, dtype, workspace, workspace_bytes, stream, act, act_out, ldact);
}

// backward of one rank group (see fwd_group for A_g / B_g); gA_g / gB_g point at the group's slice of the gradients.
// `act2` (GELU' on gx) must only be requested for the last group: gx is complete then.
static void bwd_group(const void* gy, const void* x, const void* tT_saved, const void* A_g, const void* B_g, bool pre,
                      void* gx_inout, float* gA_g, float* gB_g, long long M, int in_features, int out_features, int rank,
                      long long ldgy, long long ldx, long long ldgx, const Strides& s, float scale, const DropKey& dk,
                      int dtype, int accumulate, char* ws, hipStream_t st, int a2, void* hpre, long long ldpre,
                      const Q8Out* q8 = nullptr) {
    const Geo gq = geo_of(rank, dtype);
    const int RP = gq.RP, RT = gq.RT, RG = gq.RG, hr = gq.hl ? 1 : 0, hc = gq.hl ? 2 : 0;
    const bool hl = gq.hl;
    const long long Mp = round_up(M, 64);
    const bool f32 = dtype == SAM3_LORA_F32;
    const BwdWs w = bwd_ws(M, in_features, out_features, rank, dtype);
    const PackedLayout pl = packed_layout(in_features, out_features, rank, dtype);
    void* W1b = pre ? (void*)((char*)A_g + pl.w1b) : (void*)(ws + w.w1b);
    void* W2tb = pre ? (void*)((char*)A_g + pl.w2tb) : (void*)(ws + w.w2tb);
    void* W1a = pre ? (void*)((char*)A_g + pl.w1) : (void*)(ws + w.w1a);
    float* PB = (float*)(ws + w.pb);
    float* PA = (float*)(ws + w.pa);
    // W1b[RP][out] = B_c ; W2tb[in][RP] = A_c ; W1a[RP][in] = A_c^T (only to recompute t)
    if (!pre && stage_on(SAM3_LORA_STAGE_PACK)) {
        PackJob jobs[3] = {{(const float*)B_g, W1b, RP, out_features, rank, out_features, s.b_sr, s.b_so, 0, f32, hr},
                           {(const float*)A_g, W2tb, in_features, RP, in_features, rank, s.a_si, s.a_sr, 0, f32, hc},
                           {(const float*)A_g, W1a, RP, in_features, rank, in_features, s.a_sr, s.a_si, 0, f32, hr}};
        launch_pack(jobs, tT_saved ? 2 : 3, st);
    }
    const bool s1 = stage_on(SAM3_LORA_STAGE_T1), s2 = stage_on(SAM3_LORA_STAGE_T2);
    const bool s3b = stage_on(SAM3_LORA_STAGE_T3_GB), s3a = stage_on(SAM3_LORA_STAGE_T3_GA);
    bool one_pass = false, ga_in_pass = false, v2 = false;
    if (f32) {
        const float* T32 = (const float*)tT_saved;
        if (!T32) {     // no saved t: recompute t = drop(x) . A_c
            float* Ts = (float*)(ws + w.t);
            launch32_t1(x, ldx, (const float*)W1a, Ts, M, Mp, in_features, RT, st, dk);
            T32 = Ts;
        }
        float* GT32 = (float*)(ws + w.gt);
        if (s1) launch32_t1(gy, ldgy, (const float*)W1b, GT32, M, Mp, out_features, RT, st);                       // gt = gy . B_c^T
        if (gB_g && s3b) launch32_t3(gy, ldgy, T32, PB, M, Mp, out_features, w.pB, RT, SAM3_LORA_STAGE_T3_GB, st);   // gB = t^T . gy
        if (gA_g && s3a) launch32_t3(x, ldx, GT32, PA, M, Mp, in_features, w.pA, RT, SAM3_LORA_STAGE_T3_GA, st, dk); // gA^T = gt^T . drop(x)
        if (gx_inout && s2)
            launch32_t2(gx_inout, ldgx, GT32, (const float*)W2tb, M, in_features, scale, RT, st, dk, a2, hpre, ldpre);
    } else {
        bf16_t* GT = (bf16_t*)(ws + w.gt);
        bf16_t* GTT = (bf16_t*)(ws + w.gtt);
        const bf16_t* TT = (const bf16_t*)tT_saved;
        if (!TT) {  // no saved t: recompute t = x . A_c (one more pass over x)
            bf16_t* TTs = (bf16_t*)(ws + w.tt);
            float* t1p = RG == 16 ? (float*)(ws + w.gtp) : nullptr;     // free until k_t3e runs (stream order)
            launch_t1<bf16_t>(x, ldx, (const bf16_t*)W1a, (bf16_t*)(ws + w.t), TTs, M, Mp, in_features, RT, hl, st, dk, t1p);
            TT = TTs;
        }
        // weight gradients wanted: gy is read ONCE.  Version 2 (hi + lo, one rank group of <= 32: k_t3w) emits gt beside the gB partials;
        // r <= 16 single-rounded (or SAM3_LORA_BWD_V2=0): k_t3e
        const bool once = gB_g && s1 && s3b && !env_flag("SAM3_LORA_TWO_PASS_GY");
        v2 = once && hl && (RG == 16 || RG == 32) && bwd_v2_enabled() && M < (1LL << 31) && ldgy < (1LL << 31);
        one_pass = v2 || (once && RG == 16);
        float* GTP = (float*)(ws + w.gtp);
        if (v2) {
            const bool final_images = w.pW.nchunks == 1;
            launch_t3w<bf16_t>(gy, ldgy, TT, PB, M, Mp, out_features, w.pW, (const bf16_t*)W1b, GTP, GT, GTT, final_images, st, RG / 16);
            if (!final_images) {
                ProfScope ps(SAM3_LORA_STAGE_GT_REDUCE, out_features, st);
                if (RG == 32) hipLaunchKernelGGL((k_gt_reduce<true, 2>), dim3((unsigned)((Mp * 8 + 255) / 256)), dim3(256), 0, st, (const float*)GTP, w.pW.nchunks, GT, GTT, Mp);
                else hipLaunchKernelGGL((k_gt_reduce<true, 1>), dim3((unsigned)((Mp * 4 + 255) / 256)), dim3(256), 0, st, (const float*)GTP, w.pW.nchunks, GT, GTT, Mp);
            }
        } else if (one_pass) {
            launch_t3_emit<bf16_t>(gy, ldgy, TT, PB, M, Mp, out_features, w.pE, hl, (const bf16_t*)W1b, GTP, GT, GTT, st);
        } else {
            if (s1) launch_t1<bf16_t>(gy, ldgy, (const bf16_t*)W1b, GT, GTT, M, Mp, out_features, RT, hl, st, DropKey{0u, 0u, 0},
                                      RG == 16 ? GTP : nullptr);                                                         // gt = gy . B_c^T
            if (gB_g && s3b) launch_t3<bf16_t>(gy, ldgy, TT, PB, M, Mp, out_features, w.pB, RT, hl, SAM3_LORA_STAGE_T3_GB, st);   // gB = t^T . gy
        }
        // GELU'-fused backward of the hi + lo kernels: gA from act(h) INSIDE the pass over gx (k_t2<GA>), no second read of x
        ga_in_pass = gA_g && s3a && s2 && gx_inout && a2 == 2 && hpre && hl && (RG == 16 || RG == 32) && (!q8 || (RG == 16 && !dk.thr)) &&
                     (x == nullptr || ga_in_t2_enabled());
        // (x == NULL with the in-pass form switched off by a partial debug stage mask: nothing can read the input -- skip, the
        // header says a partial mask leaves the outputs meaningless)
        if (gA_g && s3a && !ga_in_pass && x)
            launch_t3<bf16_t>(x, ldx, GTT, PA, M, Mp, in_features, w.pA, RT, hl, SAM3_LORA_STAGE_T3_GA, st, dk);      // gA^T = gt^T . x
    }
    // partial layouts: PB[rs][r][out] -> gB_c[r][out] ; PA[rs][r][in] -> gA_c[in][r]
    const bool want_reduce = (gA_g || gB_g) && stage_on(SAM3_LORA_STAGE_REDUCE);
    ReduceRide ride{};
    if (want_reduce) {
        ride.j0 = ReduceJob{PB, gB_g, v2 ? w.pW.NR : one_pass ? w.pE.NR : w.pB.NR, RG, out_features, rank, s.b_sr, s.b_so};
        ride.j1 = ReduceJob{PA, gA_g, ga_in_pass ? ga_row_blocks(M) : w.pA.NR, RG, in_features, rank, s.a_sr, s.a_si};
        ride.scale = scale;
        ride.accumulate = accumulate;
        const long long nb = (long long)rank * out_features, na = (long long)rank * in_features;
        ride.nblk = (int)(((nb > na ? nb : na) + 63) / 64);
    }
    // the reduction rides on the bf16 rank-r update of gx (the last kernel of the call) when there is one
    // (with the gA partials produced BY that kernel the sum cannot ride on it: it follows as its own launch)
    const bool riding = want_reduce && !f32 && gx_inout && s2 && !ga_in_pass && !env_flag("SAM3_LORA_NO_RIDE");
    const GaEmit ga{(const bf16_t*)(ws + w.gtt), PA};
    if (!f32 && gx_inout && s2)
        launch_t2<bf16_t>(gx_inout, ldgx, (bf16_t*)(ws + w.gt), (const bf16_t*)W2tb, M, in_features, scale, RT, hl, st, dk, a2, hpre,
                          ldpre, riding ? &ride : nullptr, a2 ? q8 : nullptr, ga_in_pass ? &ga : nullptr);
    if (want_reduce && !riding) {
        dim3 grid((unsigned)ride.nblk, 2);
        ProfScope ps(SAM3_LORA_STAGE_REDUCE, in_features + out_features, st);
        hipLaunchKernelGGL(k_reduce, grid, dim3(256), 0, st, ride.j0, ride.j1, scale, accumulate);
    }
}

static int bwd_impl(const void* gy, const void* x, const void* tT_saved, const void* A, const void* B, void* gx_inout,
                    float* gA_accum, float* gB_accum, int64_t M, int in_features, int out_features, int rank,
                    int64_t ldgy, int64_t ldx, int64_t ldgx, int layout, float scaling, float drop_p, uint64_t seed,
                    uint64_t offset, int dtype, int accumulate, void* workspace, size_t workspace_bytes, void* stream,
                    int act, const void* pre_act, int64_t ldpre, const Q8Out* q8 = nullptr) {
    if (drop_p < 0.f || drop_p > 1.f) { g_err[0] = 0; return fail(SAM3_LORA_EINVAL, "drop_p must be in [0, 1] (got %g)", drop_p); }
    g_err[0] = 0;
    int rc;
    const bool pre = (layout & SAM3_LORA_PREPACKED) != 0;
    layout &= ~SAM3_LORA_PREPACKED;
    if ((rc = check_common(M, in_features, out_features, rank, layout, dtype))) return rc;
    if (q8 && (rc = check_q8(q8, in_features, rank, dtype, act))) return rc;
    if (q8 && drop_p > 0.f) return fail(SAM3_LORA_ENOTSUP, "fp8 output is not combined with the dropout mask of the input gradient");
    if (q8 && !gx_inout) return fail(SAM3_LORA_EINVAL, "fp8 output needs gx_inout");
    if ((rc = check_act(gy, ldgy, out_features, dtype, "gy"))) return rc;
    if (x) {
        if ((rc = check_act(x, ldx, in_features, dtype, "x"))) return rc;
    } else {        // the layer's input is act(pre_act): recomputed inside the activation-derivative pass
        if (act != SAM3_LORA_ACT_GELU || !pre_act || !gx_inout || !tT_saved)
            return fail(SAM3_LORA_EINVAL, "x may be NULL only in sam3_lora_bwd_act with pre_act, gx_inout and the saved t^T given");
        if (!ga_in_pass_supported(rank, dtype, drop_p))
            return fail(SAM3_LORA_ENOTSUP, "x == NULL (input recomputed from pre_act) needs bf16 and one rank group (rank <= 32) of the hi + lo "
                                           "kernels (sam3_lora_bwd_act_recomputes_input)");
    }
    if (gx_inout && (rc = check_act(gx_inout, ldgx, in_features, dtype, "gx_inout"))) return rc;
    if (!A || (!B && !pre)) return fail(SAM3_LORA_EINVAL, "A or B is NULL");
    if (pre && ((uintptr_t)A & 255)) return fail(SAM3_LORA_EINVAL, "packed operands must be 256-byte aligned");
    float inv_keep; const DropKey dk = make_dropkey(drop_p, seed, offset, in_features, &inv_keep);
    if (tT_saved && ((uintptr_t)tT_saved & 15)) return fail(SAM3_LORA_EINVAL, "tT_saved must be 16-byte aligned");
    if (act != SAM3_LORA_ACT_NONE && act != SAM3_LORA_ACT_GELU) return fail(SAM3_LORA_EINVAL, "unknown activation %d", act);
    if (act && !gx_inout) return fail(SAM3_LORA_EINVAL, "an activation derivative needs gx_inout");
    if (act && (rc = check_act(pre_act, ldpre, in_features, dtype, "pre_act"))) return rc;
    const size_t need = bwd_ws(M, in_features, out_features, group_rank(rank, 0, dtype), dtype).total;
    if (!workspace || workspace_bytes < need)
        return fail(SAM3_LORA_ENOMEM, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    if (((uintptr_t)workspace & 255)) return fail(SAM3_LORA_EINVAL, "workspace must be 256-byte aligned");

    const Strides s = strides_of(layout, in_features, out_features, rank);
    const int ng = n_groups(rank, dtype), gs = group_size(rank, dtype);
    const char* blob = (const char*)A;
    const char* tT = (const char*)tT_saved;
    for (int g = 0; g < ng; ++g) {
        const int rg = group_rank(rank, g, dtype);
        const void* Ag = pre ? (const void*)blob : (const void*)((const float*)A + (long long)gs * g * s.a_sr);
        const void* Bg = pre ? nullptr : (const void*)((const float*)B + (long long)gs * g * s.b_sr);
        float* gAg = gA_accum ? gA_accum + (long long)gs * g * s.a_sr : nullptr;
        float* gBg = gB_accum ? gB_accum + (long long)gs * g * s.b_sr : nullptr;
        const int a2 = (act && g == ng - 1) ? 2 : 0;
        bwd_group(gy, x, tT, Ag, Bg, pre, gx_inout, gAg, gBg, M, in_features, out_features, rg, ldgy, ldx, ldgx, s,
                  scaling * inv_keep, dk, dtype, accumulate, (char*)workspace, (hipStream_t)stream, a2,
                  const_cast<void*>(pre_act), ldpre, q8);
        if (pre) blob += packed_layout(in_features, out_features, rg, dtype).total;
        if (tT) tT += saved_t_group_bytes(M, rg, dtype);
    }
    return launch_ok("sam3_lora_bwd");
}

int sam3_lora_bwd(const void* gy, const void* x, const void* tT_saved, const void* A, const void* B, void* gx_inout,
                  float* gA_accum, float* gB_accum, int64_t M, int in_features, int out_features, int rank,
                  int64_t ldgy, int64_t ldx, int64_t ldgx, int layout, float scaling, float drop_p, uint64_t seed,
                  uint64_t offset, int dtype, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    return bwd_impl(gy, x, tT_saved, A, B, gx_inout, gA_accum, gB_accum, M, in_features, out_features, rank, ldgy, ldx, ldgx,
                    layout, scaling, drop_p, seed, offset, dtype, accumulate, workspace, workspace_bytes, stream,
                    SAM3_LORA_ACT_NONE, nullptr, 0);
}

int sam3_lora_bwd_act_recomputes_input(int rank, int dtype, float drop_p) {
    return ga_in_pass_supported(rank, dtype, drop_p) ? 1 : 0;
}

int sam3_lora_bwd_act(const void* gy, const void* x, const void* tT_saved, const void* A, const void* B, void* gx_inout,
                      float* gA_accum, float* gB_accum, int64_t M, int in_features, int out_features, int rank,
                      int64_t ldgy, int64_t ldx, int64_t ldgx, int layout, float scaling, float drop_p, uint64_t seed,
                      uint64_t offset, int dtype, int accumulate, void* workspace, size_t workspace_bytes, void* stream,
                      int act, const void* pre_act, int64_t ldpre) {
    return bwd_impl(gy, x, tT_saved, A, B, gx_inout, gA_accum, gB_accum, M, in_features, out_features, rank, ldgy, ldx, ldgx,
                    layout, scaling, drop_p, seed, offset, dtype, accumulate, workspace, workspace_bytes, stream, act, pre_act,
                    ldpre);
}

int sam3_lora_fwd_act_q8(const void* x, const void* A, const void* B, void* y_inout, void* tT_out, int64_t M,
                         int in_features, int out_features, int rank, int64_t ldx, int64_t ldy, int layout, float scaling,
                         float drop_p, uint64_t seed, uint64_t offset, int dtype, void* workspace, size_t workspace_bytes,
                         void* stream, int act, void* act_out, int64_t ldact, void* q8_out, int64_t ldq, int fmt,
                         const float* amax_in, float* amax_out, float* scale_out) {
    const Q8Out q8{(unsigned char*)q8_out, (long long)ldq, amax_in, amax_out, scale_out, fmt};
    return fwd_impl(x, A, B, y_inout, tT_out, M, in_features, out_features, rank, ldx, ldy, layout, scaling, drop_p, seed,
                    offset, dtype, workspace, workspace_bytes, stream, act, act_out, ldact, &q8);
}

int sam3_lora_bwd_act_q8(const void* gy, const void* x, const void* tT_saved, const void* A, const void* B, void* gx_inout,
                         float* gA_accum, float* gB_accum, int64_t M, int in_features, int out_features, int rank,
                         int64_t ldgy, int64_t ldx, int64_t ldgx, int layout, float scaling, float drop_p, uint64_t seed,
                         uint64_t offset, int dtype, int accumulate, void* workspace, size_t workspace_bytes, void* stream,
                         int act, const void* pre_act, int64_t ldpre, void* q8_out, int64_t ldq, int fmt,
                         const float* amax_in, float* amax_out, float* scale_out) {
    const Q8Out q8{(unsigned char*)q8_out, (long long)ldq, amax_in, amax_out, scale_out, fmt};
    return bwd_impl(gy, x, tT_saved, A, B, gx_inout, gA_accum, gB_accum, M, in_features, out_features, rank, ldgy, ldx, ldgx,
                    layout, scaling, drop_p, seed, offset, dtype, accumulate, workspace, workspace_bytes, stream, act, pre_act,
                    ldpre, &q8);
}

// ---- the adapter inside the frozen GEMM (fused_linear.inc)
// per-device facts the persistent kernel needs: CU count (grid size) and whether a workgroup may hold all 160 KB of LDS (gfx950).
// Cached per device ordinal.  A launch asks for the device its STREAM belongs to (a process may drive several devices and the
// current device need not be the tensors'); the shape query, which has no stream, asks for the current device.  `cus` is published
// last with release ordering and read with acquire: a thread that sees it non-zero also sees `lds_ok`.
struct FusedDevInfo {
    std::atomic<int> cus{0}, lds_ok{-1};
};
static FusedDevInfo g_fused_dev[64];
static FusedDevInfo* fused_dev_info(void* stream = nullptr, bool have_stream = false) {
    int dev = -1;
    if (have_stream && stream != nullptr) {
        hipDevice_t sd = 0;
        if (hipStreamGetDevice((hipStream_t)stream, &sd) == hipSuccess) dev = (int)sd;
    }
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= 64) dev = 0;
    FusedDevInfo* d = &g_fused_dev[dev];
    if (d->cus.load(std::memory_order_acquire) == 0) {
        int cus = 0, lds = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        const bool known = hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess;
        d->lds_ok.store((!known || lds >= (int)fl::TileGeo<fl::CfgBig>::LDS_BYTES) ? 1 : 0, std::memory_order_relaxed);
        d->cus.store(cus, std::memory_order_release);
    }
    return d;
}
static int fused_cu_count(void* stream) { return fused_dev_info(stream, true)->cus.load(std::memory_order_acquire); }

int sam3_lora_linear_fwd_supported(int in_features, int out_features, int rank, int dtype) {
    if (!(dtype == SAM3_LORA_BF16 && rank >= 1 && rank <= 32 && n_groups(rank, dtype) == 1 && in_features > 0 && out_features > 0 &&
          in_features % 64 == 0 && out_features % 8 == 0))
        return 0;
    // k_fused_linear declares 160 KB of static LDS: the current device must grant that to one workgroup (gfx950 does).  Without a
    // device (the build check) the shape answer stands.
    FusedDevInfo* d = fused_dev_info();
    return d->lds_ok.load(std::memory_order_relaxed) != 0 ? 1 : 0;
}

size_t sam3_lora_linear_fwd_workspace_bytes(int64_t M, int in_features, int out_features, int rank, int dtype) {
    if (check_common(M, in_features, out_features, rank, 0, dtype)) return 0;
    if (!sam3_lora_linear_fwd_supported(in_features, out_features, rank, dtype)) return 0;
    return fwd_ws(M, in_features, out_features, rank, dtype).total + al256((size_t)out_features * 128 * 2);
}

struct LinearMirror {    // sam3_lora_linear_dgrad_act: the epilogue multiplies by act'(pre_act) instead of applying an activation
    const void* pre_act; long long ldpre;
};
struct LinearF8 {        // the fp8 frozen-W form of sam3_lora_linear_fwd (sam3_lora_linear_fwd_q8)
    const void* x_q8; long long ldxq;
    const void* w_q8; long long ldwq;
    const float* scale_x; const float* scale_w;
    Q8Out q8;               // fp8 image of the activation output (q == nullptr: none)
};

static int linear_fwd_impl(const void* x, const void* W, const void* bias, const void* A, const void* B, void* y_out, void* tT_out,
                           int64_t M, int in_features, int out_features, int rank, int64_t ldx, int64_t ldw, int64_t ldy,
                           int layout, float scaling, float drop_p, uint64_t seed, uint64_t offset, int dtype, void* workspace,
                           size_t workspace_bytes, void* stream, int act, void* act_out, int64_t ldact, const LinearF8* f8,
                           const LinearMirror* mirror = nullptr) {
    g_err[0] = 0;
    if (drop_p < 0.f || drop_p > 1.f) return fail(SAM3_LORA_EINVAL, "drop_p must be in [0, 1] (got %g)", drop_p);
    int rc;
    const bool pre = (layout & SAM3_LORA_PREPACKED) != 0;
    layout &= ~SAM3_LORA_PREPACKED;
    if ((rc = check_common(M, in_features, out_features, rank, layout, dtype))) return rc;
    if (!sam3_lora_linear_fwd_supported(in_features, out_features, rank, dtype))
        return fail(SAM3_LORA_ENOTSUP, "fused linear: bf16 activations, rank <= 32, in_features %% 64 == 0 (got in %d, rank %d, dtype %d)",
                    in_features, rank, dtype);
    if ((rc = check_act(x, ldx, in_features, dtype, "x"))) return rc;
    if (!f8 && (rc = check_act(W, ldw, in_features, dtype, "W"))) return rc;
    if (f8) {
        if (in_features % 128) return fail(SAM3_LORA_ENOTSUP, "fused fp8 linear: in_features %% 128 == 0 (got %d)", in_features);
        if (!f8->x_q8 || !f8->w_q8 || !f8->scale_x || !f8->scale_w) return fail(SAM3_LORA_EINVAL, "fused fp8 linear: NULL image or scale");
        if (f8->ldxq < in_features || f8->ldwq < in_features || (f8->ldxq & 15) || (f8->ldwq & 15) || ((uintptr_t)f8->x_q8 & 15) || ((uintptr_t)f8->w_q8 & 15))
            return fail(SAM3_LORA_EINVAL, "fused fp8 linear: image base pointers and row pitches must be 16-byte aligned");
        if (256LL * f8->ldxq >= (1LL << 31) || 256LL * f8->ldwq >= (1LL << 31))
            return fail(SAM3_LORA_ENOTSUP, "fused fp8 linear: row pitches beyond 8 M bytes are not addressable by the tile descriptors");
        if (f8->q8.q) {
            if (act != SAM3_LORA_ACT_GELU) return fail(SAM3_LORA_EINVAL, "fp8 output rides on the activation output only");
            if (!f8->q8.amax_in || !f8->q8.amax_out || !f8->q8.scale_out) return fail(SAM3_LORA_EINVAL, "fp8 output: NULL pointer");
            if (f8->q8.fmt != SAM3_FP8_E4M3 && f8->q8.fmt != SAM3_FP8_E5M2) return fail(SAM3_LORA_EINVAL, "fp8 output: unknown format %d", f8->q8.fmt);
            if (f8->q8.ld < out_features || (f8->q8.ld & 7) || ((uintptr_t)f8->q8.q & 7)) return fail(SAM3_LORA_EINVAL, "fp8 output: row pitch / base must be 8-byte aligned");
        }
    }
    if ((rc = check_act(y_out, ldy, out_features, dtype, "y_out"))) return rc;
    if (bias && ((uintptr_t)bias & 7)) return fail(SAM3_LORA_EINVAL, "bias must be 8-byte aligned");
    if (!A || (!B && !pre)) return fail(SAM3_LORA_EINVAL, "A or B is NULL");
    if (pre && ((uintptr_t)A & 255)) return fail(SAM3_LORA_EINVAL, "packed operands must be 256-byte aligned");
    if (act != SAM3_LORA_ACT_NONE && act != SAM3_LORA_ACT_GELU) return fail(SAM3_LORA_EINVAL, "unknown activation %d", act);
    if (act && !mirror && (rc = check_act(act_out, ldact, out_features, dtype, "act_out"))) return rc;
    if (mirror) {
        if (f8 || drop_p != 0.f) return fail(SAM3_LORA_ENOTSUP, "dgrad: bf16 operands, no dropout mask on the branch");
        if (act != SAM3_LORA_ACT_GELU) return fail(SAM3_LORA_EINVAL, "dgrad: the activation whose derivative is applied must be given");
        if ((rc = check_act(mirror->pre_act, mirror->ldpre, out_features, dtype, "pre_act"))) return rc;
        if (256LL * mirror->ldpre * 2 >= (1LL << 31)) return fail(SAM3_LORA_ENOTSUP, "fused linear: row pitches beyond 4 M elements are not addressable by the tile descriptors");
    }
    if (tT_out && ((uintptr_t)tT_out & 15)) return fail(SAM3_LORA_EINVAL, "tT_out must be 16-byte aligned");
    if (256LL * ldx * 2 >= (1LL << 31) || (!f8 && 256LL * ldw * 2 >= (1LL << 31)) || 256LL * ldy * 2 >= (1LL << 31) || (act && !mirror && 256LL * ldact * 2 >= (1LL << 31)))
        return fail(SAM3_LORA_ENOTSUP, "fused linear: row pitches beyond 4 M elements are not addressable by the tile descriptors");
    const FwdWs w = fwd_ws(M, in_features, out_features, rank, dtype);
    const size_t need = w.total + al256((size_t)out_features * 128 * 2);
    if (!workspace || workspace_bytes < need)
        return fail(SAM3_LORA_ENOMEM, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    if (((uintptr_t)workspace & 255)) return fail(SAM3_LORA_EINVAL, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float inv_keep;
    const DropKey dk = make_dropkey(drop_p, seed, offset, in_features, &inv_keep);
    const Strides s = strides_of(layout, in_features, out_features, rank);
    const Geo gq = geo_of(rank, dtype);
    const int RP = gq.RP, RT = gq.RT, hr = gq.hl ? 1 : 0, hc = gq.hl ? 2 : 0;
    const long long Mp = round_up(M, 64);
    const PackedLayout pl = packed_layout(in_features, out_features, rank, dtype);
    void* W1 = pre ? (void*)((char*)A + pl.w1) : (void*)(ws + w.w1);
    void* W2t = pre ? (void*)((char*)A + pl.w2t) : (void*)(ws + w.w2t);
    if (!pre && stage_on(SAM3_LORA_STAGE_PACK)) {
        PackJob ja{(const float*)A, W1, RP, in_features, rank, in_features, s.a_sr, s.a_si, 0, 0, hr};
        PackJob jb{(const float*)B, W2t, out_features, RP, out_features, rank, s.b_so, s.b_sr, 0, 0, hc};
        launch_pack(ja, jb, st);
    }
    bf16_t* T = (bf16_t*)(ws + w.t);
    bf16_t* TT = tT_out ? (bf16_t*)tT_out : (bf16_t*)(ws + w.tt);
    bf16_t* Wext = (bf16_t*)(ws + w.total);
    if (stage_on(SAM3_LORA_STAGE_T1))
        launch_t1<bf16_t>(x, ldx, (const bf16_t*)W1, T, TT, M, Mp, in_features, RT, gq.hl, st, dk,
                          gq.RG == 16 ? (float*)(ws + w.t1p) : nullptr);
    if (stage_on(SAM3_LORA_STAGE_PACK)) {
        ProfScope ps(SAM3_LORA_STAGE_PACK, out_features, st);
        const int groups = gq.hl ? RP / 8 : 8;
        hipLaunchKernelGGL(fl::k_wext, dim3((unsigned)((out_features * groups + 255) / 256)), dim3(256), 0, st, (const bf16_t*)W2t, Wext,
                           out_features, RP, gq.hl ? 1 : 0, scaling * inv_keep, f8 ? f8->scale_x : nullptr, f8 ? f8->scale_w : nullptr);
    }
    if (stage_on(SAM3_LORA_STAGE_FUSED)) {
        fl::Args fa;
        fa.X = (const bf16_t*)x; fa.ldx = ldx;
        fa.W = (const bf16_t*)W; fa.ldw = ldw;
        fa.T = T; fa.Wext = Wext;
        fa.bias = (const bf16_t*)bias;
        fa.Y = (bf16_t*)y_out; fa.ldy = ldy;
        fa.A = (bf16_t*)act_out; fa.lda = ldact;
        fa.H = mirror ? (const bf16_t*)mirror->pre_act : nullptr; fa.ldh = mirror ? mirror->ldpre : 0;
        fa.M = M; fa.Mp = Mp; fa.N = out_features; fa.K = in_features;
        fa.X8 = nullptr; fa.W8 = nullptr; fa.ldx8 = fa.ldw8 = 0; fa.sx = fa.sw = nullptr;
        fa.q8 = Q8Out{nullptr, 0, nullptr, nullptr, nullptr, 0};
        if (f8) {
            fa.X8 = (const unsigned char*)f8->x_q8; fa.ldx8 = f8->ldxq;
            fa.W8 = (const unsigned char*)f8->w_q8; fa.ldw8 = f8->ldwq;
            fa.sx = f8->scale_x; fa.sw = f8->scale_w;
            fa.q8 = f8->q8;
        }
        const int bm = fl::CfgBig::BM, bn = fl::CfgBig::BN;
        fa.tiles_m = (int)((M + bm - 1) / bm);
        {   // a last column of tiles at most half a tile wide runs as "half tiles" (fl::TileSeq); SAM3_LORA_FUSED_HALF=0: as full ones
            const int rem = out_features % bn;
            fa.half_col = (rem > 0 && rem <= bn / 2 && env_int("SAM3_LORA_FUSED_HALF", 1) != 0) ? 1 : 0;
            fa.ncf = fa.half_col ? out_features / bn : (out_features + bn - 1) / bn;
        }
        const long long ntiles = (long long)fa.tiles_m * (fa.ncf + fa.half_col);
        long long grid = env_int("SAM3_LORA_FUSED_WGS", (long long)fused_cu_count(stream));
        if (grid > ntiles) grid = ntiles;
        if (grid < 1) grid = 1;
        const int trow = RP * 2;
        ProfScope ps(SAM3_LORA_STAGE_FUSED, out_features, st);
#define SAM3_FL_LAUNCH(CFG_, ACT_, TROW_) \
        hipLaunchKernelGGL((fl::k_fused_linear<fl::CFG_, ACT_, TROW_>), dim3((unsigned)grid), dim3(fl::TileGeo<fl::CFG_>::NTHREADS), 0, st, fa)
#define SAM3_FL_CFG(ACT_, TROW_) SAM3_FL_LAUNCH(CfgBig, ACT_, TROW_)
        const int probe = (int)env_int("SAM3_LORA_FUSED_PROBE", 0);
        if (mirror) {
#define SAM3_FL_MIRROR(TROW_) hipLaunchKernelGGL((fl::k_fused_linear<fl::CfgBig, 2, TROW_>), dim3((unsigned)grid), dim3(512), 0, st, fa)
            if (trow == 128) SAM3_FL_MIRROR(128); else if (trow == 64) SAM3_FL_MIRROR(64); else SAM3_FL_MIRROR(32);
#undef SAM3_FL_MIRROR
        } else if (f8) {
#define SAM3_FL_F8(ACT_, TROW_) hipLaunchKernelGGL((fl::k_fused_linear<fl::CfgBig, ACT_, TROW_, 0, true>), dim3((unsigned)grid), dim3(512), 0, st, fa)
            if (act) { if (trow == 128) SAM3_FL_F8(1, 128); else if (trow == 64) SAM3_FL_F8(1, 64); else SAM3_FL_F8(1, 32); }
            else { if (trow == 128) SAM3_FL_F8(0, 128); else if (trow == 64) SAM3_FL_F8(0, 64); else SAM3_FL_F8(0, 32); }
#undef SAM3_FL_F8
        } else if (probe && trow == 64) {      // measurement aid (fl::k_fused_linear's PROBE): fill alone / matrix pipe alone
#define SAM3_FL_PROBE(P_) do { if (act) hipLaunchKernelGGL((fl::k_fused_linear<fl::CfgBig, 1, 64, P_>), dim3((unsigned)grid), dim3(512), 0, st, fa); \
                               else hipLaunchKernelGGL((fl::k_fused_linear<fl::CfgBig, 0, 64, P_>), dim3((unsigned)grid), dim3(512), 0, st, fa); } while (0)
            if (probe == 1) SAM3_FL_PROBE(1); else if (probe == 2) SAM3_FL_PROBE(2); else if (probe == 3) SAM3_FL_PROBE(3);
            else if (probe == 5) SAM3_FL_PROBE(5); else if (probe == 6) SAM3_FL_PROBE(6); else SAM3_FL_PROBE(4);
#undef SAM3_FL_PROBE
        } else if (act) {
            if (trow == 128) SAM3_FL_CFG(1, 128); else if (trow == 64) SAM3_FL_CFG(1, 64); else SAM3_FL_CFG(1, 32);
        } else {
            if (trow == 128) SAM3_FL_CFG(0, 128); else if (trow == 64) SAM3_FL_CFG(0, 64); else SAM3_FL_CFG(0, 32);
        }
#undef SAM3_FL_CFG
#undef SAM3_FL_LAUNCH
    }
    return launch_ok("sam3_lora_linear_fwd");
}

int sam3_lora_linear_fwd(const void* x, const void* W, const void* bias, const void* A, const void* B, void* y_out, void* tT_out,
                         int64_t M, int in_features, int out_features, int rank, int64_t ldx, int64_t ldw, int64_t ldy,
                         int layout, float scaling, float drop_p, uint64_t seed, uint64_t offset, int dtype, void* workspace,
                         size_t workspace_bytes, void* stream, int act, void* act_out, int64_t ldact) {
    return linear_fwd_impl(x, W, bias, A, B, y_out, tT_out, M, in_features, out_features, rank, ldx, ldw, ldy, layout, scaling, drop_p, seed,
                           offset, dtype, workspace, workspace_bytes, stream, act, act_out, ldact, nullptr);
}

int sam3_lora_linear_dgrad_act(const void* gy, const void* Wt, const void* A, const void* B, void* gx_out, int64_t M, int in_features,
                               int out_features, int rank, int64_t ldgy, int64_t ldwt, int64_t ldgx, int layout, float scaling, int dtype,
                               void* workspace, size_t workspace_bytes, void* stream, int act, const void* pre_act, int64_t ldpre) {
    // the transposed adapter: gx = gy Wt^T + s (gy B_c^T) A_c^T is the forward of a layer [out -> in] whose weight is Wt[in, out] and whose
    // adapter has A' = B_c^T, B' = A_c^T -- i.e. the caller's B and A tensors read in the OTHER layout
    g_err[0] = 0;
    if (layout & SAM3_LORA_PREPACKED) return fail(SAM3_LORA_ENOTSUP, "dgrad: the operand blob holds the forward's images; pass the fp32 masters");
    if (layout != SAM3_LORA_LAYOUT_ROOT && layout != SAM3_LORA_LAYOUT_PACKAGE) return fail(SAM3_LORA_EINVAL, "unknown layout %d", layout);
    const LinearMirror mir{pre_act, (long long)ldpre};
    const int other = layout == SAM3_LORA_LAYOUT_ROOT ? SAM3_LORA_LAYOUT_PACKAGE : SAM3_LORA_LAYOUT_ROOT;
    return linear_fwd_impl(gy, Wt, nullptr, B, A, gx_out, nullptr, M, out_features, in_features, rank, ldgy, ldwt, ldgx, other, scaling, 0.f, 0, 0,
                           dtype, workspace, workspace_bytes, stream, act, nullptr, 0, nullptr, &mir);
}

int sam3_lora_linear_fwd_q8(const void* x, const void* x_q8, int64_t ldxq, const float* scale_x, const void* w_q8, int64_t ldwq,
                            const float* scale_w, const void* bias, const void* A, const void* B, void* y_out, void* tT_out, int64_t M,
                            int in_features, int out_features, int rank, int64_t ldx, int64_t ldy, int layout, float scaling, float drop_p,
                            uint64_t seed, uint64_t offset, int dtype, void* workspace, size_t workspace_bytes, void* stream, int act,
                            void* act_out, int64_t ldact, void* q8_out, int64_t ldq, int fmt, const float* amax_in, float* amax_out,
                            float* scale_out) {
    const LinearF8 f8{x_q8, (long long)ldxq, w_q8, (long long)ldwq, scale_x, scale_w,
                      Q8Out{(unsigned char*)q8_out, (long long)ldq, amax_in, amax_out, scale_out, fmt}};
    return linear_fwd_impl(x, nullptr, bias, A, B, y_out, tT_out, M, in_features, out_features, rank, ldx, 0, ldy, layout, scaling, drop_p, seed,
                           offset, dtype, workspace, workspace_bytes, stream, act, act_out, ldact, &f8);
}

int sam3_lora_merge(const float* W, const float* A, const float* B, float* Wm, int in_features, int out_features,
                    int rank, int layout, float scaling, void* stream) {
    g_err[0] = 0;
    int rc;
    if ((rc = check_common(1, 8, 8, rank, layout, SAM3_LORA_F32))) return rc;
    if (in_features <= 0 || out_features <= 0) return fail(SAM3_LORA_EINVAL, "bad shape");
    if (!W || !A || !B || !Wm) return fail(SAM3_LORA_EINVAL, "NULL pointer");
    const Strides s = strides_of(layout, in_features, out_features, rank);
    const long long nel = (long long)in_features * out_features;
    hipLaunchKernelGGL(k_merge, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, A, B, Wm,
                       in_features, out_features, rank, s.a_si, s.a_sr, s.b_sr, s.b_so, scaling);
    return launch_ok("sam3_lora_merge");
}

}  // extern "C"
