// sam3_lora_amd -- GroupNorm (+ ReLU) on channels-last feature maps (gfx950): C-ABI of include/sam3_seg_amd.h.
//
// The pixel decoder of the mask head (sam3/model/maskformer_segmentation.py:205-222) normalises [8, 256, 288, 288] and
// [8, 256, 144, 144] maps between MIOpen convolutions that all run channels-last.  Per (image, group) the statistics run
// over HW x C/G elements (2.65 M at the 288^2 level); ATen gives that one workgroup per (image, group).  Here:
//
//   k_gn_stats     grid (pixel chunks, N): every workgroup streams its pixels' full channel rows (16-byte vectors, a
//                  thread's vector lies inside one group), reduces in LDS in fixed order and writes one
//                  (count, mean, M2) triple per group;
//   k_gn_finalize  one thread per (image, group): Chan's pairwise combination of the chunk triples in chunk order
//                  -> (mean, rstd);
//   k_gn_apply     y = act(x * (rstd gamma) + (beta - mean rstd gamma)), channels-last in and out;
//   k_gn_bwd_stats / k_gn_bwd_finalize / k_gn_bwd_apply   the input gradient for frozen gamma / beta in the same three
//                  steps (a = sum g', b = sum g' xhat per (image, group); the ReLU mask is recomputed from x).
//
// HBM-bound elementwise work: forward reads x twice and writes y once (the second read mostly hits the 256 MB
// Infinity Cache at these sizes), backward reads x and gy twice and writes gx.  No atomics anywhere.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "sam3_seg_amd.h"

typedef unsigned short bf16_t;

namespace {
thread_local char g_err[256] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
constexpr int EINVAL_ = -22, ENOTSUP_ = -95, ENOMEM_ = -12, EIO_ = -5;

__host__ __device__ inline int chunks_of(long long HW) {
    long long c = (HW + 255) / 256;
    return (int)(c < 1 ? 1 : (c > 256 ? 256 : c));
}
}  // namespace

// ---------------------------------------------------------------------------------------------- 16-byte vectors --
template <typename T> struct V16;
template <> struct V16<bf16_t> {
    static constexpr int N = 8;
    typedef uint4 raw;
    static __device__ __forceinline__ void unpack(const raw& r, float* f) {
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ unsigned pk(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 t = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, t);
    }
    static __device__ __forceinline__ raw pack(const float* f) {
        raw r;
        r.x = pk(f[0], f[1]); r.y = pk(f[2], f[3]); r.z = pk(f[4], f[5]); r.w = pk(f[6], f[7]);
        return r;
    }
};
template <> struct V16<float> {
    static constexpr int N = 4;
    typedef float4 raw;
    static __device__ __forceinline__ void unpack(const raw& r, float* f) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
    static __device__ __forceinline__ raw pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
};

// thread geometry shared by every kernel: TPP threads cover one pixel's C channels, PPB = 256 / TPP pixels per sweep
struct Geo {
    int tp, pr, TPP, PPB, c0, g;
    long long p0, p1;
};
template <typename T>
__device__ __forceinline__ Geo geo_of(int C, int G, long long HW, int chunks) {
    Geo q;
    q.TPP = C / V16<T>::N;
    q.PPB = 256 / q.TPP;
    q.tp = threadIdx.x % q.TPP;
    q.pr = threadIdx.x / q.TPP;
    q.c0 = q.tp * V16<T>::N;
    q.g = q.c0 / (C / G);
    const long long per = (HW + chunks - 1) / chunks;
    q.p0 = (long long)blockIdx.x * per;
    q.p1 = q.p0 + per < HW ? q.p0 + per : HW;
    return q;
}

// fixed-order sum, per group, of two per-thread values over the workgroup; result valid in threads g < G
template <typename T>
__device__ __forceinline__ void group_reduce(float& a, float& b, const Geo& q, int C, int G, float (*sm)[2]) {
    sm[threadIdx.x][0] = a;
    sm[threadIdx.x][1] = b;
    __syncthreads();
    if ((int)threadIdx.x < G) {
        const int TPG = q.TPP / G;          // threads of one pixel that belong to one group
        float sa = 0.f, sb = 0.f;
        for (int pr = 0; pr < q.PPB; ++pr)
            for (int j = 0; j < TPG; ++j) {
                const int t = pr * q.TPP + threadIdx.x * TPG + j;
                sa += sm[t][0];
                sb += sm[t][1];
            }
        a = sa;
        b = sb;
    }
}

// ----------------------------------------------------------------------------------------------------- forward --
template <typename T>
__global__ __launch_bounds__(256) void k_gn_stats(const T* __restrict__ x, float* __restrict__ part, long long HW, int C,
                                                  int G, int chunks) {
    __shared__ float sm[256][2];
    const Geo q = geo_of<T>(C, G, HW, chunks);
    const int n = blockIdx.y;
    const T* xb = x + (long long)n * HW * C + q.c0;
    float s = 0.f, ss = 0.f;
    for (long long p = q.p0 + q.pr; p < q.p1; p += q.PPB) {
        float f[V16<T>::N];
        V16<T>::unpack(*reinterpret_cast<const typename V16<T>::raw*>(xb + p * C), f);
#pragma unroll
        for (int i = 0; i < V16<T>::N; ++i) { s += f[i]; ss += f[i] * f[i]; }
    }
    group_reduce<T>(s, ss, q, C, G, sm);
    if ((int)threadIdx.x < G) {
        const float cnt = (float)((q.p1 > q.p0 ? q.p1 - q.p0 : 0) * (C / G));
        const float mean = cnt > 0.f ? s / cnt : 0.f;
        float m2 = ss - s * mean;
        m2 = m2 > 0.f ? m2 : 0.f;
        float* o = part + (((long long)n * chunks + blockIdx.x) * G + threadIdx.x) * 3;
        o[0] = cnt; o[1] = mean; o[2] = m2;
    }
}

__global__ __launch_bounds__(64) void k_gn_finalize(const float* __restrict__ part, float* __restrict__ stats, int NG, int G,
                                                    int chunks, float eps) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= NG) return;
    const int n = i / G, g = i % G;
    float na = 0.f, mean = 0.f, m2 = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const float* p = part + (((long long)n * chunks + c) * G + g) * 3;
        const float nb = p[0];
        if (nb <= 0.f) continue;
        const float d = p[1] - mean, nt = na + nb;
        mean += d * (nb / nt);
        m2 += p[2] + d * d * (na * nb / nt);
        na = nt;
    }
    const float var = na > 0.f ? m2 / na : 0.f;
    stats[2 * i] = mean;
    stats[2 * i + 1] = rsqrtf(var + eps);
}

template <typename T, bool RELU>
__global__ __launch_bounds__(256) void k_gn_apply(const T* __restrict__ x, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ stats,
                                                  T* __restrict__ y, long long HW, int C, int G, int chunks) {
    const Geo q = geo_of<T>(C, G, HW, chunks);
    const int n = blockIdx.y;
    const float mean = stats[2 * (n * G + q.g)], rstd = stats[2 * (n * G + q.g) + 1];
    float sc[V16<T>::N], sh[V16<T>::N];
#pragma unroll
    for (int i = 0; i < V16<T>::N; ++i) {
        sc[i] = rstd * gamma[q.c0 + i];
        sh[i] = beta[q.c0 + i] - mean * sc[i];
    }
    const long long base = (long long)n * HW * C + q.c0;
    for (long long p = q.p0 + q.pr; p < q.p1; p += q.PPB) {
        float f[V16<T>::N];
        V16<T>::unpack(*reinterpret_cast<const typename V16<T>::raw*>(x + base + p * C), f);
#pragma unroll
        for (int i = 0; i < V16<T>::N; ++i) {
            f[i] = f[i] * sc[i] + sh[i];
            if (RELU) f[i] = f[i] > 0.f ? f[i] : 0.f;
        }
        *reinterpret_cast<typename V16<T>::raw*>(y + base + p * C) = V16<T>::pack(f);
    }
}

// ---------------------------------------------------------------------------------------------------- backward --
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void k_gn_bwd_stats(const T* __restrict__ x, const T* __restrict__ gy,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ stats, float* __restrict__ part,
                                                      long long HW, int C, int G, int chunks) {
    __shared__ float sm[256][2];
    const Geo q = geo_of<T>(C, G, HW, chunks);
    const int n = blockIdx.y;
    const float mean = stats[2 * (n * G + q.g)], rstd = stats[2 * (n * G + q.g) + 1];
    float gm[V16<T>::N], sc[V16<T>::N], sh[V16<T>::N];
#pragma unroll
    for (int i = 0; i < V16<T>::N; ++i) {
        gm[i] = gamma[q.c0 + i];
        sc[i] = rstd * gm[i];
        sh[i] = beta[q.c0 + i] - mean * sc[i];
    }
    const long long base = (long long)n * HW * C + q.c0;
    float a = 0.f, b = 0.f;
    for (long long p = q.p0 + q.pr; p < q.p1; p += q.PPB) {
        float f[V16<T>::N], d[V16<T>::N];
        V16<T>::unpack(*reinterpret_cast<const typename V16<T>::raw*>(x + base + p * C), f);
        V16<T>::unpack(*reinterpret_cast<const typename V16<T>::raw*>(gy + base + p * C), d);
#pragma unroll
        for (int i = 0; i < V16<T>::N; ++i) {
            float gp = d[i] * gm[i];
            if (RELU && !(f[i] * sc[i] + sh[i] > 0.f)) gp = 0.f;
            a += gp;
            b += gp * ((f[i] - mean) * rstd);
        }
    }
    group_reduce<T>(a, b, q, C, G, sm);
    if ((int)threadIdx.x < G) {
        float* o = part + (((long long)n * chunks + blockIdx.x) * G + threadIdx.x) * 2;
        o[0] = a; o[1] = b;
    }
}

__global__ __launch_bounds__(64) void k_gn_bwd_finalize(const float* __restrict__ part, float* __restrict__ ab, int NG, int G,
                                                        int chunks, float inv_m) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= NG) return;
    const int n = i / G, g = i % G;
    float a = 0.f, b = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const float* p = part + (((long long)n * chunks + c) * G + g) * 2;
        a += p[0];
        b += p[1];
    }
    ab[2 * i] = a * inv_m;
    ab[2 * i + 1] = b * inv_m;
}

template <typename T, bool RELU>
__global__ __launch_bounds__(256) void k_gn_bwd_apply(const T* __restrict__ x, const T* __restrict__ gy,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ stats, const float* __restrict__ ab,
                                                      T* __restrict__ gx, long long HW, int C, int G, int chunks) {
    const Geo q = geo_of<T>(C, G, HW, chunks);
    const int n = blockIdx.y;
    const float mean = stats[2 * (n * G + q.g)], rstd = stats[2 * (n * G + q.g) + 1];
    const float am = ab[2 * (n * G + q.g)], bm = ab[2 * (n * G + q.g) + 1];
    float gm[V16<T>::N], sc[V16<T>::N], sh[V16<T>::N];
#pragma unroll
    for (int i = 0; i < V16<T>::N; ++i) {
        gm[i] = gamma[q.c0 + i];
        sc[i] = rstd * gm[i];
        sh[i] = beta[q.c0 + i] - mean * sc[i];
    }
    const long long base = (long long)n * HW * C + q.c0;
    for (long long p = q.p0 + q.pr; p < q.p1; p += q.PPB) {
        float f[V16<T>::N], d[V16<T>::N];
        V16<T>::unpack(*reinterpret_cast<const typename V16<T>::raw*>(x + base + p * C), f);
        V16<T>::unpack(*reinterpret_cast<const typename V16<T>::raw*>(gy + base + p * C), d);
#pragma unroll
        for (int i = 0; i < V16<T>::N; ++i) {
            float gp = d[i] * gm[i];
            if (RELU && !(f[i] * sc[i] + sh[i] > 0.f)) gp = 0.f;
            const float xh = (f[i] - mean) * rstd;
            d[i] = rstd * (gp - am - xh * bm);
        }
        *reinterpret_cast<typename V16<T>::raw*>(gx + base + p * C) = V16<T>::pack(d);
    }
}

// -------------------------------------------------------------------------------------------------------- host --
namespace {
int check(int N, long long HW, int C, int G, int dtype) {
    if (N < 0 || HW < 0 || C <= 0 || G <= 0 || C % G) return fail(EINVAL_, "bad sizes N=%d HW=%lld C=%d G=%d", N, HW, C, G);
    if (dtype != 0 && dtype != 1) return fail(EINVAL_, "dtype must be 0 (bf16) or 1 (fp32), got %d", dtype);
    const int vec = dtype == 0 ? 8 : 4;
    const int tpp = C / vec;
    if (C % vec || (C / G) % vec || tpp > 256 || (tpp & (tpp - 1)) || G > 256)
        return fail(ENOTSUP_, "GroupNorm kernels need C/G a multiple of %d and C/%d a power of two <= 256 (C=%d, G=%d)", vec, vec, C, G);
    return 0;
}
size_t ws_bytes(int N, long long HW, int G) {
    return ((size_t)N * chunks_of(HW) * G * 3 + (size_t)N * G * 2) * sizeof(float);
}
int launched(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(EIO_, "%s launch failed: %s", what, hipGetErrorString(e));
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// decoder: box-relative position bias of the image cross-attention (sam3/model/decoder.py:357-407, "boxRPB = log").
// For every query box and every token row / column: the signed log-scaled offsets to the box's two edges go through a
// two-layer MLP per axis (2 -> hidden -> heads, ReLU); the bias of token (y, x) is the sum of its row and column terms.
// The reference builds this from ~75 small operators per decoder layer (K = 2 batched GEMMs, broadcast adds over
// [B, Q, 72, 256], a 26 MB permute + copy); the boxes are detached and the MLPs frozen, so no gradient is needed:
// one workgroup per (image, query) evaluates both MLPs into LDS and writes its [heads, H*W] slab once, head-major.
// Roundings follow the layer dtype where the operator chain rounds: MLP input, hidden activation, MLP output, the sum.
// ------------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) {
    const float f[2] = {v, 0.f};
    return __uint_as_float(V16<bf16_t>::pk(f[0], f[1]) << 16);
}
template <typename T> __device__ __forceinline__ float ldw(const T* p, long long i);
template <> __device__ __forceinline__ float ldw<float>(const float* p, long long i) { return p[i]; }
template <> __device__ __forceinline__ float ldw<bf16_t>(const bf16_t* p, long long i) { return __uint_as_float((unsigned)p[i] << 16); }
template <typename T> __device__ __forceinline__ void stw(T* p, long long i, float v);
template <> __device__ __forceinline__ void stw<float>(float* p, long long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stw<bf16_t>(bf16_t* p, long long i, float v) {
    p[i] = (bf16_t)(V16<bf16_t>::pk(v, 0.f) & 0xffffu);
}

template <typename T> __device__ __forceinline__ void st4(T* p, float a, float b, float c, float d);     // p: 4-element aligned
template <> __device__ __forceinline__ void st4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(V16<bf16_t>::pk(a, b), V16<bf16_t>::pk(c, d));
}

struct RpbMlp {            // one axis: W1 [hidden, 2], b1 [hidden], W2 [heads, hidden], b2 [heads] in the layer dtype
    const void *w1, *b1, *w2, *b2;
};

template <typename T>
__global__ __launch_bounds__(512) void k_rpb_bias(const float* __restrict__ boxes /* [Q, B, 4] cxcywh */, RpbMlp mx, RpbMlp my,
                                                  T* __restrict__ out /* [B, heads, Q + pr, H * W] */, int B, int Q, int H,
                                                  int W, int hidden, int heads, int presence_row, int log_scale) {
    // LDS: term [heads][H + W] | per axis (x then y): first layer [hidden] x (w1[j][0], w1[j][1], b1[j], -), second layer
    //      transposed [hidden][groups] 16-byte words (heads padded to a multiple of 4) | partial sums [nsplit][H + W][4 groups]
    extern __shared__ float4 sm4[];
    const int items = H + W, groups = (heads + 3) / 4, NT = blockDim.x;
    float* term = reinterpret_cast<float*>(sm4);
    float4* l1 = sm4 + (items * heads + 3) / 4;          // [2][hidden]
    float4* l2 = l1 + 2 * hidden;                        // [2][hidden][groups]
    float4* part = l2 + 2 * hidden * groups;             // [nsplit][items][groups]
    const int b = blockIdx.y, qo = blockIdx.x;           // qo: output row (0 = presence row when presence_row)
    const int QO = Q + presence_row;
    const long long HW = (long long)H * W;
    T* ob = out + ((long long)b * heads * QO + qo) * HW;  // + h * QO * HW per head
    if (presence_row && qo == 0) {
        for (int h = 0; h < heads; ++h)
            for (long long p = threadIdx.x; p < HW; p += NT) stw<T>(ob + (long long)h * QO * HW, p, 0.f);
        return;
    }
    for (int e = threadIdx.x; e < 2 * hidden; e += NT) {
        const RpbMlp m = e < hidden ? mx : my;
        const int jj = e < hidden ? e : e - hidden;
        const T *w1 = (const T*)m.w1, *b1 = (const T*)m.b1, *w2 = (const T*)m.w2;
        l1[e] = make_float4(ldw<T>(w1, 2 * jj), ldw<T>(w1, 2 * jj + 1), ldw<T>(b1, jj), 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < groups) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = 4 * k + u < heads ? ldw<T>(w2, (long long)(4 * k + u) * hidden + jj) : 0.f;
                l2[e * groups + k] = make_float4(v[0], v[1], v[2], v[3]);
            }
    }
    __syncthreads();
    const int q = qo - presence_row;
    const float* bx = boxes + ((long long)q * B + b) * 4;
    const float cx = bx[0], cy = bx[1], w = bx[2], hh = bx[3];
    const float x0 = cx - 0.5f * w, y0 = cy - 0.5f * hh, x1 = cx + 0.5f * w, y1 = cy + 0.5f * hh;
    // nsplit threads share one (row | column): each takes every nsplit-th hidden unit, partial sums meet in LDS
    int nsplit = NT / items;
    nsplit = nsplit < 1 ? 1 : (nsplit > 4 ? 4 : nsplit);
    for (int base = 0; base < items; base += NT) {        // (more rows + columns than threads: several sweeps, nsplit = 1)
        const int split = nsplit > 1 ? (int)threadIdx.x / items : 0;
        const int item = nsplit > 1 ? (int)threadIdx.x % items : base + (int)threadIdx.x;
        if (item < items && split < nsplit) {
            const bool row = item < H;
            const int i = row ? item : item - H;
            const float c = row ? (float)i / (float)H : (float)i / (float)W;
            float d0 = c - (row ? y0 : x0), d1 = c - (row ? y1 : x1);
            if (log_scale) {
                d0 *= 8.f; d1 *= 8.f;
                d0 = (d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f)) * log2f(fabsf(d0) + 1.f) / 3.f;
                d1 = (d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f)) * log2f(fabsf(d1) + 1.f) / 3.f;
            }
            d0 = rnd<T>(d0); d1 = rnd<T>(d1);
            float4 acc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4* a1 = l1 + (row ? hidden : 0);
            const float4* a2 = l2 + (row ? hidden : 0) * groups;
            for (int j = split; j < hidden; j += nsplit) {
                const float4 f = a1[j];
                float hid = rnd<T>(f.x * d0 + f.y * d1 + f.z);
                hid = hid > 0.f ? hid : 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < groups) {
                        const float4 wv = a2[j * groups + k];
                        acc[k].x += wv.x * hid; acc[k].y += wv.y * hid; acc[k].z += wv.z * hid; acc[k].w += wv.w * hid;
                    }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < groups) part[(split * items + item) * groups + k] = acc[k];
        }
        if (nsplit > 1) break;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < items * heads; e += NT) {            // fixed-order sum of the partials, bias, rounding
        const int item = e / heads, h = e % heads;
        const T* b2 = (const T*)(item < H ? my.b2 : mx.b2);
        float v = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) v += reinterpret_cast<const float*>(part + (sp * items + item) * groups)[h];
        term[h * items + item] = rnd<T>(v + ldw<T>(b2, h));           // head-major: conflict-free reads below
    }
    __syncthreads();
    // write the slab: a thread owns QUADS of neighbouring columns of one row (8-byte stores in bf16, 16-byte in fp32);
    // its (row, quad) position advances by blockDim quads per step without divisions
    const int HWI = items;
    if (W % 4 == 0 && H % 4 == 0) {                        // (then every 16-byte LDS word below is aligned too)
        const int qpr = W / 4;                              // quads per row
        const int dy = NT / qpr, dxq = NT % qpr;
        for (int h = 0; h < heads; ++h) {
            T* oh = ob + (long long)h * QO * HW;
            const float* ty = term + h * HWI;
            const float* tx = ty + H;
            int y = (int)threadIdx.x / qpr, xq = (int)threadIdx.x % qpr;
            while (y < H) {
                const float r = ty[y];
                const int x = 4 * xq;
                const float4 c = *reinterpret_cast<const float4*>(tx + x);
                st4<T>(oh + (long long)y * W + x, r + c.x, r + c.y, r + c.z, r + c.w);
                y += dy; xq += dxq;
                if (xq >= qpr) { xq -= qpr; ++y; }
            }
        }
    } else {
        for (int h = 0; h < heads; ++h) {
            T* oh = ob + (long long)h * QO * HW;
            for (long long p = threadIdx.x; p < HW; p += NT) {
                const int y = (int)(p / W), x = (int)(p % W);
                stw<T>(oh, p, term[h * HWI + y] + term[h * HWI + H + x]);
            }
        }
    }
}

extern "C" {

const char* sam3_seg_last_error(void) { return g_err; }

int sam3_gn_nhwc_supported(int C, int G, int dtype) { return check(1, 1, C, G, dtype); }

size_t sam3_gn_nhwc_workspace_bytes(int N, int64_t HW, int C, int G) {
    (void)C;
    return N > 0 && HW > 0 && G > 0 ? ws_bytes(N, HW, G) : 0;
}

int sam3_gn_nhwc_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int N, int64_t HW,
                     int C, int G, float eps, int relu, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    g_err[0] = 0;
    if (int rc = check(N, HW, C, G, dtype)) return rc;
    if (N == 0 || HW == 0) return 0;
    if (!x || !gamma || !beta || !y || !stats) return fail(EINVAL_, "null pointer");
    if (((uintptr_t)x | (uintptr_t)y) & 15) return fail(EINVAL_, "x / y must be 16-byte aligned");
    if (!workspace || workspace_bytes < ws_bytes(N, HW, G))
        return fail(ENOMEM_, "workspace too small: %zu < %zu", workspace_bytes, ws_bytes(N, HW, G));
    hipStream_t st = (hipStream_t)stream;
    const int chunks = chunks_of(HW);
    float* part = (float*)workspace;
    dim3 grid(chunks, N);
    if (dtype == 0) hipLaunchKernelGGL(k_gn_stats<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, part, (long long)HW, C, G, chunks);
    else hipLaunchKernelGGL(k_gn_stats<float>, grid, dim3(256), 0, st, (const float*)x, part, (long long)HW, C, G, chunks);
    hipLaunchKernelGGL(k_gn_finalize, dim3((N * G + 63) / 64), dim3(64), 0, st, (const float*)part, stats, N * G, G, chunks, eps);
#define APPLY(T, R) hipLaunchKernelGGL((k_gn_apply<T, R>), grid, dim3(256), 0, st, (const T*)x, gamma, beta, (const float*)stats, (T*)y, (long long)HW, C, G, chunks)
    if (dtype == 0) { if (relu) APPLY(bf16_t, true); else APPLY(bf16_t, false); }
    else { if (relu) APPLY(float, true); else APPLY(float, false); }
#undef APPLY
    return launched("sam3_gn_nhwc_fwd");
}

int sam3_gn_nhwc_bwd(const void* x, const void* gy, const float* gamma, const float* beta, const float* stats, void* gx,
                     int N, int64_t HW, int C, int G, int relu, int dtype, void* workspace, size_t workspace_bytes,
                     void* stream) {
    g_err[0] = 0;
    if (int rc = check(N, HW, C, G, dtype)) return rc;
    if (N == 0 || HW == 0) return 0;
    if (!x || !gy || !gamma || !beta || !stats || !gx) return fail(EINVAL_, "null pointer");
    if (((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gx) & 15) return fail(EINVAL_, "x / gy / gx must be 16-byte aligned");
    if (!workspace || workspace_bytes < ws_bytes(N, HW, G))
        return fail(ENOMEM_, "workspace too small: %zu < %zu", workspace_bytes, ws_bytes(N, HW, G));
    hipStream_t st = (hipStream_t)stream;
    const int chunks = chunks_of(HW);
    float* part = (float*)workspace;
    float* ab = part + (size_t)N * chunks * G * 3;
    const float inv_m = 1.f / ((float)HW * (float)(C / G));
    dim3 grid(chunks, N);
#define BSTATS(T, R) hipLaunchKernelGGL((k_gn_bwd_stats<T, R>), grid, dim3(256), 0, st, (const T*)x, (const T*)gy, gamma, beta, stats, part, (long long)HW, C, G, chunks)
#define BAPPLY(T, R) hipLaunchKernelGGL((k_gn_bwd_apply<T, R>), grid, dim3(256), 0, st, (const T*)x, (const T*)gy, gamma, beta, stats, (const float*)ab, (T*)gx, (long long)HW, C, G, chunks)
    if (dtype == 0) { if (relu) BSTATS(bf16_t, true); else BSTATS(bf16_t, false); }
    else { if (relu) BSTATS(float, true); else BSTATS(float, false); }
    hipLaunchKernelGGL(k_gn_bwd_finalize, dim3((N * G + 63) / 64), dim3(64), 0, st, (const float*)part, ab, N * G, G, chunks, inv_m);
    if (dtype == 0) { if (relu) BAPPLY(bf16_t, true); else BAPPLY(bf16_t, false); }
    else { if (relu) BAPPLY(float, true); else BAPPLY(float, false); }
#undef BSTATS
#undef BAPPLY
    return launched("sam3_gn_nhwc_bwd");
}

int sam3_rpb_bias_fwd(const float* boxes, const void* const* mlp_x, const void* const* mlp_y, void* out, int B, int Q, int H,
                      int W, int hidden, int heads, int presence_row, int log_scale, int dtype, void* stream) {
    g_err[0] = 0;
    if (B < 0 || Q < 0 || H <= 0 || W <= 0 || hidden <= 0 || heads <= 0) return fail(EINVAL_, "bad sizes");
    if (heads > 16) return fail(ENOTSUP_, "at most 16 heads (got %d)", heads);
    if (dtype != 0 && dtype != 1) return fail(EINVAL_, "dtype must be 0 (bf16) or 1 (fp32), got %d", dtype);
    const int items = H + W, groups = (heads + 3) / 4;
    int threads = ((2 * items + 63) / 64) * 64;          // two threads per row / column when that fits a workgroup
    threads = threads < 256 ? 256 : (threads > 512 ? 512 : threads);
    int nsplit = threads / items;
    nsplit = nsplit < 1 ? 1 : (nsplit > 4 ? 4 : nsplit);
    const size_t lds = ((size_t)((items * heads + 3) / 4) + (size_t)2 * hidden * (1 + groups)
                        + (size_t)nsplit * items * groups) * sizeof(float4);
    if (lds > 60 * 1024) return fail(ENOTSUP_, "feature map / hidden width too large for the LDS staging (%zu bytes)", lds);
    if (B == 0 || Q + presence_row == 0) return 0;
    if (!boxes || !mlp_x || !mlp_y || !out) return fail(EINVAL_, "null pointer");
    for (int i = 0; i < 4; ++i)
        if (!mlp_x[i] || !mlp_y[i]) return fail(EINVAL_, "null MLP tensor");
    RpbMlp mx{mlp_x[0], mlp_x[1], mlp_x[2], mlp_x[3]}, my{mlp_y[0], mlp_y[1], mlp_y[2], mlp_y[3]};
    dim3 grid((unsigned)(Q + presence_row), (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(k_rpb_bias<bf16_t>, grid, dim3(threads), lds, st, boxes, mx, my, (bf16_t*)out, B, Q, H, W, hidden, heads,
                           presence_row, log_scale);
    else
        hipLaunchKernelGGL(k_rpb_bias<float>, grid, dim3(threads), lds, st, boxes, mx, my, (float*)out, B, Q, H, W, hidden, heads,
                           presence_row, log_scale);
    return launched("sam3_rpb_bias_fwd");
}

}  // extern "C"
