// sam3_lora_amd -- mask-loss kernels (gfx950): C-ABI of include/sam3_loss_amd.h.
//
// The step right after the adapter path: for every matched instance the reference upsamples the
// 288 x 288 mask logits to the 1008 x 1008 target bilinearly, then evaluates a sigmoid focal loss and a dice loss on
// the full-resolution tensor (sam3/train/loss/loss_fns.py:679-707, :159-176 focal formula, :79-123 dice) -- about twenty
// elementwise / reduction passes over [N, 1008, 1008] fp32 in PyTorch (the reference's Triton kernels cover the focal
// part only).  Here the upsampled tensor never exists:
//
//   k_mask_fwd   one pass over the TARGET resolution: each workgroup interpolates its tile from an LDS copy of the
//                logits patch, evaluates focal / p*t / p / t per pixel and writes four partial sums per tile;
//   k_mask_sum   fixed-order sum of the tile partials -> S[n] = (sum focal, sum p t, sum p, sum t);
//   k_mask_bwd   gradient of any function of S back to the LOW-resolution logits as a GATHER (each logit pixel sums the
//                ~7 x 7 target pixels whose bilinear stencil touches it): no atomics, bit-reproducible.
//
// HBM traffic per instance: the 1 MB boolean target once per direction + the 166 KB of logits; the PyTorch form moves
// ~80 MB.  Interpolation follows torch's upsample_bilinear2d(align_corners=False): src = (dst + 0.5) * in/out - 0.5,
// clamped at 0, second tap clamped at the last row / column, all in fp32.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "sam3_loss_amd.h"

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
}  // namespace

__device__ __forceinline__ float ld_logit(const float* p) { return *p; }
__device__ __forceinline__ float ld_logit(const bf16_t* p) { return __uint_as_float((unsigned)(*p) << 16); }
__device__ __forceinline__ void st_grad(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_grad(bf16_t* p, float v) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 t = {(__bf16)v, (__bf16)0.f};
    *p = (bf16_t)(__builtin_bit_cast(unsigned, t) & 0xffffu);
}

struct Tap {
    int i0, i1;
    float l;      // weight of i1; 1 - l for i0
};
__device__ __forceinline__ Tap tap_of(int dst, float scale, int in_size) {
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    Tap t;
    t.i0 = (int)s;                                   // floor (s >= 0)
    t.i0 = t.i0 < in_size - 1 ? t.i0 : in_size - 1;
    t.i1 = t.i0 < in_size - 1 ? t.i0 + 1 : t.i0;
    t.l = s - (float)t.i0;
    return t;
}

// per-pixel terms on the upsampled logit x with target t in {0, 1}
__device__ __forceinline__ void pixel_terms(float x, float t, float alpha, float gamma, float& focal, float& p) {
    p = 1.f / (1.f + __expf(-x));
    const float ce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));          // BCE with logits
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float q = 1.f - pt;
    const float mod = gamma == 2.f ? q * q : powf(q, gamma);
    focal = ce * mod;
    if (alpha >= 0.f) focal *= alpha * t + (1.f - alpha) * (1.f - t);
}
// d focal / d x  (see derivation in DESIGN.md): a_t * s * ( -q^(gamma+1) - gamma * ce * p_t * q^gamma ), s = 2t - 1
__device__ __forceinline__ float focal_grad(float x, float t, float alpha, float gamma, float& p) {
    p = 1.f / (1.f + __expf(-x));
    const float ce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float q = 1.f - pt;
    const float qg = gamma == 2.f ? q * q : powf(q, gamma);
    float g = (2.f * t - 1.f) * (-qg * q - gamma * ce * pt * qg);
    if (alpha >= 0.f) g *= alpha * t + (1.f - alpha) * (1.f - t);
    return g;
}

constexpr int TY = 32, TX = 64;          // target-resolution tile of one workgroup (256 threads, 8 pixels each)
constexpr int PATCH_MAX = 48 * 80;       // logits patch a tile can need (any scale >= 1/2 fits; checked on the host)

template <typename ST>
__global__ __launch_bounds__(256) void k_mask_fwd(const ST* __restrict__ src, const unsigned char* __restrict__ tgt,
                                                  float* __restrict__ part, int h, int w, int H, int W, float sy, float sx,
                                                  float alpha, float gamma, int tiles_x, int tiles) {
    __shared__ float patch[PATCH_MAX];
    __shared__ float red[4][4];
    const int n = blockIdx.y, tile = blockIdx.x;
    const int Y0 = (tile / tiles_x) * TY, X0 = (tile % tiles_x) * TX;
    const int Y1 = min(Y0 + TY, H) - 1, X1 = min(X0 + TX, W) - 1;
    const Tap ya = tap_of(Y0, sy, h), yb = tap_of(Y1, sy, h), xa = tap_of(X0, sx, w), xb = tap_of(X1, sx, w);
    const int py0 = ya.i0, px0 = xa.i0, ph = yb.i1 - py0 + 1, pw = xb.i1 - px0 + 1;
    const ST* s = src + (long long)n * h * w;
    for (int e = threadIdx.x; e < ph * pw; e += 256) patch[e] = ld_logit(s + (long long)(py0 + e / pw) * w + px0 + e % pw);
    __syncthreads();
    const unsigned char* t = tgt + (long long)n * H * W;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int X = X0 + lx;
    if (X < W) {
        const Tap tx = tap_of(X, sx, w);
#pragma unroll
        for (int r = 0; r < TY / 4; ++r) {
            const int Y = Y0 + ly + 4 * r;
            if (Y < H) {
                const Tap ty = tap_of(Y, sy, h);
                const float* r0 = patch + (ty.i0 - py0) * pw - px0;
                const float* r1 = patch + (ty.i1 - py0) * pw - px0;
                const float top = r0[tx.i0] + tx.l * (r0[tx.i1] - r0[tx.i0]);
                const float bot = r1[tx.i0] + tx.l * (r1[tx.i1] - r1[tx.i0]);
                const float x = top + ty.l * (bot - top);
                const float tv = t[(long long)Y * W + X] ? 1.f : 0.f;
                float f, p;
                pixel_terms(x, tv, alpha, gamma, f, p);
                a0 += f; a1 += p * tv; a2 += p; a3 += tv;
            }
        }
    }
    // wave reduction (fixed shuffle tree), then the four waves in fixed order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_down(a0, o, 64); a1 += __shfl_down(a1, o, 64);
        a2 += __shfl_down(a2, o, 64); a3 += __shfl_down(a3, o, 64);
    }
    if (lx == 0) { red[ly][0] = a0; red[ly][1] = a1; red[ly][2] = a2; red[ly][3] = a3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        part[((long long)n * tiles + tile) * 4 + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
}

__global__ __launch_bounds__(256) void k_mask_sum(const float* __restrict__ part, float* __restrict__ S, int tiles) {
    __shared__ float red[256][4];
    const int n = blockIdx.x;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = threadIdx.x; t < tiles; t += 256) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((long long)n * tiles + t) * 4);
        a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = a[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[threadIdx.x][k] += red[threadIdx.x + o][k];
        __syncthreads();
    }
    if (threadIdx.x < 4) S[n * 4 + threadIdx.x] = red[0][threadIdx.x];
}

// backward: workgroup = a GY x GX block of LOGIT pixels of instance n; the per-target-pixel gradient
//   g(Y, X) = c_focal * dfocal/dx + (c_pt * t + c_p) * p (1 - p)
// is evaluated once per target pixel of the region the block's stencils reach (into LDS), then every logit pixel gathers
//   gsrc[i, j] = sum_{Y, X} wy(Y, i) * wx(X, j) * g(Y, X).
constexpr int GY = 8, GX = 16;
constexpr int REG_MAX = 64 * 96;         // target-resolution region of one block (scale <= 4 fits; checked on the host)

template <typename ST, typename GT>
__global__ __launch_bounds__(128) void k_mask_bwd(const ST* __restrict__ src, const unsigned char* __restrict__ tgt,
                                                  const float* __restrict__ coef, GT* __restrict__ gsrc, int h, int w, int H,
                                                  int W, float sy, float sx, float inv_sy, float inv_sx, float alpha,
                                                  float gamma, int blocks_x) {
    __shared__ float patch[(GY + 2) * (GX + 2)];
    __shared__ float greg[REG_MAX];
    const int n = blockIdx.y;
    const int i0 = (blockIdx.x / blocks_x) * GY, j0 = (blockIdx.x % blocks_x) * GX;
    // logits patch [i0-1, i0+GY] x [j0-1, j0+GX], clamped
    const int pi0 = max(i0 - 1, 0), pj0 = max(j0 - 1, 0);
    const int pi1 = min(i0 + GY, h - 1), pj1 = min(j0 + GX, w - 1);
    const int ph = pi1 - pi0 + 1, pw = pj1 - pj0 + 1;
    const ST* s = src + (long long)n * h * w;
    for (int e = threadIdx.x; e < ph * pw; e += 128) patch[e] = ld_logit(s + (long long)(pi0 + e / pw) * w + pj0 + e % pw);
    // target pixels whose stencil can touch rows [i0, i0+GY): src coordinate in (i0 - 1, i0 + GY)
    //   (Y + 0.5) * sy - 0.5 > i0 - 1  <=>  Y > (i0 - 0.5) / sy - 0.5
    int Ya = (int)floorf(((float)i0 - 0.5f) * inv_sy - 0.5f), Yb = (int)ceilf(((float)(i0 + GY) + 0.5f) * inv_sy - 0.5f);
    int Xa = (int)floorf(((float)j0 - 0.5f) * inv_sx - 0.5f), Xb = (int)ceilf(((float)(j0 + GX) + 0.5f) * inv_sx - 0.5f);
    if (i0 == 0) Ya = 0;                 // clamped coordinates: everything above / left of the first centre maps to row 0
    if (j0 == 0) Xa = 0;
    if (i0 + GY >= h) Yb = H - 1;
    if (j0 + GX >= w) Xb = W - 1;
    Ya = max(Ya, 0); Xa = max(Xa, 0); Yb = min(Yb, H - 1); Xb = min(Xb, W - 1);
    const int rh = Yb - Ya + 1, rw = Xb - Xa + 1;
    const float cf = coef[n * 4 + 0], cpt = coef[n * 4 + 1], cp = coef[n * 4 + 2];
    const unsigned char* t = tgt + (long long)n * H * W;
    __syncthreads();
    for (int e = threadIdx.x; e < rh * rw; e += 128) {
        const int Y = Ya + e / rw, X = Xa + e % rw;
        const Tap ty = tap_of(Y, sy, h), tx = tap_of(X, sx, w);
        float g = 0.f;
        // taps outside the loaded patch belong to other blocks' pixels only; such target pixels contribute with weight
        // zero to this block, so any finite value will do
        if (ty.i0 >= pi0 && ty.i1 <= pi1 && tx.i0 >= pj0 && tx.i1 <= pj1) {
            const float* r0 = patch + (ty.i0 - pi0) * pw - pj0;
            const float* r1 = patch + (ty.i1 - pi0) * pw - pj0;
            const float top = r0[tx.i0] + tx.l * (r0[tx.i1] - r0[tx.i0]);
            const float bot = r1[tx.i0] + tx.l * (r1[tx.i1] - r1[tx.i0]);
            const float x = top + ty.l * (bot - top);
            const float tv = t[(long long)Y * W + X] ? 1.f : 0.f;
            float p;
            const float gf = focal_grad(x, tv, alpha, gamma, p);
            g = cf * gf + (cpt * tv + cp) * p * (1.f - p);
        }
        greg[e] = g;
    }
    __syncthreads();
    const int li = threadIdx.x / GX, lj = threadIdx.x % GX;
    const int i = i0 + li, j = j0 + lj;
    if (i >= h || j >= w) return;
    float acc = 0.f;
    // only the ~2/sy target rows / columns whose stencil can touch (i, j): src coordinate in (i - 1, i + 1)
    const int ylo = max(Ya, (int)floorf(((float)i - 0.5f) * inv_sy - 0.5f));
    const int yhi = i >= h - 1 ? Yb : min(Yb, (int)ceilf(((float)i + 1.5f) * inv_sy - 0.5f));
    const int xlo = max(Xa, (int)floorf(((float)j - 0.5f) * inv_sx - 0.5f));
    const int xhi = j >= w - 1 ? Xb : min(Xb, (int)ceilf(((float)j + 1.5f) * inv_sx - 0.5f));
    for (int Y = (i == 0 ? Ya : ylo); Y <= yhi; ++Y) {
        const Tap ty = tap_of(Y, sy, h);
        float wy = 0.f;
        if (ty.i0 == i) wy += 1.f - ty.l;
        if (ty.i1 == i) wy += ty.l;          // i0 == i1 at the last row: both taps land on it, weights sum to 1
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int X = (j == 0 ? Xa : xlo); X <= xhi; ++X) {
            const Tap tx = tap_of(X, sx, w);
            float wx = 0.f;
            if (tx.i0 == j) wx += 1.f - tx.l;
            if (tx.i1 == j) wx += tx.l;
            if (wx != 0.f) row += wx * greg[(Y - Ya) * rw + (X - Xa)];
        }
        acc += wy * row;
    }
    st_grad(gsrc + ((long long)n * h + i) * w + j, acc);
}

// ------------------------------------------------------------------------------------------------------------------
// matched box pairs: IoU and generalised IoU of a[i] (prediction, xyxy) with b[i] (target, xyxy), and the gradient of
// any function of the two with respect to a.  One thread per pair; stands in for ~35 tiny elementwise operators per
// decoder output (loss_fns.py "Boxes" :345-400 through box_ops.generalized_box_iou's diagonal), which cost host time,
// not device time.  Sub-gradients follow autograd's: clamp(min=0) passes the gradient at 0, min / max split it at ties.
// ------------------------------------------------------------------------------------------------------------------
struct PairTerms {
    float iw, ih, hw, hh, inter, uni, hull, area_a;
};
__device__ __forceinline__ PairTerms pair_terms(const float4 a, const float4 b) {
    PairTerms t;
    t.area_a = (a.z - a.x) * (a.w - a.y);
    const float area_b = (b.z - b.x) * (b.w - b.y);
    t.iw = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
    t.ih = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    t.hw = fmaxf(fmaxf(a.z, b.z) - fminf(a.x, b.x), 0.f);
    t.hh = fmaxf(fmaxf(a.w, b.w) - fminf(a.y, b.y), 0.f);
    t.inter = t.iw * t.ih;
    t.hull = t.hw * t.hh;
    t.uni = t.area_a + area_b - t.inter;
    return t;
}

__global__ __launch_bounds__(64) void k_box_pair_fwd(const float4* __restrict__ a, const float4* __restrict__ b,
                                                     float2* __restrict__ out, int T) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= T) return;
    const PairTerms t = pair_terms(a[i], b[i]);
    const float iou = t.inter / t.uni;
    out[i] = make_float2(iou, iou - (t.hull - t.uni) / t.hull);
}

// weight of `x` in min(x, y) / max(x, y): 1 when it is the selected operand, 1/2 at a tie, 0 otherwise
__device__ __forceinline__ float sel_min(float x, float y) { return x < y ? 1.f : (x == y ? 0.5f : 0.f); }
__device__ __forceinline__ float sel_max(float x, float y) { return x > y ? 1.f : (x == y ? 0.5f : 0.f); }

__global__ __launch_bounds__(64) void k_box_pair_bwd(const float4* __restrict__ a, const float4* __restrict__ b,
                                                     const float2* __restrict__ coef, float4* __restrict__ ga, int T) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= T) return;
    const float4 A = a[i], B = b[i];
    const PairTerms t = pair_terms(A, B);
    const float c_iou = coef[i].x, c_giou = coef[i].y;
    // out = c_iou * iou + c_giou * (iou - 1 + uni / hull); iou = inter / uni
    const float k = c_iou + c_giou;
    const float g_inter_direct = k / t.uni;                                   // d/d inter at fixed uni
    const float g_uni = -k * t.inter / (t.uni * t.uni) + c_giou / t.hull;       // d/d uni
    const float g_hull = -c_giou * t.uni / (t.hull * t.hull);
    const float g_inter = g_inter_direct - g_uni;                             // uni = area_a + area_b - inter
    const float g_area = g_uni;
    // inter = iw * ih, iw = clamp(min(a.z, b.z) - max(a.x, b.x), 0)
    const float riw = fminf(A.z, B.z) - fmaxf(A.x, B.x), rih = fminf(A.w, B.w) - fmaxf(A.y, B.y);
    const float g_iw = riw >= 0.f ? g_inter * t.ih : 0.f, g_ih = rih >= 0.f ? g_inter * t.iw : 0.f;
    const float rhw = fmaxf(A.z, B.z) - fminf(A.x, B.x), rhh = fmaxf(A.w, B.w) - fminf(A.y, B.y);
    const float g_hw = rhw >= 0.f ? g_hull * t.hh : 0.f, g_hh = rhh >= 0.f ? g_hull * t.hw : 0.f;
    const float wa = A.z - A.x, ha = A.w - A.y;
    float4 g;
    g.x = -g_area * ha - g_iw * sel_max(A.x, B.x) - g_hw * sel_min(A.x, B.x);
    g.y = -g_area * wa - g_ih * sel_max(A.y, B.y) - g_hh * sel_min(A.y, B.y);
    g.z = g_area * ha + g_iw * sel_min(A.z, B.z) + g_hw * sel_max(A.z, B.z);
    g.w = g_area * wa + g_ih * sel_min(A.w, B.w) + g_hh * sel_max(A.w, B.w);
    ga[i] = g;
}

extern "C" {

const char* sam3_loss_last_error(void) { return g_err; }

size_t sam3_mask_loss_workspace_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const long long tiles = (long long)((H + TY - 1) / TY) * ((W + TX - 1) / TX);
    return (size_t)N * tiles * 4 * sizeof(float);
}

static int check(const void* src, const void* tgt, int N, int h, int w, int H, int W, int dtype) {
    if (!src || !tgt) return fail(-22, "NULL pointer");
    if (N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return fail(-22, "bad shape N=%d h=%d w=%d H=%d W=%d", N, h, w, H, W);
    if (dtype != 0 && dtype != 1) return fail(-22, "unknown dtype %d", dtype);
    // tile geometry: a target tile needs a logits patch of at most 48 x 80, a logits block a target region of 64 x 96
    const float sy = (float)h / H, sx = (float)w / W;
    if ((TY * sy + 3) * (TX * sx + 3) > PATCH_MAX || ((GY + 2) / sy + 3) * ((GX + 2) / sx + 3) > REG_MAX)
        return fail(-95, "resize ratio outside the supported range (%d x %d -> %d x %d)", h, w, H, W);
    return 0;
}

int sam3_mask_loss_fwd(const void* src, const void* tgt, float* sums, int N, int h, int w, int H, int W, float alpha,
                       float gamma, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    g_err[0] = 0;
    int rc;
    if ((rc = check(src, tgt, N, h, w, H, W, dtype))) return rc;
    if (!sums) return fail(-22, "sums is NULL");
    const size_t need = sam3_mask_loss_workspace_bytes(N, H, W);
    if (!workspace || workspace_bytes < need) return fail(-12, "workspace too small: need %zu, got %zu", need, workspace_bytes);
    const int tiles_x = (W + TX - 1) / TX, tiles = tiles_x * ((H + TY - 1) / TY);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)tiles, (unsigned)N);
    if (dtype == 0)
        hipLaunchKernelGGL((k_mask_fwd<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)src, (const unsigned char*)tgt,
                           (float*)workspace, h, w, H, W, sy, sx, alpha, gamma, tiles_x, tiles);
    else
        hipLaunchKernelGGL((k_mask_fwd<float>), grid, dim3(256), 0, st, (const float*)src, (const unsigned char*)tgt,
                           (float*)workspace, h, w, H, W, sy, sx, alpha, gamma, tiles_x, tiles);
    hipLaunchKernelGGL(k_mask_sum, dim3((unsigned)N), dim3(256), 0, st, (const float*)workspace, sums, tiles);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-5, "sam3_mask_loss_fwd: %s", hipGetErrorString(e));
}

int sam3_mask_loss_bwd(const void* src, const void* tgt, const float* coef, void* gsrc, int N, int h, int w, int H, int W,
                       float alpha, float gamma, int dtype, int grad_dtype, void* stream) {
    g_err[0] = 0;
    int rc;
    if ((rc = check(src, tgt, N, h, w, H, W, dtype))) return rc;
    if (!coef || !gsrc) return fail(-22, "NULL pointer");
    if (grad_dtype != 0 && grad_dtype != 1) return fail(-22, "unknown grad dtype %d", grad_dtype);
    const int blocks_x = (w + GX - 1) / GX, blocks = blocks_x * ((h + GY - 1) / GY);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)blocks, (unsigned)N);
#define L(ST, GT)                                                                                                        \
    hipLaunchKernelGGL((k_mask_bwd<ST, GT>), grid, dim3(128), 0, st, (const ST*)src, (const unsigned char*)tgt, coef, (GT*)gsrc, \
                       h, w, H, W, sy, sx, 1.f / sy, 1.f / sx, alpha, gamma, blocks_x)
    if (dtype == 0) { if (grad_dtype == 0) L(bf16_t, bf16_t); else L(bf16_t, float); }
    else { if (grad_dtype == 0) L(float, bf16_t); else L(float, float); }
#undef L
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-5, "sam3_mask_loss_bwd: %s", hipGetErrorString(e));
}

int sam3_box_pair_fwd(const float* a, const float* b, float* out, int T, void* stream) {
    g_err[0] = 0;
    if (T < 0) return fail(-22, "bad pair count %d", T);
    if (T == 0) return 0;
    if (!a || !b || !out) return fail(-22, "NULL pointer");
    if (((uintptr_t)a | (uintptr_t)b) & 15 || ((uintptr_t)out & 7)) return fail(-22, "boxes must be 16-byte aligned");
    hipLaunchKernelGGL(k_box_pair_fwd, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (const float4*)a,
                       (const float4*)b, (float2*)out, T);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-5, "sam3_box_pair_fwd: %s", hipGetErrorString(e));
}

int sam3_box_pair_bwd(const float* a, const float* b, const float* coef, float* ga, int T, void* stream) {
    g_err[0] = 0;
    if (T < 0) return fail(-22, "bad pair count %d", T);
    if (T == 0) return 0;
    if (!a || !b || !coef || !ga) return fail(-22, "NULL pointer");
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)ga) & 15 || ((uintptr_t)coef & 7)) return fail(-22, "boxes must be 16-byte aligned");
    hipLaunchKernelGGL(k_box_pair_bwd, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (const float4*)a,
                       (const float4*)b, (const float2*)coef, (float4*)ga, T);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-5, "sam3_box_pair_bwd: %s", hipGetErrorString(e));
}

}  // extern "C"
