// sam3_lora_amd -- fp8 activation quantiser (gfx950): C-ABI of include/sam3_fp8_amd.h.
// One pass: 16 elements per thread per iteration (two 16-byte bf16 loads -> one 16-byte fp8 store), hardware
// conversion (v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32: OCP encodings on gfx950), amax by workgroup reduction + one atomic max
// per workgroup on the bit pattern (non-negative floats order like unsigned integers; max is order-independent -> deterministic).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "sam3_fp8_amd.h"
#include "fp8_common.inc"

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

__device__ __forceinline__ void load16(const bf16_t* p, float (&v)[16]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 8);
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void load16(const float* p, float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(p + 4 * q);
        v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
    }
}

template <typename XT, int FMT>
__global__ __launch_bounds__(256) void k_fp8_quantize(const XT* __restrict__ x, unsigned char* __restrict__ out,
                                                      const float* __restrict__ amax_in, float* __restrict__ amax_out,
                                                      float* __restrict__ scale_out, long long n16) {
    const Q8Out o{out, 0, amax_in, amax_out, scale_out, FMT};
    const Q8Scale sc = q8_begin(o, blockIdx.x == 0 && threadIdx.x == 0, blockIdx.x);
    float seen = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        float v[16];
        load16(x + i * 16, v);
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = q8_pack4<FMT>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], sc, seen);
        *reinterpret_cast<uint4*>(out + i * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    // at most one atomic per WORKGROUP, spread over the amax slots
    __shared__ float wmax[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) seen = q8_nanmax(seen, __shfl_down(seen, off, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = seen;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = q8_nanmax(q8_nanmax(wmax[0], wmax[1]), q8_nanmax(wmax[2], wmax[3]));
        q8_raise_slot(amax_out, m, blockIdx.x, sc.have);
    }
}

extern "C" {

const char* sam3_fp8_last_error(void) { return g_err; }

int sam3_fp8_quantize(const void* x, void* out, const float* amax_in, float* amax_out, float* scale_out, int64_t n,
                      int src_dtype, int fmt, void* stream) {
    g_err[0] = 0;
    if (!x || !out || !amax_in || !amax_out || !scale_out) return fail(-22, "NULL pointer");
    if (n <= 0 || (n % 16)) return fail(-22, "n must be a positive multiple of 16 (got %lld)", (long long)n);
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return fail(-22, "x and out must be 16-byte aligned");
    if ((src_dtype != 0 && src_dtype != 1) || (fmt != SAM3_FP8_E4M3 && fmt != SAM3_FP8_E5M2))
        return fail(-22, "unknown dtype %d / format %d", src_dtype, fmt);
    const long long n16 = n / 16;
    long long blocks = (n16 + 255) / 256;
    if (blocks > 256 * 4) blocks = 256 * 4;           // 4 workgroups per CU, grid-stride beyond (1024 atomics per call)
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)blocks);
#define L(XT, F) hipLaunchKernelGGL((k_fp8_quantize<XT, F>), grid, dim3(256), 0, st, (const XT*)x, (unsigned char*)out, amax_in, amax_out, scale_out, n16)
    if (src_dtype == 0) { if (fmt == SAM3_FP8_E4M3) L(bf16_t, SAM3_FP8_E4M3); else L(bf16_t, SAM3_FP8_E5M2); }
    else { if (fmt == SAM3_FP8_E4M3) L(float, SAM3_FP8_E4M3); else L(float, SAM3_FP8_E5M2); }
#undef L
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-5, "sam3_fp8_quantize: %s", hipGetErrorString(e));
}

}  // extern "C"
