// Does a K = 16 bf16 MFMA that accumulates onto the result of a K = 32 bf16 MFMA (emitted back to back) give the right answer on
// gfx950?  (profiles/r03b_mfma_k16_after_k32.txt: a kernel with that pair failed 82 parity tests.)  Each lane feeds random bf16
// operands; `chained` = mfma16(a4, b4, mfma32(a8, b8, 0)); `split` = mfma32(a8, b8, 0) + mfma16(a4, b4, 0) added on the VALU.
// The two differ by fp32 re-association only (<= ~1e-6 relative) when the hardware / compiler pair is right.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain_probe.hip -o tools/probes/_mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>   // 0: the compiler interleaves the eight chains; 1: each dependent pair adjacent; 2: adjacent + 16 s_nop states
__global__ void k(const uint4* A8, const uint4* B8, const uint2* A4, const uint2* B4, f32x4* chained, f32x4* split, int iters) {
    const int lane = threadIdx.x, blk = blockIdx.x;
    f32x4 c[8], s[8];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = ((blk * iters + it) * 8 + j) * 64 + lane;
            const uint4 a8 = A8[idx], b8 = B8[idx];
            const uint2 a4 = A4[idx], b4 = B4[idx];
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);     // MODE 1 / 2: the dependent pair ADJACENT in the instruction stream
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), d, 0, 0, 0);
            if (MODE == 2) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // MODE 2: sixteen wait states between them, by hand
            d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a4), __builtin_bit_cast(s16x4, b4), d, 0, 0, 0);
            if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
            c[j] = d;
            f32x4 e = {0.f, 0.f, 0.f, 0.f}, f = {0.f, 0.f, 0.f, 0.f};
            e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), e, 0, 0, 0);
            f = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a4), __builtin_bit_cast(s16x4, b4), f, 0, 0, 0);
            s[j] = e + f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = ((blk * iters + it) * 8 + j) * 64 + lane;
            chained[idx] = c[j];
            split[idx] = s[j];
        }
    }
}

static unsigned short bf16_of(float v) { unsigned u; memcpy(&u, &v, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
    const int blocks = 1024, iters = 16, n = blocks * iters * 8 * 64;
    std::vector<unsigned short> ha8(n * 8), hb8(n * 8), ha4(n * 4), hb4(n * 4);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : ha8) v = bf16_of(rnd());
    for (auto& v : hb8) v = bf16_of(rnd());
    for (auto& v : ha4) v = bf16_of(rnd());
    for (auto& v : hb4) v = bf16_of(rnd());
    uint4 *A8, *B8; uint2 *A4, *B4; f32x4 *C, *S;
    hipMalloc(&A8, n * 16); hipMalloc(&B8, n * 16); hipMalloc(&A4, n * 8); hipMalloc(&B4, n * 8); hipMalloc(&C, n * 16); hipMalloc(&S, n * 16);
    hipMemcpy(A8, ha8.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(B8, hb8.data(), n * 16, hipMemcpyHostToDevice);
    hipMemcpy(A4, ha4.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(B4, hb4.data(), n * 8, hipMemcpyHostToDevice);
    std::vector<float> hc(n * 4), hs(n * 4);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, A8, B8, A4, B4, C, S, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, A8, B8, A4, B4, C, S, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, A8, B8, A4, B4, C, S, iters);
            hipDeviceSynchronize();
            hipMemcpy(hc.data(), C, n * 16, hipMemcpyDeviceToHost); hipMemcpy(hs.data(), S, n * 16, hipMemcpyDeviceToHost);
            double worst = 0, mx = 0; long bad = 0;
            for (long i = 0; i < (long)n * 4; ++i) {
                const double d = fabs((double)hc[i] - hs[i]);
                worst = d > worst ? d : worst; mx = fabs(hs[i]) > mx ? fabs(hs[i]) : mx;
                if (d > 1e-3) ++bad;
            }
            printf("mode %d run %d: max |chained - split| = %.3e (max |value| %.2f), elements off by > 1e-3: %ld of %ld\n", mode, rep, worst, mx, bad, (long)n * 4);
        }
    }
    return 0;
}
