// What does v_cvt_pk_{fp8,bf8}_f32 do with out-of-range input on gfx950, with and without MODE.FP16_OVFL?
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/fp8_cvt_probe.hip -o tools/probes/_fp8_cvt_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n, int ovfl) {
    if (ovfl) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1 /* hwreg(HW_REG_MODE = 1, offset 23, size 1 -> field (size - 1) = 0) */, 1);
    const int i = threadIdx.x;
    if (i < n) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], in[i], w, false);
        int v = 0;
        v = __builtin_amdgcn_cvt_pk_bf8_f32(in[i], in[i], v, false);
        out[2 * i] = (unsigned)w & 0xff;
        out[2 * i + 1] = (unsigned)v & 0xff;
    }
}
int main() {
    const float vals[] = {1.f, 447.f, 448.f, 449.f, 464.f, 480.f, 1000.f, -1000.f, 57344.f, 60000.f, 1e30f, INFINITY, -INFINITY, NAN, 0.001f, 1e-10f};
    const int n = sizeof(vals) / sizeof(float);
    float* din; unsigned* dout;
    hipMalloc(&din, sizeof(vals)); hipMalloc(&dout, 2 * n * sizeof(unsigned));
    hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, n, ovfl);
        unsigned h[2 * 32];
        hipMemcpy(h, dout, 2 * n * sizeof(unsigned), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d\n", ovfl);
        for (int i = 0; i < n; ++i) printf("  %12g -> e4m3 0x%02x  e5m2 0x%02x\n", vals[i], h[2 * i], h[2 * i + 1]);
    }
    return 0;
}
