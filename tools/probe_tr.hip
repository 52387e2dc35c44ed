// Hardware probe (diagnostic only): what each lane receives from ds_read_b64_tr_b16 when lane L
// supplies the LDS address of elements [4L, 4L+3] (values == element index), and the C/D layout of
// v_mfma_f32_16x16x32_bf16 with A = one-hot rows.  Build: hipcc --offload-arch=gfx950 tools/probe_tr.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(short* out, float* mf) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds[l * 4]));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
    // MFMA: A[i][k] = (k == 0) ? i+1 : 0 ; B[k][n] = (k == 0) ? (n+1)*100 : 0  -> D[i][n] = (i+1)*(n+1)*100
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)0.f; b[j] = (__bf16)0.f; }
    if ((l >> 4) == 0) { a[0] = (__bf16)(float)((l & 15) + 1); b[0] = (__bf16)(float)(((l & 15) + 1) * 4); }
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
    for (int j = 0; j < 4; ++j) mf[l * 4 + j] = d[j];
}
int main() {
    short* o; float* m;
    hipMalloc(&o, 512); hipMalloc(&m, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, m);
    short h[256]; float hm[256];
    hipMemcpy(h, o, 512, hipMemcpyDeviceToHost); hipMemcpy(hm, m, 1024, hipMemcpyDeviceToHost);
    printf("tr16_b64: lane -> 4 element indices received (lane L supplied address of elements 4L..4L+3)\n");
    for (int l = 0; l < 64; ++l) printf("L%02d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    printf("mfma D regs per lane (expect D[i][n]=(i+1)*(n+1)*4 with n=lane&15, i=(lane>>4)*4+reg)\n");
    for (int l = 0; l < 64; ++l) printf("L%02d: %6.0f %6.0f %6.0f %6.0f\n", l, hm[l*4], hm[l*4+1], hm[l*4+2], hm[l*4+3]);
    return 0;
}
