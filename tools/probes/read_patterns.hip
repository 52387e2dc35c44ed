// Read-only streaming patterns over the activation tensor of BASELINE configs[1] ([41472, 4736] bf16, row pitch 9472 B), non-temporal
// 16-byte loads, XOR-reduced -- which shape of the per-wave piece (rows x contiguous bytes), how many pieces in flight and how many
// waves the HBM system rewards.  The adapter path's read-only kernels (k_t1, k_t3, k_t3w at N = 4736) sit at 5.2-5.5 TB/s.
// hipcc --offload-arch=gfx950 -O3 tools/probes/read_patterns.hip -o tools/probes/_read_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ u32x4 ldnt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
__device__ __forceinline__ unsigned fold(u32x4 v) { return v[0] ^ v[1] ^ v[2] ^ v[3]; }

__global__ void k_linear(const u32x4* p, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= fold(ldnt(p + i));
    if (acc == 0x12345678u) out[0] = acc;
}
// A wave owns a column range of CB bytes and a row range; per step it reads RPS rows x CB bytes as 8 loads per lane (8 KB): the lanes of
// one load instruction cover (1024 / CB) rows x CB bytes.  DEPTH register sets ahead (1 = one step in flight while one is consumed).
// grid (column ranges / WPC, row groups); WPC waves of a workgroup take adjacent column ranges.
template <int CB, int DEPTH>
__global__ __launch_bounds__(256) void k_piece(const char* p, int rows, int rowbytes, int rows_per_wg, unsigned* out) {
    constexpr int LPR = CB / 16;            // lanes per row
    constexpr int RPI = 64 / (LPR < 64 ? LPR : 64);       // rows per load instruction (CB <= 1024)
    constexpr int IPR = CB > 1024 ? CB / 1024 : 1;        // load instructions per row (CB > 1024)
    constexpr int RPS = CB > 1024 ? 8 / IPR : 8 * RPI;    // rows per step
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb0 = (blockIdx.x * 4 + wave) * CB;
    if (cb0 >= rowbytes) return;
    const int r0 = blockIdx.y * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    const int nstep = (r1 - r0 + RPS - 1) / RPS;
    const int lr = CB > 1024 ? 0 : lane / LPR, lc = CB > 1024 ? lane : lane % LPR;
    u32x4 reg[DEPTH + 1][8];
    unsigned acc = 0;
    auto ld = [&](int st, u32x4* r) {
        const int s = st < nstep ? st : nstep - 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int row, cb;
            if (CB > 1024) { row = r0 + s * RPS + q / IPR; cb = cb0 + (q % IPR) * 1024 + lc * 16; }
            else { row = r0 + s * RPS + q * RPI + lr; cb = cb0 + lc * 16; }
            row = min(row, rows - 1);
            cb = min(cb, rowbytes - 16);
            r[q] = ldnt(p + (size_t)row * rowbytes + cb);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ld(d, reg[d]);
    for (int st = 0; st < nstep; st += DEPTH + 1) {
#pragma unroll
        for (int d = 0; d <= DEPTH; ++d) {
            ld(st + d + DEPTH, reg[(d + DEPTH) % (DEPTH + 1)]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc ^= fold(reg[d][q]);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// k_t1's shape: a workgroup owns 64 rows and walks the row in CB-byte chunks; all 256 threads load the 64 x CB tile, write it to the
// LDS, barrier (one chunk in flight while the previous is consumed).
template <int CB>
__global__ __launch_bounds__(256) void k_wgtile(const char* p, int rows, int rowbytes, unsigned* out) {
    constexpr int LPR = CB / 16, RPP = 256 / LPR, XP = 64 / RPP;
    __shared__ u32x4 xs[2][64 * LPR];
    const int r0 = blockIdx.x * 64, lr = threadIdx.x / LPR, lc = threadIdx.x % LPR;
    const int nk = (rowbytes + CB - 1) / CB;
    u32x4 xr[XP];
    unsigned acc = 0;
    auto ld = [&](int kc) {
#pragma unroll
        for (int i = 0; i < XP; ++i) xr[i] = ldnt(p + (size_t)min(r0 + lr + RPP * i, rows - 1) * rowbytes + min(kc * CB + lc * 16, rowbytes - 16));
    };
    auto st = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XP; ++i) xs[buf][(lr + RPP * i) * LPR + lc] = xr[i];
    };
    ld(0); st(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 1 < nk) ld(kc + 1);
#pragma unroll
        for (int i = 0; i < XP; ++i) acc ^= fold(xs[kc & 1][((threadIdx.x + i * 37) % 64) * LPR + lc]);
        if (kc + 1 < nk) st((kc + 1) & 1);
        __syncthreads();
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// k_t3's ownership (a workgroup owns a 256-byte column chunk and walks DOWN a row group) with k_t1's hand-over: all 256 threads load
// a [TR rows x 256 B] tile, write it to the LDS, barrier -- one tile in flight while the previous one is consumed.
template <int TR, int CB = 256>
__global__ __launch_bounds__(256) void k_wgtile_down(const char* p, int rows, int rowbytes, int rows_per_wg, unsigned* out) {
    constexpr int LPR = CB / 16, RPP = 256 / LPR, XP = TR / RPP;
    __shared__ u32x4 xs[2][TR * LPR];
    const int cb = blockIdx.x * CB, r0 = blockIdx.y * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    const int nst = (r1 - r0 + TR - 1) / TR, lr = threadIdx.x / LPR, lc = threadIdx.x % LPR;
    u32x4 xr[XP];
    unsigned acc = 0;
    auto ld = [&](int st) {
#pragma unroll
        for (int i = 0; i < XP; ++i) xr[i] = ldnt(p + (size_t)min(r0 + st * TR + lr + RPP * i, rows - 1) * rowbytes + min(cb + lc * 16, rowbytes - 16));
    };
    auto st_ = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XP; ++i) xs[buf][(lr + RPP * i) * LPR + lc] = xr[i];
    };
    ld(0); st_(0);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        if (s + 1 < nst) ld(s + 1);
#pragma unroll
        for (int i = 0; i < XP; ++i) acc ^= fold(xs[s & 1][((threadIdx.x + i * 37) % TR) * LPR + lc]);
        if (s + 1 < nst) st_((s + 1) & 1);
        __syncthreads();
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const int rows = 41472, rowbytes = 9472;
    const size_t bytes = (size_t)rows * rowbytes;
    char* a; unsigned* o;
    hipMalloc(&a, bytes); hipMalloc(&o, 4); hipMemset(a, 1, bytes);
    char* flush; hipMalloc(&flush, (size_t)512 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0.f; const int it = 8;
        for (int i = 0; i < 2; ++i) launch();
        for (int i = 0; i < it; ++i) {
            hipMemsetAsync(flush, i, (size_t)512 << 20, 0);      // the Infinity Cache holds nothing of `a` when the kernel starts
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
        }
        printf("%-44s avg %7.1f us  %6.0f GB/s   best %7.1f us %6.0f GB/s\n", name, sum / it * 1e3, bytes / (sum / it * 1e-3) / 1e9, best * 1e3, bytes / (best * 1e-3) / 1e9);
    };
    for (int g : {2048, 4096, 16384}) { char nm[64]; snprintf(nm, 64, "linear nt grid=%d", g);
        run(nm, [&] { hipLaunchKernelGGL(k_linear, dim3(g), dim3(256), 0, 0, (const u32x4*)a, bytes / 16, o); }); }
#define PIECE(CB, D, NR) { const int nx = (rowbytes + 4 * CB - 1) / (4 * CB), rpw = (rows + NR - 1) / NR, act = ((rowbytes + CB - 1) / CB) * NR; \
        char nm[64]; snprintf(nm, 64, "piece %4d B x depth %d, %5d waves (%4d wgs)", CB, D, act, nx * NR); \
        run(nm, [&] { hipLaunchKernelGGL((k_piece<CB, D>), dim3(nx, NR), dim3(256), 0, 0, a, rows, rowbytes, rpw, o); }); }
#define SWEEP(CB) { const int per = (rowbytes + CB - 1) / CB; for (int w : {1024, 2048, 4096, 8192}) { const int NR = (w + per - 1) / per; PIECE(CB, 1, NR) } \
                    { const int NR = (2048 + per - 1) / per; PIECE(CB, 2, NR) } }
    SWEEP(256)
#define DOWN(TR, CB, NR) { const int nx = (rowbytes + CB - 1) / CB; char nm[64]; snprintf(nm, 64, "wg tile DOWN %3d rows x %4d B, %4d wgs", TR, CB, nx * NR); const int rpw = (((rows + NR - 1) / NR) + TR - 1) / TR * TR; \
        run(nm, [&] { hipLaunchKernelGGL((k_wgtile_down<TR, CB>), dim3(nx, NR), dim3(256), 0, 0, a, rows, rowbytes, rpw, o); }); }
    DOWN(64, 256, 13) DOWN(64, 256, 20) DOWN(32, 256, 27)
    DOWN(32, 512, 13) DOWN(32, 512, 26) DOWN(32, 512, 40) DOWN(32, 512, 52) DOWN(64, 512, 13) DOWN(64, 512, 26) DOWN(64, 512, 40)
    DOWN(16, 1024, 26) DOWN(16, 1024, 52) DOWN(16, 1024, 78) DOWN(32, 1024, 26) DOWN(32, 1024, 52)
    { run("wg tile 64 rows x 256 B, barrier per chunk (648 wgs)", [&] { hipLaunchKernelGGL((k_wgtile<256>), dim3((rows + 63) / 64), dim3(256), 0, 0, a, rows, rowbytes, o); });
      run("wg tile 64 rows x 512 B, barrier per chunk (648 wgs)", [&] { hipLaunchKernelGGL((k_wgtile<512>), dim3((rows + 63) / 64), dim3(256), 0, 0, a, rows, rowbytes, o); }); }
    return 0;
}
