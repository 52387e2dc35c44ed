// HBM ceilings on this box (diagnostic): pure read, pure write, copy, read-modify-write, 16 B/lane.
// hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o tools/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_read(const uint4* p, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_write(uint4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4(i, 1, 2, 3);
}
__global__ void k_copy(const uint4* a, uint4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_rmw(uint4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i]; v.x += 1; p[i] = v;
    }
}
// block-contiguous variant: each block streams its own contiguous slab (like the LoRA kernels' row ranges)
__global__ void k_read_slab(const uint4* p, size_t n, unsigned* out) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x, b0 = blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
    unsigned acc = 0;
    for (size_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// LoRA-like tiled read: workgroup = `trows` rows x full row, streamed in `tbytes`-wide column pieces
// (row pitch = rowbytes).  Shows how the width of the contiguous piece per row bounds HBM efficiency.
__global__ void k_read_tiles(const char* p, int rows, int rowbytes, int trows, int tbytes, unsigned* out) {
    const int r0 = blockIdx.x * trows;
    const int lanes_per_row = tbytes / 16, rows_per_pass = 256 / lanes_per_row;
    const int lr = threadIdx.x / lanes_per_row, lc = threadIdx.x % lanes_per_row;
    unsigned acc = 0;
    for (int kb = 0; kb < rowbytes; kb += tbytes)
        for (int rr = lr; rr < trows; rr += rows_per_pass) {
            const int r = r0 + rr, cb = kb + lc * 16;
            if (r < rows && cb < rowbytes) { uint4 v = *(const uint4*)(p + (size_t)r * rowbytes + cb); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
    if (acc == 0x12345678u) out[0] = acc;
}
// column-chunk ownership (T3-like): workgroup (cx, ry) reads rows [ry*rpw, +rpw) x cols [cx*tbytes, +tbytes)
__global__ void k_read_colchunks(const char* p, int rows, int rowbytes, int rpw, int tbytes, unsigned* out) {
    const int r0 = blockIdx.y * rpw, cb0 = blockIdx.x * tbytes;
    const int lanes_per_row = tbytes / 16, rows_per_pass = 256 / lanes_per_row;
    const int lr = threadIdx.x / lanes_per_row, lc = threadIdx.x % lanes_per_row;
    unsigned acc = 0;
    for (int rr = lr; rr < rpw; rr += rows_per_pass) {
        const int r = r0 + rr, cb = cb0 + lc * 16;
        if (r < rows && cb < rowbytes) { uint4 v = *(const uint4*)(p + (size_t)r * rowbytes + cb); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// wave-private streaming (no LDS, no barrier): a wave owns 16 rows x a K-range, reads 256-B pieces per row
// (4 dwordx4 per lane per step), explicit distance-2 prefetch with two named register sets.
__global__ __launch_bounds__(256) void k_read_wave(const char* p, int rows, int rowbytes, int ksplit, unsigned* out) {
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tile = wid / ksplit, kp = wid % ksplit;
    const int kbytes = rowbytes / ksplit, nstep = kbytes / 256;
    if (tile * 16 >= rows) return;
    const char* base = p + (size_t)(tile * 16 + (lane >> 4)) * rowbytes + (size_t)kp * kbytes + (lane & 15) * 16;
    const size_t rs4 = (size_t)4 * rowbytes;
    uint4 a[4], b[4];
    unsigned acc = 0;
    auto ld = [&](int st, uint4* r) {
        const int s = st < nstep ? st : nstep - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = *(const uint4*)(base + q * rs4 + (size_t)s * 256);
    };
    auto use = [&](const uint4* r) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc ^= r[q].x ^ r[q].y ^ r[q].z ^ r[q].w;
    };
    ld(0, a); ld(1, b);
    for (int st = 0; st < nstep; st += 2) {
        uint4 c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = a[q];
        ld(st + 2, a); use(c);
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = b[q];
        ld(st + 3, b); use(c);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = (size_t)768 << 20, n = bytes / 16;
    uint4 *a, *b; unsigned* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, int grid, double traffic, auto launch) {
        for (int i = 0; i < 3; ++i) launch(grid);
        hipEventRecord(e0);
        const int it = 10;
        for (int i = 0; i < it; ++i) launch(grid);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s grid=%5d  %8.1f us  %7.1f GB/s\n", name, grid, ms * 1e3 / it, traffic / (ms / it * 1e-3) / 1e9);
    };
    for (int grid : {1024, 2048, 4096, 16384}) {
        run("read (grid-stride)", grid, bytes, [&](int g) { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, o); });
        run("read (block slabs)", grid, bytes, [&](int g) { hipLaunchKernelGGL(k_read_slab, dim3(g), dim3(256), 0, 0, a, n, o); });
        run("write", grid, bytes, [&](int g) { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n); });
        run("copy (r+w bytes)", grid, 2.0 * bytes, [&](int g) { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n); });
        run("rmw in place (r+w bytes)", grid, 2.0 * bytes, [&](int g) { hipLaunchKernelGGL(k_rmw, dim3(g), dim3(256), 0, 0, a, n); });
    }
    {
        const int rows = 41472, rowbytes = 9472;
        const double tb = (double)rows * rowbytes;
        for (int trows : {16, 32, 64, 128})
            for (int tbytes : {256, 512, 1024, 2048}) {
                char name[64]; snprintf(name, 64, "tiles %3d rows x %4d B", trows, tbytes);
                run(name, (rows + trows - 1) / trows, tb, [&](int g) {
                    hipLaunchKernelGGL(k_read_tiles, dim3(g), dim3(256), 0, 0, (const char*)a, rows, rowbytes, trows, tbytes, o); });
            }
        for (int ksplit : {1, 2, 4}) {
            char name[64]; snprintf(name, 64, "wave-private 16r, ksplit=%d", ksplit);
            const int nwaves = (rows / 16) * ksplit;
            run(name, (nwaves + 3) / 4, tb, [&](int g) {
                hipLaunchKernelGGL(k_read_wave, dim3(g), dim3(256), 0, 0, (const char*)a, rows, rowbytes, ksplit, o); });
        }
        for (int tbytes : {256, 512, 1024})
            for (int nr : {13, 26, 54, 108}) {
                char name[64]; snprintf(name, 64, "colchunks %4d B x NR=%3d", tbytes, nr);
                const int nx = (rowbytes + tbytes - 1) / tbytes, rpw = (rows + nr - 1) / nr;
                run(name, nx * nr, tb, [&](int) {
                    hipLaunchKernelGGL(k_read_colchunks, dim3(nx, nr), dim3(256), 0, 0, (const char*)a, rows, rowbytes, rpw, tbytes, o); });
            }
    }
    return 0;
}
