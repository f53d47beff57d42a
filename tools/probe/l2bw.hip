// l2bw.hip -- how many bytes per second does ONE CU pull from L2 / HBM under chip-wide load, by path?
//   mode 0: LDS-DMA   (global_load_lds_dwordx4: what conv_igemm / conv_gemm8 / conv_ht stream their operand tiles with)
//   mode 1: VGPR      (global_load_dwordx4 into registers)
//   mode 2: both at once (half the waves each)
// source: `shared` = every workgroup walks the same 2 MB (L2-resident: the weight side of a GEMM), else each workgroup its own slice of
// 1 GB (the activation side).  Development tool; build: hipcc -O3 --offload-arch=gfx950 tools/probe/l2bw.hip -o tools/probe/l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(const char *src, size_t per_wg, size_t window, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const char *base = src + (size_t)blockIdx.x * per_wg;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool dma = MODE == 0 || (MODE == 2 && (wave & 1) == 0);
    // every wave walks the window in 8 KB strides (8 waves x 1 KB per instruction), 16 instructions per batch
    size_t off = (size_t)wave * 1024 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        if (dma) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off),
                                                 (__attribute__((address_space(3))) void *)(lds + wave * 16384 + u * 1024), 16, 0, 0);
                off += 8192;
                if (off >= window) off -= window;
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            f4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v[u] = *reinterpret_cast<const f4 *>(base + off);
                off += 8192;
                if (off >= window) off -= window;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// The activation side of a 1x1-conv GEMM as conv_gemm8 walks it: a workgroup owns 256 rows of `rowbytes` bytes; a k-step takes `seg`
// contiguous bytes of every row (512 threads: 16 bytes each, seg / 16 lanes per row), 16 bytes x 64 lanes per wave instruction; after
// rowbytes / seg k-steps the workgroup moves on to its next 256 rows.
__global__ __launch_bounds__(512) void probe_rows(const char *src, size_t total, int rowbytes, int seg, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int lpr = seg / 16, rows_per_pass = 512 / lpr, passes = 256 / rows_per_pass, ksteps = rowbytes / seg;
    const size_t tile_bytes = (size_t)256 * rowbytes, ntiles = total / tile_bytes;
    size_t tile = blockIdx.x;
    int k = 0;
    for (int it = 0; it < iters; ++it) {
        const char *base = src + (tile % ntiles) * tile_bytes + (size_t)(tid / lpr) * rowbytes + (size_t)k * seg + (tid % lpr) * 16;
        for (int p = 0; p < passes; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)p * rows_per_pass * rowbytes),
                                             (__attribute__((address_space(3))) void *)(lds + wave * 16384 + (p & 15) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        if (++k == ksteps) { k = 0; tile += gridDim.x; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const size_t big = (size_t)1 << 30;
    char *buf = nullptr;
    if (hipMalloc(&buf, big) != hipSuccess) return 1;
    hipMemset(buf, 1, big);
    float *sink = nullptr;
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ncu : {256, 32}) {
        for (int shared = 1; shared >= 0; --shared) {
            for (int mode = 0; mode < 3; ++mode) {
                const size_t per_wg = shared ? 0 : big / 256, window = shared ? ((size_t)2 << 20) : big / 256;
                auto run = [&](int n) {
                    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(ncu), dim3(512), 131072, 0, buf, per_wg, window, n, sink);
                    else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(ncu), dim3(512), 131072, 0, buf, per_wg, window, n, sink);
                    else hipLaunchKernelGGL(probe<2>, dim3(ncu), dim3(512), 131072, 0, buf, per_wg, window, n, sink);
                };
                hipFuncSetAttribute(reinterpret_cast<const void *>(probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
                hipFuncSetAttribute(reinterpret_cast<const void *>(probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
                hipFuncSetAttribute(reinterpret_cast<const void *>(probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
                run(50);
                hipEventRecord(e0, 0);
                run(iters);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)ncu * 512 * 16 * 16 * iters;
                printf("%3d workgroups  %-22s %-8s %8.1f us  %7.2f TB/s  = %6.1f GB/s per CU\n", ncu, shared ? "shared 2 MB (L2)" : "own 4 MB slice (HBM)",
                       mode == 0 ? "LDS-DMA" : (mode == 1 ? "VGPR" : "both"), ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / ncu);
                fflush(stdout);
            }
        }
    }
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe_rows), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rowbytes : {2048, 1024, 512}) {
        for (int seg : {128, 256, 512}) {
            if (seg > rowbytes) continue;
            const int n = 4000 * 128 / seg;
            hipLaunchKernelGGL(probe_rows, dim3(256), dim3(512), 131072, 0, buf, big, rowbytes, seg, 50);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe_rows, dim3(256), dim3(512), 131072, 0, buf, big, rowbytes, seg, n);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = 256.0 * 256 * seg * n;
            printf("GEMM rows: %4d-byte rows, %3d bytes per row and k-step  %8.1f us  %6.2f TB/s = %5.1f GB/s per CU\n", rowbytes, seg, ms * 1e3,
                   bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / 256);
            fflush(stdout);
        }
    }
    return 0;
}
