// copybw.hip -- what does the chip sustain when a kernel READS and WRITES HBM at once, by read : write mix?
// DESIGN.md section 8 claims the fp16 conv_stream families (4.75-4.95 TB/s of algorithmic bytes, 5 : 4 read : write for the
// residual-bearing 1x1 convs) sit at "the rate the chip copies memory"; r03's l2bw probe measured reads only (6.1-6.5 TB/s).  This
// probe pins the ceiling: NR read streams and NW write streams of 16 bytes per lane, every stream far larger than the 256 MiB
// Infinity Cache, persistent workgroups that each walk their own contiguous slice (as conv_stream does) or a grid-stride walk.
// Development tool; build: hipcc -O3 --offload-arch=gfx950 tools/probe/copybw.hip -o tools/probe/copybw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

// one "row" = NR loads + NW stores per lane; UNR rows in flight per lane
template <int NR, int NW, int UNR, bool NT>
__global__ __launch_bounds__(256) void mix(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t vec_per_stream, size_t vec_per_wg) {
    const size_t base = (size_t)blockIdx.x * vec_per_wg;
    f4 keep = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = threadIdx.x; i + (UNR - 1) * 256 < vec_per_wg; i += (size_t)UNR * 256) {
        f4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            v[u] = f4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const f4 *p = src + (size_t)r * vec_per_stream + base + i + (size_t)u * 256;
                f4 t = NT ? __builtin_nontemporal_load(p) : *p;
                v[u] += t;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (NW == 0) keep += v[u];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                f4 *q = dst + (size_t)w * vec_per_stream + base + i + (size_t)u * 256;
                if (NT) __builtin_nontemporal_store(v[u], q); else *q = v[u];
            }
        }
    }
    if (NW == 0 && keep[0] + keep[1] + keep[2] + keep[3] == 12345.678f) dst[0] = keep;
}

template <int NR, int NW, bool NT>
static void run(const f4 *src, f4 *dst, size_t vec_per_stream, int wg_per_cu, hipEvent_t e0, hipEvent_t e1) {
    const int grid = 256 * wg_per_cu;
    const size_t vec_per_wg = vec_per_stream / grid / 1024 * 1024;
    auto go = [&]() { hipLaunchKernelGGL((mix<NR, NW, 4, NT>), dim3(grid), dim3(256), 0, 0, src, dst, vec_per_stream, vec_per_wg); };
    go();
    float best = 1e30f, sum = 0.f;
    const int reps = 8;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        go();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        sum += ms;
    }
    const double bytes = (double)(NR + NW) * vec_per_wg * grid * 16.0;
    printf("read : write = %d : %d  %s  %d workgroups per CU   %8.1f us (min %8.1f)   %5.2f TB/s total (best %5.2f)   read %5.2f  write %5.2f TB/s\n", NR, NW,
           NT ? "nontemporal" : "plain      ", wg_per_cu, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12,
           bytes * NR / (NR + NW) / (sum / reps * 1e-3) / 1e12, bytes * NW / (NR + NW) / (sum / reps * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    const size_t stream_bytes = (size_t)512 << 20, vec_per_stream = stream_bytes / 16;
    f4 *src = nullptr, *dst = nullptr;
    if (hipMalloc(&src, 5 * stream_bytes) != hipSuccess || hipMalloc(&dst, 4 * stream_bytes) != hipSuccess) return 1;
    hipMemset(src, 0x3c, 5 * stream_bytes);
    hipMemset(dst, 0, 4 * stream_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wpc : {2, 4, 8}) {
        run<1, 0, false>(src, dst, vec_per_stream, wpc, e0, e1);
        run<0, 1, false>(src, dst, vec_per_stream, wpc, e0, e1);
        run<1, 1, false>(src, dst, vec_per_stream, wpc, e0, e1);
        run<5, 4, false>(src, dst, vec_per_stream, wpc, e0, e1);
        run<3, 1, false>(src, dst, vec_per_stream, wpc, e0, e1);
        run<2, 1, false>(src, dst, vec_per_stream, wpc, e0, e1);
        run<1, 1, true>(src, dst, vec_per_stream, wpc, e0, e1);
        run<5, 4, true>(src, dst, vec_per_stream, wpc, e0, e1);
    }
    return 0;
}
