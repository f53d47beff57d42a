// conv_rds.hip -- persistent, weight-stationary ROW-DECOMPOSED 3x3 convolution for the fp32 path's narrow layers.
//
// Replaces, for large pixel counts, conv_igemm's row-decomposed tiles on HRNet-w40's 40- and 80-channel branches
// (/root/reference/src/models/backbones/hrnet.py:96-221: the BasicBlock convs of the two highest-resolution branches, 3x3, stride 1,
// padding 1, C -> C).  The row decomposition (conv_igemm.hip, RD) runs a 3x1 convolution as a GEMM
//     G[m][(s, n)] = sum_{r, c} in[(y + r - 1, x), c] * W[n][c][r][s]          K' = 3 C, N' = 3 C  (120 -> 128, 240 -> 256)
// and sums the three column groups of horizontally neighbouring pixels, out[m][n] = sum_s G[m + s - 1][(s, n)]: 1.5x fewer MFMAs than
// padding C = 40 to 64 output columns and 9 C = 360 to 384.  As per-tile workgroups those GEMMs have FOUR k-steps per 128 x 128 tile:
// a tile is a prologue, four steps and a staged epilogue (0.54 of the fp32 MFMA peak, 22.8 of HRNet-w40's 91 ms).
//
// MI355X mapping: conv_stream.hip's structure.
//   * ONE workgroup per CU for the launch (grid = 256) walking 64-pixel tiles (whole image rows: 64 % W == 0); the WEIGHTS LIVE IN
//     REGISTERS (a wave keeps the MFMA A-operand fragments of its 32 GEMM columns for the whole reduction: K' / 2 = 64 / 128 VGPRs);
//   * the pixel operand arrives by LDS-DMA in pieces of [64 pixels][32 reduction elements] (a ring of four, three in flight): every
//     16-byte unit of a piece has its own source address -- pixel (y + r - 1, x), channels 4 c .. 4 c + 3 of tap r -- computed per
//     thread once per launch (the tap and channel of its unit in each piece); rows past the image and the padding of K' read the
//     zero page;
//   * one static schedule per tile: NP piece steps {counted `s_waitcnt vmcnt(N)` . barrier . DMA piece + 3 . 16 MFMAs per 32-pixel
//     block} . accumulators -> LDS (G, [64][N' + 4] fp32) . barrier . row sum + bias + residual + ReLU from LDS, 16-byte stores.
//     The residual rows are plain loads and the stores plain stores, every thread issues the same number of each (masked ones
//     touch the zero / trash pages), so N is a compile-time constant;
//   * operand roles, reduction order (k ascending in the 32x32x2 MFMA's pairs) and the epilogue's summation order
//     ((((bias + G_0) + G_1) + G_2) + residual) are conv_igemm's row-decomposed path: the results are BIT-IDENTICAL to it
//     (tests/test_gpu_parity.py::test_rds_kernel_is_bit_identical), so the launcher may choose by size.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "kernels.h"

namespace hmv {

typedef float rf32x16 __attribute__((ext_vector_type(16)));
typedef float rf32x4 __attribute__((ext_vector_type(4)));

#define HMV_RGLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int N>
__device__ __forceinline__ void rds_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <typename F, int... I>
__device__ __forceinline__ void rds_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void rds_static_for(F &&f) {
    rds_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// A tile's vector-memory instructions per thread, in issue order: NP piece DMAs (one per piece step; the DMA of step j belongs to the
// piece D steps ahead), then EO epilogue instructions (residual loads, then stores).  "Piece (t, j) landed" == "at most the
// instructions issued after its DMA are still out".
constexpr int rds_after_piece(int NP, int D, int PA, int EO, int j) {
    const int T0 = 8, g = T0 * NP + j - D, tq = g / NP, jq = g - tq * NP;   // it went out at piece step (tq, jq)
    int n = 0;
    for (int s = tq * (NP + 1) + jq + 1; s < T0 * (NP + 1) + j; ++s) n += (s % (NP + 1)) < NP ? PA : EO;
    return n;
}
static_assert(rds_after_piece(4, 3, 1, 4, 0) == 2 + 4 && rds_after_piece(4, 3, 1, 4, 3) == 2 && rds_after_piece(4, 3, 2, 6, 1) == 4 + 6, "counted by hand");

// CH real channels (40 or 80); NB = 32-column blocks of the GEMM's N' = 3 CH (4 or 8) = waves along N; MW waves along the 64 pixels.
// NWV waves per workgroup: 8, or -- where four waves hold the whole N' (40 channels) -- 4, TWO workgroups per CU: they drift apart, and
// one's epilogue (accumulators -> LDS -> row sum -> stores, no MFMA) runs under the other's piece steps
template <int CH, bool HAS_RES, int NWV>
__global__ __launch_bounds__(64 * NWV, 2) void conv_rds_f32(const ConvParams p) {
    constexpr int KR = 3 * CH, NP = (KR + 31) / 32, NB = NP, NW = NB, MW = NWV / NW, TM = 2 / MW, NT = 64 * NWV;   // K' = N' = 3 CH
    constexpr int NSLOT = 4, D = NSLOT - 1;
    constexpr int C4 = CH / 4;                         // 16-byte groups per pixel and tap
    constexpr int UNITS = 64 * C4, OPT = (UNITS + NT - 1) / NT;   // output float4s per tile; per thread
    constexpr int PA = 64 * 8 / NT;                    // pixel DMA instructions per piece and thread
    constexpr int EO = (HAS_RES ? OPT : 0) + OPT;
    constexpr int LDG = 32 * NB + 4;                   // floats per row of the staged GEMM tile
    static_assert(CH % 4 == 0 && NW * MW == NWV && TM * MW == 2 && (NW == 4 || NW == 8) && PA >= 1, "waves over 64 pixels x N' columns");
    extern __shared__ __attribute__((aligned(16))) char rsm_[];
    float *sA = reinterpret_cast<float *>(rsm_);       // [NSLOT][64][32]
    float *sG = sA + NSLOT * 64 * 32;                  // [64][LDG]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int mw = wave / NW, nw = wave - mw * NW;
    const int ntiles = p.mtiles, nstreams = (int)gridDim.x, stream = (int)blockIdx.x;
    const int ntl = ntiles > stream ? (ntiles - stream + nstreams - 1) / nstreams : 0;
    if (ntl == 0) return;

    const float *zero32 = p.zero;
    float *trash = const_cast<float *>(p.zero) + 64 + 4 * lane;
    const float *Ain = reinterpret_cast<const float *>(p.in);
    const float *Rin = reinterpret_cast<const float *>(p.res);
    float *Out = reinterpret_cast<float *>(p.out);

    // ---- weights -> registers, once: GEMM column 32 nw + l31, piece j, group q: W'[col][32 j + 8 q + 4 kh .. + 3]
    rf32x4 wreg[NP * 4];
    {
        const float *wb = reinterpret_cast<const float *>(p.wgt) + (size_t)(32 * nw + l31) * p.ldw + 4 * kh;
#pragma unroll
        for (int Q = 0; Q < NP * 4; ++Q) wreg[Q] = *reinterpret_cast<const rf32x4 *>(wb + 8 * Q);
    }
    // the bias of this thread's output units (tile-invariant: a load in the tile loop would enter the counted queue)
    rf32x4 bvec[OPT];
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
        const int u = tid + NT * i, c4 = u < UNITS ? u % C4 : 0;
        bvec[i] = *reinterpret_cast<const rf32x4 *>(p.bias + 4 * c4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < OPT; ++i) asm volatile("" : "+v"(bvec[i]));

    // ---- pixel DMA role: thread -> row (tid >> 3) + (NT / 8) i of a piece, physical chunk tid & 7 holding logical chunk kqs; in piece j that is
    // 16-byte group g = 8 j + kqs of the pixel's reduction row: tap r = g / C4, channels 4 (g % C4) ..; groups past K' are padding
    const int arow = tid >> 3, kqs = (tid & 7) ^ ((tid >> 4) & 7);
    int tap[NP], coff[NP];   // tap - 1 (row displacement; 7: padding -> zero page), element offset of the group inside a pixel
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int g = 8 * j + kqs, r = g / C4;
        tap[j] = g < 3 * C4 ? r - 1 : 7;
        coff[j] = 4 * (g - r * C4);
    }
    const int HW = p.H * p.W;
    auto issue_A = [&](int G) {
        const int tt = G >= 0 ? G / NP : -1, jj = G >= 0 ? G - tt * NP : 0, slot = G & (NSLOT - 1);
        int tj = 7, cj = 0;
#pragma unroll
        for (int j = 0; j < NP; ++j)
            if (j == jj) { tj = tap[j]; cj = coff[j]; }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = (stream + tt * nstreams) * 64 + arow + (NT / 8) * i;
            const int n = m / HW, rem = m - n * HW, y = rem / p.W, yy = y + tj;
            const bool ok = tt >= 0 && tt < ntl && m < p.M && tj != 7 && (unsigned)yy < (unsigned)p.H;
            const float *src = ok ? Ain + ((size_t)m + (size_t)((long)tj * p.W)) * p.lda + cj : zero32;
            asm volatile("" : "+v"(src));   // ONE DMA instruction per schedule entry
            HMV_RGLDS16(src, sA + ((slot * 64 + (NT / 8) * i + wave * 8) * 32));
        }
    };

    // ---- prologue: the schedule of the tile "before the first" (real DMAs where they belong to tile 0, dummies otherwise)
    rds_static_for<NP>([&](auto jc) { issue_A(-NP + decltype(jc)::value + D); });
#pragma unroll
    for (int i = 0; i < EO; ++i)
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(trash), "v"(rf32x4{0.f, 0.f, 0.f, 0.f}) : "memory");

    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    const int fsw = (l31 >> 1) & 7;
    rf32x16 acc[TM];
    for (int tt = 0; tt < ntl; ++tt) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
        rds_static_for<NP>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            rds_wait_vm<rds_after_piece(NP, D, PA, EO, j)>();                        // my share of piece (tt, j) has landed
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // everyone's has; everyone is done with piece g - 1 (and with G)
            issue_A(tt * NP + j + D);
            const float *pa = sA + ((((tt * NP + j) & (NSLOT - 1)) * 64 + mw * TM * 32 + l31) * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rf32x4 px[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a) px[a] = *reinterpret_cast<const rf32x4 *>(pa + a * 32 * 32 + (((2 * q + kh) ^ fsw) * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < TM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[4 * j + q][e], px[a][e], acc[a], 0, 0, 0);
            }
        });
        // ---- accumulators -> G: register group 4 g .. 4 g + 3 of block a = GEMM columns 32 nw + 8 g + 4 kh .. of pixel (mw TM + a) 32 + l31
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<rf32x4 *>(sG + ((mw * TM + a) * 32 + l31) * LDG + 32 * nw + 8 * g + 4 * kh) =
                    rf32x4{acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- out[m][n] = relu((((bias + G[m - 1][n]) + G[m][C + n]) + G[m + 1][2 C + n]) + residual): one float4 per unit
        const int m0 = (stream + tt * nstreams) * 64;
        rf32x4 rv[OPT];
        if constexpr (HAS_RES) {
#pragma unroll
            for (int i = 0; i < OPT; ++i) {
                const int u = tid + NT * i, ml = u / C4, c4 = u - ml * C4, m = m0 + ml;
                const float *rp = (u < UNITS && m < p.M) ? Rin + (size_t)m * p.ldr + 4 * c4 : zero32;
                asm volatile("" : "+v"(rp));
                rv[i] = *reinterpret_cast<const rf32x4 *>(rp);
            }
        }
#pragma unroll
        for (int i = 0; i < OPT; ++i) {
            const int u = tid + NT * i, ml = u < UNITS ? u / C4 : 0, c4 = u < UNITS ? u - ml * C4 : 0, m = m0 + ml;
            const int x = m % p.W;
            rf32x4 t = bvec[i];
            if (x > 0) t += *reinterpret_cast<const rf32x4 *>(sG + (ml - 1) * LDG + 4 * c4);
            t += *reinterpret_cast<const rf32x4 *>(sG + ml * LDG + CH + 4 * c4);
            if (x + 1 < p.W) t += *reinterpret_cast<const rf32x4 *>(sG + (ml + 1) * LDG + 2 * CH + 4 * c4);
            if constexpr (HAS_RES) t += rv[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = fmaxf(t[k], lo);
            float *dst = (u < UNITS && m < p.M) ? Out + (size_t)m * p.ldc + 4 * c4 : trash;
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(t) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// ====================================================================== host side
static int g_rds_mode = -1;   // -1: the launcher's rule (HMV_NO_RDS=1 disables it); 0 never; 1 whenever supported (op-level tests)
void conv_rds_set_mode(int mode) { g_rds_mode = mode; }

bool conv_rds_supported(const ConvParams &p) {
    static int off = -1;   // development knob: HMV_NO_RDS=1 keeps these layers on conv_igemm's row-decomposed tiles (A/B runs)
    if (off < 0) off = HMV_DEV_ENV("HMV_NO_RDS") ? 1 : 0;
    if (g_rds_mode == 0 || (g_rds_mode < 0 && off)) return false;
    if (p.in_f16 || p.out_f16 || p.res_f16 || (p.rd_cout != 40 && p.rd_cout != 80)) return false;
    if (p.R != 3 || p.S != 1 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 0 || p.Cout != 3 * p.rd_cout || p.Cin != p.rd_cout) return false;
    if (p.Ho != p.H || p.Wo != p.W || p.W <= 0 || 64 % p.W != 0 || (p.H * p.W) % 64 != 0) return false;
    if (p.up || p.in2 || p.ksl > 1 || p.phases > 1 || p.cwrap || p.x3_plane || p.res_split || p.out_split || p.acc_shift || p.scatter || p.rg_out) return false;
    if (p.fill && p.ldc != p.rd_cout) return false;   // (pad columns to clear: conv_igemm's epilogue does that)
    if (p.act != ACT_NONE && p.act != ACT_RELU) return false;
    const int kp = (3 * p.rd_cout + 31) / 32 * 32;
    if ((p.lda ? p.lda : p.Cin) != p.Cin || (p.ldw ? p.ldw : p.Kpad) < kp || ((p.ldw ? p.ldw : p.Kpad) & 3) || (p.ldc & 3) || (p.res && (p.ldr & 3))) return false;
    if ((long long)p.M * p.Cin >= (1ll << 31)) return false;
    if (g_rds_mode > 0) return true;
    return (long long)(p.M + 63) / 64 >= 4 * 256;   // at least four tiles per workgroup
}

template <int CH, bool HAS_RES, int NWV>
static hipError_t launch_rds_one(ConvParams p, hipStream_t s) {
    constexpr int NB = (3 * CH + 31) / 32;
    constexpr size_t lds = (size_t)4 * 64 * 32 * 4 + (size_t)64 * (32 * NB + 4) * 4;
    static_assert(lds * (8 / NWV) <= 160 * 1024, "LDS budget");
    static bool configured[64] = {};
    auto kern = conv_rds_f32<CH, HAS_RES, NWV>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    p.mtiles = (p.M + 63) / 64;
    p.ntiles = 1;
    const int grid = 256 * (8 / NWV);
    hipLaunchKernelGGL(kern, dim3(p.mtiles < grid ? p.mtiles : grid), dim3(64 * NWV), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_rds(const ConvParams &p, hipStream_t s, const char **name) {
    if (p.rd_cout == 40) {
        if (name) *name = p.res ? "conv_rds_f32<3x3,40->40,res>" : "conv_rds_f32<3x3,40->40>";
        return p.res ? launch_rds_one<40, true, 4>(p, s) : launch_rds_one<40, false, 4>(p, s);
    }
    if (name) *name = p.res ? "conv_rds_f32<3x3,80->80,res>" : "conv_rds_f32<3x3,80->80>";
    return p.res ? launch_rds_one<80, true, 8>(p, s) : launch_rds_one<80, false, 8>(p, s);
}

}  // namespace hmv
