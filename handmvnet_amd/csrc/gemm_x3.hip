// gemm_x3.hip -- token GEMMs on (hi, lo) fp16 pairs: the q / k / v projections of the fusion transformer in the fp16-kernel modes.
//
// /root/reference/src/models/layers.py:213-215 (`to_q / to_k / to_v`, Linear(d -> 1024, no bias)): in the fp16 and f32x3 modes the
// engine evaluates them fp32-equivalently on the fp16 matrix cores -- every fp32 token value travels as hi = fp16(v), lo = fp16(v - hi),
// every product as hi*hi + lo*hi + hi*lo with fp32 accumulation.  Until round 4 they ran on conv_igemm's fused split
// loop: two LDS stages, ONE DMA in flight, a drained `vmcnt(0)` per k-step.  At cfg-3 (5 376 token rows x 3 072 columns, K = 544) every
// tile choice from 128 x 128 to 256 x 256 took 66-76 us for 21 us of MFMA time (gpurun_out/r04/pl_tile*.md): the launch is bound by one
// exposed DMA latency per k-step.  This kernel is conv_m16.hip's loop for that operand format:
//   * 64 x 64 tiles on four waves (small launches: two workgroups per CU) and 128 x 128 tiles on EIGHT (4 x 2: a wave owns 32 x 64);
//     a k-step = 32 channels = one 128-byte LDS row [32 hi | 32 lo] per token / weight row; a ring of FOUR stages, three tiles in
//     flight, one counted `s_waitcnt vmcnt(N)` + barrier per step.  (First version: 128 x 128 on four waves, one wave per SIMD -- 84 us
//     where the old loop took 72: a wave that issues eight LDS-DMAs (~100 cycles each beside MFMAs) AND 48 MFMAs per step serialises
//     them in its own instruction stream.  With two waves per SIMD one wave's DMA issue runs under the other's MFMAs.)
//   * v_mfma_f32_16x16x32_f16 (the shape the chip clocks higher on, profiles/r04_probe_mfma_shape.txt): per 16 x 16 block and step
//     THREE MFMAs -- hi*hi, lo*hi, hi*lo, in that order -- on four 16-byte fragments;
//   * weights as Loader::linear_x3 packs them ([Cout][per step: 32 hi, 32 lo], scaled by 2^acc_shift); fp32 output rows.
// It takes these layers at EVERY size (gemm_x3_rule is a rule on the layer alone), so a sample's bits do not depend on its batch.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace hmv {

typedef float xf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 xf16x8 __attribute__((ext_vector_type(8)));

#define HMV_XGLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

constexpr int X3_NS = 4;   // LDS stages

// LDS rows of 128 bytes (eight 16-byte chunks: 0-3 = hi k-groups, 4-7 = lo); chunk' = chunk ^ ((row >> 1) & 7).  A fragment read's lane
// (l15, kg) takes chunk 4 g + kg of row 16 blk + l15: per row parity the eight lanes of a 16-lane service group read
// c ^ {0, 1, 6, 7} (rows 0-3, 12-15) and c ^ 1 ^ {2, 3, 4, 5} (rows 4-11): eight distinct chunks, conflict-free.
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_x3_f16(const ConvParams p) {
    constexpr int NT = 64 * WGM * WGN, RPP = NT / 8;   // threads; tile rows per DMA pass
    constexpr int WM = BM / WGM, WN = BN / WGN, PB = WM / 16, CB = WN / 16, AP = BM / RPP, BP = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0 && WM % 16 == 0 && WN % 32 == 0, "tile shape");
    static_assert(CB % 2 == 0, "a lane's channel blocks pair up into 8 consecutive columns");
    extern __shared__ __attribute__((aligned(16))) char xsm[];   // [X3_NS][BM + BN][128 bytes]
    constexpr int STAGE = (BM + BN) * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4, wm = wave / WGN, wn = wave % WGN;

    int mt, nt;
    if (p.ngroup > 0) {
        // weights larger than an XCD's 4 MB L2 (q / k / v at cfg-3: 3 072 x 1 088 halfs = 6.7 MB): with row-tile-major placement every XCD
        // re-streams ALL of them from the Infinity Cache once per row tile -- 281 of the launch's 562 MB of operand traffic, and at the
        // ~34 GB/s per CU the Infinity Cache serves (MI355X_MICROARCH.md, "Indexed rows") that IS the 70-85 us every tile size and loop
        // structure measured.  Here XCD x owns N-tiles [x g, x g + g): its weight slice (0.84 MB) stays in its L2, the g workgroups of a
        // row tile are consecutive on it (the token tile is an L2 hit for all but the first), and the token rows stream through once per XCD.
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;   // (blocks are dealt round-robin over the XCDs: speed only)
        mt = loc / p.ngroup;
        nt = xcd * p.ngroup + (loc - mt * p.ngroup);
    } else {   // the N-tiles of a row tile are consecutive workgroups of one XCD (bijective for any grid size)
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r = nblk & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        mt = lid / p.ntiles;
        nt = lid - mt * p.ntiles;
    }
    const _Float16 *zero16 = reinterpret_cast<const _Float16 *>(p.zero);
    const int nk = p.x3_plane >> 5;   // 32-channel steps

    // ---- DMA roles: thread -> row tid >> 3 of each RPP-row pass, physical chunk tid & 7 holding logical chunk (tid & 7) ^ key(row)
    const int lrow = tid >> 3, lc = (tid & 7) ^ ((lrow >> 1) & 7);
    const _Float16 *aptr[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = mt * BM + i * RPP + lrow;
        // logical chunks 0-3: the hi plane's channels 8 lc .. of the step, 4-7: the lo plane's
        aptr[i] = m < p.M ? reinterpret_cast<const _Float16 *>(p.in) + (size_t)m * p.lda + (lc < 4 ? 8 * lc : p.x3_plane + 8 * (lc - 4)) : nullptr;
    }
    // weight LDS row R = 32 u + 16 e + rho holds output column 32 u + 8 (rho >> 2) + 4 e + (rho & 3) (conv_m16.hip's permutation: a lane's
    // block pair (2 t, 2 t + 1) then holds 8 consecutive columns)
    const _Float16 *wptr[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int R = i * RPP + lrow, u = R >> 5, e = (R >> 4) & 1, rho = R & 15;
        const int col = 32 * u + 8 * (rho >> 2) + 4 * e + (rho & 3);
        wptr[i] = reinterpret_cast<const _Float16 *>(p.wgt) + (size_t)(nt * BN + col) * p.ldw + 8 * lc;
    }
    int ck = 0;   // DMA cursor (wave-uniform): step index
    auto dma = [&]() {   // the tile of step ck into stage ck & 3, then advance (past the end: dummies from the zero page)
        char *st = xsm + (ck & (X3_NS - 1)) * STAGE;
        const bool live = ck < nk;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const _Float16 *src = (live && aptr[i]) ? aptr[i] + 32 * ck : zero16;
            asm volatile("" : "+v"(src));
            HMV_XGLDS16(src, st + (i * RPP + wave * 8) * 128);
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const _Float16 *src = live ? wptr[i] + 64 * ck : zero16;
            asm volatile("" : "+v"(src));
            HMV_XGLDS16(src, st + (BM + i * RPP + wave * 8) * 128);
        }
        ++ck;
    };

    // ---- accumulators start at the bias (scaled like the weights): acc[a][cb][r], cb = 2 t + e, = column 32 t + 8 kg + 4 e + r of row 16 a + l15
    xf32x4 acc[PB][CB];
    {
        const float binit = 1.f / p.acc_scale;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const xf32x4 bq = *reinterpret_cast<const xf32x4 *>(p.bias + nt * BN + wn * WN + 32 * (cb >> 1) + 8 * kg + 4 * (cb & 1));
#pragma unroll
            for (int a = 0; a < PB; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][cb][r] = bq[r] * binit;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int a = 0; a < PB; ++a)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) asm volatile("" : "+v"(acc[a][cb]));   // the bias is in the accumulators BEFORE the first DMA

    const int fkey = (l15 >> 1) & 7;   // (rows + 16 blk: (row >> 1) & 7 unchanged)
    const int foff_a = (wm * WM + l15) * 128, foff_w = (BM + wn * WN + l15) * 128;
    const int ch_hi = (kg ^ fkey) * 16, ch_lo = ((4 + kg) ^ fkey) * 16;
    xf16x8 ah0[PB], al0[PB], wh0[CB], wl0[CB], ah1[PB], al1[PB], wh1[CB], wl1[CB];
#define X3_READ(AH, AL, WH, WL, kt_)                                                                        \
    {                                                                                                       \
        const char *st_ = xsm + ((kt_) & (X3_NS - 1)) * STAGE;                                              \
        _Pragma("unroll") for (int cb_ = 0; cb_ < CB; ++cb_) {                                              \
            WH[cb_] = *reinterpret_cast<const xf16x8 *>(st_ + foff_w + cb_ * 2048 + ch_hi);                 \
            WL[cb_] = *reinterpret_cast<const xf16x8 *>(st_ + foff_w + cb_ * 2048 + ch_lo);                 \
        }                                                                                                   \
        _Pragma("unroll") for (int a_ = 0; a_ < PB; ++a_) {                                                 \
            AH[a_] = *reinterpret_cast<const xf16x8 *>(st_ + foff_a + a_ * 2048 + ch_hi);                   \
            AL[a_] = *reinterpret_cast<const xf16x8 *>(st_ + foff_a + a_ * 2048 + ch_lo);                   \
        }                                                                                                   \
    }
    // hi*hi, then lo*hi (token lo x weight hi), then hi*lo: every accumulator sees the same order at every tile size
#define X3_MFMA(AH, AL, WH, WL)                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    _Pragma("unroll") for (int a_ = 0; a_ < PB; ++a_)                                                       \
        _Pragma("unroll") for (int cb_ = 0; cb_ < CB; ++cb_)                                                \
            acc[a_][cb_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WH[cb_], AH[a_], acc[a_][cb_], 0, 0, 0);  \
    _Pragma("unroll") for (int a_ = 0; a_ < PB; ++a_)                                                       \
        _Pragma("unroll") for (int cb_ = 0; cb_ < CB; ++cb_)                                                \
            acc[a_][cb_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WH[cb_], AL[a_], acc[a_][cb_], 0, 0, 0);  \
    _Pragma("unroll") for (int a_ = 0; a_ < PB; ++a_)                                                       \
        _Pragma("unroll") for (int cb_ = 0; cb_ < CB; ++cb_)                                                \
            acc[a_][cb_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WL[cb_], AH[a_], acc[a_][cb_], 0, 0, 0);  \
    __builtin_amdgcn_s_setprio(0);

    // ---- prologue: tiles 0, 1, 2 go out; tile 0 must have landed
    dma();
    dma();
    dma();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AP + BP)) : "memory");
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    X3_READ(ah0, al0, wh0, wl0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        // step kt (fragment set 0): tile kt + 1 has landed (tile kt + 2 may fly); every wave has read tile kt - 1 (a step ago)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AP + BP) : "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        dma();   // tile kt + 3 into the stage of tile kt - 1
        X3_READ(ah1, al1, wh1, wl1, kt + 1);
        X3_MFMA(ah0, al0, wh0, wl0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 >= nk) break;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AP + BP) : "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        dma();
        X3_READ(ah0, al0, wh0, wl0, kt + 2);
        X3_MFMA(ah1, al1, wh1, wl1);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the trailing dummies
#undef X3_READ
#undef X3_MFMA

    // ---- epilogue: blocks (2 t, 2 t + 1) of row block a = columns 32 t + 8 kg + 0 .. 7 of row 16 a + l15: two 16-byte fp32 stores
    const int nb0 = nt * BN + wn * WN;
    const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 3) & ~3);
#pragma unroll
    for (int a = 0; a < PB; ++a) {
        const int m = mt * BM + wm * WM + 16 * a + l15;
        float *orow = reinterpret_cast<float *>(p.out) + (size_t)m * p.ldc;
#pragma unroll
        for (int t = 0; t < CB / 2; ++t)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int col = nb0 + 32 * t + 8 * kg + 4 * e;
                xf32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[a][2 * t + e][r] * p.acc_scale;
                if (m < p.M && col < cend) *reinterpret_cast<xf32x4 *>(orow + col) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// gemm_x3k16_f16: the q / k / v projections at LARGE row counts (layers.py:213-215; cfg-3: 5 376 rows x 3 072 columns, K = 544 pairs).
// conv_igemm's fused split loop runs them on 128 x 128 tiles, two workgroups per CU, ONE DMA in flight per workgroup: ~36 us per tile for
// 5.9 us of MFMA time (17 k-steps of one exposed operand round trip each), 72 us per launch whatever the tile size.
// Here: 256 x 256 tiles (half the operand bytes through the CUs: 281 instead of 562 MB), eight waves (2 x 4, a wave owns 128 x 64),
// k-steps of SIXTEEN channels -- 64-byte LDS rows [16 hi | 16 lo], 32 KB per stage -- on a ring of four stages with three tiles in flight
// and one counted `s_waitcnt vmcnt(N)` + barrier per step.  Per 32 x 32 block and step three v_mfma_f32_32x32x16_f16 -- hi*hi, lo*hi,
// hi*lo -- which is EXACTLY the fused split loop's sequence per 16-channel group (weights as the A operand, bias / 2^shift as the
// accumulators' initial value, (acc * 2^-shift + 0) in the epilogue): the results are bit-identical to that loop's
// (tests/test_gpu_parity.py::test_split_pair_gemm_large_tiles_are_bit_identical), so the launcher may pick by size.
typedef float xf32x16 __attribute__((ext_vector_type(16)));
constexpr int XK_NS = 4, XK_STAGE = 512 * 64, XK_LDS = XK_NS * XK_STAGE;   // 131 072

__global__ __launch_bounds__(512, 2) void gemm_x3k16_f16(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char ksm[];   // [XK_NS][256 token rows + 256 weight rows][64 bytes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5, wm = wave >> 2, wn = wave & 3;
    int mt, nt;
    {   // the N-tiles of a row tile are consecutive workgroups of one XCD (bijective for any grid size)
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r = nblk & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        mt = lid / p.ntiles;
        nt = lid - mt * p.ntiles;
    }
    const _Float16 *zero16 = reinterpret_cast<const _Float16 *>(p.zero);
    const int nk = p.x3_plane >> 4;   // 16-channel steps

    // ---- DMA roles: thread -> row tid >> 2 of each 128-row pass, physical chunk tid & 3 holding logical chunk (tid & 3) ^ ((row >> 2) & 3):
    // logical chunks 0, 1 = the step's 16 hi channels, 2, 3 = its 16 lo channels
    const int lrow = tid >> 2, lc = (tid & 3) ^ ((lrow >> 2) & 3);
    const _Float16 *aptr[2], *wptr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mt * 256 + i * 128 + lrow;
        aptr[i] = m < p.M ? reinterpret_cast<const _Float16 *>(p.in) + (size_t)m * p.lda + (lc < 2 ? 8 * lc : p.x3_plane + 8 * (lc - 2)) : nullptr;
        // packed weights: per 32-channel step 32 hi values then 32 lo values
        wptr[i] = reinterpret_cast<const _Float16 *>(p.wgt) + (size_t)(nt * 256 + i * 128 + lrow) * p.ldw + (lc < 2 ? 8 * lc : 32 + 8 * (lc - 2));
    }
    int ck = 0;
    auto dma = [&]() {   // the tile of step ck into stage ck & 3, then advance (past the end: dummies from the zero page)
        char *st = ksm + (ck & (XK_NS - 1)) * XK_STAGE;
        const bool live = ck < nk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const _Float16 *src = (live && aptr[i]) ? aptr[i] + 16 * ck : zero16;
            asm volatile("" : "+v"(src));
            HMV_XGLDS16(src, st + (i * 128 + wave * 16) * 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const _Float16 *src = live ? wptr[i] + 64 * (ck >> 1) + 16 * (ck & 1) : zero16;
            asm volatile("" : "+v"(src));
            HMV_XGLDS16(src, st + (256 + i * 128 + wave * 16) * 64);
        }
        ++ck;
    };

    // ---- accumulators start at the bias: register 4 q + t of block (a, b) = column 32 b + 8 q + 4 kh + t of row 32 a + l31 (conv_igemm's
    // transposed-output map for 32-bit rows)
    xf32x16 acc[4][2];
    {
        const float binit = 1.f / p.acc_scale;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float *bp = p.bias + nt * 256 + wn * 64 + 32 * b;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const xf32x4 bq = *reinterpret_cast<const xf32x4 *>(bp + 8 * q + 4 * kh);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[a][b][4 * q + t] = bq[t] * binit;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(acc[a][b]));

    const int fkey = (l31 >> 2) & 3;
    const int foff_a = (wm * 128 + l31) * 64, foff_w = (256 + wn * 64 + l31) * 64;
    const int ch_hi = (kh ^ fkey) * 16, ch_lo = ((2 + kh) ^ fkey) * 16;
    xf16x8 ah0[4], al0[4], wh0[2], wl0[2], ah1[4], al1[4], wh1[2], wl1[2];
#define XK_READ(AH, AL, WH, WL, kt_)                                                                        \
    {                                                                                                       \
        const char *st_ = ksm + ((kt_) & (XK_NS - 1)) * XK_STAGE;                                           \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_) {                                                  \
            WH[b_] = *reinterpret_cast<const xf16x8 *>(st_ + foff_w + b_ * 2048 + ch_hi);                   \
            WL[b_] = *reinterpret_cast<const xf16x8 *>(st_ + foff_w + b_ * 2048 + ch_lo);                   \
        }                                                                                                   \
        _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_) {                                                  \
            AH[a_] = *reinterpret_cast<const xf16x8 *>(st_ + foff_a + a_ * 2048 + ch_hi);                   \
            AL[a_] = *reinterpret_cast<const xf16x8 *>(st_ + foff_a + a_ * 2048 + ch_lo);                   \
        }                                                                                                   \
    }
#define XK_MFMA(AH, AL, WH, WL)                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_)                                                        \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                                    \
            acc[a_][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WH[b_], AH[a_], acc[a_][b_], 0, 0, 0);     \
    _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_)                                                        \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                                    \
            acc[a_][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WH[b_], AL[a_], acc[a_][b_], 0, 0, 0);     \
    _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_)                                                        \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                                    \
            acc[a_][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WL[b_], AH[a_], acc[a_][b_], 0, 0, 0);     \
    __builtin_amdgcn_s_setprio(0);

    dma();
    dma();
    dma();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile 0 landed (two tiles of 4 DMAs may fly)
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    XK_READ(ah0, al0, wh0, wl0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");   // tile kt + 1 landed; this wave's reads of tile kt - 1 are done
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        dma();
        XK_READ(ah1, al1, wh1, wl1, kt + 1);
        XK_MFMA(ah0, al0, wh0, wl0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 >= nk) break;
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        dma();
        XK_READ(ah0, al0, wh0, wl0, kt + 2);
        XK_MFMA(ah1, al1, wh1, wl1);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef XK_READ
#undef XK_MFMA

    // ---- epilogue (conv_igemm's register path for 32-bit rows without a residual: acc * 2^-shift + 0, no activation)
    const int nb0 = nt * 256 + wn * 64;
    if (p.out_split) {
        // Rows of (hi, lo) fp16 pairs [hi ldc / 2 | lo ldc / 2] (round 4: the q / k / v projections feed attention_x3_kernel directly); conv_igemm's
        // arithmetic per value: clamp, hi = fp16(c), lo = fp16(c - hi).  A lane is a ROW here and holds columns 8 q + 4 kh .. + 3 of every
        // 8-column group q: the two lanes of a row (kh = 0 / 1) swap halves so that each ends up with whole groups -- kh = 0 the even q,
        // kh = 1 the odd ones -- and stores 16 bytes per plane (8-byte stores were measured: 78 instead of 67 us per projection).
        typedef _Float16 xf16x8 __attribute__((ext_vector_type(8)));
        const int rowc = p.ldc >> 1;
        const int cend = (p.fill || p.Cout + 7 >= rowc) ? rowc : ((p.Cout + 7) & ~7);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int m = mt * 256 + wm * 128 + 32 * a + l31;
            _Float16 *prow = reinterpret_cast<_Float16 *>(p.out) + (size_t)m * p.ldc;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    float own[4], got[4];   // this lane's half of the group it keeps; the other half, from the row's other lane
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float ve = fmaxf(acc[a][b][4 * (2 * qp) + t] * p.acc_scale + 0.f, -INFINITY);        // group 2 qp: columns + 4 kh
                        const float vo = fmaxf(acc[a][b][4 * (2 * qp + 1) + t] * p.acc_scale + 0.f, -INFINITY);    // group 2 qp + 1
                        own[t] = kh ? vo : ve;
                        got[t] = __shfl_xor(kh ? ve : vo, 32, 64);
                    }
                    const int col = nb0 + 32 * b + 8 * (2 * qp + kh);
                    xf16x8 hv, lv;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        // kh = 0: columns 0 .. 3 are its own, 4 .. 7 came from the kh = 1 lane; kh = 1: the other way round
                        const float x = (t < 4) == (kh == 0) ? own[t & 3] : got[t & 3];
                        const float c = fminf(fmaxf(x, -65504.f), 65504.f);
                        hv[t] = (_Float16)c;
                        lv[t] = (_Float16)(c - (float)hv[t]);
                    }
                    if (m < p.M && col < cend) {
                        *reinterpret_cast<xf16x8 *>(prow + col) = hv;
                        *reinterpret_cast<xf16x8 *>(prow + col + rowc) = lv;
                    }
                }
        }
        return;
    }
    const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 3) & ~3);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int m = mt * 256 + wm * 128 + 32 * a + l31;
        float *orow = reinterpret_cast<float *>(p.out) + (size_t)m * p.ldc;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = nb0 + 32 * b + 8 * q + 4 * kh;
                xf32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = fmaxf(acc[a][b][4 * q + t] * p.acc_scale + 0.f, -INFINITY);
                if (m < p.M && col < cend) *reinterpret_cast<xf32x4 *>(orow + col) = v;
            }
    }
}

// the launches it takes: the split-pair GEMM rule below with a SHORT reduction (the long ones run gemm_x3_f16), whole 256-column tiles,
// and enough of them to fill most of the chip -- a size rule, legitimate because the bits equal the fused split loop's
static int g_x3k16_mode = -1;   // -1 the size rule, 0 never, 1 whenever the shape allows (op-level tests)
void gemm_x3k16_set_mode(int mode) { g_x3k16_mode = mode; }
bool gemm_x3k16_ok(const ConvParams &p) {
    if (g_x3k16_mode == 0) return false;
    if (!p.in_f16 || (p.out_f16 && !p.out_split) || (p.out_split && (p.ldc & 15)) || p.res || p.in2 || p.up || p.ksl > 1 || p.phases > 1 || p.cwrap || !p.x3_plane || p.rd_cout ||
        p.scatter || p.rg_out || p.nx_wgt || p.pool || p.tall || p.act != ACT_NONE || p.fill)
        return false;
    const int ldw = p.ldw ? p.ldw : p.Kpad;
    if (!(p.R == 1 && p.S == 1 && p.stride == 1 && !p.pad_h && !p.pad_w && p.x3_plane % 32 == 0 && p.x3_plane < 1024 && p.Kpad == 2 * p.x3_plane &&
          p.lda >= 2 * p.x3_plane && !(p.lda & 7) && !(ldw & 7) && !(p.ldc & 3) && p.Ho == p.H && p.Wo == p.W && p.Cout % 256 == 0))
        return false;
    if (g_x3k16_mode > 0) return true;
    return (long long)((p.M + 255) / 256) * (p.Cout / 256) >= 200;
}
hipError_t launch_gemm_x3k16(ConvParams p, hipStream_t s, const char **name) {
    if (!gemm_x3k16_ok(p)) return hipErrorInvalidValue;
    if (!p.ldw) p.ldw = p.Kpad;
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_x3k16_f16), hipFuncAttributeMaxDynamicSharedMemorySize, XK_LDS);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    p.mtiles = (p.M + 255) / 256;
    p.ntiles = p.Cout / 256;
    if (name) *name = "gemm_x3k16_f16<256x256>";
    hipLaunchKernelGGL(gemm_x3k16_f16, dim3(p.mtiles * p.ntiles), dim3(512), XK_LDS, s, p);
    return hipGetLastError();
}

// ====================================================================== host side
// The layers it takes: plain GEMMs over split token rows with fp32 output rows (Loader::linear_x3's packing), no epilogue beyond the
// bias.  A rule on the layer alone -- never on the number of rows.
bool gemm_x3_rule(const ConvParams &p) {
    if (!p.in_f16 || p.out_f16 || p.out_split || p.res || p.in2 || p.up || p.ksl > 1 || p.phases > 1 || p.cwrap || !p.x3_plane || p.rd_cout ||
        p.scatter || p.rg_out || p.nx_wgt || p.pool || p.tall || p.act != ACT_NONE)
        return false;
    const int ldw = p.ldw ? p.ldw : p.Kpad;
    // Long reductions only (to_out: 1 024 channels; 38 us where four fp32 split-K slices took 78 at cfg-3).  The q / k / v projections
    // (K = d <= 576) stay on conv_igemm's fused split loop: measured on one box (gpurun_out/r04/pl_new1.md vs pl_old1.md) this kernel takes
    // 77-84 us for them where that loop takes 72 -- at 8-17 k-steps its three-tile prologue and trailing dummies cost what the deeper
    // ring saves -- and 3 us more per launch at cfg-2's 672 rows.
    if (p.x3_plane < 1024) return false;
    return p.R == 1 && p.S == 1 && p.stride == 1 && !p.pad_h && !p.pad_w && p.x3_plane % 32 == 0 && p.Kpad == 2 * p.x3_plane && p.lda >= 2 * p.x3_plane &&
           !(p.lda & 7) && !(ldw & 7) && !(p.ldc & 3) && p.Ho == p.H && p.Wo == p.W;
}

template <int BM, int BN, int WGM, int WGN>
static hipError_t launch_x3(ConvParams p, hipStream_t s) {
    constexpr int lds = X3_NS * (BM + BN) * 128;
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_x3_f16<BM, BN, WGM, WGN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.Cout + BN - 1) / BN;
    // column ranges per XCD when the weights do not fit an XCD's L2 and the N-tiles divide evenly (placement: speed only, never results)
    p.ngroup = ((p.ntiles & 7) == 0 && (size_t)p.Cout * p.Kpad * 2 > ((size_t)3 << 20)) ? p.ntiles / 8 : 0;
    hipLaunchKernelGGL((gemm_x3_f16<BM, BN, WGM, WGN>), dim3(p.mtiles * p.ntiles), dim3(64 * WGM * WGN), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_gemm_x3(ConvParams p, hipStream_t s, const char **name) {
    if (!gemm_x3_rule(p)) return hipErrorInvalidValue;
    if (!p.ldw) p.ldw = p.Kpad;
    // 128 x 128 tiles (eight waves, one 128 KB workgroup per CU) once they fill the chip, 64 x 64 tiles (four waves, two per CU) below
    if ((long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) >= 256) {
        if (name) *name = "gemm_x3_f16<128x128>";
        return launch_x3<128, 128, 4, 2>(p, s);
    }
    if (name) *name = "gemm_x3_f16<64x64>";
    return launch_x3<64, 64, 2, 2>(p, s);
}

}  // namespace hmv
