// conv_gemm8.hip -- the fp16 256 x 256 GEMM tile on a counted-vmcnt, phase-interleaved main loop.
//
// Replaces, for the fp16 path's MFMA-heavy 1x1 convolutions (/root/reference/src/models/backbones/resnet.py:124-131 conv1 of
// layer3's Bottlenecks, K = 1024; handmvnet.py:70-86 pose_net.0, K = 1024; the conv3 + downsample GEMM of a layer's first block,
// K = planes + inplanes), conv_igemm's main loop: two tile buffers, ONE `vmcnt(0)` + ONE barrier per 64-wide k-step with all
// eight waves reading their fragments, then all issuing MFMAs, in lock step.  That structure is tuned for the 64-cycle fp32
// MFMA; with 32-cycle fp16 MFMAs it leaves the matrix pipe idle for every fragment-read burst and every drained DMA
// (MFMA-busy 0.19 .. 0.53).  /opt/skills/guides/cdna_hip_programming.md section 5 gives the structure that does
// better on gfx950 (its "256^2 8-phase template"); this is that structure for our operand roles and tile decomposition:
//
//   * a k-step is FOUR phases, each {fragment reads . DMA of one half-tile . [counted wait] . barrier . 8 MFMAs . barrier}:
//         phase   MFMAs (pixel blocks a, channel block b)   fragment reads          DMA issued                      wait
//           1     a = 0,1  b = 0                            P[0,1] (8), W0 (4)      pixel half 1 of tile t + 1      vmcnt(10)
//           2     a = 0,1  b = 1                            W1 (4)                  pixel half 0 of tile t + 2      vmcnt(10)
//           3     a = 2,3  b = 1                            P[2,3] (8)              weight half 0 of tile t + 2     --
//           4     a = 2,3  b = 0                            --                      weight half 1 of tile t + 2     vmcnt(10)
//     a wave's 128 x 64 sub-tile = 4 x 2 accumulator blocks; every block gets the four k16 MFMAs of the step inside ONE phase,
//     in ascending k: the same accumulation order as conv_igemm, hence the same bits.
//   * the two wave halves (pixel rows 0-127 / 128-255) run ONE BARRIER APART: while one half issues its 8 MFMAs the other
//     reads fragments and issues DMAs, so each SIMD's matrix pipe alternates between its two waves instead of idling.
//   * `s_waitcnt vmcnt(10)`, never 0: a half-tile is two DMA instructions per thread and its slot is refilled in the phase after
//     the one that read it, so FIVE half-tiles (80 KB) are in flight across the barriers (raw s_barrier + lgkmcnt only): the
//     pixel tiles come from HBM at 2-4 us per request under chip-wide load (the first version kept two half-tiles in flight:
//     layer3 conv1 172 -> 162 us).  A half-tile is read one phase after the wait that retires it (the wait precedes a barrier that
//     every reader passes).
//   * LDS: 2 k-steps x 4 half-tiles x [128 rows][64 halfs] = 128 KB, XOR-swizzled on the DMA source side like conv_igemm.
//     Half-tiles are cut along the phase boundaries: pixel half h = blocks 2h, 2h + 1 of both wave rows; weight half h =
//     block h of all four wave columns.
// Operand roles, bias-as-initial-accumulator and the register epilogue are conv_igemm's transposed-output path.
//
// Round 4: M16 = the same tile, schedule and LDS image on v_mfma_f32_16x16x32_f16 (12-15 % less time on conv_ht's MFMA-bound loop at equal
// cycles, profiles/r04_probe_mfma_shape.txt).  A phase's 8 MFMAs of 32 cycles become 16 of 16: pixel blocks a, a + 1 are four 16-row
// blocks, channel block b two, the 64-wide k-step two k32 groups (ascending); a lane's fragment is k-group lane >> 4 of row lane & 15
// (chunk 4 g + kg of the 128-byte row: the (row >> 1) & 7 swizzle keeps a ds_read_b128's 16-lane groups conflict-free for this map too).
// LDS weight row 16 e + rho of a 32-channel block holds channel 8 (rho >> 2) + 4 e + (rho & 3) (source-side permutation), which makes a
// lane's two 16-row blocks 8 consecutive channels.  The layers that take it are chosen by SHAPE (conv_m16_rule), never by the batch:
// their small launches run conv_m16.hip's tiles, same bits.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace hmv {

typedef float gf32x16 __attribute__((ext_vector_type(16)));
typedef float gf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 gf16x8 __attribute__((ext_vector_type(8)));

#define HMV_GGLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// DUAL: the reduction is the concatenation [first source | second source] (conv3 + downsample as one GEMM, ConvParams::in2)
template <bool DUAL, bool M16 = false>
__global__ __launch_bounds__(512, 2) void conv_gemm8_f16(const ConvParams p) {
    constexpr int HT = 128 * 64;   // halfs per half-tile
    extern __shared__ __attribute__((aligned(16))) _Float16 gsm[];   // [2 k-steps][A0, A1, B0, B1][128][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    // ---- XCD-aware tile assignment (bijective for any grid size), as conv_igemm
    int mt, nt;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r = nblk & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        mt = lid / p.ntiles;
        nt = lid - mt * p.ntiles;
    }
    const _Float16 *zero = reinterpret_cast<const _Float16 *>(p.zero);
    const int nk = p.Kpad >> 6;

    // ---- DMA roles.  Thread -> row (tid >> 3) + 64 i of a half-tile, physical chunk tid & 7 holding logical chunk kqs.
    const int kqs = (tid & 7) ^ ((tid >> 4) & 7);
    // pixel half h, pass i: half-row hr = (tid >> 3) + 64 i -> wave row hr >> 6, block 2h + ((hr >> 5) & 1), line hr & 31
    int aoff[2][2];          // element offset of the row in the first source (-1: past M, zero page)
    int aoff2[2][2];         // ... in the second source
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hr = (tid >> 3) + 64 * i;
            const int m = mt * 256 + (hr >> 6) * 128 + (2 * h + ((hr >> 5) & 1)) * 32 + (hr & 31);
            aoff[h][i] = m < p.M ? m * p.lda + 8 * kqs : -1;
            aoff2[h][i] = -1;
            if constexpr (DUAL) {
                if (m < p.M) {
                    const int hw = p.Ho * p.Wo, n = m / hw, rem = m - n * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    aoff2[h][i] = ((n * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.lda2 + 8 * kqs;
                }
            }
        }
    // weight half h, pass i: half-row hr -> wave column hr >> 5, block h, line hr & 31
    const _Float16 *wsrc[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hr = (tid >> 3) + 64 * i;
            const int line = hr & 31, ch = M16 ? 8 * ((line & 15) >> 2) + 4 * (line >> 4) + (line & 3) : line;
            wsrc[h][i] = reinterpret_cast<const _Float16 *>(p.wgt) + (size_t)(nt * 256 + (hr >> 5) * 64 + h * 32 + ch) * p.ldw + 8 * kqs;
        }
    const _Float16 *Ain = reinterpret_cast<const _Float16 *>(p.in), *Ain2 = reinterpret_cast<const _Float16 *>(p.in2);

    // half-tile `which` (0 A0, 1 A1, 2 B0, 3 B1) of k-step t (t >= nk: dummies from the zero page, nobody reads them)
    auto stage = [&](int t, int which) {
        _Float16 *dst = gsm + ((t & 1) * 4 + which) * HT + wave * 8 * 64;
        const bool live = t < nk;
        const int k0 = t << 6;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const _Float16 *src = zero;
            if (which < 2) {
                if constexpr (DUAL) {
                    if (live && k0 >= p.ksplit) { if (aoff2[which][i] >= 0) src = Ain2 + (size_t)aoff2[which][i] + (k0 - p.ksplit); }
                    else if (live && aoff[which][i] >= 0) src = Ain + (size_t)aoff[which][i] + k0;
                } else {
                    if (live && aoff[which][i] >= 0) src = Ain + (size_t)aoff[which][i] + k0;
                }
            } else if (live) {
                src = wsrc[which - 2][i] + k0;
            }
            asm volatile("" : "+v"(src));   // ONE DMA instruction per schedule entry
            HMV_GGLDS16(src, dst + i * 64 * 64);
        }
    };

    // ---- accumulators start at the bias (transposed output: a register is ONE channel for all of the lane's pixels).
    // 16x16x32: acc4[a][s][b][e] = pixel rows 32 a + 16 s + l15, 16-row weight block e of channel block b: register r = channel
    // 32 b + 8 kg + 4 e + r
    gf32x16 acc[M16 ? 1 : 4][M16 ? 1 : 2];
    gf32x4 acc4[M16 ? 4 : 1][M16 ? 2 : 1][M16 ? 2 : 1][M16 ? 2 : 1];
    const int l15 = lane & 15, kg = lane >> 4;
    {
        const float binit = 1.f / p.acc_scale;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float *bp = p.bias + nt * 256 + wn * 64 + 32 * b;
            if constexpr (!M16) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const gf32x4 bq = *reinterpret_cast<const gf32x4 *>(bp + 16 * (q >> 1) + 8 * kh + 4 * (q & 1));
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[a][b][4 * q + u] = bq[u] * binit;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const gf32x4 bq = *reinterpret_cast<const gf32x4 *>(bp + 8 * kg + 4 * e);
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                            for (int u = 0; u < 4; ++u) acc4[a][sb][b][e][u] = bq[u] * binit;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias loads: the counted waits below see DMAs only

    // fragment addresses: pixel block a -> half a >> 1, half-row wm * 64 + (a & 1) * 32 + l31; weight block b -> half b,
    // half-row wn * 32 + swap23(l31) (the row swap makes registers 8j .. 8j+7 eight consecutive channels).
    // 16x16x32: rows ... + 16 s + l15 resp. wn * 32 + 16 e + l15 (the channel permutation sits in the DMA source), chunks 4 g + kg
    const int wl31 = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int fsw = M16 ? (l15 >> 1) & 7 : (l31 >> 1) & 7, fswb = M16 ? (l15 >> 1) & 7 : (wl31 >> 1) & 7;   // (rows + 16 s: (row >> 1) & 7 is unchanged)
    const int prow = (wm * 64 + (M16 ? l15 : l31)) * 64, wrow = (wn * 32 + (M16 ? l15 : wl31)) * 64;
    gf16x8 fa[2][4], fb0[4], fb1[4];   // 16x16x32: fa[a][2 s + g], fb[2 e + g]
#define G8_READ_P(t, h)                                                                                     \
    _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_)                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
            if constexpr (!M16) fa[a_][q_] = *reinterpret_cast<const gf16x8 *>(gsm + (((t) & 1) * 4 + (h)) * HT + prow + a_ * 32 * 64 + (((2 * q_ + kh) ^ fsw) * 8)); \
            else fa[a_][q_] = *reinterpret_cast<const gf16x8 *>(gsm + (((t) & 1) * 4 + (h)) * HT + prow + (a_ * 32 + (q_ >> 1) * 16) * 64 + (((4 * (q_ & 1) + kg) ^ fsw) * 8)); \
        }
#define G8_READ_W(t, h, FB)                                                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                      \
        if constexpr (!M16) FB[q_] = *reinterpret_cast<const gf16x8 *>(gsm + (((t) & 1) * 4 + 2 + (h)) * HT + wrow + (((2 * q_ + kh) ^ fswb) * 8)); \
        else FB[q_] = *reinterpret_cast<const gf16x8 *>(gsm + (((t) & 1) * 4 + 2 + (h)) * HT + wrow + (q_ >> 1) * 16 * 64 + (((4 * (q_ & 1) + kg) ^ fswb) * 8)); \
    }
#define G8_MFMA(A0, B, FB)                                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    if constexpr (!M16) {                                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                    \
            _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_)                                                \
                acc[(A0) + a_][B] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FB[q_], fa[a_][q_], acc[(A0) + a_][B], 0, 0, 0); \
    } else {                                                                                                \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_)                                                    \
            _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_)                                                \
                _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                            \
                    _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_)                                        \
                        acc4[(A0) + a_][s_][B][e_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FB[2 * e_ + g_], fa[a_][2 * s_ + g_], acc4[(A0) + a_][s_][B][e_], 0, 0, 0); \
    }                                                                                                       \
    __builtin_amdgcn_s_setprio(0);
// raw barrier, no vmcnt / lgkmcnt: the DMAs stay in flight across it; the compiler itself waits (lgkmcnt) for the fragments an
// MFMA reads, and sched_barrier(0) keeps the MFMA cluster between its two barriers
#define G8_BAR()                                                                                            \
    asm volatile("s_barrier" ::: "memory");                                                                 \
    __builtin_amdgcn_sched_barrier(0)
// Five half-tiles (80 KB) stay in flight: a half-tile's slot is refilled -- with the half-tile of k-step t + 2 -- in the phase after
// the one whose fragment reads emptied it, not a whole k-step later.  The operand tiles come from HBM (the weights from L2): at the
// 2-4 us a request takes under chip-wide load, 32 KB in flight were 10-16 GB/s per CU where HBM gives each CU 24 (tools/probe/l2bw.hip).
// Issue order, two DMAs per phase:  ... P1: A1(t+1) . P2: A0(t+2) . P3: W0(t+2) . P4: W1(t+2) . P1: A1(t+2) ...
// A half-tile is read 5 issues after its own: "at most 10 instructions out" == it has landed.  lgkmcnt(0) before a barrier: the
// fragment reads of this phase have left LDS before any wave refills what they read.
#define G8_WAIT10() asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory")
#define G8_WAITLDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    // ---- prologue: the issue order of the steps "before the first"; A0(0) and W0(0) must have landed
    stage(0, 0);
    stage(0, 2);
    stage(0, 3);
    stage(0, 1);
    stage(1, 0);
    stage(1, 2);
    stage(1, 3);
    G8_WAIT10();
    G8_BAR();
    if (wm == 1) G8_BAR();   // the second wave half runs one barrier behind the first from here on

    for (int t = 0; t < nk; ++t) {
        // phase 1
        G8_READ_W(t, 0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        G8_READ_P(t, 0);
        stage(t + 1, 1);       // (its slot: pixel half 1 of k-step t - 1, read in that step's phase 3)
        G8_WAIT10();           // weight half 1 of this k-step (read in phase 2) has landed
        G8_BAR();
        G8_MFMA(0, 0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        G8_BAR();
        // phase 2
        G8_READ_W(t, 1, fb1);
        stage(t + 2, 0);       // (pixel half 0 of this k-step was read in phase 1)
        G8_WAIT10();           // pixel half 1 of this k-step (read in phase 3) has landed
        G8_BAR();
        G8_MFMA(0, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        G8_BAR();
        // phase 3
        G8_READ_P(t, 1);
        stage(t + 2, 2);       // (weight half 0: phase 1)
        G8_WAITLDS();
        G8_BAR();
        G8_MFMA(2, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        G8_BAR();
        // phase 4
        stage(t + 2, 3);       // (weight half 1: phase 2)
        G8_WAIT10();           // pixel half 0 and weight half 0 of k-step t + 1 (read in its phase 1) have landed
        G8_BAR();
        G8_MFMA(2, 0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        G8_BAR();
    }
    if (wm == 0) G8_BAR();   // the first half waits for the second: barrier counts match
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy DMAs of k-step nk must not outlive the LDS allocation
#undef G8_READ_P
#undef G8_READ_W
#undef G8_MFMA
#undef G8_BAR
#undef G8_WAIT10
#undef G8_WAITLDS

    // ---- epilogue straight from the accumulators (conv_igemm's register path without a residual):
    //   register 8j + u of block (a, b) = channel 32 b + 16 j + 8 kh + u of pixel row 32 a + l31
    //   16x16x32: registers of blocks (a, s, b, e = 0 / 1) = channels 32 b + 8 kg + 0 .. 7 of pixel row 32 a + 16 s + l15
    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    const int nb0 = nt * 256 + wn * 64;
    const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 7) & ~7);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if constexpr (!M16) {
            const int m = mt * 256 + wm * 128 + 32 * a + l31;
            _Float16 *orow = reinterpret_cast<_Float16 *>(p.out) + (size_t)m * p.ldc;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = nb0 + 32 * b + 16 * j + 8 * kh;
                    gf16x8 hv;
#pragma unroll
                    for (int u = 0; u < 8; ++u) hv[u] = (_Float16)fmaxf(acc[a][b][8 * j + u] * p.acc_scale + 0.f, lo);
                    if (m < p.M && col < cend) *reinterpret_cast<gf16x8 *>(orow + col) = hv;
                }
        } else {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int m = mt * 256 + wm * 128 + 32 * a + 16 * sb + l15;
                _Float16 *orow = reinterpret_cast<_Float16 *>(p.out) + (size_t)m * p.ldc;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int col = nb0 + 32 * b + 8 * kg;
                    gf16x8 hv;
#pragma unroll
                    for (int u = 0; u < 8; ++u) hv[u] = (_Float16)fmaxf(acc4[a][sb][b][u >> 2][u & 3] * p.acc_scale + 0.f, lo);
                    if (m < p.M && col < cend) *reinterpret_cast<gf16x8 *>(orow + col) = hv;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_gemm8p_f16: conv_gemm8_f16<DUAL, M16 = true> as a PERSISTENT kernel (round 4, after conv_htp_f16).  One workgroup per CU (grid
// = 256) walks its tiles.  The half-tiles that a tile's last two k-steps would have requested as dummies are the NEXT tile's first seven
// -- in the prologue's own order, A0 W0 W1 A1 of step 0, A0 W0 W1 of step 1 -- so the next tile starts at phase 1 with its operands
// landed or in flight (the one-workgroup-per-tile kernel pays an HBM round trip of 2-4 us per 30-40 us tile there), and its epilogue
// stores drain under the next tile's first phases.  Placement as conv_htp_f16: XCD x owns the pixel tiles [x B, (x + 1) B), its 32
// workgroups are 32 / ntiles tile slots x ntiles channel tiles (a workgroup keeps its channel tile).  The epilogue's stores are younger
// than the DMAs in flight, so the counted waits of the next phases also retire stores: conservative, never wrong.  Same MFMA sequence
// per output as conv_gemm8_f16<DUAL, true>: bit-identical (test_gemm8_persistent_form).
template <bool DUAL>
__global__ __launch_bounds__(512) void conv_gemm8p_f16(const ConvParams p) {
    constexpr int HT = 128 * 64;   // halfs per half-tile
    extern __shared__ __attribute__((aligned(16))) _Float16 gsm[];   // [2 k-steps][A0, A1, B0, B1][128][64], then 256 floats of bias
    float *sbias = reinterpret_cast<float *>(gsm + 2 * 4 * HT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, kg = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int S = 32 / p.ntiles, nt = slot % p.ntiles, sidx = slot / p.ntiles;
    const int Bx = (p.mtiles + 7) >> 3, mt_end = min((xcd + 1) * Bx, p.mtiles);
    int mt = xcd * Bx + sidx;
    if (mt >= mt_end) return;
    const _Float16 *zero = reinterpret_cast<const _Float16 *>(p.zero);
    const int nk = p.Kpad >> 6;

    const int kqs = (tid & 7) ^ ((tid >> 4) & 7);
    // element offsets of the four pixel rows a thread moves ([half h][pass i]) in pixel tile mt_ (-1: past M -> zero page)
    auto rows_of = [&](int mt_, int (&o1)[2][2], int (&o2)[2][2]) {
        int t = tid;
        asm volatile("" : "+v"(t));   // recomputed per tile: nothing of it lives across the phases
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hr = (t >> 3) + 64 * i;
                const int m = mt_ * 256 + (hr >> 6) * 128 + (2 * h + ((hr >> 5) & 1)) * 32 + (hr & 31);
                o1[h][i] = m < p.M ? m * p.lda + 8 * kqs : -1;
                o2[h][i] = -1;
                if constexpr (DUAL) {
                    if (m < p.M) {
                        const int hw = p.Ho * p.Wo, n = m / hw, rem = m - n * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                        o2[h][i] = ((n * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.lda2 + 8 * kqs;
                    }
                }
            }
    };
    int aoff[2][2], aoff2[2][2], noff[2][2] = {{-1, -1}, {-1, -1}}, noff2[2][2] = {{-1, -1}, {-1, -1}};   // this tile's rows, the next tile's
    rows_of(mt, aoff, aoff2);
    const _Float16 *wsrc[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hr = (tid >> 3) + 64 * i;
            const int line = hr & 31, ch = 8 * ((line & 15) >> 2) + 4 * (line >> 4) + (line & 3);
            wsrc[h][i] = reinterpret_cast<const _Float16 *>(p.wgt) + (size_t)(nt * 256 + (hr >> 5) * 64 + h * 32 + ch) * p.ldw + 8 * kqs;
        }
    const _Float16 *Ain = reinterpret_cast<const _Float16 *>(p.in), *Ain2 = reinterpret_cast<const _Float16 *>(p.in2);

    int par = 0;            // LDS buffer of this tile's k-step 0 (k-steps alternate buffers across tiles)
    bool have_next = false;
    // half-tile `which` of k-step t of THIS tile; t >= nk: k-step t - nk of the next tile (none: dummies from the zero page)
    auto stage = [&](int t, int which) {
        _Float16 *dst = gsm + (((t + par) & 1) * 4 + which) * HT + wave * 8 * 64;
        const bool nxt = t >= nk;
        const bool live = !nxt || have_next;
        const int k0 = (nxt ? t - nk : t) << 6;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const _Float16 *src = zero;
            if (which < 2) {
                const int o1 = nxt ? noff[which][i] : aoff[which][i];
                if constexpr (DUAL) {
                    const int o2 = nxt ? noff2[which][i] : aoff2[which][i];
                    if (live && k0 >= p.ksplit) { if (o2 >= 0) src = Ain2 + (size_t)o2 + (k0 - p.ksplit); }
                    else if (live && o1 >= 0) src = Ain + (size_t)o1 + k0;
                } else {
                    if (live && o1 >= 0) src = Ain + (size_t)o1 + k0;
                }
            } else if (live) {
                src = wsrc[which - 2][i] + k0;
            }
            asm volatile("" : "+v"(src));   // ONE DMA instruction per schedule entry
            HMV_GGLDS16(src, dst + i * 64 * 64);
        }
    };

    if (tid < 256) sbias[tid] = p.bias[nt * 256 + tid] * (1.f / p.acc_scale);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias loads: the counted waits below see DMAs (and a tile's stores) only

    const int fsw = (l15 >> 1) & 7;
    const int prow = (wm * 64 + l15) * 64, wrow = (wn * 32 + l15) * 64;
    gf16x8 fa[2][4], fb0[4], fb1[4];   // fa[a][2 s + g], fb[2 e + g]
    gf32x4 acc4[4][2][2][2];
#define G8P_READ_P(t, h)                                                                                    \
    _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_)                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                    \
            fa[a_][q_] = *reinterpret_cast<const gf16x8 *>(gsm + ((((t) + par) & 1) * 4 + (h)) * HT + prow + (a_ * 32 + (q_ >> 1) * 16) * 64 + (((4 * (q_ & 1) + kg) ^ fsw) * 8));
#define G8P_READ_W(t, h, FB)                                                                                \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                        \
        FB[q_] = *reinterpret_cast<const gf16x8 *>(gsm + ((((t) + par) & 1) * 4 + 2 + (h)) * HT + wrow + (q_ >> 1) * 16 * 64 + (((4 * (q_ & 1) + kg) ^ fsw) * 8));
#define G8P_MFMA(A0, B, FB)                                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_)                                                        \
        _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_)                                                    \
            _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                \
                _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_)                                            \
                    acc4[(A0) + a_][s_][B][e_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FB[2 * e_ + g_], fa[a_][2 * s_ + g_], acc4[(A0) + a_][s_][B][e_], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);
#define G8P_BAR()                                                                                           \
    asm volatile("s_barrier" ::: "memory");                                                                 \
    __builtin_amdgcn_sched_barrier(0)
#define G8P_WAIT10() asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory")
#define G8P_WAITLDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    // ---- prologue of the FIRST tile (conv_gemm8_f16's)
    stage(0, 0);
    stage(0, 2);
    stage(0, 3);
    stage(0, 1);
    stage(1, 0);
    stage(1, 2);
    stage(1, 3);
    G8P_WAIT10();
    G8P_BAR();
    if (wm == 1) G8P_BAR();   // the second wave half runs one barrier behind the first from here on

    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    const int nb0 = nt * 256 + wn * 64;
    const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 7) & ~7);
    for (;;) {
        const int mt_next = mt + S;
        have_next = mt_next < mt_end;
        if (have_next) rows_of(mt_next, noff, noff2);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 2; ++e) {   // register r of block (b, e) = channel 32 b + 8 kg + 4 e + r (as conv_gemm8_f16<., true>)
                const gf32x4 bq = *reinterpret_cast<const gf32x4 *>(sbias + wn * 64 + 32 * b + 8 * kg + 4 * e);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) acc4[a][sb][b][e] = bq;
            }
        for (int t = 0; t < nk; ++t) {
            // phase 1
            G8P_READ_W(t, 0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            G8P_READ_P(t, 0);
            stage(t + 1, 1);
            G8P_WAIT10();
            G8P_BAR();
            G8P_MFMA(0, 0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            G8P_BAR();
            // phase 2
            G8P_READ_W(t, 1, fb1);
            stage(t + 2, 0);
            G8P_WAIT10();
            G8P_BAR();
            G8P_MFMA(0, 1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            G8P_BAR();
            // phase 3
            G8P_READ_P(t, 1);
            stage(t + 2, 2);
            G8P_WAITLDS();
            G8P_BAR();
            G8P_MFMA(2, 1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            G8P_BAR();
            // phase 4
            stage(t + 2, 3);
            G8P_WAIT10();
            G8P_BAR();
            G8P_MFMA(2, 0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            G8P_BAR();
        }
        // ---- epilogue of this tile (conv_gemm8_f16<., true>'s)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int m = mt * 256 + wm * 128 + 32 * a + 16 * sb + l15;
                _Float16 *orow = reinterpret_cast<_Float16 *>(p.out) + (size_t)m * p.ldc;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int col = nb0 + 32 * b + 8 * kg;
                    gf16x8 hv;
#pragma unroll
                    for (int u = 0; u < 8; ++u) hv[u] = (_Float16)fmaxf(acc4[a][sb][b][u >> 2][u & 3] * p.acc_scale + 0.f, lo);
                    if (m < p.M && col < cend) *reinterpret_cast<gf16x8 *>(orow + col) = hv;
                }
            }
        if (!have_next) break;
        mt = mt_next;
        par = (par + nk) & 1;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) { aoff[h][i] = noff[h][i]; aoff2[h][i] = noff2[h][i]; }
    }
    if (wm == 0) G8P_BAR();   // the first half waits for the second: barrier counts match
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef G8P_READ_P
#undef G8P_READ_W
#undef G8P_MFMA
#undef G8P_BAR
#undef G8P_WAIT10
#undef G8P_WAITLDS
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_gemm8r_f16: the same tile on 32-element k-steps with the two operand streams on SEPARATE waves.
//
// Vector-memory returns are in order per wave.  A wave that requests both operand tiles queues its weight lines -- L2 hits, back in
// ~1 us, 117 GB/s per CU (tools/probe/l2bw.hip) -- behind its pixel lines, which come from HBM (24 GB/s per CU, 2-4 us under chip-wide
// load): everything then pays the HBM latency, and a ring deep enough to cover it for BOTH operands does not fit LDS.  Here the first
// wave half requests pixel tiles only, FIVE k-steps ahead (80 KB in flight, a ring of six 16 KB stages), the second half weight tiles
// only, two k-steps ahead (a ring of three): each queue carries one latency class and its own counted wait (vmcnt(16) / vmcnt(4)).
// Main loop: conv_ht.hip's -- one phase per k-step {12 fragment reads . 4 DMAs . wait . barrier . 16 MFMAs . barrier}, the wave halves one
// barrier apart.  Accumulation order (k16 blocks ascending per accumulator block) and epilogue are conv_gemm8_f16's: same bits.
constexpr int G8R_NA = 6, G8R_NW = 3, G8R_LA = G8R_NA - 1, G8R_LW = G8R_NW - 1, G8R_ST = 256 * 64;   // stages; bytes per stage
constexpr int G8R_LDS = (G8R_NA + G8R_NW) * G8R_ST;                                                   // 147 456

template <bool DUAL>
__global__ __launch_bounds__(512, 2) void conv_gemm8r_f16(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char rsm[];   // [NA pixel stages][NW weight stages], each [256 rows][32 halfs]
    char *wst = rsm + G8R_NA * G8R_ST;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    int mt, nt;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r = nblk & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        mt = lid / p.ntiles;
        nt = lid - mt * p.ntiles;
    }
    const _Float16 *zero = reinterpret_cast<const _Float16 *>(p.zero);
    const int nk = p.Kpad >> 5;
    unsigned long long t_entry = 0, t0c = 0, t0r = 0, t1c = 0, t1r = 0;   // diagnostic stamps (hmv_bench_conv with HMV_BENCH_CLOCK)
    if (p.dbg) t_entry = __builtin_amdgcn_s_memrealtime();

    // ---- DMA roles: 256 threads per operand; pass i moves 16-byte unit 256 i + (tid & 255) of a stage = physical chunk lane & 3 of
    // row 64 i + ((tid & 255) >> 2), holding logical chunk (lane & 3) ^ ((row >> 2) & 3) (64-byte rows)
    const int t8 = tid & 255, srow = t8 >> 2, sch = (t8 & 3) ^ ((t8 >> 4) & 3);
    const bool pix_wave = wm == 0;
    int off1[4], off2[4];              // pixel waves: element offsets of the four rows in the two sources (-1: past M -> zero page)
    const _Float16 *wrow_src[4];       // weight waves: the four rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 64 * i + srow, m = mt * 256 + row;
        off1[i] = m < p.M ? m * p.lda + 8 * sch : -1;
        off2[i] = -1;
        if constexpr (DUAL) {
            if (m < p.M) {
                const int hw = p.Ho * p.Wo, n = m / hw, rem = m - n * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                off2[i] = ((n * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.lda2 + 8 * sch;
            }
        }
        wrow_src[i] = reinterpret_cast<const _Float16 *>(p.wgt) + (size_t)(nt * 256 + row) * p.ldw + 8 * sch;
    }
    const _Float16 *Ain = reinterpret_cast<const _Float16 *>(p.in), *Ain2 = reinterpret_cast<const _Float16 *>(p.in2);
    // this wave's stage of k-step s (past the end: dummies from the zero page, nobody reads them)
    auto issue = [&](int s) {
        const bool live = s < nk;
        const int k0 = s << 5;
        char *dst = pix_wave ? rsm + (s % G8R_NA) * G8R_ST + (wave & 3) * 1024 : wst + (s % G8R_NW) * G8R_ST + (wave & 3) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const _Float16 *src = zero;
            if (pix_wave) {
                if constexpr (DUAL) {
                    if (live && k0 >= p.ksplit) { if (off2[i] >= 0) src = Ain2 + (size_t)off2[i] + (k0 - p.ksplit); }
                    else if (live && off1[i] >= 0) src = Ain + (size_t)off1[i] + k0;
                } else {
                    if (live && off1[i] >= 0) src = Ain + (size_t)off1[i] + k0;
                }
            } else if (live) {
                src = wrow_src[i] + k0;
            }
            asm volatile("" : "+v"(src));   // ONE DMA instruction per schedule entry
            HMV_GGLDS16(src, dst + i * 4096);
        }
    };

    gf32x16 acc[4][2];
    {
        const float binit = 1.f / p.acc_scale;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float *bp = p.bias + nt * 256 + wn * 64 + 32 * b;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const gf32x4 bq = *reinterpret_cast<const gf32x4 *>(bp + 16 * (q >> 1) + 8 * kh + 4 * (q & 1));
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[a][b][4 * q + u] = bq[u] * binit;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(acc[a][b]));   // the bias is in the accumulators before the first DMA goes out

    // fragment addresses (bytes inside a stage): pixel block a = rows wm 128 + 32 a + l31; weight block b = rows wn 64 + 32 b + swap23(l31)
    const int wl31 = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    int pfo[2], wfo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pr = wm * 128 + l31, wr = wn * 64 + wl31;   // (+ 32 a / 32 b leave (row >> 2) & 3 unchanged)
        pfo[j] = (pr * 4 + ((2 * j + kh) ^ ((pr >> 2) & 3))) * 16;
        wfo[j] = (wr * 4 + ((2 * j + kh) ^ ((wr >> 2) & 3))) * 16;
    }
    gf16x8 fp[4][2], fw[2][2];
#define G8R_BAR()                                                                                           \
    asm volatile("s_barrier" ::: "memory");                                                                 \
    __builtin_amdgcn_sched_barrier(0)

    // ---- prologue: each wave half fills its own ring's lead; stage 0 of both operands must have landed
#pragma unroll
    for (int s = 0; s < G8R_LA; ++s) if (pix_wave || s < G8R_LW) issue(s);
    if (pix_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (G8R_LA - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (G8R_LW - 1)) : "memory");
    G8R_BAR();
    if (p.dbg) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
    if (wm == 1) { G8R_BAR(); }   // the second wave half runs one barrier behind the first from here on

    for (int s = 0; s < nk; ++s) {
        const char *ast = rsm + (s % G8R_NA) * G8R_ST, *bst = wst + (s % G8R_NW) * G8R_ST;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[b][j] = *reinterpret_cast<const gf16x8 *>(bst + wfo[j] + b * 32 * 64);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 2; ++j) fp[a][j] = *reinterpret_cast<const gf16x8 *>(ast + pfo[j] + a * 32 * 64);
        // the stage whose slot the previous step's reads emptied; then: my share of the next step's stage has landed, and this wave's
        // fragment reads have left LDS
        if (pix_wave) { issue(s + G8R_LA); asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (G8R_LA - 1)) : "memory"); }
        else { issue(s + G8R_LW); asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (G8R_LW - 1)) : "memory"); }
        G8R_BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[b][j], fp[a][j], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        G8R_BAR();
    }
    if (wm == 0) { G8R_BAR(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.dbg) { t1c = __builtin_amdgcn_s_memtime(); t1r = __builtin_amdgcn_s_memrealtime(); }
#undef G8R_BAR

    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    const int nb0 = nt * 256 + wn * 64;
    const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 7) & ~7);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int m = mt * 256 + wm * 128 + 32 * a + l31;
        _Float16 *orow = reinterpret_cast<_Float16 *>(p.out) + (size_t)m * p.ldc;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = nb0 + 32 * b + 16 * j + 8 * kh;
                gf16x8 hv;
#pragma unroll
                for (int u = 0; u < 8; ++u) hv[u] = (_Float16)fmaxf(acc[a][b][8 * j + u] * p.acc_scale + 0.f, lo);
                if (m < p.M && col < cend) *reinterpret_cast<gf16x8 *>(orow + col) = hv;
            }
    }
    if (p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *d = p.dbg + 8 * (size_t)blockIdx.x;
        d[0] = t1c - t0c; d[1] = t1r - t0r; d[2] = t_entry; d[3] = t0r; d[4] = t1r; d[5] = __builtin_amdgcn_s_memrealtime();
    }
}

// ====================================================================== host side
static int g_gemm8_mode = -1;   // -1: the launcher's rule (HMV_NO_GEMM8=1 disables it); 0 never; 1 whenever supported (op-level tests)
void conv_gemm8_set_mode(int mode) { g_gemm8_mode = mode; }
static int g_gemm8_persist = 1;   // 1: the persistent form from two tiles per CU up; 0: never; 2: wherever it exists (op-level identity tests)
void conv_gemm8_set_persistent(int on) { g_gemm8_persist = on; }

bool conv_gemm8_supported(const ConvParams &p) {
    static int off = -1;   // development knob: HMV_NO_GEMM8=1 keeps these convs on conv_igemm (A/B runs)
    if (off < 0) off = HMV_DEV_ENV("HMV_NO_GEMM8") ? 1 : 0;
    if (g_gemm8_mode == 0 || (g_gemm8_mode < 0 && off)) return false;
    if (!p.in_f16 || !p.out_f16 || p.res || p.R != 1 || p.S != 1 || p.pad_h || p.pad_w || p.up || p.ksl > 1 || p.phases > 1) return false;
    if (p.cwrap || p.x3_plane || p.res_split || p.out_split || p.acc_shift || p.rd_cout || p.scatter || p.rg_out) return false;
    if (p.act != ACT_NONE && p.act != ACT_RELU) return false;
    if (p.Kpad % 64 || p.Kpad < 128 || (p.ldc & 7) || ((p.lda ? p.lda : p.Cin) & 7) || ((p.ldw ? p.ldw : p.Kpad) & 7)) return false;
    // the loop walks Kpad / 64 k-steps and DMA-reads Kpad halfs per pixel row with no column mask: every reduction column must be a real
    // channel of its source (a padded K would read the next pixel's channels against zero weights: 0 x NaN, and past the buffer's end)
    if (p.K != p.Kpad) return false;
    if (p.in2) {
        if (p.ksplit > (p.lda ? p.lda : p.Cin) || p.Kpad - p.ksplit > p.lda2) return false;
        if (p.ksplit % 64 || p.ksplit <= 0 || p.ksplit >= p.Kpad || (p.lda2 & 7) || p.stride != 1) return false;
        if ((long long)p.N * p.H2 * p.W2 * p.lda2 >= (1ll << 31)) return false;
    } else if (p.stride != 1 || p.Cin != p.Kpad) {
        return false;
    }
    if ((long long)p.M * (p.lda ? p.lda : p.Cin) >= (1ll << 31)) return false;   // 32-bit element offsets of the pixel rows
    if (g_gemm8_mode > 0) return true;
    // the launcher's rule: the shapes conv_pick_tile gives the 256 x 256 tile (Cout > 128, >= 512 tiles)
    return p.Cout > 128 && (long long)((p.M + 255) / 256) * ((p.Cout + 255) / 256) >= 512;
}

hipError_t launch_conv_gemm8(ConvParams p, hipStream_t s, const char **name) {
    constexpr size_t lds = (size_t)2 * 4 * 128 * 64 * sizeof(_Float16);
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm8_f16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm8_f16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm8_f16<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm8_f16<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    p.mtiles = (p.M + 255) / 256;
    p.ntiles = (p.Cout + 255) / 256;
    // the persistent form (16x16x32 only): from two tiles per CU up, channel-tile counts that divide an XCD's 32 workgroups
    if (p.m16 && g_gemm8_persist && ((long long)p.mtiles * p.ntiles >= 512 || g_gemm8_persist == 2) && (p.ntiles == 1 || p.ntiles == 2 || p.ntiles == 4)) {
        static bool pconf[64] = {};
        if (!pconf[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm8p_f16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm8p_f16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + 1024);
            if (e != hipSuccess) return e;
            pconf[dev] = true;
        }
        if (name) *name = p.in2 ? "conv_gemm8_f16<256x256,1x1,dual,m16,persistent>" : "conv_gemm8_f16<256x256,1x1,m16,persistent>";
        if (p.in2) hipLaunchKernelGGL(conv_gemm8p_f16<true>, dim3(256), dim3(512), lds + 1024, s, p);
        else hipLaunchKernelGGL(conv_gemm8p_f16<false>, dim3(256), dim3(512), lds + 1024, s, p);
        return hipGetLastError();
    }
    if (p.m16) {   // the layer multiplies on the 16x16x32 MFMA at every batch size (conv_m16_rule: by shape)
        if (name) *name = p.in2 ? "conv_gemm8_f16<256x256,1x1,dual,m16>" : "conv_gemm8_f16<256x256,1x1,m16>";
        if (p.in2) hipLaunchKernelGGL((conv_gemm8_f16<true, true>), dim3(p.mtiles * p.ntiles), dim3(512), lds, s, p);
        else hipLaunchKernelGGL((conv_gemm8_f16<false, true>), dim3(p.mtiles * p.ntiles), dim3(512), lds, s, p);
        return hipGetLastError();
    }
    if (name) *name = p.in2 ? "conv_gemm8_f16<256x256,1x1,dual>" : "conv_gemm8_f16<256x256,1x1>";
    // development knob, OFF by default: HMV_GEMM8_RING=1 selects conv_gemm8r_f16 (wave-specialised operand streams on a k32 ring).
    // Measured 6 % SLOWER than the four-phase loop (layer3 conv1 172 vs 162 us, profiles/r03_probe_gemm8_ring.txt): these launches are
    // bound by power, not by request latency (0.68 MFMA-busy in cycles at a 1.06-1.35 GHz clock on dense operands)
    static const int ring = HMV_DEV_ENV("HMV_GEMM8_RING") ? atoi(HMV_DEV_ENV("HMV_GEMM8_RING")) : 0;
    if (ring) {
        static bool rconf[64] = {};
        if (!rconf[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm8r_f16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, G8R_LDS);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm8r_f16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, G8R_LDS);
            if (e != hipSuccess) return e;
            rconf[dev] = true;
        }
        if (p.in2) hipLaunchKernelGGL(conv_gemm8r_f16<true>, dim3(p.mtiles * p.ntiles), dim3(512), G8R_LDS, s, p);
        else hipLaunchKernelGGL(conv_gemm8r_f16<false>, dim3(p.mtiles * p.ntiles), dim3(512), G8R_LDS, s, p);
        return hipGetLastError();
    }
    if (p.in2) hipLaunchKernelGGL(conv_gemm8_f16<true>, dim3(p.mtiles * p.ntiles), dim3(512), lds, s, p);
    else hipLaunchKernelGGL(conv_gemm8_f16<false>, dim3(p.mtiles * p.ntiles), dim3(512), lds, s, p);
    return hipGetLastError();
}

}  // namespace hmv
