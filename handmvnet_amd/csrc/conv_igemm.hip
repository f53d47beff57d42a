// conv_igemm.hip -- NHWC implicit-GEMM convolution on the gfx950 fp32 matrix cores.
//
// Replaces every nn.Conv2d / nn.Linear of the reference's hot path
// (/root/reference/src/models/backbones/resnet.py:114-118,162,193; layers.py:213-215,224,
//  161-174, 318-334; handmvnet.py:70-86) with ONE kernel family:
//
//   out[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] + residual[m][n] )
//
// M = images * Ho * Wo output pixels, N = Cout, K = R*S*Cin.  K is ordered
// (32-channel chunk, r, s, channel in chunk): a k-step is one tap of one 32-channel chunk, and the
// R*S taps of a chunk are consecutive k-steps, so the shifted re-reads of a 3x3 window hit L1/L2
// instead of re-streaming the activation from HBM once per tap (measured: FETCH_SIZE 4.3x the
// algorithmic bytes with tap-major K).  The im2col matrix A is never built: every lane of an
// LDS-DMA instruction supplies the global address of one 16-byte channel vector.
//
// MI355X mapping (the measurements behind each choice are in DESIGN.md section 4)
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD): each wave owns a (32*TM)x(32*TN)
//     output tile in TM*TN 16-register accumulators.  The operand fetch is a ds_read_b128 of 4
//     consecutive k for lane (row = lane&31, k-half = lane>>5); its 4 components feed 4
//     consecutive MFMAs (the k order inside a 32-wide step is permuted identically for A and Wt).
//   * tiles travel HBM/L2 -> LDS by global_load_lds_dwordx4 (LDS-DMA): no VGPR round trip and no
//     ds_write -- the VGPR->LDS store path alone cost ~10 % of the matrix pipe in the
//     register-staged version of this kernel.
//   * LDS image: [row][32 floats], unpadded 128-byte rows; one wave-instruction fills 8 rows
//     (1 KiB, lane-linear).  ds_read_b128 bank conflicts are removed by an XOR swizzle of the
//     16-byte chunk index, chunk' = chunk ^ ((row >> 1) & 7); since the DMA destination is
//     lane-linear, the swizzle is applied to the per-lane SOURCE address and to the read.
//   * out-of-range taps / rows DMA from a 256-byte zero page: no select after the load.
//   * per 32-wide k-step: MFMA groups q0..q2, then {vmcnt(0) for tile t+1, ONE s_barrier}, then the
//     DMA of tile t+2 into the buffer just vacated and the first fragments of tile t+1, all
//     under the 16 MFMAs of group q3.  In-kernel stamps show the main loop at ~94 % matrix-pipe
//     occupancy in cycles; what is left is DVFS (the chip drops to 2.0-2.2 GHz under real
//     operand traffic), which is why the largest tile that still fills the chip wins.
//   * XCD-aware block -> tile map: the N-tiles of one M-tile are consecutive on ONE XCD, so an
//     activation tile is pulled from HBM once per XCD and re-read from that XCD's L2.
//   * epilogue fused: folded-BN shift / bias, residual add, ReLU / GELU(erf) / LeakyReLU, staged
//     through LDS so that global traffic is 16-byte vectors on full output rows.
//
// Variants of the same template (DESIGN.md section 4 has the measurements):
//   * T = _Float16: v_mfma_f32_32x32x16_f16, fp32 accumulation, byte-identical LDS image / DMA pattern (k-step 64).
//   * MODE_DENSE: K = plain (r, s, c) over the REAL channels for Cin % 32 != 0 (stem, HRNet's 40 / 80-channel tensors).
//   * PARTN ("skipN"): all-padding 32-column blocks of the last N tile are skipped (Cout = 80 / 160 / 320).
//   * RD ("rowsum"): row-decomposed 3x3 conv for narrow outputs (3x1 GEMM with (s, cout) columns + row-sum epilogue).
//   * MODE_HALO (fp16 3x3 stride-1 convs on 256-row tiles): the tile is a 16 x 16 pixel block, its 18 x 18 halo is fetched once
//     per 64-channel chunk and the nine taps read it at shifted LDS rows (same K order and bits as MODE_TAPS, 43 % fewer bytes).
//   * TILE_256x256_RING (fp16, 32-element k-step, four stages / three tiles in flight): a recorded negative result, off by default.
//   * split operands (HMV_F32X3, fp16 kernels): fp32 values as (hi, lo) fp16 pairs, hi*hi + lo*hi + hi*lo; the fused loop
//     issues the three products from one tile load (x3_plane), the cwrap loop walks hi, lo, hi as a 3x longer reduction.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kernels.h"

namespace hmv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));


#ifndef HMV_UB
#define HMV_UB 4   // output rows (16-byte vectors) in flight per thread in the epilogue
#endif
#ifndef HMV_UB_F16
#define HMV_UB_F16 4   // the same for the fp16 kernels' drain
#endif
#ifndef HMV_TOUT
#define HMV_TOUT 1     // 1: transposed-output register epilogue in the non-generic kernels (0 = the LDS-staged drain, for A/B builds)
#endif

// 256 bytes of zeros: out-of-range taps / rows DMA from here.
static float *g_zero_page[64] = {};   // one per device ordinal
static hipError_t ensure_zero_page(float **page) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!g_zero_page[dev]) {
        // 256 bytes of zeros, then 1 KiB that masked-out lanes of the burst epilogue store into (never read)
        e = hipMalloc(reinterpret_cast<void **>(&g_zero_page[dev]), 256 + 1024);
        if (e == hipSuccess) e = hipMemset(g_zero_page[dev], 0, 256 + 1024);
        if (e != hipSuccess) { g_zero_page[dev] = nullptr; return e; }
    }
    *page = g_zero_page[dev];
    return hipSuccess;
}

// A-operand addressing modes
enum { MODE_TAPS = 0,   // general R x S convolution, Cin % 32 == 0: the tap of a k-step is wave-uniform
       MODE_1X1 = 1,    // 1x1 / plain GEMM, no padding: one pointer bump per DMA, no bounds test
       MODE_DENSE = 2,  // any Cin % 4 == 0 (the stem on NHWC4 frames, HRNet's 40/80-channel branches): K is the plain
                        // (r, s, c) order over the REAL channels, so no k is spent on channel padding; every 16-byte
                        // vector carries its own (tap, channel offset), derived per lane with two multiply-highs
       MODE_HALO = 3    // 3x3 stride-1 pad-1 conv, fp16, 256-pixel tiles that are 16 x 16 BLOCKS of one image: per 64-channel
                        // chunk the 18 x 18 halo of the block is fetched ONCE into LDS (41 KB) and the nine taps read it at
                        // shifted rows, instead of nine 32 KB fetches of the shifted tile (MODE_TAPS); weights stream as before
};
constexpr int HALO_ROWS = 328;   // 18 * 18 = 324 halo pixels, padded to whole 8-row DMA pieces

// Epilogue staging: per pass every wave stages AS of its 32-row accumulator blocks, so a pass holds
// SR = WGM*AS*32 rows of BN+4 floats.  AS is the largest divisor of TM that fits the LDS budget
// without growing the allocation (much) beyond the two tile buffers.
// (KB is given in 4-byte units here: the kernels pass KB * sizeof(T) / 4)
constexpr int stage_blocks(int BM, int BN, int WGM, int KB, bool full = false) {
    const int TM = BM / WGM / 32;
    if (full) return TM;   // the row-decomposed epilogue sums neighbouring rows: the whole tile is staged at once
    int as = TM;
    while (as > 1 && (TM % as != 0 || WGM * as * 32 * (BN + 4) > 2 * (BM + BN) * KB + 1024)) --as;
    return as;
}
// residual burst: blocks per chunk = the largest divisor of `nblk` whose pieces fit the wave's share of the tile buffers
constexpr int burst_blocks(int wave_bytes, int piece_count, int nblk) {
    int cb = wave_bytes / (piece_count * 1024);
    if (cb > nblk) cb = nblk;
    while (cb > 1 && nblk % cb != 0) --cb;
    return cb;
}
// Tile buffers in flight.  Two (double buffering) everywhere except the fp16 256x256 tile with a 32-element k-step
// (TILE_256x256_RING): there FOUR 32 KB stages form a ring with three tiles in flight (96 KB instead of 64 KB, same 128 KB of
// LDS).  Built in round 2 to test whether the ~2 us that a 64 KB k-step of the two-stage loop takes is one exposed L2 latency
// per step.  It is not: with three tiles in flight the same convs run 4-8 % slower (twice the barriers), so the ~30 GB/s per
// CU is a throughput limit of the L2 -> LDS path under chip-wide load, not latency.  Off by default (HMV_F16_RING=1).
constexpr int ring_stages(bool f16, int KB, bool generic, bool rd, int BM, int BN) {
    return (f16 && KB == 32 && !generic && !rd && BM == 256 && BN == 256) ? 4 : 2;
}
constexpr int lds_floats(int BM, int BN, int WGM, int KB, bool full = false, bool staged = true, int ns = 2) {
    const int tile = ns * (BM + BN) * KB;
    if (!staged) return tile;   // register epilogue: the tile buffers are all the LDS a workgroup needs
    const int stage = WGM * stage_blocks(BM, BN, WGM, KB, full) * 32 * (BN + 4);
    return tile > stage ? tile : stage;
}

#define HMV_GLDS16(gptr, lptr)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// GENERIC = false: the backbone's epilogue (bias, optional residual, optional ReLU, rows written in
// place).  GENERIC = true adds the rarely used paths (sub-pixel output scatter, residual row remap,
// GELU / LeakyReLU, row strides that are not multiples of 4); keeping them out of the hot
// instantiation keeps its epilogue straight-line.
// T = float: v_mfma_f32_32x32x2_f32, 4 elements per 16-byte chunk, KB = 32 (or 16) elements per k-step.
// T = _Float16 (BASELINE configs[4]): v_mfma_f32_32x32x16_f16 with fp32 accumulation, 8 elements per chunk,
// KB = 64: the LDS image, the DMA pattern and the swizzle are byte-identical to the fp32 kernel; one
// 16-byte chunk is exactly one MFMA operand (k = 8*kh + j), so a q-group is ONE MFMA per block pair.
// PARTN = true (chosen when the last N-tile holds at least one all-padding 32-column block, e.g. Cout = 80 / 160 / 320
// under 128- or 256-wide tiles): a wave skips the fragment reads and MFMAs of its all-padding blocks, and the waves are
// mapped to sub-tiles so that the two waves sharing a SIMD cover complementary column ranges.
// RD = true: row-decomposed 3x3 convolution for narrow outputs (HRNet's 40-channel branch).  The GEMM runs a 3x1
// convolution (taps r only) against weights whose N index is (s, cout): N' = 3*Cout = 120 fills a 128-wide tile where
// Cout = 40 fills 40 of 64 columns, and K' = 3*Cin = 120 pads to 128 instead of 360 to 384 -- 1.5x fewer MFMAs.  The
// epilogue then sums the three column groups of horizontally neighbouring pixels, out[m][n] = sum_s G[m + s - 1][s*Cout + n],
// which needs no halo because a tile always covers whole image rows (BM % Wo == 0, checked by launch_conv).
// waves per SIMD the register allocator is asked to leave room for: the 8-wave 64-accumulator tiles (256x128 / 128x256) of the fp16
// kernels are held to 128 registers so that TWO workgroups share a CU (their load / compute / drain phases then overlap)
#ifndef HMV_OCC2
#define HMV_OCC2 1
#endif
constexpr int min_waves(int BM, int BN, int NT, bool f16, bool generic, bool rd, int KB) {
    return (HMV_OCC2 && KB == (f16 ? 32 : 16) && !generic && !rd && NT == 512 && BM * BN == 256 * 128) ? 4 : (NT == 512 ? 2 : 1);
}
// C32 (fp16, MODE_TAPS, KB = 32): the packed K order walks 32-channel chunks -- conv_ht.hip's order, so that the layers packed for
// the tall-tile kernel run here, with the same bits, when the batch is too small to fill the chip with 512-pixel tiles
template <typename T, int BM, int BN, int WGM, int WGN, int MODE, bool GENERIC, int KB, bool PARTN = false, bool RD = false, bool X3 = false, bool C32 = false>
__global__ __launch_bounds__(64 * WGM * WGN, min_waves(BM, BN, 64 * WGM * WGN, sizeof(T) == 2, GENERIC, RD, KB)) void conv_igemm(const ConvParams p) {
    constexpr bool F16 = sizeof(T) == 2;
    constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
    constexpr int CH = (F16 && !C32) ? 64 : 32;    // channel-chunk width of the packed K order
    static_assert(!C32 || (F16 && KB == 32 && MODE == MODE_TAPS && !GENERIC && !PARTN && !RD && !X3), "32-channel chunk order");
    constexpr int KB4 = KB * (int)sizeof(T) / 4;   // k-step in 4-byte units (row bytes / 4)
    constexpr int NT = 64 * WGM * WGN;   // 4 or 8 waves
    constexpr int LPR = KB / EPC;        // lanes (16-byte chunks) per tile row: 8 (128-byte rows) or 4 (64-byte rows)
    constexpr int RPW = 64 / LPR;        // rows filled by one wave-instruction (1 KiB): 8 or 16
    constexpr int RPS = NT / LPR;        // tile rows filled per DMA pass
    constexpr int NQ = LPR / 2;          // MFMA groups per k-step (two chunks, k-halves 0/1, per group)
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int AP = BM / RPS, BP = BN / RPS;
    constexpr int LDC = BN + 4;
    constexpr int AS = stage_blocks(BM, BN, WGM, KB4, RD), SR = WGM * AS * 32;
    // Transposed output (the hot instantiations): the MFMA takes the WEIGHT rows as its A operand and the pixel rows as its B
    // operand, so an accumulator block holds out^T: the lane is the pixel, the 16 registers are 16 output channels of that
    // pixel.  The epilogue then needs no LDS transposition, no barrier and no staging memory: every lane adds bias / residual to
    // its own channels and stores them as 16-byte vectors (4 consecutive fp32 channels or 8 consecutive halfs per register group).
    constexpr bool TOUT = HMV_TOUT && F16 && !GENERIC && !RD;   // measured: a win on the HBM-bound fp16 kernels, not on the fp32 ones
    static_assert(KB == CH || KB == CH / 2, "k-step");
    // X3: the fused split main loop (p.x3_plane != 0) -- its own instantiation, so that its six fragment sets do not set the
    // register budget of the plain fp16 kernels
    constexpr bool X3CAP = X3;
    static_assert(!X3 || (F16 && KB == 64 && !GENERIC && !PARTN && !RD), "fused split loop");
    constexpr bool x3n = X3;
    static_assert(BM % RPS == 0 && BN % RPS == 0 && WM % 32 == 0 && WN % 32 == 0, "tile shape");
    static_assert(TOUT || (TM % AS == 0 && SR * LDC <= lds_floats(BM, BN, WGM, KB4, RD)), "epilogue staging");
    static_assert(!RD || (!GENERIC && !PARTN), "row-decomposed epilogue");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NS = ring_stages(F16, KB, GENERIC, RD, BM, BN);
    static_assert(NS == 2 || (TOUT && !X3 && !PARTN), "ring main loop");
    constexpr bool HALO = MODE == MODE_HALO;
    static_assert(!HALO || (TOUT && KB == 64 && BM == 256 && NS == 2 && !X3 && !PARTN && NT == 512), "halo mode");
    T *sA = reinterpret_cast<T *>(smem);   // [NS][BM][KB]   (HALO: [2][HALO_ROWS][KB], one halo image per 64-channel chunk)
    T *sB = sA + (HALO ? 2 * HALO_ROWS : NS * BM) * KB;   // [NS][BN][KB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    int wm = wave / WGN, wn = wave % WGN;
    if constexpr (PARTN && NT == 512) {   // waves w and w + 4 share SIMD w & 3: give them different column ranges
        if constexpr (WGN == 2) { wm = wave >> 1; wn = (wave ^ (wave >> 2)) & 1; }
        if constexpr (WGN == 4) { wm = wave >> 2; wn = wm ? 3 - (wave & 3) : (wave & 3); }
    }
    unsigned long long t_entry = 0;
    if (p.dbg) t_entry = __builtin_amdgcn_s_memrealtime();
    if (p.stagger > 0 && (int)blockIdx.x < p.stagger_blocks) {
        // Every workgroup of a launch runs the same phases for the same time, and the first round starts together: left alone,
        // all CUs load, then all compute (HBM idle), then all drain (HBM saturated).  Offsetting the first round in 4 groups
        // keeps part of the chip computing while another part moves bytes.  Later workgroups inherit the phase of the slot
        // they take over.  Groups of 8 consecutive blocks (one per XCD) share a phase.
        const int grp = (p.stagger_mode ? ((int)blockIdx.x >> 8) : ((int)blockIdx.x >> 3)) & 3;
        if (grp) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), wait = (unsigned long long)grp * (unsigned)p.stagger;
            while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
        }
    }

    // ---- XCD-aware tile assignment (bijective for any grid size)
    int mt, nt, kslice_id = 0, phase = 0;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r = nblk & 7;
        int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        if (p.ksl > 1) {   // split-K: the slices of one output tile are consecutive blocks (they share the A rows' neighbourhood in L2)
            kslice_id = lid % p.ksl;
            lid /= p.ksl;
        }
        if constexpr (GENERIC) {
            if (p.phases > 1) {   // sub-pixel phases of one tile: consecutive blocks as well (same input rows)
                phase = lid % p.phases;
                lid /= p.phases;
            }
        }
        mt = lid / p.ntiles;
        nt = lid - mt * p.ntiles;
    }
    // The 4 sub-pixel phases (a, b) of a stride-2 transposed conv (k4 s2 p1 = four 2x2 convs) in ONE launch: phase = 2a + b has
    // its own weight block, window origin (pad 1 - a, 1 - b) and output offset (a, b).  A small batch gives each phase only
    // one workgroup per CU; together they are four, which hides the DMA latency of the short tiles.
    const int pad_h = (GENERIC && p.phases > 1) ? 1 - (phase >> 1) : p.pad_h, pad_w = (GENERIC && p.phases > 1) ? 1 - (phase & 1) : p.pad_w;
    const int ooy = (GENERIC && p.phases > 1) ? (phase >> 1) : p.ooy, oox = (GENERIC && p.phases > 1) ? (phase & 1) : p.oox;
    const T *wgt_base = reinterpret_cast<const T *>(p.wgt) + (size_t)phase * p.phase_stride;
    const size_t koffs = (size_t)kslice_id * p.kslice;                          // this block's first reduction index
    char *outv = static_cast<char *>(p.out) + (size_t)kslice_id * p.out_slice * sizeof(float);

    // ---- DMA roles: thread -> (row lrow of each RPS-row pass, physical chunk tid&7); it fetches the
    // LOGICAL chunk kqs so that the lane-linear LDS image ends up XOR-swizzled.
    // swizzle: chunk' = chunk ^ f(row), f(row) = (row >> 1) & 7 for 128-byte rows, (row >> 2) & 3 for 64-byte rows
    // (makes every 16-lane ds_read_b128 group hit 16 distinct 16-byte slots of the 256-byte bank row)
    const int lrow = tid / LPR;
    const int kqs = LPR == 8 ? ((tid & 7) ^ ((tid >> 4) & 7)) : ((tid & 3) ^ ((tid >> 4) & 3));
    const T *zero = reinterpret_cast<const T *>(p.zero);
    // element offset of this lane's 16-byte vector inside a k-step of the A row: 8 consecutive channels per chunk; in the
    // fused split loop chunks 0-3 are 32 hi channels and chunks 4-7 the same 32 channels of the lo plane
    const int koff = x3n ? (kqs & 3) * EPC + (kqs >> 2) * p.x3_plane : EPC * kqs;
    const int kadv = x3n ? KB / 2 : KB;   // channels consumed per k-step
    const T *aptr[AP];
    int astep[AP], hi0[AP], wi0[AP];
    const int HoWo = p.Ho * p.Wo;
    // HALO: tile mt = block (by, bx) of image n; this thread's halo rows r = 64 i + lrow <-> input pixel (16 by - 1 + r / 18,
    // 16 bx - 1 + r % 18); pixels outside the image (and the 4 pad rows) come from the zero page
    const T *hsrc[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t hpix0 = 0;   // index of the block's first output pixel
    if constexpr (HALO) {
        const int tx = p.W >> 4, timg = (p.H >> 4) * tx;
        const int n = mt / timg, trem = mt - n * timg, by = trem / tx, bx = trem - by * tx;
        const T *img = reinterpret_cast<const T *>(p.in) + (size_t)n * p.H * p.W * p.lda + koff;
        hpix0 = ((size_t)n * p.H + by * 16) * p.W + bx * 16;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int r = 64 * i + lrow, hy = r / 18, hx = r - 18 * hy;
            const int iy = by * 16 - 1 + hy, ix = bx * 16 - 1 + hx;
            if (r < 324 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) hsrc[i] = img + ((size_t)iy * p.W + ix) * p.lda;
        }
    }
#pragma unroll
    for (int i = 0; HALO ? false : i < AP; ++i) {
        const int m = mt * BM + i * RPS + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const T *base = reinterpret_cast<const T *>(p.in) + koffs + (size_t)n * p.H * p.W * p.lda;
        hi0[i] = ok ? ho * p.stride - pad_h : -(1 << 28);   // out-of-range rows fail every bounds test
        wi0[i] = wo * p.stride - pad_w;
        if (MODE == MODE_1X1) {
            if (ok) { hi0[i] >>= p.up; wi0[i] >>= p.up; }   // nearest-neighbour upsampled input (HRNet fuse)
            aptr[i] = ok ? base + (hi0[i] * p.W + wi0[i]) * p.lda + koff : zero;
            astep[i] = ok ? kadv : 0;
        } else if (MODE == MODE_TAPS) {
            aptr[i] = base + (ok ? (hi0[i] * p.W + wi0[i]) * p.lda + koff : 0);
            astep[i] = 0;
        } else {
            aptr[i] = base;
            astep[i] = 0;
        }
    }
    const T *wptr[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i)
        wptr[i] = wgt_base + koffs + (size_t)(nt * BN + i * RPS + lrow) * p.ldw + EPC * kqs;

    f32x16 acc[TM][TN];
    // 16-bit result rows (fp16 / (hi, lo) pairs): decides the channel order of the transposed-output accumulators (below)
    const bool out16 = F16 && (p.out_f16 || p.out_split);
    if constexpr (TOUT) {
        // transposed output: a register is ONE output channel for all of the lane's pixels, so the folded-BN shift / bias is
        // simply the accumulator's initial value (scaled up for split layers, whose epilogue scales the accumulator back)
        const float binit = F16 ? 1.f / p.acc_scale : 1.f;   // a power of two: exact
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const float *bp = p.bias + nt * BN + wn * WN + 32 * b;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bq = *reinterpret_cast<const f32x4 *>(bp + (out16 ? 16 * (q >> 1) + 8 * kh + 4 * (q & 1) : 8 * q + 4 * kh));
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[a][b][4 * q + t] = bq[t] * binit;
            }
        }
    } else {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    }

    int tnr = TN;   // 32-column blocks of this wave that hold at least one real output channel (wave-uniform)
    if constexpr (PARTN) {
        tnr = (p.Cout - (nt * BN + wn * WN) + 31) / 32;
        tnr = __builtin_amdgcn_readfirstlane(tnr < 0 ? 0 : (tnr > TN ? TN : tnr));
    }
    int cr = 0, cs = 0, cc = 0, cdelta = 0, ck = 0;   // load cursor (wave-uniform)
    int dtap = 0, dchunk = 0;                          // HALO: (tap, 64-channel chunk) of the next tile the DMA cursor issues
    const int nk = (p.ksl > 1 ? p.kslice : p.Kpad) / KB;

    // issue the DMA of the cursor tile into LDS buffer `buf`, then advance the cursor
#define HMV_DMA(buf)                                                                                        \
    {                                                                                                       \
        int sr_ = 0, ss_ = 0;                                                                               \
        int sc_ = 0;                                                                                        \
        if (MODE == MODE_DENSE) { /* lane-private (tap, channel offset) of this 16-byte vector */           \
            /* fused split loop: 4 (tap, channel) vectors per k-step, chunks 4-7 fetch the same vectors from the lo plane */ \
            const unsigned g_ = (X3CAP && x3n) ? (unsigned)(ck / (2 * EPC) + (kqs & 3)) : (unsigned)(ck / EPC + kqs); \
            const unsigned tap_ = p.cpt == 1 ? g_ : __umulhi(g_, p.cpt_magic);   /* g / (Cin / EPC) */      \
            sc_ = (int)(g_ - tap_ * (unsigned)p.cpt) * EPC;                                                 \
            if (F16 && p.cwrap && sc_ >= p.cwrap) sc_ -= p.cwrap;   /* split operands: third plane = hi again */ \
            if (X3CAP && x3n) sc_ += (kqs >> 2) * p.x3_plane;                                               \
            sr_ = p.S == 1 ? (int)tap_ : (int)__umulhi(tap_, p.s_magic);         /* tap / S */              \
            ss_ = (int)tap_ - sr_ * p.S;                                                                    \
            if (tap_ >= (unsigned)(p.R * p.S)) sr_ = 1 << 29;                                               \
        }                                                                                                   \
        if (HALO && dtap == 0) { /* first tap of a 64-channel chunk: its halo image, once */                \
            _Pragma("unroll") for (int i = 0; i < 6; ++i)                                                   \
                if (i < 5 || wave == 0)                                                                     \
                    HMV_GLDS16(hsrc[i] ? hsrc[i] + dchunk * CH : zero, sA + ((dchunk & 1) * HALO_ROWS + i * 64 + wave * 8) * KB); \
        }                                                                                                   \
        if (HALO) { if (++dtap == 9) { dtap = 0; ++dchunk; } }                                              \
        _Pragma("unroll") for (int i = 0; HALO ? false : i < AP; ++i) {                                     \
            const T *src_;                                                                                  \
            if (MODE == MODE_1X1) {                                                                         \
                src_ = aptr[i];                                                                             \
                aptr[i] += astep[i];                                                                        \
            } else if (MODE == MODE_TAPS) {                                                                 \
                const bool ok_ = (unsigned)(hi0[i] + cr) < (unsigned)p.H && (unsigned)(wi0[i] + cs) < (unsigned)p.W; \
                src_ = ok_ ? aptr[i] + cdelta : zero;                                                       \
            } else {                                                                                        \
                const int hi_ = (hi0[i] + sr_) >> p.up, wi_ = (wi0[i] + ss_) >> p.up;                       \
                const bool ok_ = (unsigned)hi_ < (unsigned)p.H && (unsigned)wi_ < (unsigned)p.W;            \
                src_ = ok_ ? aptr[i] + (hi_ * p.W + wi_) * p.lda + sc_ : zero;                              \
            }                                                                                               \
            HMV_GLDS16(src_, sA + ((buf) * BM + i * RPS + wave * RPW) * KB);                                \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < BP; ++i) {                                                    \
            HMV_GLDS16(wptr[i], sB + ((buf) * BN + i * RPS + wave * RPW) * KB);                             \
            wptr[i] += KB;                                                                                  \
        }                                                                                                   \
        ck += KB;                                                                                           \
        if (MODE == MODE_1X1 && p.in2 && ck == p.ksplit) { /* concatenated reduction: the rest comes from the second source */ \
            _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                                \
                const int m_ = mt * BM + i * RPS + lrow;                                                    \
                if (m_ < p.M) {                                                                             \
                    const int n_ = m_ / HoWo, rem_ = m_ - n_ * HoWo, ho_ = rem_ / p.Wo, wo_ = rem_ - ho_ * p.Wo; \
                    aptr[i] = reinterpret_cast<const T *>(p.in2) + ((size_t)n_ * p.H2 * p.W2 +            \
                              (size_t)(ho_ * p.stride2) * p.W2 + wo_ * p.stride2) * p.lda2 + koff;          \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
        if (F16 && MODE == MODE_1X1 && ck == p.cwrap) { /* split operands: after hi, lo walk the hi plane again */ \
            _Pragma("unroll") for (int i = 0; i < AP; ++i) if (astep[i]) aptr[i] -= p.cwrap;                \
        }                                                                                                   \
        if (MODE == MODE_TAPS) { /* K order: 32-channel chunk slowest, taps fastest (see ConvParams) */     \
            if (KB == CH / 2 && (ck & (CH / 2))) { /* second half of the same tap */                        \
                cdelta += CH / 2;                                                                           \
            } else {                                                                                        \
                if (KB == CH / 2) cdelta -= CH / 2;                                                         \
                cdelta += p.lda;                                                                            \
                if (++cs == p.S) {                                                                          \
                    cs = 0;                                                                                 \
                    cdelta += (p.W - p.S) * p.lda;                                                          \
                    if (++cr == p.R) { cr = 0; cc += (X3CAP && x3n) ? CH / 2 : CH; if (F16 && cc == p.cwrap) cc = 0; cdelta = cc; } \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
    }
    // operand fetch: lane (row l31, k-half kh) reads logical chunk 2q+kh at its swizzled position
    const int fsw = LPR == 8 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
    const T *arow = sA + (wm * WM + l31) * KB;
    // Transposed output with 16-bit results: MFMA row i = (e & 3) + 8 * (e >> 2) + 4 * kh must carry channel
    // (e & 7) + 8 * kh + 16 * (e >> 3) for a lane's registers 8j .. 8j+7 to be 8 CONSECUTIVE channels (one 16-byte store), which
    // is the row index with bits 2 and 3 swapped: lane l31 reads weight row swap23(l31) of its block.  The swap maps each
    // 16-lane ds_read_b128 group onto itself, so the conflict-free property of the XOR swizzle is kept.
    const int wl31 = (TOUT && out16) ? ((l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1)) : l31;
    const int fswb = LPR == 8 ? ((wl31 >> 1) & 7) : ((wl31 >> 2) & 3);
    const T *brow = sB + (wn * WN + wl31) * KB;
    // HALO: the pixel operand of block a at tap (dy, dx) is halo row (py + dy) * 18 + px + dx of the chunk's halo image, with
    // the swizzle of THAT row; (py, px) = the lane's pixel in the 16 x 16 block (block a = pixel rows 2a, 2a + 1 of the wave's
    // WM / 16 rows).  hrow / hsw are set once per k-step (HMV_HALO_SET) for the step whose fragments are read next.
    const T *hrow[TM];
    int hsw[TM];
    const int hbase = (wm * (WM / 16) + (l31 >> 4)) * 18 + (l31 & 15);
#define HMV_HALO_SET(tap, chunk)                                                                            \
    {                                                                                                       \
        const int dy_ = (tap) / 3, toff_ = dy_ * 18 + ((tap) - 3 * dy_);                                    \
        _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                    \
            const int hr_ = hbase + 36 * a + toff_;                                                         \
            hrow[a] = sA + (((chunk) & 1) * HALO_ROWS + hr_) * KB;                                          \
            hsw[a] = (hr_ >> 1) & 7;                                                                        \
        }                                                                                                   \
    }
#define HMV_FRAGS(FA, FB, buf, q)                                                                           \
    {                                                                                                       \
        const int ch_ = ((2 * (q) + kh) ^ fsw) * EPC;                                                       \
        _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                    \
            if (HALO) FA[a] = *reinterpret_cast<const f32x4 *>(hrow[a] + ((2 * (q) + kh) ^ hsw[a]) * EPC);  \
            else if (!PARTN || tnr > 0) FA[a] = *reinterpret_cast<const f32x4 *>(arow + ((buf) * BM + a * 32) * KB + ch_); \
        }                                                                                                   \
        const int chb_ = ((2 * (q) + kh) ^ fswb) * EPC;                                                     \
        _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                      \
            if (!PARTN || b < tnr) FB[b] = *reinterpret_cast<const f32x4 *>(brow + ((buf) * BN + b * 32) * KB + chb_); \
    }
#define HMV_FRAG_A(FA, buf, q)                                                                              \
    {                                                                                                       \
        const int ch_ = ((2 * (q) + kh) ^ fsw) * EPC;                                                       \
        _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                      \
            FA[a] = *reinterpret_cast<const f32x4 *>(arow + ((buf) * BM + a * 32) * KB + ch_);              \
    }
#define HMV_FRAG_B(FB, buf, q)                                                                              \
    {                                                                                                       \
        const int ch_ = ((2 * (q) + kh) ^ fswb) * EPC;                                                      \
        _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                      \
            FB[b] = *reinterpret_cast<const f32x4 *>(brow + ((buf) * BN + b * 32) * KB + ch_);              \
    }
#define HMV_MFMA(FA, FB)                                                                                    \
    {                                                                                                       \
        if constexpr (F16) {                                                                                \
            _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                  \
                _Pragma("unroll") for (int b = 0; b < TN; ++b)                                              \
                    if (!PARTN || b < tnr)                                                                  \
                    acc[a][b] = TOUT ? __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, FB[b]),  \
                                                                       __builtin_bit_cast(f16x8, FA[a]), acc[a][b], 0, 0, 0) \
                                     : __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, FA[a]),  \
                                                                       __builtin_bit_cast(f16x8, FB[b]), acc[a][b], 0, 0, 0); \
        } else {                                                                                            \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                   \
                _Pragma("unroll") for (int a = 0; a < TM; ++a)                                              \
                    _Pragma("unroll") for (int b = 0; b < TN; ++b)                                          \
                        if (!PARTN || b < tnr)                                                              \
                        acc[a][b] = TOUT ? __builtin_amdgcn_mfma_f32_32x32x2f32(FB[b][e], FA[a][e], acc[a][b], 0, 0, 0) \
                                         : __builtin_amdgcn_mfma_f32_32x32x2f32(FA[a][e], FB[b][e], acc[a][b], 0, 0, 0); \
        }                                                                                                   \
    }

    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    unsigned long long t0c = 0, t0r = 0;
    if (p.dbg) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (TOUT) {
        // ---- software L2 prefetch.  A workgroup keeps at most one k-step of operands in flight, and its epilogue meets the
        // residual tile cold: every phase then pays a full HBM round trip (2-3 us under load) per ~64 KB.  gfx950 has no
        // prefetch instruction, so one 4-byte LDS-DMA load per 128-byte line (destination: a 256-byte dummy slot behind
        // the tile buffers, never read) pulls (a) the residual tile and (b) the next k-steps of the activation rows into
        // this XCD's L2 right now, while the first operand tiles are on their way; the real accesses later hit L2.
        // Issued BEFORE the first DMA: loads return in order, so the waits below need no new counts.
        if (p.prefetch && !HALO) {
            float *sdummy = smem + NS * (BM + BN) * KB4;
            if (p.res) {
                constexpr int ebr = F16 ? 2 : 4;                       // residual element bytes (a split row holds two fp16 planes)
                constexpr int lpr = (BN * ebr + 127) / 128;           // 128-byte lines per tile row and plane
                const int planes = (F16 && p.res_split) ? 2 : 1, rowc = (F16 && p.res_split) ? p.ldr >> 1 : p.ldr;
                for (int pl = 0; pl < planes; ++pl)
#pragma unroll
                    for (int L0 = 0; L0 < BM * lpr; L0 += NT) {
                        const int L = L0 + tid, row = L / lpr, piece = L - row * lpr;
                        const int m = mt * BM + row, col = nt * BN + piece * (128 / ebr);
                        const bool ok = L < BM * lpr && m < p.M && col < rowc;
                        const char *src = ok ? reinterpret_cast<const char *>(p.res) + ((size_t)m * p.ldr + (size_t)pl * rowc + col) * ebr
                                             : reinterpret_cast<const char *>(p.zero);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                         (__attribute__((address_space(3))) void *)sdummy, 4, 0, 0);
                    }
            }
            if constexpr (MODE == MODE_1X1 && !X3) {
                // activation rows: the LPR lanes that DMA one row take one further 128-byte line of it each (k-steps 2 ..)
                constexpr int j0 = 2 * KB * (int)sizeof(T) / 128;
                // (a dual-source launch walks only ksplit channels of the first source's rows: never prefetch past them)
                const int off = (j0 + tid % LPR) * 128, kbytes = (p.in2 ? p.ksplit : p.Kpad) * (int)sizeof(T);
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    const bool ok = astep[i] != 0 && off < kbytes && !p.cwrap;
                    const char *src = ok ? reinterpret_cast<const char *>(aptr[i] - koff) + off : reinterpret_cast<const char *>(p.zero);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)sdummy, 4, 0, 0);
                }
            }
        }
    }
    // wait until at most `tiles_` of the DMAs issued so far are still in flight (loads retire in order; wave-uniform argument)
#define HMV_WAIT_TILES(tiles_)                                                                              \
    {                                                                                                       \
        const int w_ = (tiles_);                                                                            \
        if (w_ >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (AP + BP)) : "memory");                   \
        else if (w_ == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AP + BP)) : "memory");              \
        else if (w_ == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + BP) : "memory");                    \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
    }
    if constexpr (NS == 2) {
        HMV_DMA(0);
        if (nk > 1) {
            HMV_DMA(1);
            if constexpr (HALO) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BP) : "memory");   // weights 0 and halo 0 landed, weights 1 may fly
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + BP) : "memory");   // tile 0 landed, tile 1 may fly
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        static_assert(NS == 2 || 3 * (AP + BP) < 64, "vmcnt range");
        const int pre = nk < NS ? nk : NS;   // tiles 0 .. NS-1 go out together; tile 0 must have landed
        for (int t = 0; t < pre; ++t) HMV_DMA(t);
        HMV_WAIT_TILES(pre - 1);
    }
    asm volatile("s_barrier" ::: "memory");
    bool done = false;
    if constexpr (X3CAP) {
        if (x3n) {
            // Fused split reduction: an LDS row holds [hi k0-15 | hi k16-31 | lo k0-15 | lo k16-31] (groups 0..3) of 32
            // channels for A and for W.  Per half h: hi*hi (A[h], W[h]), lo*hi (A[2+h], W[h]), hi*lo (A[h], W[2+h]).
            f32x4 ah[TM], al[TM], bh[TN], bl[TN], ah2[TM], bh2[TN];
            HMV_FRAG_A(ah, 0, 0);
            HMV_FRAG_B(bh, 0, 0);
            for (int kt = 0; kt < nk; ++kt) {
                const int buf = kt & 1;
                HMV_FRAG_A(al, buf, 2);
                HMV_MFMA(ah, bh);
                __builtin_amdgcn_sched_barrier(0);
                HMV_FRAG_B(bl, buf, 2);
                HMV_MFMA(al, bh);
                __builtin_amdgcn_sched_barrier(0);
                HMV_FRAG_A(ah2, buf, 1);
                HMV_FRAG_B(bh2, buf, 1);
                HMV_MFMA(ah, bl);
                __builtin_amdgcn_sched_barrier(0);
                HMV_FRAG_A(al, buf, 3);
                HMV_MFMA(ah2, bh2);
                __builtin_amdgcn_sched_barrier(0);
                HMV_FRAG_B(bl, buf, 3);
                HMV_MFMA(al, bh2);
                __builtin_amdgcn_sched_barrier(0);
                // tile kt+1 (the only DMA in flight) must have landed; everyone is done reading `buf`
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (kt + 2 < nk) HMV_DMA(buf);
                if (kt + 1 < nk) { HMV_FRAG_A(ah, buf ^ 1, 0); HMV_FRAG_B(bh, buf ^ 1, 0); }
                HMV_MFMA(ah2, bl);
                __builtin_amdgcn_sched_barrier(0);
            }
            done = true;
        }
    }
    int htap = 0, hchunk = 0;   // HALO: (tap, chunk) of the k-step whose fragments are read next
    if constexpr (HALO) HMV_HALO_SET(0, 0);
    if (!done) HMV_FRAGS(fa0, fb0, 0, 0);

    for (int kt = 0; !done && kt < nk; ++kt) {
        const int buf = kt & (NS - 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {   // fragment sets alternate by the parity of q (NQ is even)
            if (q + 1 < NQ) {
                if (q & 1) { HMV_FRAGS(fa0, fb0, buf, q + 1); } else { HMV_FRAGS(fa1, fb1, buf, q + 1); }
            } else if constexpr (NS == 2) {
                // tile kt+1 (the only DMA in flight) must have landed; everyone is done reading `buf`
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (kt + 2 < nk) HMV_DMA(buf);
                if constexpr (HALO) {   // the next k-step: next tap of the same halo image, or tap 0 of the next chunk's
                    if (++htap == 9) { htap = 0; ++hchunk; }
                    HMV_HALO_SET(htap, hchunk);
                }
                if (kt + 1 < nk) HMV_FRAGS(fa0, fb0, buf ^ 1, 0);
            } else {
                // ring: tiles up to kt+NS-1 are out; tile kt+1 must have landed, kt+2 .. kt+NS-1 (as far as they exist) may fly
                const int fly = nk - kt - 2;
                HMV_WAIT_TILES(fly < 0 ? 0 : (fly > NS - 2 ? NS - 2 : fly));
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everyone is done reading `buf`
                if (kt + NS < nk) HMV_DMA(buf);
                if (kt + 1 < nk) HMV_FRAGS(fa0, fb0, (kt + 1) & (NS - 1), 0);
            }
            if (q & 1) { HMV_MFMA(fa1, fb1); } else { HMV_MFMA(fa0, fb0); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef HMV_DMA
#undef HMV_WAIT_TILES
#undef HMV_HALO_SET
#undef HMV_FRAGS
#undef HMV_FRAG_A
#undef HMV_FRAG_B
#undef HMV_MFMA
    unsigned long long t1c = 0, t1r = 0;
    if (p.dbg) { t1c = __builtin_amdgcn_s_memtime(); t1r = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (TOUT) {
        // ---- epilogue straight from the accumulators (transposed output, see TOUT above).  Lane (l31, kh) owns pixel
        // m = tile row l31 of each 32-pixel block a; block b's 16 registers are 16 channels of that pixel:
        //   32-bit rows: register 4q + t  = channel 8q + 4kh + t            (4 consecutive channels per register group)
        //   16-bit rows: register 8j + t  = channel 16j + 8kh + t           (8 consecutive channels, weight rows read swapped)
        // The bias is already in the accumulators (their initial value); the residual of a later block is in flight while a
        // block is finished and stored.  No LDS, no barrier: waves drain independently.
        const int nb0 = nt * BN + wn * WN;
        const bool has_res = p.res != nullptr;
        const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
        const int mrow0 = mt * BM + wm * WM + l31;
        constexpr int NBLK = TM * TN;
        // ---- residual burst.  After the last barrier of the main loop nobody reads the tile buffers any more, so each wave
        // turns its 1/NW share of them into a private landing zone and requests the residual of ALL its blocks at once by
        // LDS-DMA (16 bytes per lane, the lane's own pixel and channel group as source): up to 128 KB per CU in flight with
        // no registers spent, instead of a few loads per wave.  A block is then finished as soon as its pieces have landed
        // (counted vmcnt: the stores issued meanwhile are younger than the DMAs they must not wait for) and read back with
        // one ds_read_b128 per piece (lane-linear image, conflict free).  Wave-private: no barrier.
        if (has_res && p.burst) {
            constexpr int WLB = NS * (BM + BN) * KB * (int)sizeof(T) / (NT / 64);   // bytes of tile buffer per wave
            char *wl = reinterpret_cast<char *>(smem) + wave * WLB;
            float *trash = const_cast<float *>(p.zero) + 64 + 4 * lane;   // 16 bytes per lane behind the zero page
            if constexpr (!F16) {
                constexpr int PC = 4, CB = burst_blocks(WLB, PC, NBLK);   // pieces / block, blocks / chunk
                static_assert(CB >= 1 && NBLK % CB == 0, "residual landing zone");
                const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 3) & ~3);
#pragma unroll
                for (int c0 = 0; c0 < NBLK; c0 += CB) {
#pragma unroll
                    for (int i = 0; i < CB; ++i) {
                        const int g = c0 + i, a = g / TN, b = g % TN, m = mrow0 + 32 * a;
#pragma unroll
                        for (int q = 0; q < PC; ++q) {
                            const int col = nb0 + 32 * b + 8 * q + 4 * kh;
                            const float *src = (m < p.M && col < cend) ? reinterpret_cast<const float *>(p.res) + (size_t)m * p.ldr + col : p.zero;
                            HMV_GLDS16(src, wl + (i * PC + q) * 1024);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < CB; ++i) {
                        const int g = c0 + i, a = g / TN, b = g % TN, m = mrow0 + 32 * a;
                        // pieces of block i landed when at most the (CB - 1 - i) * PC younger DMAs + the i * PC stores since are out
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((CB - 1) * PC) : "memory");
                        float *orow = reinterpret_cast<float *>(outv) + (size_t)m * p.ldc;
#pragma unroll
                        for (int q = 0; q < PC; ++q) {
                            const int col = nb0 + 32 * b + 8 * q + 4 * kh;
                            const f32x4 r = *reinterpret_cast<const f32x4 *>(wl + (i * PC + q) * 1024 + lane * 16);
                            f32x4 t;
#pragma unroll
                            for (int j = 0; j < 4; ++j) t[j] = fmaxf(acc[a][b][4 * q + j] + r[j], lo);
                            // every lane stores (the vmcnt arithmetic above counts on it): lanes outside the tensor hit the trash page
                            float *dst = (m < p.M && col < cend) ? orow + col : trash;
                            *reinterpret_cast<f32x4 *>(dst) = t;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zone is rewritten by the next chunk's DMA
                }
                if (p.dbg && tid == 0) {
                    unsigned long long *d = p.dbg + 8 * (size_t)blockIdx.x;
                    d[0] = t1c - t0c; d[1] = t1r - t0r; d[2] = t_entry; d[3] = t0r; d[4] = t1r; d[5] = __builtin_amdgcn_s_memrealtime();
                    unsigned hwid, xcc;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                    d[6] = hwid; d[7] = xcc;
                }
                return;
            } else if (out16 && (p.res_split != 0) == (p.out_split != 0)) {
                const int rowc = p.out_split ? p.ldc >> 1 : p.ldc, rplane = p.ldr >> 1;
                const int cend = (p.fill || p.Cout + 3 >= rowc) ? rowc : ((p.Cout + 7) & ~7);
                auto run = [&](auto split_tag) {
                    constexpr bool SPLIT = decltype(split_tag)::value;
                    constexpr int PC = SPLIT ? 4 : 2, CB = burst_blocks(WLB, PC, NBLK);
                    static_assert(CB >= 1 && NBLK % CB == 0, "residual landing zone");
#pragma unroll
                    for (int c0 = 0; c0 < NBLK; c0 += CB) {
#pragma unroll
                        for (int i = 0; i < CB; ++i) {
                            const int g = c0 + i, a = g / TN, b = g % TN, m = mrow0 + 32 * a;
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int col = nb0 + 32 * b + 16 * j + 8 * kh;
                                const bool lv = m < p.M && col < cend;
                                const _Float16 *src = lv ? reinterpret_cast<const _Float16 *>(p.res) + (size_t)m * p.ldr + col
                                                         : reinterpret_cast<const _Float16 *>(p.zero);
                                HMV_GLDS16(src, wl + (i * PC + j) * 1024);
                                if constexpr (SPLIT) HMV_GLDS16(lv ? src + rplane : src, wl + (i * PC + 2 + j) * 1024);
                            }
                        }
#pragma unroll
                        for (int i = 0; i < CB; ++i) {
                            const int g = c0 + i, a = g / TN, b = g % TN, m = mrow0 + 32 * a;
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((CB - 1) * PC) : "memory");   // PC stores per block follow PC DMAs
                            _Float16 *orow = reinterpret_cast<_Float16 *>(outv) + (size_t)m * p.ldc;
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int col = nb0 + 32 * b + 16 * j + 8 * kh;
                                const f16x8 rh = *reinterpret_cast<const f16x8 *>(wl + (i * PC + j) * 1024 + lane * 16);
                                f16x8 rl = {0, 0, 0, 0, 0, 0, 0, 0};
                                if constexpr (SPLIT) rl = *reinterpret_cast<const f16x8 *>(wl + (i * PC + 2 + j) * 1024 + lane * 16);
                                float t[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    float r = (float)rh[u];
                                    if constexpr (SPLIT) r += (float)rl[u];
                                    t[u] = fmaxf(acc[a][b][8 * j + u] * p.acc_scale + r, lo);
                                }
                                const bool st = m < p.M && col < cend;
                                if constexpr (SPLIT) {
                                    f16x8 hv, lv8;
#pragma unroll
                                    for (int u = 0; u < 8; ++u) {
                                        const float c = fminf(fmaxf(t[u], -65504.f), 65504.f);
                                        hv[u] = (_Float16)c;
                                        lv8[u] = (_Float16)(c - (float)hv[u]);
                                    }
                                    // every lane stores twice (the vmcnt arithmetic counts on it): lanes outside the tensor hit the trash page
                                    _Float16 *dst = st ? orow + col : reinterpret_cast<_Float16 *>(trash);
                                    *reinterpret_cast<f16x8 *>(dst) = hv;
                                    *reinterpret_cast<f16x8 *>(st ? dst + rowc : dst) = lv8;
                                } else {
                                    f16x8 hv;
#pragma unroll
                                    for (int u = 0; u < 8; ++u) hv[u] = (_Float16)t[u];
                                    *reinterpret_cast<f16x8 *>(st ? orow + col : reinterpret_cast<_Float16 *>(trash)) = hv;
                                }
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                };
                if (p.res_split) run(std::true_type{}); else run(std::false_type{});
                if (p.dbg && tid == 0) {
                    unsigned long long *d = p.dbg + 8 * (size_t)blockIdx.x;
                    d[0] = t1c - t0c; d[1] = t1r - t0r; d[2] = t_entry; d[3] = t0r; d[4] = t1r; d[5] = __builtin_amdgcn_s_memrealtime();
                    unsigned hwid, xcc;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                    d[6] = hwid; d[7] = xcc;
                }
                return;
            }
        }
        // ---- register path (no residual, or a residual / output format pair the burst does not cover).
        // blocks are walked in the order g = a * TN + b; the residual of block g + PF is requested before block g is
        // finished (ring of PF + 1 register sets, indices static after unrolling)
        constexpr int PF = 1, RING = PF + 1;
        // output row of tile row m: the row itself, or (HALO) pixel (l >> 4, l & 15) of the tile's 16 x 16 block
        auto prow = [&](int m) -> size_t {
            if constexpr (HALO) { const int l = m - mt * BM; return hpix0 + (size_t)(l >> 4) * p.W + (l & 15); }
            else return (size_t)m;
        };
        if (!out16) {
            const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 3) & ~3);
            f32x4 rv[RING][4];
#pragma unroll
            for (int g = -PF; g < NBLK; ++g) {
                if (g + PF < NBLK) {   // residual of block g + PF (the zero page stands in for "no residual" / rows past M)
                    const int gg = g + PF, a = gg / TN, b = gg % TN;
                    const int m = mrow0 + 32 * a;
                    const bool live = has_res && m < p.M;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = nb0 + 32 * b + 8 * q + 4 * kh;
                        const bool lv = live && col < cend;
                        if constexpr (F16) {
                            const _Float16 *rp = lv ? reinterpret_cast<const _Float16 *>(p.res) + (size_t)m * p.ldr + col
                                                    : reinterpret_cast<const _Float16 *>(p.zero);
                            const f16x4 hv = *reinterpret_cast<const f16x4 *>(rp);
                            rv[gg % RING][q] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                            if (p.res_split) {
                                const f16x4 l4 = *reinterpret_cast<const f16x4 *>(lv ? rp + (p.ldr >> 1) : rp);
                                rv[gg % RING][q] += f32x4{(float)l4[0], (float)l4[1], (float)l4[2], (float)l4[3]};
                            }
                        } else {
                            const float *rp = lv ? reinterpret_cast<const float *>(p.res) + (size_t)m * p.ldr + col : p.zero;
                            rv[gg % RING][q] = *reinterpret_cast<const f32x4 *>(rp);
                        }
                    }
                }
                if (g >= 0) {
                    const int a = g / TN, b = g % TN;
                    const int m = mrow0 + 32 * a;
                    float *orow = reinterpret_cast<float *>(outv) + prow(m) * p.ldc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = nb0 + 32 * b + 8 * q + 4 * kh;
                        f32x4 t;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if constexpr (F16) t[j] = fmaxf(acc[a][b][4 * q + j] * p.acc_scale + rv[g % RING][q][j], lo);   // scale 1.0 unless pre-scaled weights
                            else t[j] = fmaxf(acc[a][b][4 * q + j] + rv[g % RING][q][j], lo);
                        }
                        if (m < p.M && col < cend) *reinterpret_cast<f32x4 *>(orow + col) = t;
                    }
                }
            }
        } else {
            if constexpr (F16) {
                const int rowc = p.out_split ? p.ldc >> 1 : p.ldc;   // columns of one plane / of the row
                const int cend = (p.fill || p.Cout + 3 >= rowc) ? rowc : ((p.Cout + 7) & ~7);
                const int rplane = p.ldr >> 1;
                f16x8 rh[RING][2], rl[RING][2];
#pragma unroll
                for (int g = -PF; g < NBLK; ++g) {
                    if (g + PF < NBLK) {
                        const int gg = g + PF, a = gg / TN, b = gg % TN;
                        const int m = mrow0 + 32 * a;
                        const bool live = has_res && m < p.M;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int col = nb0 + 32 * b + 16 * j + 8 * kh;
                            const bool lv = live && col < cend;
                            const _Float16 *rp = lv ? reinterpret_cast<const _Float16 *>(p.res) + (size_t)m * p.ldr + col
                                                    : reinterpret_cast<const _Float16 *>(p.zero);
                            rh[gg % RING][j] = *reinterpret_cast<const f16x8 *>(rp);
                            if (p.res_split) rl[gg % RING][j] = *reinterpret_cast<const f16x8 *>(lv ? rp + rplane : rp);
                        }
                    }
                    if (g >= 0) {
                        const int a = g / TN, b = g % TN;
                        const int m = mrow0 + 32 * a;
                        _Float16 *orow = reinterpret_cast<_Float16 *>(outv) + prow(m) * p.ldc;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int col = nb0 + 32 * b + 16 * j + 8 * kh;
                            float t[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                float r = (float)rh[g % RING][j][u];
                                if (p.res_split) r += (float)rl[g % RING][j][u];
                                t[u] = fmaxf(acc[a][b][8 * j + u] * p.acc_scale + r, lo);
                            }
                            if (!(m < p.M && col < cend)) continue;
                            if (p.out_split) {   // fp32 value -> (hi, lo) fp16 pair, hi + lo == value to ~2^-22
                                f16x8 hv, lv8;
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const float c = fminf(fmaxf(t[u], -65504.f), 65504.f);
                                    hv[u] = (_Float16)c;
                                    lv8[u] = (_Float16)(c - (float)hv[u]);
                                }
                                *reinterpret_cast<f16x8 *>(orow + col) = hv;
                                *reinterpret_cast<f16x8 *>(orow + col + rowc) = lv8;
                            } else {
                                f16x8 hv;
#pragma unroll
                                for (int u = 0; u < 8; ++u) hv[u] = (_Float16)t[u];
                                *reinterpret_cast<f16x8 *>(orow + col) = hv;
                            }
                        }
                    }
                }
            }
        }
        if (p.dbg && tid == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long *d = p.dbg + 8 * (size_t)blockIdx.x;
            d[0] = t1c - t0c; d[1] = t1r - t0r; d[2] = t_entry; d[3] = t0r; d[4] = t1r;
            d[5] = __builtin_amdgcn_s_memrealtime(); d[6] = hwid; d[7] = xcc;
        }
        return;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- epilogue: staged through LDS in passes of AS 32-row blocks per wave, then 16-byte row stores.
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    float *sC = smem;
    const int n0 = nt * BN;
    const bool vec = ((p.ldc & 3) == 0) && (p.res == nullptr || (p.ldr & 3) == 0);
#pragma unroll
    for (int pass = 0; pass < TM / AS; ++pass) {
        if (pass > 0) __syncthreads();   // the previous pass has been drained
#pragma unroll
        for (int aa = 0; aa < AS; ++aa)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int srow = (wm * AS + aa) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    sC[srow * LDC + wn * WN + b * 32 + l31] = acc[pass * AS + aa][b][e];
                }
        __syncthreads();
        // staged row sr belongs to wave-row sr / (AS*32), block (sr / 32) % AS, line sr % 32
        auto tile_row = [&](int sr) { return (sr / (AS * 32)) * WM + (pass * AS + (sr / 32) % AS) * 32 + (sr & 31); };
        auto out_row = [&](int m) -> size_t {   // sub-pixel scatter of the transposed convolution
            const int n = m / HoWo, rem = m - n * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            return ((size_t)n * (p.Ho * p.osy) + (ho * p.osy + ooy)) * (size_t)(p.Wo * p.osx) + (wo * p.osx + oox);
        };
        if constexpr (RD) {
            // AS == TM: staged row == tile row.  One thread per 4 output channels of one pixel.
            const int cv = p.rd_cout >> 2;
            const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
            for (int idx = tid; idx < BM * cv; idx += NT) {
                const int r = idx / cv, c4 = idx - r * cv;
                const int m = mt * BM + r;
                if (m >= p.M) continue;
                const int wo = m % p.Wo;
                f32x4 t = *reinterpret_cast<const f32x4 *>(p.bias + 4 * c4);
#pragma unroll
                for (int sx = 0; sx < 3; ++sx)
                    if ((unsigned)(wo + sx - 1) < (unsigned)p.Wo)
                        t += *reinterpret_cast<const f32x4 *>(&sC[(r + sx - 1) * LDC + sx * p.rd_cout + 4 * c4]);
                if (p.res) {
                    if (p.res_f16) {
                        const f16x4 hv = *reinterpret_cast<const f16x4 *>(reinterpret_cast<const _Float16 *>(p.res) + (size_t)m * p.ldr + 4 * c4);
                        t += f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                    } else {
                        t += *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(p.res) + (size_t)m * p.ldr + 4 * c4);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = fmaxf(t[j], lo);
                if (p.out_f16)
                    *reinterpret_cast<f16x4 *>(reinterpret_cast<_Float16 *>(outv) + (size_t)m * p.ldc + 4 * c4) =
                        f16x4{(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
                else
                    *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(outv) + (size_t)m * p.ldc + 4 * c4) = t;
            }
        } else if (vec) {
            if constexpr (!F16) {   // the fp32 kernels keep their hand-tuned 4-wide drain (any restructuring here costs ~3 %)
            constexpr int TPR = BN / 4, RPP = NT / TPR, NPASS = SR / RPP, UB = NPASS < HMV_UB ? NPASS : HMV_UB;
            static_assert(TOUT || (SR % RPP == 0 && NPASS % UB == 0), "staging pass shape");   // (the register-epilogue kernels never get here)
            const int c4 = tid % TPR, r0 = tid / TPR;
            const int col = n0 + 4 * c4;
            // columns [Cout, round4(Cout)) hold exact zeros (zero-padded weights and bias): writing them is
            // harmless whenever the row stride leaves room, which lets Cout = 21 use vector stores too.
            const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 3) & ~3);
            if (col < cend) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + col);
                const bool has_res = p.res != nullptr;
                const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
#pragma unroll
                for (int g = 0; g < NPASS; g += UB) {
                    f32x4 v[UB], rv[UB];
                    size_t orow[UB];
                    bool okr[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int sr = r0 + (g + u) * RPP, m = mt * BM + tile_row(sr);
                        okr[u] = m < p.M;
                        orow[u] = (size_t)m;
                        size_t rrow = (size_t)m;
                        if (GENERIC) {
                            if (p.scatter) orow[u] = out_row(m);
                            if (p.rg_out) rrow = (size_t)(m / p.rg_out) * p.rg_in + (m % p.rg_out);
                        }
                        if (p.res_f16) {   // fp16 residual: 4 halfs = 8 bytes
                            const _Float16 *rp = (has_res && okr[u]) ? reinterpret_cast<const _Float16 *>(p.res) + rrow * p.ldr + col
                                                                      : reinterpret_cast<const _Float16 *>(p.zero);
                            const f16x4 hv = *reinterpret_cast<const f16x4 *>(rp);
                            rv[u] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                            if (p.res_split) {   // [hi | lo] pair: the lo plane sits ldr / 2 further (the zero page is all zeros)
                                const f16x4 lv = *reinterpret_cast<const f16x4 *>((has_res && okr[u]) ? rp + (p.ldr >> 1) : rp);
                                rv[u] += f32x4{(float)lv[0], (float)lv[1], (float)lv[2], (float)lv[3]};
                            }
                        } else {
                            const float *rp = (has_res && okr[u]) ? reinterpret_cast<const float *>(p.res) + rrow * p.ldr + col : p.zero;
                            rv[u] = *reinterpret_cast<const f32x4 *>(rp);
                        }
                        v[u] = *reinterpret_cast<const f32x4 *>(&sC[sr * LDC + 4 * c4]);
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        f32x4 t = v[u] + bv + rv[u];
                        if constexpr (F16) t = v[u] * p.acc_scale + bv + rv[u];   // 1.0 unless the layer's weights were pre-scaled
                        if (GENERIC) {
                            if (p.act == ACT_GELU) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) t[j] = 0.5f * t[j] * (1.f + erff(t[j] * 0.70710678118654752440f));
                            } else if (p.act == ACT_LEAKY) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) t[j] = t[j] > 0.f ? t[j] : 0.01f * t[j];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[j] = fmaxf(t[j], lo);
                        if (okr[u]) {
                            if (p.out_split) {   // fp32 value -> (hi, lo) fp16 pair, hi + lo == value to ~2^-22
                                f16x4 hv, lv;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const float c = fminf(fmaxf(t[j], -65504.f), 65504.f);
                                    hv[j] = (_Float16)c;
                                    lv[j] = (_Float16)(c - (float)hv[j]);
                                }
                                _Float16 *op = reinterpret_cast<_Float16 *>(outv) + orow[u] * p.ldc + col;
                                *reinterpret_cast<f16x4 *>(op) = hv;
                                *reinterpret_cast<f16x4 *>(op + (p.ldc >> 1)) = lv;
                            } else if (p.out_f16)
                                *reinterpret_cast<f16x4 *>(reinterpret_cast<_Float16 *>(outv) + orow[u] * p.ldc + col) =
                                    f16x4{(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
                            else
                                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(outv) + orow[u] * p.ldc + col) = t;
                        }
                    }
                }
            }
            } else {
            // W output columns per thread: 4 (one 16-byte fp32 vector) or, for 16-bit outputs of the fp16 kernels, 8 (so that
            // fp16 / (hi, lo) rows are written and fp16 residuals read as 16-byte vectors too: 8-byte accesses capped the
            // residual-bearing fp16 layers at ~3 TB/s)
            auto drain = [&](auto wtag) {
                constexpr int W = decltype(wtag)::value, W4 = W / 4;
                constexpr int TPR = BN / W, RPP = NT / TPR, NPASS = SR / RPP, UB = NPASS < HMV_UB_F16 ? NPASS : HMV_UB_F16;
                static_assert(TOUT || (SR % RPP == 0 && NPASS % UB == 0), "staging pass shape");   // (the register-epilogue kernels never get here)
                const int cw = tid % TPR, r0 = tid / TPR;
                const int col = n0 + W * cw;
                // columns [Cout, round4(Cout)) hold exact zeros (zero-padded weights and bias): writing them is
                // harmless whenever the row stride leaves room, which lets Cout = 21 use vector stores too.
                const int rowc = p.out_split ? p.ldc >> 1 : p.ldc;   // columns of one plane / of the row
                const int cend = (p.fill || p.Cout + 3 >= rowc) ? rowc : ((p.Cout + W - 1) & ~(W - 1));
                if (col >= cend) return;
                f32x4 bv[W4];
#pragma unroll
                for (int q = 0; q < W4; ++q) bv[q] = *reinterpret_cast<const f32x4 *>(p.bias + col + 4 * q);
                const bool has_res = p.res != nullptr;
                const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
#pragma unroll
                for (int g = 0; g < NPASS; g += UB) {
                    f32x4 v[UB][W4], rv[UB][W4];
                    size_t orow[UB];
                    bool okr[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int sr = r0 + (g + u) * RPP, m = mt * BM + tile_row(sr);
                        okr[u] = m < p.M;
                        orow[u] = (size_t)m;
                        size_t rrow = (size_t)m;
                        if (GENERIC) {
                            if (p.scatter) orow[u] = out_row(m);
                            if (p.rg_out) rrow = (size_t)(m / p.rg_out) * p.rg_in + (m % p.rg_out);
                        }
                        if (p.res_f16) {   // fp16 residual (the zero page stands in for "no residual" / rows past M)
                            const bool live = has_res && okr[u];
                            const _Float16 *rp = live ? reinterpret_cast<const _Float16 *>(p.res) + rrow * p.ldr + col
                                                      : reinterpret_cast<const _Float16 *>(p.zero);
#pragma unroll
                            for (int q = 0; q < W4; ++q) {
                                const f16x4 hv = *reinterpret_cast<const f16x4 *>(rp + 4 * q);
                                rv[u][q] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                            }
                            if (p.res_split) {   // [hi | lo] pair: the lo plane sits ldr / 2 further
                                const _Float16 *lp = live ? rp + (p.ldr >> 1) : rp;
#pragma unroll
                                for (int q = 0; q < W4; ++q) {
                                    const f16x4 lv = *reinterpret_cast<const f16x4 *>(lp + 4 * q);
                                    rv[u][q] += f32x4{(float)lv[0], (float)lv[1], (float)lv[2], (float)lv[3]};
                                }
                            }
                        } else {
                            const float *rp = (has_res && okr[u]) ? reinterpret_cast<const float *>(p.res) + rrow * p.ldr + col : p.zero;
#pragma unroll
                            for (int q = 0; q < W4; ++q) rv[u][q] = *reinterpret_cast<const f32x4 *>(rp + 4 * q);
                        }
#pragma unroll
                        for (int q = 0; q < W4; ++q) v[u][q] = *reinterpret_cast<const f32x4 *>(&sC[sr * LDC + W * cw + 4 * q]);
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        f32x4 t[W4];
#pragma unroll
                        for (int q = 0; q < W4; ++q) {
                            t[q] = v[u][q] + bv[q] + rv[u][q];
                            if constexpr (F16) t[q] = v[u][q] * p.acc_scale + bv[q] + rv[u][q];   // 1.0 unless the weights were pre-scaled
                            if (GENERIC) {
                                if (p.act == ACT_GELU) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) t[q][j] = 0.5f * t[q][j] * (1.f + erff(t[q][j] * 0.70710678118654752440f));
                                } else if (p.act == ACT_LEAKY) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) t[q][j] = t[q][j] > 0.f ? t[q][j] : 0.01f * t[q][j];
                                }
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) t[q][j] = fmaxf(t[q][j], lo);
                        }
                        if (!okr[u]) continue;
                        if (p.out_split) {   // fp32 value -> (hi, lo) fp16 pair, hi + lo == value to ~2^-22
                            _Float16 hv[W], lv[W];
#pragma unroll
                            for (int j = 0; j < W; ++j) {
                                const float c = fminf(fmaxf(t[j >> 2][j & 3], -65504.f), 65504.f);
                                hv[j] = (_Float16)c;
                                lv[j] = (_Float16)(c - (float)hv[j]);
                            }
                            _Float16 *op = reinterpret_cast<_Float16 *>(outv) + orow[u] * p.ldc + col;
                            if constexpr (W == 8) {
                                *reinterpret_cast<f16x8 *>(op) = f16x8{hv[0], hv[1], hv[2], hv[3], hv[4], hv[5], hv[6], hv[7]};
                                *reinterpret_cast<f16x8 *>(op + (p.ldc >> 1)) = f16x8{lv[0], lv[1], lv[2], lv[3], lv[4], lv[5], lv[6], lv[7]};
                            } else {
                                *reinterpret_cast<f16x4 *>(op) = f16x4{hv[0], hv[1], hv[2], hv[3]};
                                *reinterpret_cast<f16x4 *>(op + (p.ldc >> 1)) = f16x4{lv[0], lv[1], lv[2], lv[3]};
                            }
                        } else if (p.out_f16) {
                            _Float16 *op = reinterpret_cast<_Float16 *>(outv) + orow[u] * p.ldc + col;
                            if constexpr (W == 8)
                                *reinterpret_cast<f16x8 *>(op) = f16x8{(_Float16)t[0][0], (_Float16)t[0][1], (_Float16)t[0][2], (_Float16)t[0][3],
                                                                       (_Float16)t[W4 - 1][0], (_Float16)t[W4 - 1][1], (_Float16)t[W4 - 1][2], (_Float16)t[W4 - 1][3]};
                            else
                                *reinterpret_cast<f16x4 *>(op) = f16x4{(_Float16)t[0][0], (_Float16)t[0][1], (_Float16)t[0][2], (_Float16)t[0][3]};
                        } else {
#pragma unroll
                            for (int q = 0; q < W4; ++q)
                                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(outv) + orow[u] * p.ldc + col + 4 * q) = t[q];
                        }
                    }
                }
            };
            bool wide = false;
            if constexpr (F16 && !GENERIC && BN % 8 == 0 && (NT % (BN / 8)) == 0)
                wide = (p.out_f16 || p.out_split) && ((p.out_split ? p.ldc >> 1 : p.ldc) & 7) == 0 &&
                       (p.res == nullptr || ((p.res_split ? p.ldr >> 1 : p.ldr) & 7) == 0);
            if constexpr (F16 && !GENERIC && BN % 8 == 0 && (NT % (BN / 8)) == 0) {
                if (wide) drain(std::integral_constant<int, 8>{});
                else drain(std::integral_constant<int, 4>{});
            } else {
                drain(std::integral_constant<int, 4>{});
            }
            }
        } else if (GENERIC) {   // scalar fallback (row strides that are not multiples of 4, e.g. the 21x3 output)
            for (int idx = tid; idx < SR * BN; idx += NT) {
                const int sr = idx / BN, c = idx - sr * BN;
                const int m = mt * BM + tile_row(sr), col = n0 + c;
                if (m >= p.M || col >= p.Cout) continue;
                const size_t orow = p.scatter ? out_row(m) : (size_t)m;
                const size_t rrow = p.rg_out ? (size_t)(m / p.rg_out) * p.rg_in + (m % p.rg_out) : (size_t)m;
                float v = sC[sr * LDC + c] + p.bias[col];
                if (p.res) v += p.res_f16 ? (float)reinterpret_cast<const _Float16 *>(p.res)[rrow * p.ldr + col]
                                          : reinterpret_cast<const float *>(p.res)[rrow * p.ldr + col];
                if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (p.act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
                else if (p.act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
                if (p.out_f16) reinterpret_cast<_Float16 *>(outv)[orow * p.ldc + col] = (_Float16)v;
                else reinterpret_cast<float *>(outv)[orow * p.ldc + col] = v;
            }
        }
    }
    // diagnostic stamps (tools/timeline.py): main-loop shader cycles / 100 MHz ticks, block timeline, CU id
    if (p.dbg && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *d = p.dbg + 8 * (size_t)blockIdx.x;
        d[0] = t1c - t0c; d[1] = t1r - t0r; d[2] = t_entry; d[3] = t0r; d[4] = t1r;
        d[5] = __builtin_amdgcn_s_memrealtime(); d[6] = hwid; d[7] = xcc;
    }
}

// ====================================================================== host side
int conv_tile_bn(ConvTile t) {
    switch (t) {
        case TILE_128x32: return 32;
        case TILE_128x64: case TILE_64x64: return 64;
        case TILE_128x256: case TILE_256x256: case TILE_128x256_K16: case TILE_256x256_RING: return 256;
        default: return 128;
    }
}

// Family name = one rocprofv3 symbol: conv_igemm<T, BM, BN, WGM, WGN, MODE, false, KB>
static const char *kTileShape[TILE_COUNT] = {"128x32", "128x64", "128x128", "256x128", "128x256", "256x256", "128x128,k16",
                                              "128x256,k16", "256x128,k16", "64x64", "256x128,k16,w8", "256x256,ring"};
static const char *tile_name(const char *dtype, ConvTile t, int mode, bool partn = false, bool rd = false) {
    static char names[2][TILE_COUNT][3][3][56];
    if (t < 0 || t >= TILE_COUNT || mode < 0 || mode > 2) return "conv_igemm<?>";
    const int d = dtype[1] == '1';   // "f16" / "f32"
    char *n = names[d][t][mode][rd ? 2 : (partn ? 1 : 0)];
    if (!n[0])
        snprintf(n, sizeof(names[0][0][0][0]), "conv_igemm_%s<%s,%s%s>", dtype, kTileShape[t],
                 mode == 0 ? "taps" : (mode == 1 ? "1x1" : "dense"), rd ? ",rowsum" : (partn ? ",skipN" : ""));
    return n;
}
// fp16 has no half-step variants of the big tiles: they run as their full-step tile
const char *conv_tile_name_f16(ConvTile t, int mode) {
    return tile_name("f16", t, mode);
}
const char *conv_tile_name(ConvTile t, int mode) { return tile_name("f32", t, mode); }

// the tile a dense-K launch really uses (fewer instantiations than the chunked modes)
ConvTile conv_dense_tile(ConvTile t, bool f16) {
    switch (t) {
        case TILE_128x32: case TILE_128x64: case TILE_64x64: case TILE_256x128: return t;
        case TILE_256x128_K16: case TILE_256x256: case TILE_256x256_RING: return TILE_256x128;
        default: return TILE_128x128;
    }
}
// true when the tile has a block-skipping instantiation and the last N-tile has an all-padding 32-column block
bool conv_partial_n(ConvTile t, int Cout) {
    if (t != TILE_128x128 && t != TILE_256x128 && t != TILE_256x256) return false;
    const int bn = conv_tile_bn(t);
    return (Cout + bn - 1) / bn * bn - Cout >= 32;
}

ConvTile conv_pick_tile(int M, int Cout, int K, bool f16, bool has_res) {
    static int forced = -2;   // development knob: HMV_FORCE_TILE=<ConvTile> for layers with Cout > 64
    if (forced == -2) { const char *e = HMV_DEV_ENV("HMV_FORCE_TILE"); forced = e ? atoi(e) : -1; }
    static const bool force_all = HMV_DEV_ENV("HMV_FORCE_TILE_ALL") != nullptr;   // ... and for the narrow layers too
    if ((Cout > 64 || force_all) && forced >= 0 && forced < TILE_COUNT) return (ConvTile)forced;
    // Measured on MI355X (tools/conv_sweep.py): the matrix pipe is DVFS/power limited, so the tile with
    // the least L2->LDS traffic per FLOP wins as long as it still fills the 256 CUs for several rounds.
    // tiny-K expanding convs (layer1/2 conv3 + residual) are epilogue/HBM-bound: 4 small blocks per CU
    // overlap one block's residual read / store with the others' short main loops
    // fp32 only: with 16-byte vectors on both sides of the fp16 epilogue the big tile wins there too (forced-tile
    // A/B runs of bench.py --dtype f16 / f32x3 --per-layer)
    if (!f16 && Cout >= 256 && K <= 128 && M >= 65536) return TILE_128x128_K16;
    // residual-bearing expanding convs with a medium reduction (layer3 conv3: 256 -> 1024 + residual): a 256x256 workgroup owns
    // its CU, so its 20 us drain overlaps nothing (75 us main loop: 100 TF).  Two 8-wave 256x128 workgroups per CU (k-step 16,
    // 48 KB of tile buffers, <= 128 registers) alternate: one drains while the other computes -- 1.44 -> 1.30 ms per launch
    // although each main loop is less efficient (tools/stagger_probe.py, profiles/r02_probe_w8.txt)
    static const bool no_w8 = HMV_DEV_ENV("HMV_NO_W8") != nullptr;   // development knob (A/B runs)
    if (!no_w8 && !f16 && has_res && K <= 256 && Cout >= 512 && (long long)((M + 255) / 256) * ((Cout + 127) / 128) >= 2048)
        return TILE_256x128_K16W8;
    // ... and the squeezing conv1 of layer2 (256 / 512 -> 128, no residual): one N-tile, eight k-steps of 32 or sixteen -- the same
    // short-tile regime; forced-tile runs of bench.py --per-layer (round 3): 692 -> 595 us and 300 -> 276 us
    if (!no_w8 && !f16 && !has_res && K >= 256 && K <= 512 && Cout == 128 && (long long)((M + 255) / 256) >= 1024)
        return TILE_256x128_K16W8;
    // channel counts that are not multiples of 128 (HRNet-w40: 160, 320), measured with tools/hr_sweep.py: one 256-wide
    // N-tile with its all-padding blocks skipped beats two 128-wide tiles whose second one is mostly DMA latency;
    // 64-wide tiles beat 128-wide ones when the last 128-wide tile would be at most half real.
    // round 3, fp32, re-measured per layer on HRNet-w40 (tools/gpu_r03_hr320.sh: forced-tile runs of bench.py --workload hr40
    // --per-layer): 80 and 160 output channels are whole multiples of NO wider tile, and 32-wide tiles spend nothing on padding
    // columns -- 160-channel 3x3 convs 303 -> 282 us, the 80- / 160-channel fuse layers 131 -> 96, 224 -> 156, 242 -> 212 us (the
    // one long-reduction case, 3x3 256 -> 80, keeps its 256 x 128 tile: 1 014 vs 1 057 us); 320 channels over 16 384 pixels run as
    // 640 tiles of 128 x 64 -- 2.5 per CU, a ragged single round -- or 1 280 tiles of 64 x 64, five per CU: 308 -> 268 us
    if (!f16 && Cout > 64 && Cout <= 192 && Cout % 64 != 0 && M >= 16384 && K < 2048) return TILE_128x32;
    if (!f16 && Cout > 256 && Cout % 128 != 0 && Cout % 128 <= 64 && M >= 16384 && (long long)((M + 127) / 128) * ((Cout + 63) / 64) < 1024)
        return TILE_64x64;
    if (Cout > 128 && Cout <= 192 && M >= 16384) return (M + 255) / 256 >= 256 ? TILE_256x256 : TILE_128x64;
    if (Cout > 256 && M >= 16384 && Cout % 128 != 0 && Cout % 128 <= 64) return TILE_128x64;
    if (Cout > 128 && (long long)((M + 255) / 256) * ((Cout + 255) / 256) >= 512) return TILE_256x256;
    if (Cout > 64 && (long long)((M + 255) / 256) * ((Cout + 127) / 128) >= 512) return TILE_256x128;
    // small problems (few frames, or the token GEMMs of the fusion transformer): 64x64 tiles give 4x the
    // workgroups and 4x shorter k-steps, which is what matters when the 128-wide tiling cannot fill 256 CUs
    // (long fp32 reductions whose 128 x 64 tiling still gives every CU a tile: layer3's 3x3 convs of one 8-view sample, 93.5 -> 90 us)
    if (!f16 && K >= 2048 && Cout >= 256 && (long long)((M + 127) / 128) * ((Cout + 127) / 128) < 256 &&
        (long long)((M + 127) / 128) * ((Cout + 63) / 64) >= 256)
        return TILE_128x64;
    if (Cout > 32 && (long long)((M + 127) / 128) * ((Cout + 127) / 128) < 256) return TILE_64x64;
    // ... and the expanding 1x1 convs of a single multi-view sample (fp32 Bottleneck conv3, K <= 256: little MFMA work per 128 x 128
    // tile, so 4x the workgroups win up to 1 024 of the big tiles: batch-1 layer3 conv3 50 -> 46 us, layer1 conv3 26 -> 21 us)
    if (!f16 && K <= 256 && Cout >= 256 && (long long)((M + 127) / 128) * ((Cout + 127) / 128) < 1024) return TILE_64x64;
    if (Cout > 64) return TILE_128x128;
    if (Cout > 32) return TILE_128x64;
    return TILE_128x32;
}

template <typename T, int BM, int BN, int WGM, int WGN, int MODE, bool GENERIC, int KB, bool PARTN = false, bool RD = false, bool X3 = false, bool C32 = false>
static hipError_t launch_one(ConvParams p, hipStream_t s) {
    if constexpr (!X3 && sizeof(T) == 2 && KB == 64 && !GENERIC && !PARTN && !RD && MODE != MODE_HALO) {
        if (p.x3_plane) return launch_one<T, BM, BN, WGM, WGN, MODE, GENERIC, KB, PARTN, RD, true>(p, s);
    }
    static bool configured[64] = {};   // per device ordinal
    constexpr bool tout = HMV_TOUT && sizeof(T) == 2 && !GENERIC && !RD;   // register epilogue: no staging memory (+ the 256-byte prefetch dummy slot)
    constexpr int ns = ring_stages(sizeof(T) == 2, KB, GENERIC, RD, BM, BN);
    const size_t lds = MODE == MODE_HALO ? (size_t)2 * (HALO_ROWS + BN) * KB * sizeof(T) + 256   // two halo images + two weight stages
                                         : (size_t)lds_floats(BM, BN, WGM, KB * (int)sizeof(T) / 4, RD, !tout, ns) * sizeof(float) + (tout ? 256 : 0);
    auto kern = conv_igemm<T, BM, BN, WGM, WGN, MODE, GENERIC, KB, PARTN, RD, X3, C32>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.Cout + BN - 1) / BN;
    if (p.mtiles * p.ntiles < 4 * p.stagger_blocks) p.stagger = 0;   // too few rounds for a start-up offset to pay
    hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles * (p.ksl > 1 ? p.ksl : 1) * (p.phases > 1 ? p.phases : 1)), dim3(64 * WGM * WGN), lds, s, p);
    return hipGetLastError();
}

template <typename T, int BM, int BN, int WGM, int WGN, int KB>
static hipError_t launch_modes(const ConvParams &p, bool one, bool generic, hipStream_t s) {
    if (generic) return one ? launch_one<T, BM, BN, WGM, WGN, MODE_1X1, true, KB>(p, s) : launch_one<T, BM, BN, WGM, WGN, MODE_TAPS, true, KB>(p, s);
    return one ? launch_one<T, BM, BN, WGM, WGN, MODE_1X1, false, KB>(p, s) : launch_one<T, BM, BN, WGM, WGN, MODE_TAPS, false, KB>(p, s);
}
template <typename T, int BM, int BN, int WGM, int WGN, int KB>
static hipError_t launch_plain(const ConvParams &p, bool one, hipStream_t s) {
    return one ? launch_one<T, BM, BN, WGM, WGN, MODE_1X1, false, KB>(p, s) : launch_one<T, BM, BN, WGM, WGN, MODE_TAPS, false, KB>(p, s);
}

hipError_t launch_conv(ConvParams p, ConvTile tile, hipStream_t s, const char **name) {
    if (name) *name = "conv_igemm<none>";
    if (p.M <= 0) return hipSuccess;
    {
        float *page = nullptr;
        hipError_t e = ensure_zero_page(&page);
        if (e != hipSuccess) return e;
        p.zero = page;
    }
    if (!p.lda) p.lda = p.Cin;
    if (!p.ldw) p.ldw = p.Kpad;
    {   // development knob: HMV_STAGGER=<10-ns ticks per phase group>[,<blocks of the first round>]
        static int st_ticks = -1, st_blocks = 256, st_mode = 0;
        if (st_ticks < 0) {
            const char *e = HMV_DEV_ENV("HMV_STAGGER");
            st_ticks = e ? atoi(e) : 0;
            if (e) { const char *c = strchr(e, ','); if (c) { st_blocks = atoi(c + 1); c = strchr(c + 1, ','); if (c) st_mode = atoi(c + 1); } }
        }
        p.stagger = st_ticks;
        p.stagger_blocks = st_blocks;
        p.stagger_mode = st_mode;
        static int pf = -1;   // development knob, OFF by default: HMV_PREFETCH=1 enables the software L2 prefetch (A/B runs)
        if (pf < 0) { const char *e = HMV_DEV_ENV("HMV_PREFETCH"); pf = e ? atoi(e) : 0; }
        p.prefetch = pf;
        static int burst = -1;   // development knob: HMV_BURST=0 keeps the residual on the register-ring path (A/B runs)
        if (burst < 0) { const char *e = HMV_DEV_ENV("HMV_BURST"); burst = e ? atoi(e) : 1; }
        p.burst = burst;
    }
    p.acc_scale = ldexpf(1.f, -p.acc_shift);
    bool generic = p.scatter || p.rg_out || p.act == ACT_GELU || p.act == ACT_LEAKY || (p.ldc & 3) ||
                   (p.res && (p.ldr & 3));
    // the register epilogue of the hot kernels moves 16-bit rows as 8-element vectors: planes that are not multiples of 8
    // columns take the staged (generic) epilogue
    if (HMV_TOUT && p.in_f16 && (p.out_f16 || p.out_split) &&
        (((p.out_split ? p.ldc >> 1 : p.ldc) & 7) || (p.res && ((p.res_split ? p.ldr >> 1 : p.ldr) & 7))))
        generic = true;
    if (p.phases > 1 && (p.phases != 4 || !p.scatter || p.R != 2 || p.S != 2 || p.stride != 1 || p.ksl > 1 || p.in2 || p.rd_cout || p.phase_stride == 0))
        return hipErrorInvalidValue;   // the merged launch exists for the four 2x2 phases of the k4 s2 p1 transposed conv only
    const bool one = p.R == 1 && p.S == 1 && p.pad_h == 0 && p.pad_w == 0;
    // a chained 1x1 conv (ConvParams::nx_*) exists in conv_stream.hip only: the caller asked conv_stream_chain_ok first
    if (p.nx_wgt && (generic || p.tall || !conv_stream_chain_ok(p, p.nx_cout))) return hipErrorInvalidValue;
    // ... and a fused max pool in conv_hs.hip only
    if (p.pool && (generic || p.tall || !conv_hs_supported(p))) return hipErrorInvalidValue;
    // weights packed for the tall-tile 3x3 kernel (conv_ht.hip; K order (32-channel chunk, r, s, c % 32)): that kernel when its 512-pixel x
    // 128-channel tiles fill the chip, else the 64 x 64 / 128 x 128 tiles of this file walking the SAME order -- same operand roles,
    // same accumulation sequence, same epilogue arithmetic, so the bits do not depend on which of the two ran (tests/test_gpu_parity.py)
    if (p.tall) {
        if (generic || !conv_ht_shape_ok(p.R, p.S, p.stride, p.pad_h, p.Cin, p.Cout, p.H, p.W) || !p.in_f16 || !p.out_f16 || p.res || p.in2 ||
            p.x3_plane || p.cwrap || p.rd_cout || p.ksl > 1 || p.phases > 1 || p.up)
            return hipErrorInvalidValue;
        static int ht_min = -1;   // development knob: HMV_HT_MIN_TILES=<n> (default 256: one tile per CU)
        if (ht_min < 0) { const char *e = HMV_DEV_ENV("HMV_HT_MIN_TILES"); ht_min = e ? atoi(e) : 256; }
        const long long ht_tiles = (long long)p.N * (p.H >> 4) * (p.W >> 5) * (p.Cout / 128);
        if (conv_ht_mode() > 0 || (conv_ht_mode() < 0 && ht_tiles >= ht_min)) return launch_conv_ht(p, s, name);
        // small launches: conv_m16.hip's 64 x 64 / 128 x 128 tiles on the same 16x16x32 MFMA (same bits as conv_ht<..., m16>); the c32
        // instantiations below are the 32x32x16 partner of the shape A/B (conv_ht_set_shape(0), op-level tests only)
        if (conv_ht_shape()) return launch_conv_m16(p, s, name);
        if ((long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) < 256) {
            if (name) *name = "conv_igemm_f16<64x64,taps,c32>";
            return launch_one<_Float16, 64, 64, 2, 2, MODE_TAPS, false, 32, false, false, false, true>(p, s);
        }
        if (name) *name = "conv_igemm_f16<128x128,taps,c32>";
        return launch_one<_Float16, 128, 128, 2, 2, MODE_TAPS, false, 32, false, false, false, true>(p, s);
    }
    // token GEMMs over (hi, lo) fp16 pairs with fp32 output rows (the fusion blocks' q / k / v projections in the fp16-kernel modes):
    // gemm_x3.hip at every size (a four-stage ring on the 16x16x32 MFMA instead of the fused split loop's one DMA in flight)
    if (!generic && gemm_x3_rule(p)) return launch_gemm_x3(p, s, name);
    if (!generic && gemm_x3k16_ok(p)) return launch_gemm_x3k16(p, s, name);
    // MFMA-heavy fp16 1x1 convs without a residual (layer3's conv1, pose_net.0): the 16x16x32 MFMA at EVERY batch size -- the
    // phase-interleaved 256 x 256 tile where its tiles fill the chip, conv_m16.hip's small tiles elsewhere (same bits)
    if (!generic && conv_m16_rule(p)) {
        p.m16 = 1;
        return conv_gemm8_supported(p) ? launch_conv_gemm8(p, s, name) : launch_conv_m16(p, s, name);
    }
    // short-reduction residual 1x1 convs over many pixels (fp16 Bottleneck conv3): the persistent weight-stationary kernel
    if (!generic && conv_stream_supported(p)) return launch_conv_stream(p, s, name);
    // MFMA-heavy fp16 1x1 convs without a residual on 256 x 256 tiles: the counted-vmcnt, phase-interleaved main loop
    if (!generic && conv_gemm8_supported(p)) return launch_conv_gemm8(p, s, name);
    // few-channel fp16 layers (3x3 64 -> 64, the space-to-depth stem) over many pixels: weights in registers, halo images streamed
    if (!generic && conv_hs_supported(p)) return launch_conv_hs(p, s, name);
    if (p.ksl > 1 && (!one || p.in_f16 || p.in2 || p.up || p.res || p.act != ACT_NONE || p.kslice <= 0 || p.kslice % 32 != 0 ||
                      p.ksl * p.kslice != p.Kpad || p.Cin % 32 != 0))
        return hipErrorInvalidValue;   // split-K: plain fp32 GEMM slices, epilogue left to the reduction kernel
    if (p.in2 && (!one || p.stride != 1 || p.up || p.cwrap || p.x3_plane || p.rd_cout || p.ksplit <= 0 || p.ksplit >= p.K ||
                  p.ksplit % (p.in_f16 ? 64 : 32) != 0 || p.Cin % (p.in_f16 ? 64 : 32) != 0))
        return hipErrorInvalidValue;   // the second source exists in the chunked 1x1 mode only
    if (generic && (tile == TILE_256x128 || tile == TILE_128x256 || tile == TILE_256x256 || tile == TILE_256x256_RING || tile == TILE_128x128_K16 || tile == TILE_256x128_K16W8 ||
                    tile == TILE_128x256_K16 || tile == TILE_256x128_K16))
        tile = TILE_128x128;   // the rarely used epilogue paths exist only for the 4-wave tiles
    const int ch = p.in_f16 ? 64 : 32, epc = p.in_f16 ? 8 : 4;
    const bool dense = p.Cin % ch != 0;
    // the fused split loop exists in the non-generic fp16 kernels with a 64-element k-step, chunked modes
    if (p.x3_plane && (!p.in_f16 || generic || p.rd_cout || p.cwrap || p.x3_plane % 8 != 0 || p.Cin != 2 * p.x3_plane ||
                       (!dense && p.x3_plane % 32 != 0) || tile == TILE_128x128_K16 || tile == TILE_128x256_K16 || tile == TILE_256x128_K16 || tile == TILE_256x256_RING))
        return hipErrorInvalidValue;
    // split operands: fp16 kernels only; the generic epilogue handles them on its vector path only
    if ((p.cwrap || p.res_split || p.out_split || p.acc_shift) &&
        (!p.in_f16 || p.rd_cout || (p.ldc & 3) || (p.res && (p.ldr & 3)) || p.act == ACT_GELU || p.act == ACT_LEAKY))
        return hipErrorInvalidValue;
    // row-decomposed fp32 convs over many pixels (HRNet-w40's 40- / 80-channel branches): the persistent weight-stationary kernel
    if (p.rd_cout && !generic && conv_rds_supported(p)) return launch_conv_rds(p, s, name);
    if (p.rd_cout) {   // row-decomposed 3x3 (see conv_igemm): the caller passes the 3x1 GEMM (R = 3, S = 1, Cout = 3 * rd_cout)
        if (generic || p.R != 3 || p.S != 1 || p.stride != 1 || p.pad_h != 1 || p.pad_w != 0 || p.Cout != 3 * p.rd_cout || p.Cout > 256 ||
            (p.rd_cout & 3) || (p.ldc & 3) || (p.res && (p.ldr & 3)) || 128 % p.Wo != 0 || p.Ho != p.H || p.Wo != p.W)
            return hipErrorInvalidValue;
        if (dense) {
            if (p.Cin % epc != 0 || p.lda % epc != 0 || p.Kpad > (1 << 16)) return hipErrorInvalidValue;
            p.cpt = p.Cin / epc;
            p.cpt_magic = p.cpt > 1 ? (unsigned)(0xFFFFFFFFu / (unsigned)p.cpt) + 1u : 0u;
            p.s_magic = 0u;
        }
        const bool wide = p.Cout > 128;   // 3 * 80 = 240 columns: a 128x256 tile (8 waves)
        if (name) *name = tile_name(p.in_f16 ? "f16" : "f32", wide ? TILE_128x256 : TILE_128x128, dense ? 2 : 0, false, true);
#define HMV_RD(T_, KB_)                                                                                                  \
        if (wide) return dense ? launch_one<T_, 128, 256, 2, 4, MODE_DENSE, false, KB_, false, true>(p, s)                  \
                               : launch_one<T_, 128, 256, 2, 4, MODE_TAPS, false, KB_, false, true>(p, s);                  \
        return dense ? launch_one<T_, 128, 128, 2, 2, MODE_DENSE, false, KB_, false, true>(p, s)                            \
                     : launch_one<T_, 128, 128, 2, 2, MODE_TAPS, false, KB_, false, true>(p, s);
        if (p.in_f16) { HMV_RD(_Float16, 64) }
        HMV_RD(float, 32)
#undef HMV_RD
    }
    if (dense) {
        if (p.Cin % epc != 0 || p.lda % epc != 0 || generic || p.Kpad > (1 << 16)) return hipErrorInvalidValue;
        p.cpt = (p.x3_plane ? p.x3_plane : p.Cin) / epc;   // 16-byte vectors per tap (of one plane in the fused split loop)
        p.cpt_magic = p.cpt > 1 ? (unsigned)(0xFFFFFFFFu / (unsigned)p.cpt) + 1u : 0u;
        p.s_magic = p.S > 1 ? (unsigned)(0xFFFFFFFFu / (unsigned)p.S) + 1u : 0u;
        if (p.up && (p.R != 1 || p.S != 1 || p.pad_h || p.pad_w || p.stride != 1)) return hipErrorInvalidValue;
        tile = conv_dense_tile(tile, p.in_f16 != 0);
    }
    // fp16 256x256 (the plain kernels of the backbone) on the four-stage ring main loop (ring_stages above): measured 4-8 %
    // SLOWER than the two-stage loop (profiles/r02_probe_ring.txt), so it stays behind a development knob
    static int ring = -1;   // HMV_F16_RING=1 selects it (A/B runs)
    if (ring < 0) { const char *e = HMV_DEV_ENV("HMV_F16_RING"); ring = e ? atoi(e) : 0; }
    if (ring && p.in_f16 && tile == TILE_256x256 && !generic && !dense && !p.x3_plane && !p.cwrap && !p.rd_cout && p.ksl <= 1)
        tile = TILE_256x256_RING;
    // fp16 3x3 stride-1 pad-1 convs on 256 x 256 tiles: 16 x 16 pixel blocks with the halo image in LDS (MODE_HALO)
    static int halo = -1;   // development knob: HMV_NO_HALO=1 keeps the nine-fetch MODE_TAPS loop (A/B runs)
    if (halo < 0) halo = HMV_DEV_ENV("HMV_NO_HALO") ? 0 : 1;
    const bool use_halo = halo && p.in_f16 && (tile == TILE_256x256 || tile == TILE_256x128) && !generic && !dense && !p.x3_plane && !p.cwrap && !p.rd_cout &&
                          p.ksl <= 1 && !p.in2 && !p.res && !p.up && p.R == 3 && p.S == 3 && p.stride == 1 && p.pad_h == 1 && p.pad_w == 1 &&
                          p.Ho == p.H && p.Wo == p.W && p.H % 16 == 0 && p.W % 16 == 0 && p.Cin % 64 == 0 && p.lda == p.Cin &&
                          p.M % 256 == 0;
    if (use_halo) {
        if (name) *name = tile == TILE_256x256 ? "conv_igemm_f16<256x256,halo>" : "conv_igemm_f16<256x128,halo>";
        return tile == TILE_256x256 ? launch_one<_Float16, 256, 256, 2, 4, MODE_HALO, false, 64>(p, s)
                                    : launch_one<_Float16, 256, 128, 4, 2, MODE_HALO, false, 64>(p, s);
    }
    // last N-tile with >= 32 all-padding columns: the block-skipping instantiations (fp32, the three big tiles)
    const int mode = dense ? MODE_DENSE : (one ? MODE_1X1 : MODE_TAPS);
    static int no_skip = -1;   // development knob: HMV_NO_SKIPN=1 disables the block-skipping instantiations (A/B runs)
    if (no_skip < 0) { const char *e = HMV_DEV_ENV("HMV_NO_SKIPN"); no_skip = e ? atoi(e) : 0; }
    const bool partn = !no_skip && !p.in_f16 && !generic && conv_partial_n(tile, p.Cout) && !(dense && tile == TILE_256x256);
    if (name) *name = p.in_f16 ? conv_tile_name_f16(tile, mode) : tile_name("f32", tile, mode, partn);
    if (partn) {
#define HMV_PARTN(BM_, BN_, WGM_, WGN_)                                                                              \
        switch (mode) {                                                                                                 \
            case MODE_DENSE: return launch_one<float, BM_, BN_, WGM_, WGN_, MODE_DENSE, false, 32, true>(p, s);         \
            case MODE_1X1: return launch_one<float, BM_, BN_, WGM_, WGN_, MODE_1X1, false, 32, true>(p, s);             \
            default: return launch_one<float, BM_, BN_, WGM_, WGN_, MODE_TAPS, false, 32, true>(p, s);                  \
        }
        if (tile == TILE_128x128) { HMV_PARTN(128, 128, 2, 2) }
        if (tile == TILE_256x128) { HMV_PARTN(256, 128, 4, 2) }
        // (an 8 x 1 wave layout -- every wave 32 pixels x all columns, so that the five real blocks of a 160-column layer load all SIMDs
        // alike -- cannot stage its epilogue: eight 32-row blocks x 260 columns are 266 KB of LDS)
        if (tile == TILE_256x256) { HMV_PARTN(256, 256, 2, 4) }
#undef HMV_PARTN
    }
    if (dense) {   // dense K order over the real channels (stem, HRNet's 40 / 80-channel tensors)
        if (p.in_f16) {
            // 129 .. 192 output channels (HRNet-w40's 160-channel branch): ONE 192-wide N-tile instead of two 128-wide ones whose
            // second is three quarters padding (6 instead of 8 MFMA column blocks per pixel block)
            static const bool no192 = HMV_DEV_ENV("HMV_NO_N192") != nullptr;   // development knob (A/B runs)
            if (tile == TILE_256x128 && p.Cout > 128 && p.Cout <= 192 && !generic && !no192) {
                // (128 x 192 tiles, two 4-wave workgroups per CU, measured 28 % slower on the one-round launches of HRNet-w40: 3.82 vs 2.98 ms)
                if (name) *name = "conv_igemm_f16<256x192,dense>";
                return launch_one<_Float16, 256, 192, 4, 2, MODE_DENSE, false, 64>(p, s);
            }
            switch (tile) {
                case TILE_128x32: return launch_one<_Float16, 128, 32, 4, 1, MODE_DENSE, false, 64>(p, s);
                case TILE_128x64: return launch_one<_Float16, 128, 64, 2, 2, MODE_DENSE, false, 64>(p, s);
                case TILE_64x64: return launch_one<_Float16, 64, 64, 2, 2, MODE_DENSE, false, 64>(p, s);
                case TILE_256x128: return launch_one<_Float16, 256, 128, 4, 2, MODE_DENSE, false, 64>(p, s);
                default: return launch_one<_Float16, 128, 128, 2, 2, MODE_DENSE, false, 64>(p, s);
            }
        }
        switch (tile) {
            case TILE_128x32: return launch_one<float, 128, 32, 4, 1, MODE_DENSE, false, 32>(p, s);
            case TILE_128x64: return launch_one<float, 128, 64, 2, 2, MODE_DENSE, false, 32>(p, s);
            case TILE_64x64: return launch_one<float, 64, 64, 2, 2, MODE_DENSE, false, 32>(p, s);
            case TILE_256x128: return launch_one<float, 256, 128, 4, 2, MODE_DENSE, false, 32>(p, s);
            default: return launch_one<float, 128, 128, 2, 2, MODE_DENSE, false, 32>(p, s);
        }
    }
    if (p.in_f16) {   // fp16 operands (k-step 64); the half-step variants map to their full-step tile
        switch (tile) {
            case TILE_128x32: return launch_modes<_Float16, 128, 32, 4, 1, 64>(p, one, generic, s);
            case TILE_128x64: return launch_modes<_Float16, 128, 64, 2, 2, 64>(p, one, generic, s);
            case TILE_64x64: return launch_modes<_Float16, 64, 64, 2, 2, 64>(p, one, generic, s);
            case TILE_128x128: return launch_modes<_Float16, 128, 128, 2, 2, 64>(p, one, generic, s);
            case TILE_128x128_K16: return launch_plain<_Float16, 128, 128, 2, 2, 32>(p, one, s);   // 64-byte rows, 4 blocks/CU
            case TILE_256x128: return launch_plain<_Float16, 256, 128, 4, 2, 64>(p, one, s);
            case TILE_128x256: return launch_plain<_Float16, 128, 256, 2, 4, 64>(p, one, s);
            // 64-byte rows: 48 KB of tile buffers and <= 128 registers -> two 8-wave workgroups per CU
            case TILE_256x128_K16: return launch_plain<_Float16, 256, 128, 4, 2, 32>(p, one, s);
            case TILE_128x256_K16: return launch_plain<_Float16, 128, 256, 2, 4, 32>(p, one, s);
            case TILE_256x256: return launch_plain<_Float16, 256, 256, 2, 4, 64>(p, one, s);
            case TILE_256x256_RING: return launch_plain<_Float16, 256, 256, 2, 4, 32>(p, one, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (tile) {
        case TILE_128x32: return launch_modes<float, 128, 32, 4, 1, 32>(p, one, generic, s);
        case TILE_128x64: {
            // 64-channel 3x3 convs over many pixels (layer1 conv2): ONE 8-wave workgroup per 256 pixels instead of two 4-wave ones
            // per 128 -- 664 -> 640 us (round 3; HMV_NO_T256x64=1 for A/B runs).  Same accumulation order: same bits.
            static const bool no25664 = HMV_DEV_ENV("HMV_NO_T256x64") != nullptr;
            if (!no25664 && !generic && !one && p.Cout == 64 && p.M >= 524288) {
                if (name) *name = "conv_igemm_f32<256x64,taps>";
                return launch_plain<float, 256, 64, 4, 2, 32>(p, false, s);
            }
            return launch_modes<float, 128, 64, 2, 2, 32>(p, one, generic, s);
        }
        case TILE_128x128: return launch_modes<float, 128, 128, 2, 2, 32>(p, one, generic, s);
        case TILE_256x128: return launch_plain<float, 256, 128, 4, 2, 32>(p, one, s);
        case TILE_128x256: return launch_plain<float, 128, 256, 2, 4, 32>(p, one, s);
        case TILE_256x256: return launch_plain<float, 256, 256, 2, 4, 32>(p, one, s);
        case TILE_128x128_K16: return launch_plain<float, 128, 128, 2, 2, 16>(p, one, s);
        case TILE_128x256_K16: return launch_plain<float, 128, 256, 2, 2, 16>(p, one, s);
        case TILE_256x128_K16: return launch_plain<float, 256, 128, 2, 2, 16>(p, one, s);
        case TILE_64x64: return launch_modes<float, 64, 64, 2, 2, 32>(p, one, generic, s);
        case TILE_256x128_K16W8: return launch_plain<float, 256, 128, 4, 2, 16>(p, one, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace hmv
