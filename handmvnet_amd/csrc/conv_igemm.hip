// conv_igemm.hip -- NHWC implicit-GEMM convolution on the gfx950 fp32 matrix cores.
//
// Replaces every nn.Conv2d / nn.Linear of the reference's hot path
// (/root/reference/src/models/backbones/resnet.py:114-118,162,193; layers.py:213-215,224,
//  161-174, 318-334; handmvnet.py:70-86) with ONE kernel family:
//
//   out[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] + residual[m][n] )
//
// M = images * Ho * Wo output pixels, N = Cout, K = R*S*Cin with k = (r, s, c) so that
// consecutive k are consecutive NHWC channels.  The im2col matrix A is never built: each
// thread gathers 16-byte channel vectors straight from the NHWC activation, with the
// (r, s) tap of a 32-wide k-step being wave-uniform scalar work.
//
// MI355X mapping
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD): each wave owns a (32*TM)x(32*TN)
//     output tile in TM*TN 16-register accumulators.
//   * A and Wt tiles are staged through LDS as [row][k] with rows padded by one 16-byte
//     access (stride 36 floats): the MFMA operand fetch is a conflict-free ds_read_b128 of
//     4 consecutive k for lane (row = lane&31, k-half = lane>>5); the 4 components feed 4
//     consecutive MFMAs (the k order inside a step is permuted identically for A and Wt).
//   * double-buffered LDS + register prefetch: global loads of step t+1 are in flight while
//     step t runs on the matrix pipe; one barrier per k-step.
//   * XCD-aware block -> tile map: the N-tiles of one M-tile are consecutive on ONE XCD, so
//     an activation tile is pulled from HBM once per XCD and re-read from that XCD's L2.
//   * epilogue fused: folded-BN shift / bias, residual add, ReLU / GELU(erf) / LeakyReLU;
//     each accumulator register stores two full 128-byte row segments per wave.
#include <cstdlib>

#include "kernels.h"

namespace hmv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

// 256 bytes of zeros: out-of-range taps / rows read from here, so that the global load result
// needs no post-processing and its s_waitcnt can sink below the MFMAs to the LDS write.
static float *g_zero_page = nullptr;
static hipError_t ensure_zero_page() {
    if (g_zero_page) return hipSuccess;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&g_zero_page), 256);
    if (e == hipSuccess) e = hipMemset(g_zero_page, 0, 256);
    return e;
}

// GENERIC = false: the backbone's epilogue (bias, optional residual, optional ReLU, rows written in
// place).  GENERIC = true adds the rarely used paths (sub-pixel output scatter, residual row remap,
// GELU / LeakyReLU); keeping them out of the hot instantiation keeps its epilogue straight-line.
template <int BM, int BN, int WGM, int WGN, bool SMALLC, bool GENERIC>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_f32(const ConvParams p) {
    constexpr int NT = 64 * WGM * WGN;   // threads per workgroup (4 or 8 waves)
    constexpr int RPS = NT / 8;          // tile rows staged per pass (8 threads per 32-float row segment)
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int AP = BM / RPS, BP = BN / RPS;
    constexpr int LDC = BN + 4;  // epilogue staging row stride (floats)
    static_assert(BM % RPS == 0 && BN % RPS == 0, "tile rows must be a multiple of the staging pass");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile is made of 32x32 MFMA blocks");
    // the epilogue stages the output tile through the (dead) tile buffers, in EH row-halves if it must
    constexpr int EH = (BM * LDC <= 2 * (BM + BN) * LDS_LD) ? 1 : 2;
    constexpr int HR = BM / EH;  // rows per staged half
    static_assert(HR * LDC <= 2 * (BM + BN) * LDS_LD, "epilogue staging must fit in the tile buffers");
    static_assert(HR % WM == 0, "a wave's rows must not straddle staging halves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;                    // [2][BM][LDS_LD]
    float *sB = smem + 2 * BM * LDS_LD;  // [2][BN][LDS_LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;

    // ---- XCD-aware tile assignment (bijective for any grid size)
    int mt, nt;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r = nblk & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        mt = lid / p.ntiles;
        nt = lid - mt * p.ntiles;
    }

    // ---- per-thread gather roles: 8 threads cover one 32-float k-row segment
    const int lrow = tid >> 3, kq = tid & 7;
    const float *abase[AP];
    int hi0[AP], wi0[AP], aoff[AP];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = mt * BM + i * RPS + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        abase[i] = p.in + (size_t)n * p.H * p.W * p.lda;
        hi0[i] = ok ? ho * p.stride - p.pad_h : -(1 << 28);  // out-of-range rows fail the bounds test
        wi0[i] = wo * p.stride - p.pad_w;
        aoff[i] = ok ? (hi0[i] * p.W + wi0[i]) * p.lda + (SMALLC ? 0 : 4 * kq) : 0;
    }
    const float *wrow[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) wrow[i] = p.wgt + (size_t)(nt * BN + i * RPS + lrow) * p.ldw + 4 * kq;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    f32x4 ra[AP], rb[BP];
    const int nk = p.Kpad / BK;
    const float *zero = p.zero;

#define HMV_LOAD_TILE(kt)                                                                         \
    {                                                                                             \
        const int k0 = (kt) * BK;                                                                 \
        int r_, s_, delta_;                                                                       \
        if (!SMALLC) { /* the (r, s) tap of a k-step is wave-uniform scalar work */               \
            const int tap = k0 / p.Cin;                                                           \
            r_ = tap / p.S;                                                                       \
            s_ = tap - r_ * p.S;                                                                  \
            delta_ = (r_ * p.W + s_) * p.lda + (k0 - tap * p.Cin);                                \
        } else { /* Cin == 4: one tap per 16-byte vector */                                       \
            const int tap = (k0 >> 2) + kq;                                                       \
            r_ = tap / p.S;                                                                       \
            s_ = tap - r_ * p.S;                                                                  \
            if (tap >= p.R * p.S) r_ = 1 << 28;                                                   \
            delta_ = (r_ * p.W + s_) * p.lda;                                                     \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                          \
            const bool ok = (unsigned)(hi0[i] + r_) < (unsigned)p.H && (unsigned)(wi0[i] + s_) < (unsigned)p.W; \
            const float *src = (ok && !(p.tune & 32)) ? abase[i] + (aoff[i] + delta_) : zero;     \
            ra[i] = *reinterpret_cast<const f32x4 *>(src);                                        \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < BP; ++i) rb[i] =                                    \
            *reinterpret_cast<const f32x4 *>((p.tune & 32) ? zero : wrow[i] + k0);                \
    }

#define HMV_STORE_TILE(buf)                                                                       \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                            \
            *reinterpret_cast<f32x4 *>(&sA[((buf) * BM + i * RPS + lrow) * LDS_LD + 4 * kq]) = ra[i]; \
        _Pragma("unroll") for (int i = 0; i < BP; ++i)                                            \
            *reinterpret_cast<f32x4 *>(&sB[((buf) * BN + i * RPS + lrow) * LDS_LD + 4 * kq]) = rb[i]; \
    }

    if (p.tune & 1) {  // experiment: de-phase co-resident workgroups
        if (__builtin_popcount(blockIdx.x & 1023) & 1) __builtin_amdgcn_s_sleep(32);
    }
    HMV_LOAD_TILE(0);
    HMV_STORE_TILE(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more && !(p.tune & 4)) HMV_LOAD_TILE(kt + 1);
        const float *a0 = &sA[(buf * BM + wm * WM + l31) * LDS_LD + 4 * kh];
        const float *b0 = &sB[(buf * BN + wn * WN + l31) * LDS_LD + 4 * kh];
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 af[TM], bf[TN];
            if (!(p.tune & 16) || kt == 0) {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const f32x4 *>(a0 + a * 32 * LDS_LD + 8 * q);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const f32x4 *>(b0 + b * 32 * LDS_LD + 8 * q);
            } else {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = ra[a % AP];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = rb[b % BP];
            }
            if (p.tune & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
            if (p.tune & 2) __builtin_amdgcn_s_setprio(0);
        }
        if (!(p.tune & 8)) {
            if (more && !(p.tune & 128)) HMV_STORE_TILE(buf ^ 1);
            if (!(p.tune & 64)) __syncthreads();
        }
    }
#undef HMV_LOAD_TILE
#undef HMV_STORE_TILE

    // ---- fused epilogue, staged through LDS so that global traffic is 16-byte vectors on full rows.
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    float *sC = smem;  // [HR][LDC]; the tile buffers are dead after the loop's last barrier
#pragma unroll
    for (int half = 0; half < EH; ++half) {
    if (half > 0) __syncthreads();  // the previous half has been drained
    if ((wm * WM) / HR == half) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm * WM - half * HR + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    sC[row * LDC + wn * WN + b * 32 + l31] = acc[a][b][e];
                }
    }
    __syncthreads();

    const int m0 = mt * BM + half * HR, n0 = nt * BN;
    const bool vec = ((p.ldc & 3) == 0) && (p.res == nullptr || (p.ldr & 3) == 0);
    if (vec) {
        constexpr int TPR = BN / 4;          // threads per row
        constexpr int RPP = NT / TPR;        // rows per pass
        constexpr int NPASS = HR / RPP;
        constexpr int UB = NPASS < 4 ? NPASS : 4;  // rows in flight per thread
        const int c4 = tid % TPR, r0 = tid / TPR;
        const int col = n0 + 4 * c4;
        // columns [Cout, round4(Cout)) hold exact zeros (zero-padded weights and bias): writing them is
        // harmless whenever the row stride leaves room, which lets Cout = 21 use vector stores too.
        const int cend = p.Cout + 3 < p.ldc ? ((p.Cout + 3) & ~3) : p.ldc;
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
                    const int r = r0 + (g + u) * RPP, m = m0 + r;
                    okr[u] = m < p.M;
                    orow[u] = (size_t)m;
                    size_t rrow = (size_t)m;
                    if (GENERIC) {
                        if (p.scatter) {
                            const int n = m / HoWo, rem = m - n * HoWo;
                            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                            orow[u] = ((size_t)n * (p.Ho * p.osy) + (ho * p.osy + p.ooy)) * (size_t)(p.Wo * p.osx) + (wo * p.osx + p.oox);
                        }
                        if (p.rg_out) rrow = (size_t)(m / p.rg_out) * p.rg_in + (m % p.rg_out);
                    }
                    const float *rp = (has_res && okr[u]) ? p.res + rrow * p.ldr + col : zero;
                    rv[u] = *reinterpret_cast<const f32x4 *>(rp);
                    v[u] = *reinterpret_cast<const f32x4 *>(&sC[r * LDC + 4 * c4]);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    f32x4 t = v[u] + bv + rv[u];
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
                    if (okr[u]) *reinterpret_cast<f32x4 *>(p.out + orow[u] * p.ldc + col) = t;
                }
            }
        }
    } else if (GENERIC) {  // scalar fallback (row strides that are not multiples of 4, e.g. the 21x3 output)
        for (int idx = tid; idx < HR * BN; idx += NT) {
            const int r = idx / BN, c = idx - r * BN;
            const int m = m0 + r, col = n0 + c;
            if (m >= p.M || col >= p.Cout) continue;
            size_t orow = (size_t)m, rrow = (size_t)m;
            if (p.scatter) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                orow = ((size_t)n * (p.Ho * p.osy) + (ho * p.osy + p.ooy)) * (size_t)(p.Wo * p.osx) + (wo * p.osx + p.oox);
            }
            if (p.rg_out) rrow = (size_t)(m / p.rg_out) * p.rg_in + (m % p.rg_out);
            float v = sC[r * LDC + c] + p.bias[col];
            if (p.res) v += p.res[rrow * p.ldr + col];
            if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
            else if (p.act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
            else if (p.act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
            p.out[orow * p.ldc + col] = v;
        }
    }
    }  // staging halves
}

// ====================================================================== v3: software-pipelined main loop
// Same math, layouts and epilogue as conv_igemm_f32 above, but each wave keeps the matrix pipe fed
// across the per-step seams:
//   * global loads run TWO tiles ahead in two named register sets (tile t+2 is requested at the
//     start of step t and written to LDS in the middle of step t+1), spread over the 4 MFMA groups;
//   * the LDS write of tile t+1 sits in the middle of step t (its loads were issued a full step
//     earlier), not at the end in front of the barrier;
//   * ONE raw s_barrier per step, placed BEFORE the last MFMA group: the fragments of that group are
//     already in registers, and the first fragments of step t+1 are fetched right behind the barrier,
//     so barrier skew and LDS latency hide under 16 MFMAs (1024 cycles);
//   * tap / address arithmetic is incremental (no per-step division); 1x1 convolutions and plain
//     GEMMs (32 of the 46 r50 convs) take the IS1X1 path: one pointer bump per load, no bounds test.
template <int BM, int BN, int WGM, int WGN, bool IS1X1, bool GENERIC>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_f32_v3(const ConvParams p) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int RPS = NT / 8;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int AP = BM / RPS, BP = BN / RPS, NL = AP + BP;
    constexpr int LDC = BN + 4;
    constexpr int EH = (BM * LDC <= 2 * (BM + BN) * LDS_LD) ? 1 : 2;
    constexpr int HR = BM / EH;
    static_assert(BM % RPS == 0 && BN % RPS == 0, "tile rows must be a multiple of the staging pass");
    static_assert(HR * LDC <= 2 * (BM + BN) * LDS_LD && HR % WM == 0, "epilogue staging");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;
    float *sB = smem + 2 * BM * LDS_LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;

    int mt, nt;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r = nblk & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        mt = lid / p.ntiles;
        nt = lid - mt * p.ntiles;
    }

    const int lrow = tid >> 3, kq = tid & 7;
    const float *zero = p.zero;
    const float *aptr[AP];   // IS1X1: running pointer (or the zero page, with step 0)
    int astep[AP];           // IS1X1: floats to advance per k-step
    int hi0[AP], wi0[AP];    // generic: top-left tap of the row's receptive field
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = mt * BM + i * RPS + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const float *base = p.in + (size_t)n * p.H * p.W * p.lda;
        hi0[i] = ok ? ho * p.stride - p.pad_h : -(1 << 28);
        wi0[i] = wo * p.stride - p.pad_w;
        if (IS1X1) {
            aptr[i] = ok ? base + (hi0[i] * p.W + wi0[i]) * p.lda + 4 * kq : zero;
            astep[i] = ok ? BK : 0;
        } else {
            aptr[i] = base + (ok ? (hi0[i] * p.W + wi0[i]) * p.lda + 4 * kq : 0);
            astep[i] = 0;
        }
    }
    const float *wptr[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) wptr[i] = p.wgt + (size_t)(nt * BN + i * RPS + lrow) * p.ldw + 4 * kq;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // load cursor (wave-uniform): tap (cr, cs), channel offset cc of the NEXT tile to request
    int cr = 0, cs = 0, cc = 0, cdelta = 0;
    f32x4 ra0[AP], rb0[BP], ra1[AP], rb1[BP];   // two named register sets (static indexing only)
    const int nk = p.Kpad / BK;

    // request load #idx (0..AP-1: A rows, AP..NL-1: weight rows) of the cursor tile into set SET
#define V3_LOAD(SET, idx)                                                                                       \
    {                                                                                                           \
        if ((idx) < AP) {                                                                                       \
            const int i_ = (idx) < AP ? (idx) : 0;                                                          \
            const float *src_;                                                                                  \
            if (IS1X1) {                                                                                        \
                src_ = aptr[i_];                                                                                \
                aptr[i_] += astep[i_];                                                                          \
            } else {                                                                                            \
                const bool ok_ = (unsigned)(hi0[i_] + cr) < (unsigned)p.H && (unsigned)(wi0[i_] + cs) < (unsigned)p.W; \
                src_ = ok_ ? aptr[i_] + cdelta : zero;                                                          \
            }                                                                                                   \
            if (SET == 0) ra0[i_] = *reinterpret_cast<const f32x4 *>(src_);                                     \
            else ra1[i_] = *reinterpret_cast<const f32x4 *>(src_);                                              \
        } else {                                                                                                \
            const int j_ = (idx) >= AP ? (idx) - AP : 0;                                                    \
            if (SET == 0) rb0[j_] = *reinterpret_cast<const f32x4 *>(wptr[j_]);                                 \
            else rb1[j_] = *reinterpret_cast<const f32x4 *>(wptr[j_]);                                          \
            wptr[j_] += BK;                                                                                     \
        }                                                                                                       \
    }
    // all loads of group g (0..3) of the cursor tile
#define V3_LOAD_GROUP(SET, g)                                                                                   \
    {                                                                                                           \
        _Pragma("unroll") for (int idx_ = 0; idx_ < NL; ++idx_)                                                 \
            if (idx_ * 4 / NL == (g)) V3_LOAD(SET, idx_);                                                       \
    }
#define V3_ADVANCE()                                                                                            \
    {                                                                                                           \
        if (!IS1X1) {                                                                                           \
            cc += BK;                                                                                           \
            cdelta += BK;                                                                                       \
            if (cc >= p.Cin) {                                                                                  \
                cc = 0;                                                                                         \
                cdelta -= p.Cin;                                                                                \
                if (++cs == p.S) { cs = 0; ++cr; cdelta += (p.W - p.S) * p.lda; }                               \
                cdelta += p.lda;                                                                                \
            }                                                                                                   \
        }                                                                                                       \
    }
#define V3_STORE(SET, buf)                                                                                      \
    {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                                          \
            *reinterpret_cast<f32x4 *>(&sA[((buf) * BM + i * RPS + lrow) * LDS_LD + 4 * kq]) = (SET == 0 ? ra0[i] : ra1[i]); \
        _Pragma("unroll") for (int i = 0; i < BP; ++i)                                                          \
            *reinterpret_cast<f32x4 *>(&sB[((buf) * BN + i * RPS + lrow) * LDS_LD + 4 * kq]) = (SET == 0 ? rb0[i] : rb1[i]); \
    }
#define V3_FRAGS(FA, FB, buf, q)                                                                                \
    {                                                                                                           \
        const float *a0_ = &sA[((buf) * BM + wm * WM + l31) * LDS_LD + 4 * kh + 8 * (q)];                       \
        const float *b0_ = &sB[((buf) * BN + wn * WN + l31) * LDS_LD + 4 * kh + 8 * (q)];                       \
        _Pragma("unroll") for (int a = 0; a < TM; ++a) FA[a] = *reinterpret_cast<const f32x4 *>(a0_ + a * 32 * LDS_LD); \
        _Pragma("unroll") for (int b = 0; b < TN; ++b) FB[b] = *reinterpret_cast<const f32x4 *>(b0_ + b * 32 * LDS_LD); \
    }
#define V3_MFMA(FA, FB)                                                                                         \
    {                                                                                                           \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                           \
            _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                      \
                _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                  \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[a][e], FB[b][e], acc[a][b], 0, 0, 0);   \
    }
    // one k-step: tile kt sits in LDS buffer SET, tile kt+1 in register set SET^1, tile kt+2 is requested into SET
#define V3_STEP(SET, kt)                                                                                        \
    {                                                                                                           \
        const bool next1_ = (kt) + 1 < nk, next2_ = (kt) + 2 < nk;                                              \
        if (next2_) V3_LOAD_GROUP(SET, 0);                                                                      \
        V3_FRAGS(fa1, fb1, SET, 1);                                                                             \
        V3_MFMA(fa0, fb0);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (next2_) V3_LOAD_GROUP(SET, 1);                                                                      \
        if (next1_) V3_STORE(SET ^ 1, SET ^ 1);                                                                 \
        V3_FRAGS(fa0, fb0, SET, 2);                                                                             \
        V3_MFMA(fa1, fb1);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (next2_) V3_LOAD_GROUP(SET, 2);                                                                      \
        V3_FRAGS(fa1, fb1, SET, 3);                                                                             \
        V3_MFMA(fa0, fb0);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (next2_) { V3_LOAD_GROUP(SET, 3); V3_ADVANCE(); }                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                         \
        if (next1_) V3_FRAGS(fa0, fb0, SET ^ 1, 0);                                                             \
        V3_MFMA(fa1, fb1);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    }

    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    // ---- prologue: tile 0 -> set 0 -> LDS buffer 0 ; tile 1 -> set 1 (in flight)
    V3_LOAD_GROUP(0, 0); V3_LOAD_GROUP(0, 1); V3_LOAD_GROUP(0, 2); V3_LOAD_GROUP(0, 3);
    V3_ADVANCE();
    V3_STORE(0, 0);
    if (nk > 1) {
        V3_LOAD_GROUP(1, 0); V3_LOAD_GROUP(1, 1); V3_LOAD_GROUP(1, 2); V3_LOAD_GROUP(1, 3);
        V3_ADVANCE();
    }
    __syncthreads();
    V3_FRAGS(fa0, fb0, 0, 0);

    for (int kt = 0; kt < nk; kt += 2) {
        V3_STEP(0, kt);
        if (kt + 1 < nk) V3_STEP(1, kt + 1);
    }
#undef V3_LOAD
#undef V3_LOAD_GROUP
#undef V3_ADVANCE
#undef V3_STORE
#undef V3_FRAGS
#undef V3_MFMA
#undef V3_STEP
    __syncthreads();   // every wave is done with the tile buffers

    // ---- epilogue (identical to conv_igemm_f32)
    float *sC = smem;
#pragma unroll
    for (int half = 0; half < EH; ++half) {
    if (half > 0) __syncthreads();
    if ((wm * WM) / HR == half) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm * WM - half * HR + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    sC[row * LDC + wn * WN + b * 32 + l31] = acc[a][b][e];
                }
    }
    __syncthreads();
    const int m0 = mt * BM + half * HR, n0 = nt * BN;
    const bool vec = ((p.ldc & 3) == 0) && (p.res == nullptr || (p.ldr & 3) == 0);
    if (vec) {
        constexpr int TPR = BN / 4, RPP = NT / TPR, NPASS = HR / RPP, UB = NPASS < 4 ? NPASS : 4;
        const int c4 = tid % TPR, r0 = tid / TPR;
        const int col = n0 + 4 * c4;
        const int cend = p.Cout + 3 < p.ldc ? ((p.Cout + 3) & ~3) : p.ldc;
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
                    const int r = r0 + (g + u) * RPP, m = m0 + r;
                    okr[u] = m < p.M;
                    orow[u] = (size_t)m;
                    size_t rrow = (size_t)m;
                    if (GENERIC) {
                        if (p.scatter) {
                            const int n = m / HoWo, rem = m - n * HoWo;
                            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                            orow[u] = ((size_t)n * (p.Ho * p.osy) + (ho * p.osy + p.ooy)) * (size_t)(p.Wo * p.osx) + (wo * p.osx + p.oox);
                        }
                        if (p.rg_out) rrow = (size_t)(m / p.rg_out) * p.rg_in + (m % p.rg_out);
                    }
                    const float *rp = (has_res && okr[u]) ? p.res + rrow * p.ldr + col : zero;
                    rv[u] = *reinterpret_cast<const f32x4 *>(rp);
                    v[u] = *reinterpret_cast<const f32x4 *>(&sC[r * LDC + 4 * c4]);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    f32x4 t = v[u] + bv + rv[u];
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
                    if (okr[u]) *reinterpret_cast<f32x4 *>(p.out + orow[u] * p.ldc + col) = t;
                }
            }
        }
    } else if (GENERIC) {
        for (int idx = tid; idx < HR * BN; idx += NT) {
            const int r = idx / BN, c = idx - r * BN;
            const int m = m0 + r, col = n0 + c;
            if (m >= p.M || col >= p.Cout) continue;
            size_t orow = (size_t)m, rrow = (size_t)m;
            if (p.scatter) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                orow = ((size_t)n * (p.Ho * p.osy) + (ho * p.osy + p.ooy)) * (size_t)(p.Wo * p.osx) + (wo * p.osx + p.oox);
            }
            if (p.rg_out) rrow = (size_t)(m / p.rg_out) * p.rg_in + (m % p.rg_out);
            float v = sC[r * LDC + c] + p.bias[col];
            if (p.res) v += p.res[rrow * p.ldr + col];
            if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
            else if (p.act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
            else if (p.act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
            p.out[orow * p.ldc + col] = v;
        }
    }
    }  // staging halves
}

// ====================================================================== v4: LDS-DMA main loop
// Tiles go HBM/L2 -> LDS by `global_load_lds_dwordx4` (no VGPR round trip, no ds_write: the
// VGPR->LDS store path cost ~10 % of the matrix pipe in the register-staged kernels).
//   * LDS image: [row][32 floats] unpadded (128-byte rows).  One wave-instruction fills 8 rows
//     (1 KiB, lane-linear).  Bank conflicts of the ds_read_b128 operand fetch are removed by an XOR
//     swizzle on the 16-byte chunk index, chunk' = chunk ^ ((row >> 1) & 7); because the DMA
//     destination is lane-linear the swizzle is applied to the per-lane SOURCE address and to the read.
//   * out-of-range taps / rows DMA from a zero page.
//   * per k-step: MFMA groups q0..q2, then {vmcnt(0) for tile t+1, s_barrier}, then the DMA of tile
//     t+2 is issued into the buffer just vacated and the first fragments of tile t+1 are fetched,
//     all under the 16 MFMAs of group q3.
#define HMV_GLDS16(gptr, lptr)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// LDS budget of the v4 kernel: the two tile buffers, or one epilogue staging pass if that is larger.
constexpr int v4_stage_blocks(int BM, int BN, int WGM) {
    const int TM = BM / WGM / 32;
    int as = TM;   // largest divisor of TM whose staging pass fits next to nothing else in 140 KB
    while (as > 1 && (WGM * as * 32 * (BN + 4) * 4 > 140 * 1024 || TM % as != 0)) --as;
    // do not stage more than the tile buffers already provide unless a single block forces it
    while (as > 1 && WGM * as * 32 * (BN + 4) > 2 * (BM + BN) * BK + 1024) --as, as = (TM % as == 0) ? as : as - 1;
    return as < 1 ? 1 : as;
}
constexpr int v4_lds_floats(int BM, int BN, int WGM) {
    const int tile = 2 * (BM + BN) * BK;
    const int stage = WGM * v4_stage_blocks(BM, BN, WGM) * 32 * (BN + 4);
    return tile > stage ? tile : stage;
}

template <int BM, int BN, int WGM, int WGN, bool IS1X1, bool GENERIC>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_f32_v4(const ConvParams p) {
    constexpr int NT = 64 * WGM * WGN, NW = WGM * WGN;
    constexpr int RPS = NT / 8;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int AP = BM / RPS, BP = BN / RPS;
    constexpr int LDC = BN + 4;
    // epilogue staging: per pass every wave stages AS of its 32-row blocks -> SR = WGM*AS*32 staged rows
    constexpr int LDS_FLOATS = v4_lds_floats(BM, BN, WGM);
    constexpr int AS = v4_stage_blocks(BM, BN, WGM);
    constexpr int SR = WGM * AS * 32;
    static_assert(BM % RPS == 0 && BN % RPS == 0 && TM % AS == 0 && SR * LDC <= LDS_FLOATS, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;                  // [2][BM][32]
    float *sB = smem + 2 * BM * BK;    // [2][BN][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    unsigned long long t_entry = 0;
    if (p.dbg) t_entry = __builtin_amdgcn_s_memrealtime();

    int mt, nt;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r = nblk & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        mt = lid / p.ntiles;
        nt = lid - mt * p.ntiles;
    }

    // DMA roles: thread -> (row lrow of each 32-row pass, physical chunk tid&7); it fetches the
    // LOGICAL chunk kqs so that the lane-linear LDS image ends up XOR-swizzled.
    const int lrow = tid >> 3;
    const int kqs = (tid & 7) ^ ((tid >> 4) & 7);
    const float *zero = p.zero;
    const float *aptr[AP];
    int astep[AP], hi0[AP], wi0[AP];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = mt * BM + i * RPS + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const float *base = p.in + (size_t)n * p.H * p.W * p.lda;
        hi0[i] = ok ? ho * p.stride - p.pad_h : -(1 << 28);
        wi0[i] = wo * p.stride - p.pad_w;
        if (IS1X1) {
            aptr[i] = ok ? base + (hi0[i] * p.W + wi0[i]) * p.lda + 4 * kqs : zero;
            astep[i] = ok ? BK : 0;
        } else {
            aptr[i] = base + (ok ? (hi0[i] * p.W + wi0[i]) * p.lda + 4 * kqs : 0);
            astep[i] = 0;
        }
    }
    const float *wptr[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) wptr[i] = p.wgt + (size_t)(nt * BN + i * RPS + lrow) * p.ldw + 4 * kqs;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    int cr = 0, cs = 0, cc = 0, cdelta = 0;   // load cursor (wave-uniform)
    const int nk = p.Kpad / BK;

    // issue the DMA of the cursor tile into LDS buffer `buf`, then advance the cursor
#define V4_DMA(buf)                                                                                         \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                                    \
            const float *src_;                                                                              \
            if (IS1X1) {                                                                                    \
                src_ = aptr[i];                                                                             \
                aptr[i] += astep[i];                                                                        \
            } else {                                                                                        \
                const bool ok_ = (unsigned)(hi0[i] + cr) < (unsigned)p.H && (unsigned)(wi0[i] + cs) < (unsigned)p.W; \
                src_ = ok_ ? aptr[i] + cdelta : zero;                                                       \
            }                                                                                               \
            if (p.tune & 32) src_ = zero;                                                                   \
            HMV_GLDS16(src_, sA + ((buf) * BM + i * RPS + wave * 8) * BK);                                  \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < BP; ++i) {                                                    \
            HMV_GLDS16((p.tune & 32) ? zero : wptr[i], sB + ((buf) * BN + i * RPS + wave * 8) * BK);        \
            wptr[i] += BK;                                                                                  \
        }                                                                                                   \
        if (!IS1X1) {                                                                                       \
            cc += BK;                                                                                       \
            cdelta += BK;                                                                                   \
            if (cc >= p.Cin) {                                                                              \
                cc = 0;                                                                                     \
                cdelta -= p.Cin;                                                                            \
                if (++cs == p.S) { cs = 0; ++cr; cdelta += (p.W - p.S) * p.lda; }                           \
                cdelta += p.lda;                                                                            \
            }                                                                                               \
        }                                                                                                   \
    }
    // operand fetch: lane (row l31, k-half kh) reads logical chunk 2q+kh at its swizzled position
    const int fsw = (l31 >> 1) & 7;
    const float *arow = sA + (wm * WM + l31) * BK;
    const float *brow = sB + (wn * WN + l31) * BK;
#define V4_FRAGS(FA, FB, buf, q)                                                                            \
    {                                                                                                       \
        const int ch_ = ((2 * (q) + kh) ^ fsw) * 4;                                                         \
        _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                      \
            FA[a] = *reinterpret_cast<const f32x4 *>(arow + ((buf) * BM + a * 32) * BK + ch_);              \
        _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                      \
            FB[b] = *reinterpret_cast<const f32x4 *>(brow + ((buf) * BN + b * 32) * BK + ch_);              \
    }
#define V4_MFMA(FA, FB)                                                                                     \
    {                                                                                                       \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                       \
            _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                  \
                _Pragma("unroll") for (int b = 0; b < TN; ++b)                                              \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[a][e], FB[b][e], acc[a][b], 0, 0, 0); \
    }

    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    unsigned long long t0c = 0, t0r = 0;
    if (p.dbg) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
    V4_DMA(0);
    if (nk > 1) {
        V4_DMA(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + BP) : "memory");   // tile 0 landed, tile 1 may fly
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    V4_FRAGS(fa0, fb0, 0, 0);

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        V4_FRAGS(fa1, fb1, buf, 1);
        V4_MFMA(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        V4_FRAGS(fa0, fb0, buf, 2);
        V4_MFMA(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        V4_FRAGS(fa1, fb1, buf, 3);
        V4_MFMA(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        // tile kt+1 (the only DMA in flight) must have landed; everyone is done reading `buf`
        if (p.tune & 256) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nk && !(p.tune & 4)) V4_DMA(buf);
        if (kt + 1 < nk) V4_FRAGS(fa0, fb0, buf ^ 1, 0);
        V4_MFMA(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef V4_DMA
#undef V4_FRAGS
#undef V4_MFMA
    unsigned long long t1c = 0, t1r = 0;
    if (p.dbg) { t1c = __builtin_amdgcn_s_memtime(); t1r = __builtin_amdgcn_s_memrealtime(); }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- epilogue: staged through LDS in passes of AS 32-row blocks per wave, then 16-byte row stores
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
        // staged row sr belongs to wave-row (sr / (AS*32)), block aa = (sr / 32) % AS, line sr % 32
        auto tile_row = [&](int sr) { return (sr / (AS * 32)) * WM + (pass * AS + (sr / 32) % AS) * 32 + (sr & 31); };
        if (vec) {
            constexpr int TPR = BN / 4, RPP = NT / TPR, NPASS = SR / RPP, UB = NPASS < 4 ? NPASS : 4;
            static_assert(SR % RPP == 0 && NPASS % UB == 0, "staging pass shape");
            const int c4 = tid % TPR, r0 = tid / TPR;
            const int col = n0 + 4 * c4;
            const int cend = p.Cout + 3 < p.ldc ? ((p.Cout + 3) & ~3) : p.ldc;
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
                            if (p.scatter) {
                                const int n = m / HoWo, rem = m - n * HoWo;
                                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                                orow[u] = ((size_t)n * (p.Ho * p.osy) + (ho * p.osy + p.ooy)) * (size_t)(p.Wo * p.osx) + (wo * p.osx + p.oox);
                            }
                            if (p.rg_out) rrow = (size_t)(m / p.rg_out) * p.rg_in + (m % p.rg_out);
                        }
                        const float *rp = (has_res && okr[u]) ? p.res + rrow * p.ldr + col : zero;
                        rv[u] = *reinterpret_cast<const f32x4 *>(rp);
                        v[u] = *reinterpret_cast<const f32x4 *>(&sC[sr * LDC + 4 * c4]);
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        f32x4 t = v[u] + bv + rv[u];
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
                        if (okr[u]) *reinterpret_cast<f32x4 *>(p.out + orow[u] * p.ldc + col) = t;
                    }
                }
            }
        } else if (GENERIC) {
            for (int idx = tid; idx < SR * BN; idx += NT) {
                const int sr = idx / BN, c = idx - sr * BN;
                const int m = mt * BM + tile_row(sr), col = n0 + c;
                if (m >= p.M || col >= p.Cout) continue;
                size_t orow = (size_t)m, rrow = (size_t)m;
                if (p.scatter) {
                    const int n = m / HoWo, rem = m - n * HoWo;
                    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    orow = ((size_t)n * (p.Ho * p.osy) + (ho * p.osy + p.ooy)) * (size_t)(p.Wo * p.osx) + (wo * p.osx + p.oox);
                }
                if (p.rg_out) rrow = (size_t)(m / p.rg_out) * p.rg_in + (m % p.rg_out);
                float v = sC[sr * LDC + c] + p.bias[col];
                if (p.res) v += p.res[rrow * p.ldr + col];
                if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (p.act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
                else if (p.act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
                p.out[orow * p.ldc + col] = v;
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
}

int conv_tile_bn(ConvTile t) {
    switch (t) {
        case TILE_128x64: case TILE_V4_128x64: return 64;
        case TILE_128x32: case TILE_V4_128x32: return 32;
        case TILE_V4_128x256: case TILE_V4_256x256: return 256;
        default: return 128;
    }
}

const char *conv_tile_name(ConvTile t, bool smallc) {
    if (smallc) return "conv_igemm_f32<128x64,smallc>";
    switch (t) {
        case TILE_128x128: return "conv_igemm_f32<128x128>";
        case TILE_128x64: return "conv_igemm_f32<128x64>";
        case TILE_128x32: return "conv_igemm_f32<128x32>";
        case TILE_128x128_8W: return "conv_igemm_f32<128x128,8w>";
        case TILE_256x128_8W: return "conv_igemm_f32<256x128,8w>";
        case TILE_V3_128x128: return "conv_igemm_f32_v3<128x128>";
        case TILE_V4_128x128: return "conv_igemm_f32_v4<128x128>";
        case TILE_V4_128x64: return "conv_igemm_f32_v4<128x64>";
        case TILE_V4_128x32: return "conv_igemm_f32_v4<128x32>";
        case TILE_V4_256x128: return "conv_igemm_f32_v4<256x128>";
        case TILE_V4_128x256: return "conv_igemm_f32_v4<128x256>";
        case TILE_V4_256x256: return "conv_igemm_f32_v4<256x256>";
        default: return "conv_igemm_f32<?>";
    }
}

ConvTile conv_pick_tile(int M, int Cout) {
    static int forced = -2;
    if (forced == -2) { const char *e = getenv("HMV_FORCE_TILE"); forced = e ? atoi(e) : -1; }
    if (Cout > 64 && forced >= 0 && forced < TILE_COUNT) return (ConvTile)forced;
    // Measured on MI355X (tools/conv_sweep.py): the matrix pipe is DVFS/power limited, so the tile with
    // the least L2->LDS traffic per FLOP wins as long as it still fills the 256 CUs for several rounds.
    if (Cout > 128 && (long long)((M + 255) / 256) * ((Cout + 255) / 256) >= 512) return TILE_V4_256x256;
    if (Cout > 64 && (long long)((M + 255) / 256) * ((Cout + 127) / 128) >= 512) return TILE_V4_256x128;
    if (Cout > 64) return TILE_V4_128x128;
    if (Cout > 32) return TILE_V4_128x64;
    return TILE_V4_128x32;
}

template <int BM, int BN, int WGM, int WGN, bool SMALLC, bool GENERIC>
static hipError_t launch_one(ConvParams p, hipStream_t s) {
    static bool configured = false;
    const size_t lds = 2ull * (BM + BN) * LDS_LD * sizeof(float);
    auto kern = conv_igemm_f32<BM, BN, WGM, WGN, SMALLC, GENERIC>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = true;
    }
    {
        hipError_t e = ensure_zero_page();
        if (e != hipSuccess) return e;
    }
    p.zero = g_zero_page;
    {
        static int tune = -1;
        if (tune < 0) { const char *e = getenv("HMV_TUNE"); tune = e ? atoi(e) : 0; }
        p.tune = tune;
    }
    if (!p.lda) p.lda = p.Cin;
    if (!p.ldw) p.ldw = p.Kpad;
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(64 * WGM * WGN), lds, s, p);
    return hipGetLastError();
}

template <int BM, int BN, int WGM, int WGN, bool IS1X1, bool GENERIC>
static hipError_t launch_v3(ConvParams p, hipStream_t s) {
    static bool configured = false;
    const size_t lds = 2ull * (BM + BN) * LDS_LD * sizeof(float);
    auto kern = conv_igemm_f32_v3<BM, BN, WGM, WGN, IS1X1, GENERIC>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = true;
    }
    {
        hipError_t e = ensure_zero_page();
        if (e != hipSuccess) return e;
    }
    p.zero = g_zero_page;
    p.tune = 0;
    if (!p.lda) p.lda = p.Cin;
    if (!p.ldw) p.ldw = p.Kpad;
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(64 * WGM * WGN), lds, s, p);
    return hipGetLastError();
}

template <int BM, int BN, int WGM, int WGN, bool IS1X1, bool GENERIC>
static hipError_t launch_v4(ConvParams p, hipStream_t s) {
    static bool configured = false;
    const size_t lds = (size_t)v4_lds_floats(BM, BN, WGM) * sizeof(float);
    auto kern = conv_igemm_f32_v4<BM, BN, WGM, WGN, IS1X1, GENERIC>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = true;
    }
    {
        hipError_t e = ensure_zero_page();
        if (e != hipSuccess) return e;
    }
    p.zero = g_zero_page;
    {
        static int tune = -1;
        if (tune < 0) { const char *e = getenv("HMV_TUNE"); tune = e ? atoi(e) : 0; }
        p.tune = tune;
    }
    if (!p.lda) p.lda = p.Cin;
    if (!p.ldw) p.ldw = p.Kpad;
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(64 * WGM * WGN), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv(ConvParams p, ConvTile tile, hipStream_t s) {
    if (p.M <= 0) return hipSuccess;
    const bool generic = p.scatter || p.rg_out || p.act == ACT_GELU || p.act == ACT_LEAKY || (p.ldc & 3) ||
                         (p.res && (p.ldr & 3));
    if (p.Cin < BK) {  // stem: Cin == 4
        if (p.Cin != 4 || generic) return hipErrorInvalidValue;
        return launch_one<128, 64, 2, 2, true, false>(p, s);
    }
    if (p.Cin % BK != 0) return hipErrorInvalidValue;
    if (generic && (tile == TILE_V4_256x128 || tile == TILE_V4_128x256 || tile == TILE_V4_256x256 ||
                    tile == TILE_128x128_8W || tile == TILE_256x128_8W))
        tile = TILE_V4_128x128;   // the rarely used epilogue paths exist only for the 4-wave tiles
    switch (tile) {
        case TILE_128x128:
            return generic ? launch_one<128, 128, 2, 2, false, true>(p, s) : launch_one<128, 128, 2, 2, false, false>(p, s);
        case TILE_128x64:
            return generic ? launch_one<128, 64, 2, 2, false, true>(p, s) : launch_one<128, 64, 2, 2, false, false>(p, s);
        case TILE_128x32:
            return generic ? launch_one<128, 32, 4, 1, false, true>(p, s) : launch_one<128, 32, 4, 1, false, false>(p, s);
        case TILE_128x128_8W:
            return generic ? hipErrorInvalidValue : launch_one<128, 128, 2, 4, false, false>(p, s);
        case TILE_256x128_8W:
            return generic ? hipErrorInvalidValue : launch_one<256, 128, 4, 2, false, false>(p, s);
        case TILE_V4_128x128: {
            const bool one = p.R == 1 && p.S == 1 && p.pad_h == 0 && p.pad_w == 0;
            if (generic) return one ? launch_v4<128, 128, 2, 2, true, true>(p, s) : launch_v4<128, 128, 2, 2, false, true>(p, s);
            return one ? launch_v4<128, 128, 2, 2, true, false>(p, s) : launch_v4<128, 128, 2, 2, false, false>(p, s);
        }
        case TILE_V4_128x64: {
            const bool one = p.R == 1 && p.S == 1 && p.pad_h == 0 && p.pad_w == 0;
            if (generic) return one ? launch_v4<128, 64, 2, 2, true, true>(p, s) : launch_v4<128, 64, 2, 2, false, true>(p, s);
            return one ? launch_v4<128, 64, 2, 2, true, false>(p, s) : launch_v4<128, 64, 2, 2, false, false>(p, s);
        }
        case TILE_V4_128x32: {
            const bool one = p.R == 1 && p.S == 1 && p.pad_h == 0 && p.pad_w == 0;
            if (generic) return one ? launch_v4<128, 32, 4, 1, true, true>(p, s) : launch_v4<128, 32, 4, 1, false, true>(p, s);
            return one ? launch_v4<128, 32, 4, 1, true, false>(p, s) : launch_v4<128, 32, 4, 1, false, false>(p, s);
        }
        case TILE_V4_256x128:
            if (generic) return hipErrorInvalidValue;
            return (p.R == 1 && p.S == 1 && p.pad_h == 0 && p.pad_w == 0) ? launch_v4<256, 128, 4, 2, true, false>(p, s)
                                                                         : launch_v4<256, 128, 4, 2, false, false>(p, s);
        case TILE_V4_128x256:
            if (generic) return hipErrorInvalidValue;
            return (p.R == 1 && p.S == 1 && p.pad_h == 0 && p.pad_w == 0) ? launch_v4<128, 256, 2, 4, true, false>(p, s)
                                                                         : launch_v4<128, 256, 2, 4, false, false>(p, s);
        case TILE_V4_256x256:
            if (generic) return hipErrorInvalidValue;
            return (p.R == 1 && p.S == 1 && p.pad_h == 0 && p.pad_w == 0) ? launch_v4<256, 256, 2, 4, true, false>(p, s)
                                                                         : launch_v4<256, 256, 2, 4, false, false>(p, s);
        case TILE_V3_128x128: {
            const bool one = p.R == 1 && p.S == 1 && p.pad_h == 0 && p.pad_w == 0;
            if (generic) return one ? launch_v3<128, 128, 2, 2, true, true>(p, s) : launch_v3<128, 128, 2, 2, false, true>(p, s);
            return one ? launch_v3<128, 128, 2, 2, true, false>(p, s) : launch_v3<128, 128, 2, 2, false, false>(p, s);
        }
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace hmv
