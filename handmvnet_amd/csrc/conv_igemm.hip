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
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvParams p) {
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int AP = BM / 32, BP = BN / 32;
    constexpr int LDC = BN + 4;  // epilogue staging row stride (floats)
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile is made of 32x32 MFMA blocks");
    static_assert(BM * LDC <= 2 * (BM + BN) * LDS_LD, "epilogue staging must fit in the tile buffers");
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
        const int m = mt * BM + i * 32 + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        abase[i] = p.in + (size_t)n * p.H * p.W * p.Cin;
        hi0[i] = ok ? ho * p.stride - p.pad_h : -(1 << 28);  // out-of-range rows fail the bounds test
        wi0[i] = wo * p.stride - p.pad_w;
        aoff[i] = ok ? (hi0[i] * p.W + wi0[i]) * p.Cin + (SMALLC ? 0 : 4 * kq) : 0;
    }
    const float *wrow[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) wrow[i] = p.wgt + (size_t)(nt * BN + i * 32 + lrow) * p.Kpad + 4 * kq;

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
            delta_ = (r_ * p.W + s_) * p.Cin + (k0 - tap * p.Cin);                                \
        } else { /* Cin == 4: one tap per 16-byte vector */                                       \
            const int tap = (k0 >> 2) + kq;                                                       \
            r_ = tap / p.S;                                                                       \
            s_ = tap - r_ * p.S;                                                                  \
            if (tap >= p.R * p.S) r_ = 1 << 28;                                                   \
            delta_ = (r_ * p.W + s_) * 4;                                                         \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < AP; ++i) {                                          \
            const bool ok = (unsigned)(hi0[i] + r_) < (unsigned)p.H && (unsigned)(wi0[i] + s_) < (unsigned)p.W; \
            const float *src = ok ? abase[i] + (aoff[i] + delta_) : zero;                         \
            ra[i] = *reinterpret_cast<const f32x4 *>(src);                                        \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < BP; ++i) rb[i] =                                    \
            *reinterpret_cast<const f32x4 *>(wrow[i] + k0);                                       \
    }

#define HMV_STORE_TILE(buf)                                                                       \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < AP; ++i)                                            \
            *reinterpret_cast<f32x4 *>(&sA[((buf) * BM + i * 32 + lrow) * LDS_LD + 4 * kq]) = ra[i]; \
        _Pragma("unroll") for (int i = 0; i < BP; ++i)                                            \
            *reinterpret_cast<f32x4 *>(&sB[((buf) * BN + i * 32 + lrow) * LDS_LD + 4 * kq]) = rb[i]; \
    }

    HMV_LOAD_TILE(0);
    HMV_STORE_TILE(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) HMV_LOAD_TILE(kt + 1);
        const float *a0 = &sA[(buf * BM + wm * WM + l31) * LDS_LD + 4 * kh];
        const float *b0 = &sB[(buf * BN + wn * WN + l31) * LDS_LD + 4 * kh];
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const f32x4 *>(a0 + a * 32 * LDS_LD + 8 * q);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const f32x4 *>(b0 + b * 32 * LDS_LD + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][e], bf[b][e], acc[a][b], 0, 0, 0);
        }
        if (more) HMV_STORE_TILE(buf ^ 1);
        __syncthreads();
    }
#undef HMV_LOAD_TILE
#undef HMV_STORE_TILE

    // ---- fused epilogue, staged through LDS so that global traffic is 16-byte vectors on full rows.
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    float *sC = smem;  // [BM][LDC]; the tile buffers are dead after the loop's last barrier
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * WM + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                sC[row * LDC + wn * WN + b * 32 + l31] = acc[a][b][e];
            }
    __syncthreads();

    const int m0 = mt * BM, n0 = nt * BN;
    const bool vec = ((p.ldc & 3) == 0) && (p.res == nullptr || (p.ldr & 3) == 0);
    if (vec) {
        constexpr int TPR = BN / 4;          // threads per row
        constexpr int RPP = 256 / TPR;       // rows per pass
        constexpr int NPASS = BM / RPP;
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
        for (int idx = tid; idx < BM * BN; idx += 256) {
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
}

int conv_tile_bn(ConvTile t) { return t == TILE_128x128 ? 128 : (t == TILE_128x64 ? 64 : 32); }

const char *conv_tile_name(ConvTile t, bool smallc) {
    if (smallc) return "conv_igemm_f32<128x64,smallc>";
    switch (t) {
        case TILE_128x128: return "conv_igemm_f32<128x128>";
        case TILE_128x64: return "conv_igemm_f32<128x64>";
        default: return "conv_igemm_f32<128x32>";
    }
}

ConvTile conv_pick_tile(int M, int Cout) {
    (void)M;
    if (Cout > 64) return TILE_128x128;
    if (Cout > 32) return TILE_128x64;
    return TILE_128x32;
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
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(256), lds, s, p);
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
    switch (tile) {
        case TILE_128x128:
            return generic ? launch_one<128, 128, 2, 2, false, true>(p, s) : launch_one<128, 128, 2, 2, false, false>(p, s);
        case TILE_128x64:
            return generic ? launch_one<128, 64, 2, 2, false, true>(p, s) : launch_one<128, 64, 2, 2, false, false>(p, s);
        default:
            return generic ? launch_one<128, 32, 4, 1, false, true>(p, s) : launch_one<128, 32, 4, 1, false, false>(p, s);
    }
}

}  // namespace hmv
