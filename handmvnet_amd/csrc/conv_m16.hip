// conv_m16.hip -- the small-launch companion of the fp16 kernels that multiply on v_mfma_f32_16x16x32_f16 (round 4).
//
// Round 4 measured the MFMA-shape lever of MI355X_MICROARCH.md ("DVFS give-back" item 7) on this path's own MFMA-bound kernel
// (profiles/r04_probe_mfma_shape.txt): conv_ht on the 16x16x32 fp16 MFMA takes 12-15 % less time than on 32x32x16 at equal cycles
// per FLOP.  A 16x16x32 MFMA sums 32 reduction elements inside ONE instruction where two 32x32x16 MFMAs sum 16 each, so its bits
// differ (last fp16 bit of ~1 % of the outputs), and a sample's bits must not depend on its batch: a layer that runs on the
// 16x16x32 form at large batches runs on it at EVERY batch.  This file is that "every other batch" kernel for
//   * the tall-tile 3x3 layers (conv2 of layer2 / layer3's Bottlenecks, /root/reference/src/models/backbones/resnet.py:114-118,
//     132-134; K order (32-channel chunk, r, s, c % 32), the packing of conv_ht.hip),
//   * plain 1x1 convs without a residual whose reduction is a multiple of 32 channels.
// 64 x 64 and 128 x 128 tiles, four waves (2 x 2), a k-step = 32 channels (one tap) = ONE MFMA per 16 x 16 block; operand roles
// (weights = A, pixels = B: the lane is the pixel), bias as the accumulators' initial value, k-ascending accumulation and the
// epilogue arithmetic are conv_ht's, so the two agree bit for bit (tests/test_gpu_parity.py::test_tall_tile_kernel).
// Loop: a ring of four LDS stages, three tiles in flight, ONE counted `s_waitcnt vmcnt(N)` + barrier per k-step (N static: tiles
// past the end are fetched from the zero page so that the count holds to the last step).
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace hmv {

typedef float mf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 mf16x8 __attribute__((ext_vector_type(8)));

#define HMV_MGLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

constexpr int M16_NS = 4;   // LDS stages (tiles kt .. kt + 3)

// LDS image: rows of 32 halfs (64 bytes, four 16-byte chunks); chunk' = chunk ^ (((row >> 2) & 1) << 1).  A fragment read's lane
// (l15, kg) takes chunk kg of row 16 blk + l15; a ds_read_b128's 16-lane service group is rows 0-3, 12-15 of k-group g with rows 4-11
// of k-group g ^ 1, and this key puts the 16 on distinct bank quads (tools/probe/swizzle_search.py).
__device__ __forceinline__ int m16_key(int row) { return ((row >> 2) & 1) << 1; }

// TAPS: 3x3 stride 1 pad 1 in conv_ht's K order; else a plain 1x1 (stride 1) over rows of lda halfs.  DUAL (1x1 only): the reduction is
// the concatenation [first source | second source sampled at (ho, wo) * stride2] (conv3 + downsample as one GEMM, ConvParams::in2)
template <int BM, int BN, bool TAPS, bool DUAL = false>
__global__ __launch_bounds__(256) void conv_m16_f16(const ConvParams p) {
    static_assert(!(TAPS && DUAL), "two sources: 1x1 only");
    constexpr int WM = BM / 2, WN = BN / 2, PB = WM / 16, CB = WN / 16, AP = BM / 64, BP = BN / 64;
    static_assert(CB % 2 == 0, "a lane's channel blocks pair up into 8 consecutive channels");
    extern __shared__ __attribute__((aligned(16))) char msm[];   // [M16_NS][BM + BN][64 bytes]
    constexpr int STAGE = (BM + BN) * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4, wm = wave >> 1, wn = wave & 1;

    int mt, nt;
    {   // the N-tiles of a pixel tile are consecutive workgroups of one XCD (bijective for any grid size)
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r = nblk & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        mt = lid / p.ntiles;
        nt = lid - mt * p.ntiles;
    }
    const _Float16 *zero16 = reinterpret_cast<const _Float16 *>(p.zero);
    const int nk = TAPS ? 9 * (p.Cin >> 5) : (p.Kpad >> 5);

    // ---- DMA roles: thread -> row tid >> 2 of each 64-row pass, physical chunk tid & 3 holding logical chunk (tid & 3) ^ key(row)
    const int lrow = tid >> 2, lc = (tid & 3) ^ m16_key(lrow);
    const _Float16 *aptr[AP], *aptr2[AP];
    int hi0[AP], wi0[AP];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = mt * BM + i * 64 + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        if constexpr (TAPS) {
            const int n = mm / HoWo, rem = mm - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hi0[i] = ok ? ho - 1 : -(1 << 28);   // rows past M fail every bounds test
            wi0[i] = wo - 1;
            aptr[i] = reinterpret_cast<const _Float16 *>(p.in) + ((size_t)(n * p.H + ho - 1) * p.W + (wo - 1)) * p.lda + 8 * lc;
        } else {
            hi0[i] = wi0[i] = 0;
            aptr[i] = ok ? reinterpret_cast<const _Float16 *>(p.in) + (size_t)mm * p.lda + 8 * lc : nullptr;
        }
        aptr2[i] = nullptr;
        if constexpr (DUAL) {
            if (ok) {
                const int n = mm / HoWo, rem = mm - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                aptr2[i] = reinterpret_cast<const _Float16 *>(p.in2) + ((size_t)(n * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.lda2 + 8 * lc;
            }
        }
    }
    // weight LDS row R = 32 u + 16 e + rho holds channel 32 u + 8 (rho >> 2) + 4 e + (rho & 3): block pair (2 t, 2 t + 1) of a lane then
    // holds 8 consecutive channels (the MFMA's D row of lane group kg, register r is rho = 4 kg + r)
    const _Float16 *wptr[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int R = i * 64 + lrow, u = R >> 5, e = (R >> 4) & 1, rho = R & 15;
        const int ch = 32 * u + 8 * (rho >> 2) + 4 * e + (rho & 3);
        wptr[i] = reinterpret_cast<const _Float16 *>(p.wgt) + (size_t)(nt * BN + ch) * p.ldw + 8 * lc;
    }
    int cr = 0, cs = 0, cchunk = 0, ck = 0;   // DMA cursor (wave-uniform): tap (cr, cs) of 32-channel chunk cchunk; ck = step index
    auto dma = [&]() {   // the tile of step ck into stage ck & 3, then advance (past the end: dummies from the zero page)
        char *st = msm + (ck & (M16_NS - 1)) * STAGE;
        const bool live = ck < nk;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const _Float16 *src;
            if constexpr (TAPS) {
                const bool ok = live && (unsigned)(hi0[i] + cr) < (unsigned)p.H && (unsigned)(wi0[i] + cs) < (unsigned)p.W;
                src = ok ? aptr[i] + ((size_t)cr * p.W + cs) * p.lda + 32 * cchunk : zero16;
            } else if constexpr (DUAL) {
                const int k0 = 32 * ck;
                src = (live && aptr[i]) ? (k0 < p.ksplit ? aptr[i] + k0 : aptr2[i] + (k0 - p.ksplit)) : zero16;
            } else {
                src = (live && aptr[i]) ? aptr[i] + 32 * ck : zero16;
            }
            asm volatile("" : "+v"(src));
            HMV_MGLDS16(src, st + (i * 64 + wave * 16) * 64);
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const _Float16 *src = live ? wptr[i] + 32 * ck : zero16;
            asm volatile("" : "+v"(src));
            HMV_MGLDS16(src, st + (BM + i * 64 + wave * 16) * 64);
        }
        ++ck;
        if constexpr (TAPS) { if (++cs == 3) { cs = 0; if (++cr == 3) { cr = 0; ++cchunk; } } }
    };

    // ---- accumulators start at the bias: acc[a][cb][r], cb = 2 t + e, = channel 32 t + 8 kg + 4 e + r of pixel 16 a + l15
    mf32x4 acc[PB][CB];
    {
        const float binit = 1.f / p.acc_scale;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const mf32x4 bq = *reinterpret_cast<const mf32x4 *>(p.bias + nt * BN + wn * WN + 32 * (cb >> 1) + 8 * kg + 4 * (cb & 1));
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

    const int foff_p = (wm * WM + l15) * 64 + (kg ^ m16_key(l15)) * 16;        // + 16 a rows
    const int foff_w = (BM + wn * WN + l15) * 64 + (kg ^ m16_key(l15)) * 16;   // + 16 cb rows
    mf16x8 fp0[PB], fw0[CB], fp1[PB], fw1[CB];
#define M16_READ(FP, FW, kt_)                                                                               \
    {                                                                                                       \
        const char *st_ = msm + ((kt_) & (M16_NS - 1)) * STAGE;                                             \
        _Pragma("unroll") for (int cb_ = 0; cb_ < CB; ++cb_) FW[cb_] = *reinterpret_cast<const mf16x8 *>(st_ + foff_w + cb_ * 1024); \
        _Pragma("unroll") for (int a_ = 0; a_ < PB; ++a_) FP[a_] = *reinterpret_cast<const mf16x8 *>(st_ + foff_p + a_ * 1024); \
    }
#define M16_MFMA(FP, FW)                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    _Pragma("unroll") for (int a_ = 0; a_ < PB; ++a_)                                                       \
        _Pragma("unroll") for (int cb_ = 0; cb_ < CB; ++cb_)                                                \
            acc[a_][cb_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FW[cb_], FP[a_], acc[a_][cb_], 0, 0, 0);  \
    __builtin_amdgcn_s_setprio(0);

    // ---- prologue: tiles 0, 1, 2 go out; tile 0 must have landed
    dma();
    dma();
    dma();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AP + BP)) : "memory");
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    M16_READ(fp0, fw0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        // step kt (fragment set 0): tile kt + 1 has landed (tile kt + 2 may fly); every wave has read tile kt (and, a step ago, kt - 1)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AP + BP) : "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        dma();   // tile kt + 3 into the stage of tile kt - 1
        M16_READ(fp1, fw1, kt + 1);
        M16_MFMA(fp0, fw0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 >= nk) break;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AP + BP) : "memory");
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        dma();
        M16_READ(fp0, fw0, kt + 2);
        M16_MFMA(fp1, fw1);
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the trailing dummies
#undef M16_READ
#undef M16_MFMA

    // ---- epilogue straight from the accumulators: blocks (2 t, 2 t + 1) of pixel block a = channels 32 t + 8 kg + 0 .. 7 of pixel 16 a + l15
    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    const int nb0 = nt * BN + wn * WN;
    const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 7) & ~7);
#pragma unroll
    for (int a = 0; a < PB; ++a) {
        const int m = mt * BM + wm * WM + 16 * a + l15;
        _Float16 *orow = reinterpret_cast<_Float16 *>(p.out) + (size_t)m * p.ldc;
#pragma unroll
        for (int t = 0; t < CB / 2; ++t) {
            const int col = nb0 + 32 * t + 8 * kg;
            mf16x8 hv;
#pragma unroll
            for (int u = 0; u < 8; ++u) hv[u] = (_Float16)fmaxf(acc[a][2 * t + (u >> 2)][u & 3] * p.acc_scale + 0.f, lo);
            if (m < p.M && col < cend) *reinterpret_cast<mf16x8 *>(orow + col) = hv;
        }
    }
}

// ====================================================================== host side
bool conv_m16_supported(const ConvParams &p) {
    if (!p.in_f16 || !p.out_f16 || p.res || p.up || p.ksl > 1 || p.phases > 1 || p.cwrap || p.x3_plane || p.out_split || p.acc_shift ||
        p.rd_cout || p.scatter || p.rg_out || p.nx_wgt || p.pool || (p.act != ACT_NONE && p.act != ACT_RELU))
        return false;
    const int lda = p.lda ? p.lda : p.Cin, ldw = p.ldw ? p.ldw : p.Kpad;
    if (p.in2) {   // [t2 | x]: whole 32-channel steps on either side of the seam, every reduction column a real channel of its source
        return p.R == 1 && p.S == 1 && p.stride == 1 && !p.pad_h && !p.pad_w && p.K == p.Kpad && p.Kpad % 32 == 0 && p.ksplit > 0 &&
               p.ksplit < p.Kpad && p.ksplit % 32 == 0 && lda >= p.ksplit && p.lda2 >= p.Kpad - p.ksplit && !(lda & 7) && !(p.lda2 & 7) &&
               !(ldw & 7) && !(p.ldc & 7) && p.Cout % 64 == 0 && p.stride2 >= 1;
    }
    if ((lda & 7) || (ldw & 7) || (p.ldc & 7) || p.Cin % 32 || p.Cout % 64 || lda < p.Cin) return false;
    if ((long long)p.N * p.H * p.W * lda >= (1ll << 31)) return false;
    if (p.R == 3 && p.S == 3) return p.stride == 1 && p.pad_h == 1 && p.pad_w == 1 && p.Ho == p.H && p.Wo == p.W && p.Kpad >= 9 * p.Cin;
    return p.R == 1 && p.S == 1 && p.stride == 1 && !p.pad_h && !p.pad_w && p.K == p.Cin && p.Kpad == p.Cin;
}

// 1x1 layers on the 16x16x32 MFMA: the MFMA-heavy squeezing / head convs without a residual (layer3's conv1 1024 -> 256 and 512 -> 256,
// pose_net.0 1024 -> 512: resnet.py:124-131, handmvnet.py:70-86) -- long reductions (K >= 128, whole 64-channel k-steps), wide outputs.
// Everything in the rule is a property of the layer (shape, epilogue, storage), nothing of the launch size.
static int g_m16_rule = 1;
void conv_m16_set_rule(int on) { g_m16_rule = on; }
bool conv_m16_rule(const ConvParams &p) {
    // (two-source launches: only the long one, layer3.0's conv3 + downsample, K = 256 + 512; the K = 128 / 384 ones of layer1.0 / layer2.0
    // are HBM-bound and belong to conv_stream.hip at every size that matters)
    return g_m16_rule && conv_m16_supported(p) && p.R == 1 && p.Kpad % 64 == 0 && p.Kpad >= (p.in2 ? 768 : 128) && p.Cout > 128 && !p.tall && !p.fill;
}

template <int BM, int BN, bool TAPS, bool DUAL = false>
static hipError_t launch_m16(ConvParams p, hipStream_t s) {
    constexpr int lds = M16_NS * (BM + BN) * 64;
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_m16_f16<BM, BN, TAPS, DUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = p.Cout / BN;
    hipLaunchKernelGGL((conv_m16_f16<BM, BN, TAPS, DUAL>), dim3(p.mtiles * p.ntiles), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_m16(ConvParams p, hipStream_t s, const char **name) {
    if (!conv_m16_supported(p)) return hipErrorInvalidValue;
    if (!p.lda) p.lda = p.Cin;
    if (!p.ldw) p.ldw = p.Kpad;
    const bool taps = p.R == 3;
    // 64 x 64 tiles while 128 x 128 ones would leave CUs idle (the rule of conv_igemm's small-launch tiles)
    const bool small = p.Cout % 128 != 0 || (long long)((p.M + 127) / 128) * (p.Cout / 128) < 256;
    if (p.in2) {
        if (name) *name = small ? "conv_m16_f16<64x64,1x1,dual>" : "conv_m16_f16<128x128,1x1,dual>";
        return small ? launch_m16<64, 64, false, true>(p, s) : launch_m16<128, 128, false, true>(p, s);
    }
    if (small) {
        if (name) *name = taps ? "conv_m16_f16<64x64,taps,c32>" : "conv_m16_f16<64x64,1x1>";
        return taps ? launch_m16<64, 64, true>(p, s) : launch_m16<64, 64, false>(p, s);
    }
    if (name) *name = taps ? "conv_m16_f16<128x128,taps,c32>" : "conv_m16_f16<128x128,1x1>";
    return taps ? launch_m16<128, 128, true>(p, s) : launch_m16<128, 128, false>(p, s);
}

}  // namespace hmv
