// conv_ht.hip -- 3x3 convolution on TALL tiles (512 pixels x 128 channels) for the fp16 path's MFMA-heavy 3x3 layers.
//
// Replaces, where the feature map tiles into 16 x 32 pixel blocks, conv_igemm's 256 x 256 halo tiles for the conv2 of layer2 /
// layer3's Bottlenecks (/root/reference/src/models/backbones/resnet.py:114-118, 132-134: 3x3, stride 1, padding 1, 128 -> 128 and
// 256 -> 256 channels at H/8).  Those launches are bound by the bytes a tile moves into the CU (their energy: DESIGN.md section 8), and four fifths of
// them are WEIGHTS, re-streamed once per tile: 9 taps x 4 chunks x 32 KB = 1 152 of 1 440 KB.  The weight
// bytes of a layer are (M / BM) K N 2 whatever BN is, while the halo image makes the pixel side cheap (1.27 C bytes per pixel
// and chunk instead of 9 C), so the tile that moves the fewest bytes per output at 65 536 accumulators is tall: 512 x 128 moves
// 576 (weights) + 314 (halos) + 128 (output) = 1 018 KB for the same 65 536 outputs, -29 %.
//
//   * tile = one 16 x 32 block of one image x 128 output channels; 8 waves as 4 (pixel rows 4 wm .. 4 wm + 3) x 2 (64 channels):
//     a wave's 32-pixel MFMA block is ONE image row of the block, its lane the column.
//   * the reduction runs in 32-channel sub-chunks: per sub-chunk the 18 x 34 halo of the block is fetched once (39 KB, two images
//     in LDS), the nine taps read it at shifted pixels; a k-step = one tap of one sub-chunk = 2 k16 MFMA steps, its weights
//     [128][32] = 8 KB = ONE 16-byte DMA per thread into a ring of eight stages (seven steps ahead).
//   * main loop: conv_gemm8's structure with one phase per k-step {fragment reads . DMA . counted `s_waitcnt vmcnt(N)` . barrier .
//     16 MFMAs . barrier}, the two wave halves one barrier apart, raw barriers, nothing drained inside the loop.  N is static per
//     tap: the instructions issued after the weight stage that must have landed (one weight DMA per step, one halo DMA on taps 0-4).
//   * K order (32-channel sub-chunk, tap, channel) differs from conv_igemm's (64-channel chunk, tap, channel): same products, another
//     summation order.  So this kernel is chosen by the CONFIGURATION (layer shape and map size), never by the batch: a layer it
//     takes, it takes at every batch size, and a sample's bits stay independent of its batch.  The engine packs those layers'
//     weights in this order (Loader::conv, `tall`).
// Operand roles, bias-as-initial-accumulator and the register epilogue are conv_igemm's transposed-output path.
//
// Round 4: the kernel is a template on the MFMA SHAPE (MI355X_MICROARCH.md "DVFS give-back" item 7: where the chip
// lowers its clock under load, the clock it holds can depend on the shape).  M16 = v_mfma_f32_16x16x32_f16: the wave's 128 pixels x 64
// channels are 8 x 4 blocks of 16 x 16, a k-step (one tap of one 32-channel sub-chunk) is ONE MFMA per block (32 MFMAs of 16 cycles
// instead of 16 of 32), a lane's 16-byte fragment is k-group lane >> 4 of pixel / weight row lane & 15.  Same DMA schedule, same
// LDS bytes read per step (12 ds_read_b128), same accumulator count; the LDS images are swizzled with another key (below) because
// a ds_read_b128's 16-lane service groups then mix two k-groups.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace hmv {

typedef float tf32x16 __attribute__((ext_vector_type(16)));
typedef float tf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 tf16x8 __attribute__((ext_vector_type(8)));

#define HMV_TGLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

constexpr int HT_HW = 34, HT_HROWS = 18 * 34;      // halo of a 16 x 32 block
constexpr int HT_HP = 5;                           // halo DMA instructions per thread and image (612 pixels x 4 chunks / 512)
constexpr int HT_IMG = HT_HP * 8192;               // bytes of a halo image slot
constexpr int HT_WST = 8192;                       // bytes of a weight stage [128][32] halfs
constexpr int HT_NWS = 8;                          // weight stages in LDS (a power of two); a stage is issued HT_NWS - 1 steps ahead
constexpr int HT_LEAD = HT_NWS - 1;
constexpr int HT_LDS = 2 * HT_IMG + HT_NWS * HT_WST;   // 147 456

// DMAs issued after the weight stage of step s + 1 (issued first thing in step s + 1 - HT_LEAD) when step s waits for it: that
// step's halo DMA (taps 0-4 carry one) and the weight and halo DMAs of the HT_LEAD - 1 steps since
constexpr int ht_h(int tap) { return ((tap % 9) + 9) % 9 < HT_HP ? 1 : 0; }
constexpr int ht_after_w(int tap) {
    int n = HT_LEAD - 1;
    for (int i = 0; i < HT_LEAD; ++i) n += ht_h(tap - i);
    // the last step of a sub-chunk also needs the next halo image, whose last DMA went out on tap HT_HP - 1: 8 - (HT_HP - 1) weight
    // DMAs have followed it
    if (tap == 8 && n > 8 - (HT_HP - 1)) n = 8 - (HT_HP - 1);
    return n;
}
static_assert(ht_after_w(2) < 60, "vmcnt is a 6-bit field");
template <int TAP>
__device__ __forceinline__ void ht_wait() {   // ... and this wave's fragment reads have left LDS (the slots they read are refilled next)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ht_after_w(TAP)) : "memory");
}

// LDS swizzle keys (XORed into the 16-byte chunk index of a 64-byte row).  32x32x16: (index >> 2) & 3 -- a read's 16 lanes are 16
// consecutive pixels / rows of ONE k-half.  16x16x32: ((index >> 2) & 1) << 1 -- a read's 16-lane service group {0-3, 12-15, 20-27} is
// pixels 0-3 and 12-15 of k-group g with pixels 4-11 of k-group g ^ 1; this key (found by exhaustive search over keys of period 8,
// tools/probe/swizzle_search.py) keeps all 16 on distinct bank quads for the tap shifts 0, 1, 2 and for the weight rows' 8-row pitch.
// Weight rows: a 16x16 block's fragment rows are 4-row groups 8 rows apart (the epilogue wants a lane's two blocks to hold 8
// consecutive channels), so their key is ((row >> 3) & 1) << 1.
template <bool M16> __device__ __forceinline__ int ht_pkey(int hx) { return M16 ? (((hx >> 2) & 1) << 1) : ((hx >> 2) & 3); }
template <bool M16> __device__ __forceinline__ int ht_wkey(int row) { return M16 ? (((row >> 3) & 1) << 1) : ((row >> 2) & 3); }

template <bool M16>
__global__ __launch_bounds__(512) void conv_ht_f16(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char tsm[];   // [2 halo images][4 weight stages]
    char *wst = tsm + 2 * HT_IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    unsigned long long t_entry = 0, t0c = 0, t0r = 0, t1c = 0, t1r = 0;   // diagnostic stamps (hmv_bench_conv with HMV_BENCH_CLOCK)
    if (p.dbg) t_entry = __builtin_amdgcn_s_memrealtime();

    // ---- tile: the N-tiles of a pixel block are consecutive workgroups of one XCD (bijective XCD map as conv_igemm)
    int blk, nt;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r = nblk & 7;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        blk = lid / p.ntiles;
        nt = lid - blk * p.ntiles;
    }
    const int tyn = p.H >> 4, txn = p.W >> 5, per_img = tyn * txn;
    const int n = blk / per_img, brem = blk - n * per_img, by = brem / txn, bx = brem - by * txn;
    const _Float16 *zero16 = reinterpret_cast<const _Float16 *>(p.zero);
    const int nchunk = p.Cin >> 5, nstep = 9 * nchunk;

    // ---- DMA roles.  Halo: pass i moves 16-byte unit L = 512 i + tid = chunk L & 3 of halo pixel L >> 2 (the chunk it fetches is
    // XOR-swizzled with (halo column >> 2) & 3: 64-byte rows, and a fragment read's 16 lanes are 16 consecutive columns of one row);
    // out-of-image pixels and units past the halo come from the zero page.
    const _Float16 *hsrc[HT_HP];
#pragma unroll
    for (int i = 0; i < HT_HP; ++i) {
        const int L = 512 * i + tid, hp = L >> 2, hy = hp / HT_HW, hx = hp - hy * HT_HW;
        const int iy = by * 16 - 1 + hy, ix = bx * 32 - 1 + hx;
        const bool ok = hp < HT_HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        hsrc[i] = ok ? reinterpret_cast<const _Float16 *>(p.in) + ((size_t)(n * p.H + iy) * p.W + ix) * p.lda + 8 * ((L & 3) ^ ht_pkey<M16>(hx)) : nullptr;
    }
    // weights: thread -> row tid >> 2 of the 128-row stage, physical chunk tid & 3 holding logical chunk (tid & 3) ^ ((row >> 2) & 3)
    const _Float16 *wsrc = reinterpret_cast<const _Float16 *>(p.wgt) + (size_t)(nt * 128 + (tid >> 2)) * p.ldw + 8 * ((tid & 3) ^ ht_wkey<M16>(tid >> 2));

    auto issue_w = [&](int s) {   // the weight stage of step s (past the end: a dummy from the zero page, nobody reads it)
        const _Float16 *src = s < nstep ? wsrc + 32 * s : zero16;
        asm volatile("" : "+v"(src));
        HMV_TGLDS16(src, wst + (s & (HT_NWS - 1)) * HT_WST + wave * 1024);
    };
    auto issue_h = [&](int c, int i) {   // DMA i of the halo image of sub-chunk c (past the last: a dummy)
        const _Float16 *src = (c < nchunk && hsrc[i]) ? hsrc[i] + 32 * c : zero16;
        asm volatile("" : "+v"(src));
        HMV_TGLDS16(src, tsm + (c & 1) * HT_IMG + i * 8192 + wave * 1024);
    };

    // ---- accumulators start at the bias.  32x32x16: acc[a][b] = pixel row a x 32-channel block b, register 8 j + u = channel
    // 32 b + 16 j + 8 kh + u.  16x16x32: acc4[a][h][cb] = pixel row a, column half h x 16-row block cb = 2 t + e, register u of lane
    // group g = lane >> 4 = channel 32 t + 8 g + 4 e + u (so blocks 2 t, 2 t + 1 give a lane 8 consecutive channels)
    tf32x16 acc[M16 ? 1 : 4][M16 ? 1 : 2];
    tf32x4 acc4[M16 ? 4 : 1][M16 ? 2 : 1][M16 ? 4 : 1];
    const int l15 = lane & 15, kg = lane >> 4;
    {
        const float binit = 1.f / p.acc_scale;
        if constexpr (!M16) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float *bp = p.bias + nt * 128 + wn * 64 + 32 * b;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const tf32x4 bq = *reinterpret_cast<const tf32x4 *>(bp + 16 * (q >> 1) + 8 * kh + 4 * (q & 1));
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[a][b][4 * q + u] = bq[u] * binit;
                }
            }
        } else {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const tf32x4 bq = *reinterpret_cast<const tf32x4 *>(p.bias + nt * 128 + wn * 64 + 32 * (cb >> 1) + 8 * kg + 4 * (cb & 1));
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc4[a][h][cb][u] = bq[u] * binit;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!M16) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(acc[a][b]));   // the bias is in the accumulators BEFORE the first DMA is issued
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) asm volatile("" : "+v"(acc4[a][h][cb]));
    }

    // fragment addresses.  32x32x16: pixel block a = image row 4 wm + a of the block, lane = column; weight block b = rows
    // wn 64 + 32 b + swap23(l31).  16x16x32: pixel block (a, h) = columns 16 h + l15 of that row, k-group kg; weight block cb = 2 t + e =
    // rows wn 64 + 32 t + 8 (l15 >> 2) + 4 e + (l15 & 3), k-group kg
    const int wl31 = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int wrow = M16 ? wn * 64 + 8 * (l15 >> 2) + (l15 & 3) : wn * 64 + wl31;
    const int wsw = ht_wkey<M16>(wrow);   // (+ 32 t, + 4 e leave the key unchanged)
    // halo pixel of (block row a, column c) at tap (dy, dx): (4 wm + a + dy) * 34 + c + dx; its swizzle depends on c + dx only,
    // so the 3 x 2 byte offsets below plus compile-time displacements address every pixel fragment
    int poff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (!M16) poff[dx][j] = ((4 * wm) * HT_HW + l31) * 64 + ((2 * j + kh) ^ ht_pkey<false>(l31 + dx)) * 16;
            else poff[dx][j] = ((4 * wm) * HT_HW + 16 * j + l15) * 64 + (kg ^ ht_pkey<true>(16 * j + l15 + dx)) * 16;   // j = column half
        }
    tf16x8 fp[4][2], fw[2][2];   // 32x32x16: [a][k-half], [b][k-half]; 16x16x32: [a][column half], [t][e]
#define HT_READ(s_, tap_)                                                                                   \
    {                                                                                                       \
        const char *img_ = tsm + (((s_) / 9) & 1) * HT_IMG;                                                 \
        const char *ws_ = wst + ((s_) & (HT_NWS - 1)) * HT_WST;                                             \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                              \
                if constexpr (!M16) fw[b_][j_] = *reinterpret_cast<const tf16x8 *>(ws_ + ((wrow + 32 * b_) * 4 + ((2 * j_ + kh) ^ wsw)) * 16); \
                else fw[b_][j_] = *reinterpret_cast<const tf16x8 *>(ws_ + ((wrow + 32 * b_ + 4 * j_) * 4 + (kg ^ wsw)) * 16); \
            }                                                                                               \
        _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_)                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                \
                fp[a_][j_] = *reinterpret_cast<const tf16x8 *>(img_ + poff[(tap_) % 3][j_] + ((a_ + (tap_) / 3) * HT_HW + (tap_) % 3) * 64); \
    }
#define HT_MFMA()                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    if constexpr (!M16) {                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                    \
            _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_)                                                \
                _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                            \
                    acc[a_][b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[b_][j_], fp[a_][j_], acc[a_][b_], 0, 0, 0); \
    } else {                                                                                                \
        _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_)                                                    \
            _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                \
                _Pragma("unroll") for (int cb_ = 0; cb_ < 4; ++cb_)                                         \
                    acc4[a_][h_][cb_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[cb_ >> 1][cb_ & 1], fp[a_][h_], acc4[a_][h_][cb_], 0, 0, 0); \
    }                                                                                                       \
    __builtin_amdgcn_s_setprio(0);
#define HT_BAR()                                                                                            \
    asm volatile("s_barrier" ::: "memory");                                                                 \
    __builtin_amdgcn_sched_barrier(0)

    // ---- prologue: the halo of sub-chunk 0, then HT_LEAD "steps before the first" that issue what those steps would have issued --
    // the weight stages of steps 0 .. HT_LEAD - 1 and, on the taps that carry one, a dummy halo DMA (zero page into image 1, which
    // the real halo of sub-chunk 1 overwrites later: same wave, same addresses, in order) -- so the loop's static counts hold from step 0
#pragma unroll
    for (int i = 0; i < HT_HP; ++i) issue_h(0, i);
#pragma unroll
    for (int f = -HT_LEAD; f < 0; ++f) {
        issue_w(f + HT_LEAD);
        if (ht_h(f)) {
            const _Float16 *src = zero16;
            asm volatile("" : "+v"(src));
            HMV_TGLDS16(src, tsm + HT_IMG + wave * 1024);
        }
    }
    ht_wait<8>();   // as step -1 would: stage 0 (and the halo before it) has landed
    HT_BAR();
    if (p.dbg) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
    if (wm >= 2) { HT_BAR(); }   // the second wave half runs one barrier behind the first from here on

    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int s = 9 * c + tap;
            HT_READ(s, tap);
            issue_w(s + HT_LEAD);
            if (tap < HT_HP) issue_h(c + 1, tap);
            // the weight stage of step s + 1 (read in the next phase) has landed; with it every older halo DMA
            switch (tap) {
                case 0: ht_wait<0>(); break; case 1: ht_wait<1>(); break; case 2: ht_wait<2>(); break;
                case 3: ht_wait<3>(); break; case 4: ht_wait<4>(); break; case 5: ht_wait<5>(); break;
                case 6: ht_wait<6>(); break; case 7: ht_wait<7>(); break; default: ht_wait<8>(); break;
            }
            HT_BAR();
            HT_MFMA();
            __builtin_amdgcn_sched_barrier(0);
            HT_BAR();
        }
    }
    if (wm < 2) { HT_BAR(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.dbg) { t1c = __builtin_amdgcn_s_memtime(); t1r = __builtin_amdgcn_s_memrealtime(); }
#undef HT_READ
#undef HT_MFMA
#undef HT_BAR

    // ---- epilogue straight from the accumulators.  32x32x16: register 8j + u of block (a, b) = channel 32 b + 16 j + 8 kh + u of pixel
    // (row a, column l31); 16x16x32: registers of blocks (2 t, 2 t + 1) of (a, h) = channels 32 t + 8 kg + 0 .. 7 of pixel (row a, column 16 h + l15)
    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    const int nb0 = nt * 128 + wn * 64;
    const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 7) & ~7);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if constexpr (!M16) {
            const size_t pix = (size_t)(n * p.H + by * 16 + 4 * wm + a) * p.W + bx * 32 + l31;
            _Float16 *orow = reinterpret_cast<_Float16 *>(p.out) + pix * p.ldc;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = nb0 + 32 * b + 16 * j + 8 * kh;
                    tf16x8 hv;
#pragma unroll
                    for (int u = 0; u < 8; ++u) hv[u] = (_Float16)fmaxf(acc[a][b][8 * j + u] * p.acc_scale + 0.f, lo);
                    if (col < cend) *reinterpret_cast<tf16x8 *>(orow + col) = hv;
                }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const size_t pix = (size_t)(n * p.H + by * 16 + 4 * wm + a) * p.W + bx * 32 + 16 * h + l15;
                _Float16 *orow = reinterpret_cast<_Float16 *>(p.out) + pix * p.ldc;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int col = nb0 + 32 * t + 8 * kg;
                    tf16x8 hv;
#pragma unroll
                    for (int u = 0; u < 8; ++u) hv[u] = (_Float16)fmaxf(acc4[a][h][2 * t + (u >> 2)][u & 3] * p.acc_scale + 0.f, lo);
                    if (col < cend) *reinterpret_cast<tf16x8 *>(orow + col) = hv;
                }
            }
        }
    }
    if (p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *d = p.dbg + 8 * (size_t)blockIdx.x;
        d[0] = t1c - t0c; d[1] = t1r - t0r; d[2] = t_entry; d[3] = t0r; d[4] = t1r; d[5] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_htp_f16: conv_ht_f16<M16> as a PERSISTENT kernel (round 4; VERDICT r3 lead: "request the next tile's first operands before the
// epilogue").  One workgroup per CU (grid = 256) walks its tiles; the weight stages and the first halo image that a tile's last steps
// would have fetched as dummies are the NEXT tile's -- its seven leading weight stages and chunk-0 halo are in flight while this tile's
// last taps multiply and its epilogue stores -- so a tile starts at step 0 with the same static vmcnt counts and no prologue, and a CU
// never idles between a workgroup's exit and the next one's first DMA round trip (prologue + epilogue were 7 of 61 us per tile).
// Placement: XCD x owns the pixel blocks [x B, (x + 1) B), B = ceil(blocks / 8); its 32 workgroups are 32 / ntiles block slots x ntiles
// channel tiles (a workgroup keeps its channel tile: weights pointer and bias registers are launch constants); the N-tiles of a block run
// side by side on one XCD as before.  Same MFMA sequence per output as conv_ht_f16<true>: bit-identical (test_tall_tile_kernel).
__global__ __launch_bounds__(512) void conv_htp_f16(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char tsm[];
    char *wst = tsm + 2 * HT_IMG;
    float *sbias = reinterpret_cast<float *>(tsm + HT_LDS);   // this workgroup's 128 initial accumulator values (bias / acc_scale)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;          // (blocks are dealt round-robin over the XCDs: speed only)
    const int S = 32 / p.ntiles, nt = slot % p.ntiles, sidx = slot / p.ntiles;
    const int Bx = (p.mtiles + 7) >> 3, blk_end = min((xcd + 1) * Bx, p.mtiles);
    int blk = xcd * Bx + sidx;
    if (blk >= blk_end) return;
    const int tyn = p.H >> 4, txn = p.W >> 5, per_img = tyn * txn;
    const _Float16 *zero16 = reinterpret_cast<const _Float16 *>(p.zero);
    const int nchunk = p.Cin >> 5, nstep = 9 * nchunk;

    // halo source pointer of DMA i of pixel block (n, by, bx) (nullptr: out of the image / past the halo -> zero page)
    auto halo_ptr = [&](int n, int by, int bx, int i) -> const _Float16 * {
        int t = tid;
        asm volatile("" : "+v"(t));   // recomputed where it is used: nothing of it lives across the steps
        const int L = 512 * i + t, hp = L >> 2, hy = hp / HT_HW, hx = hp - hy * HT_HW;
        const int iy = by * 16 - 1 + hy, ix = bx * 32 - 1 + hx;
        const bool ok = hp < HT_HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        return ok ? reinterpret_cast<const _Float16 *>(p.in) + ((size_t)(n * p.H + iy) * p.W + ix) * p.lda + 8 * ((L & 3) ^ ht_pkey<true>(hx)) : nullptr;
    };
    int bn = blk / per_img, by = (blk - bn * per_img) / txn, bx = blk - bn * per_img - by * txn;   // this tile's block (uniform)
    const _Float16 *hcur[HT_HP];
#pragma unroll
    for (int i = 0; i < HT_HP; ++i) hcur[i] = halo_ptr(bn, by, bx, i);
    const _Float16 *wsrc = reinterpret_cast<const _Float16 *>(p.wgt) + (size_t)(nt * 128 + (tid >> 2)) * p.ldw + 8 * ((tid & 3) ^ ht_wkey<true>(tid >> 2));

    if (tid < 128) sbias[tid] = p.bias[nt * 128 + tid] * (1.f / p.acc_scale);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the bias loads are not counted by the schedule below)

    const int wrow = wn * 64 + 8 * (l15 >> 2) + (l15 & 3), wsw = ht_wkey<true>(wrow);
    int poff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int j = 0; j < 2; ++j) poff[dx][j] = ((4 * wm) * HT_HW + 16 * j + l15) * 64 + (kg ^ ht_pkey<true>(16 * j + l15 + dx)) * 16;

    int gs = 0, gc = 0;   // running step / chunk counters: LDS slots are (gs & 7) and (gc & 1) across tiles
    auto issue_wsrc = [&](const _Float16 *src, int g) {
        asm volatile("" : "+v"(src));
        HMV_TGLDS16(src, wst + (g & (HT_NWS - 1)) * HT_WST + wave * 1024);
    };
    auto issue_hsrc = [&](const _Float16 *src, int c, int i) {
        asm volatile("" : "+v"(src));
        HMV_TGLDS16(src, tsm + (c & 1) * HT_IMG + i * 8192 + wave * 1024);
    };
    // ---- prologue of the FIRST tile (as conv_ht_f16): its chunk-0 halo, then the steps "before the first"
#pragma unroll
    for (int i = 0; i < HT_HP; ++i) issue_hsrc(hcur[i] ? hcur[i] : zero16, 0, i);
#pragma unroll
    for (int f = -HT_LEAD; f < 0; ++f) {
        issue_wsrc(wsrc + 32 * (f + HT_LEAD), f + HT_LEAD);
        if (ht_h(f)) issue_hsrc(zero16, 1, 0);
    }
    ht_wait<8>();
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (wm >= 2) { asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }

    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    const int nb0 = nt * 128 + wn * 64;
    const int cend = (p.fill || p.Cout + 3 >= p.ldc) ? p.ldc : ((p.Cout + 7) & ~7);
    tf32x4 acc4[4][2][4];
    tf16x8 fp[4][2], fw[2][2];
    for (;;) {
        const int nblk_ = blk + S;
        const bool have_next = nblk_ < blk_end;
        const int nn = nblk_ / per_img, nby = (nblk_ - nn * per_img) / txn, nbx = nblk_ - nn * per_img - nby * txn;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {   // channel 32 t + 8 kg + 4 e + u, cb = 2 t + e (as conv_ht_f16<true>)
            const tf32x4 bq = *reinterpret_cast<const tf32x4 *>(sbias + wn * 64 + 32 * (cb >> 1) + 8 * kg + 4 * (cb & 1));
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc4[a][h][cb] = bq;
        }
        for (int c = 0; c < nchunk; ++c) {
            const bool lastc = c == nchunk - 1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int s = 9 * c + tap;
                {   // fragment reads of this step
                    const char *img_ = tsm + (gc & 1) * HT_IMG;
                    const char *ws_ = wst + (gs & (HT_NWS - 1)) * HT_WST;
#pragma unroll
                    for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
                        for (int j_ = 0; j_ < 2; ++j_)
                            fw[b_][j_] = *reinterpret_cast<const tf16x8 *>(ws_ + ((wrow + 32 * b_ + 4 * j_) * 4 + (kg ^ wsw)) * 16);
#pragma unroll
                    for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
                        for (int j_ = 0; j_ < 2; ++j_)
                            fp[a_][j_] = *reinterpret_cast<const tf16x8 *>(img_ + poff[tap % 3][j_] + ((a_ + tap / 3) * HT_HW + tap % 3) * 64);
                }
                {   // the weight stage HT_LEAD steps ahead: this tile's, else the next tile's, else a dummy
                    const int idx = s + HT_LEAD;
                    const _Float16 *src = idx < nstep ? wsrc + 32 * idx : (have_next ? wsrc + 32 * (idx - nstep) : zero16);
                    issue_wsrc(src, gs + HT_LEAD);
                }
                if (tap < HT_HP) {   // the next halo image: this tile's next sub-chunk, else the next tile's sub-chunk 0, else a dummy
                    const _Float16 *src;
                    if (!lastc) src = hcur[tap] ? hcur[tap] + 32 * (c + 1) : zero16;
                    else {
                        if (have_next) hcur[tap] = halo_ptr(nn, nby, nbx, tap);   // (this tile is done with pointer `tap`)
                        src = (have_next && hcur[tap]) ? hcur[tap] : zero16;
                    }
                    issue_hsrc(src, gc + 1, tap);
                }
                switch (tap) {
                    case 0: ht_wait<0>(); break; case 1: ht_wait<1>(); break; case 2: ht_wait<2>(); break;
                    case 3: ht_wait<3>(); break; case 4: ht_wait<4>(); break; case 5: ht_wait<5>(); break;
                    case 6: ht_wait<6>(); break; case 7: ht_wait<7>(); break; default: ht_wait<8>(); break;
                }
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
                    for (int h_ = 0; h_ < 2; ++h_)
#pragma unroll
                        for (int cb_ = 0; cb_ < 4; ++cb_)
                            acc4[a_][h_][cb_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[cb_ >> 1][cb_ & 1], fp[a_][h_], acc4[a_][h_][cb_], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                ++gs;
            }
            ++gc;
        }
        // ---- epilogue of this tile (its stores are younger than the next tile's operands already in flight: the counted waits of the
        // next steps then also retire the oldest of them -- conservative, never wrong)
        {
            const int n = bn;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const size_t pix = (size_t)(n * p.H + by * 16 + 4 * wm + a) * p.W + bx * 32 + 16 * h + l15;
                    _Float16 *orow = reinterpret_cast<_Float16 *>(p.out) + pix * p.ldc;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int col = nb0 + 32 * t + 8 * kg;
                        tf16x8 hv;
#pragma unroll
                        for (int u = 0; u < 8; ++u) hv[u] = (_Float16)fmaxf(acc4[a][h][2 * t + (u >> 2)][u & 3] * p.acc_scale + 0.f, lo);
                        if (col < cend) *reinterpret_cast<tf16x8 *>(orow + col) = hv;
                    }
                }
        }
        if (!have_next) break;
        blk = nblk_; bn = nn; by = nby; bx = nbx;
    }
    if (wm < 2) { asm volatile("s_barrier" ::: "memory"); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ====================================================================== host side
static int g_ht_mode = -1;   // -1: launch_conv's rule (enough tiles to fill the chip); 0 never, 1 always (op-level tests)
void conv_ht_set_mode(int mode) { g_ht_mode = mode; }
int conv_ht_mode() { return g_ht_mode; }
// MFMA shape of the tall-tile layers: 1 = 16x16x32 (the engine's: 12-15 % less time, profiles/r04_probe_mfma_shape.txt), 0 = 32x32x16 (the
// A/B partner, hmv_op_conv2d_f16 kernel_sel 5 / 6).  A property of the BUILD, not of a launch: every batch size runs the same shape.
static int g_ht_m16 = 1;
void conv_ht_set_shape(int m16) { g_ht_m16 = m16; }
static int g_ht_persist = 1;   // 1: the persistent form from two tiles per CU up; 0: never; 2: wherever it exists (op-level identity tests)
void conv_ht_set_persistent(int on) { g_ht_persist = on; }
int conv_ht_shape() { return g_ht_m16; }

// shape rule (the engine asks it at weight-packing time and at launch: the same answer for every batch)
bool conv_ht_shape_ok(int R, int S, int stride, int pad, int Cin, int Cout, int H, int W) {
    static int off = -1;   // development knob: HMV_NO_HT=1 keeps these layers on conv_igemm's 256 x 256 halo tiles (A/B runs)
    if (off < 0) off = HMV_DEV_ENV("HMV_NO_HT") ? 1 : 0;
    return !off && R == 3 && S == 3 && stride == 1 && pad == 1 && Cin % 32 == 0 && Cin >= 64 && Cout % 128 == 0 && H % 16 == 0 && W % 32 == 0 &&
           H > 0 && W > 0;
}

hipError_t launch_conv_ht(ConvParams p, hipStream_t s, const char **name) {
    if (!p.in_f16 || !p.out_f16 || p.res || !conv_ht_shape_ok(p.R, p.S, p.stride, p.pad_h, p.Cin, p.Cout, p.H, p.W) || p.pad_w != 1 || p.Ho != p.H ||
        p.Wo != p.W || p.up || p.in2 || p.ksl > 1 || p.phases > 1 || p.cwrap || p.x3_plane || p.out_split || p.acc_shift || p.rd_cout ||
        p.scatter || p.rg_out || (p.act != ACT_NONE && p.act != ACT_RELU) || (p.lda & 7) || p.lda < p.Cin || (p.ldw & 7) || (p.ldc & 7) || p.Kpad < 9 * p.Cin)
        return hipErrorInvalidValue;
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_ht_f16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, HT_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_ht_f16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, HT_LDS);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    p.mtiles = p.N * (p.H >> 4) * (p.W >> 5);
    p.ntiles = p.Cout / 128;
    // the persistent form (16x16x32 only: the engine's shape): from two tiles per CU up, channel tiles that divide an XCD's 32 workgroups
    if (g_ht_m16 && g_ht_persist && ((long long)p.mtiles * p.ntiles >= 512 || g_ht_persist == 2) && (p.ntiles == 1 || p.ntiles == 2 || p.ntiles == 4)) {
        static bool pconf[64] = {};
        if (!pconf[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_htp_f16), hipFuncAttributeMaxDynamicSharedMemorySize, HT_LDS + 512);
            if (e != hipSuccess) return e;
            pconf[dev] = true;
        }
        if (name) *name = "conv_ht_f16<512x128,3x3,m16,persistent>";
        hipLaunchKernelGGL(conv_htp_f16, dim3(256), dim3(512), HT_LDS + 512, s, p);
        return hipGetLastError();
    }
    if (g_ht_m16) {
        if (name) *name = "conv_ht_f16<512x128,3x3,m16>";
        hipLaunchKernelGGL(conv_ht_f16<true>, dim3(p.mtiles * p.ntiles), dim3(512), HT_LDS, s, p);
        return hipGetLastError();
    }
    if (name) *name = "conv_ht_f16<512x128,3x3>";
    hipLaunchKernelGGL(conv_ht_f16<false>, dim3(p.mtiles * p.ntiles), dim3(512), HT_LDS, s, p);
    return hipGetLastError();
}

}  // namespace hmv
