// conv_hs.hip -- persistent, weight-stationary R x S convolution for few-channel layers of the fp16 path ("halo stream").
//
// Replaces, for large pixel counts, the stride-1 convolutions whose whole weight tensor fits one workgroup's registers:
//   * the 3x3 conv2 of layer1's Bottlenecks, 64 -> 64 channels at H/4 (/root/reference/src/models/backbones/resnet.py:114-118,
//     132-134) -- and with them every 64-channel 3x3 of the BasicBlock ResNets (resnet.py:77-106) on the fp16 path;
//   * the stem conv1 7x7 s2 (resnet.py:218), which the engine runs as a 4x4 stride-1 convolution over 2x2 space-to-depth
//     frames of 16 channels.
// conv_igemm gives these layers 128 x 64 tiles: per tile it re-reads the shifted pixel tile once per tap (9 x 16 KB) and
// re-streams the weights (72 KB) for 16 KB of input and 16 KB of output -- 0.24 of the HBM roofline on algorithmic bytes.
//
// MI355X mapping (conv_stream.hip's structure with a halo image in place of a GEMM row tile)
//   * ONE workgroup per CU for the whole launch, walking 16 x 16 output blocks; the WEIGHTS LIVE IN REGISTERS (a wave keeps
//     the MFMA A-operand fragments of its 32 * TN output channels for all R * S taps: 144 VGPRs for 3x3 x 64 channels);
//   * per block the (16 + R - 1) x (16 + S - 1) halo of input pixels arrives ONCE by LDS-DMA (41 KB for 3x3 x 64 channels,
//     XOR-swizzled on the source side) into a ring of NSLOT images; the R * S taps read it at shifted rows, so a pixel is
//     fetched 1.27x (halo) instead of 9x;
//   * one static schedule per block, identical for every wave: wait(halo landed) . barrier . DMA halo of block t + D .
//     [DMA residual of block t + 1] . ceil(R * S * CPP / 2) k16 MFMA steps . wait(residual) . epilogue; every wait is a counted
//     `s_waitcnt vmcnt(N)` (sched_* in conv_stream.hip's sense); blocks past the stream's end use the zero / trash pages;
//   * operand roles, K order (tap-major, channels inside), bias-as-initial-accumulator and the epilogue arithmetic are
//     conv_igemm's, so the results are BIT-IDENTICAL to conv_igemm's (tests/test_gpu_parity.py::test_stream_kernel_is_bit_identical).
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "kernels.h"

namespace hmv {

typedef float hf32x16 __attribute__((ext_vector_type(16)));
typedef float hf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));

#define HMV_HGLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int N>
__device__ __forceinline__ void hs_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename F, int... I>
__device__ __forceinline__ void hs_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void hs_static_for(F &&f) {
    hs_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// A block's vector-memory instructions, in issue order: HP halo DMAs (block t + D), RB residual DMAs (block t + 1), OS stores.
// "X landed" == "at most the instructions issued after X are still out".
constexpr int hs_after_halo(int D, int HP, int RB, int OS) { return RB + OS + (D - 1) * (HP + RB + OS); }   // waited at the top of block t
constexpr int hs_after_residual(int HP, int RB, int OS) { return OS + HP + RB; }                              // waited before the epilogue of t

// R x S taps over pixels of CPP 16-byte chunks (8 channels each): K is the plain (r, s, c) order, so chunk g = (tap g / CPP, channels
// 8 (g % CPP) ..) and k16 step Q multiplies chunks 2Q (lanes 0-31) and 2Q + 1 (lanes 32-63) -- which may belong to two taps when
// CPP is odd (HRNet-w40's 40-channel pixels).  Wave grid MW x NW over (pixel blocks x channel blocks); TM 32-pixel blocks (= two
// rows of the BH x 16 output block) and TN 32-channel blocks per wave; BH = 2 MW TM is 16 (eight waves) or, where the weights of
// a 32-channel block take most of a wave's registers (80 -> 80 channels: 180 of them), 8 rows under THREE waves (one per
// 32-channel block, one wave per SIMD: each may then use the 512 registers of its SIMD lane).
// POOL (the stem): the launch also applies the MaxPool2d(3, stride 2, padding 1) that follows conv1 + BN + ReLU (resnet.py:218-221).  Blocks
// of 16 x 16 conv outputs are laid 14 apart (origin 14 b - 1), so each holds every conv output that the 7 x 7 pooled pixels 7 b .. 7 b + 6
// need: the epilogue puts the block's fp16 outputs into an LDS tile instead of HBM, one barrier, then 392 threads (pooled pixel, 8
// channels) take the maximum over the window positions that lie inside the conv map -- maxpool3s2_f16_kernel's arithmetic on the same
// fp16 values, so the pooled map is BIT-IDENTICAL to conv + pool as two launches -- and store ONE 16-byte vector each.  The conv map
// (8x the pooled map's bytes) is never written or read; 1.3x the conv FLOPs are recomputed on a kernel that is nowhere near MFMA-bound.
template <int R, int S, int CPP, int TM, int TN, int MW, int NW, int NSLOT, bool HAS_RES, bool POOL = false>
__global__ __launch_bounds__(64 * MW * NW, ((MW * NW >= 4) ? 2 : 1)) void conv_hs_f16(const ConvParams p) {
    constexpr int NWV = MW * NW, NT = 64 * NWV;
    constexpr int BH = 2 * MW * TM;                    // output rows of a block
    constexpr int HH = BH + R - 1, HW = 16 + S - 1, HROWS = HH * HW;
    constexpr int NCH = R * S * CPP;                   // 16-byte chunks of the reduction
    constexpr int HP = (HROWS * CPP + NT - 1) / NT;    // DMA instructions per halo image and thread
    constexpr int SLOT = HP * NT * 16;                 // bytes of one halo image (padded to whole passes)
    constexpr int D = NSLOT - 1;
    constexpr int RB = HAS_RES ? TM * TN * 2 : 0, OS = POOL ? 512 / NT : TM * TN * 2;
    static_assert(!POOL || (!HAS_RES && BH == 16 && (NT == 512 || NT == 256) && TN * NW == 2), "pooled epilogue: 16 x 16 blocks of 64 channels, 512 pooling tasks");
    constexpr int ZW = TM * TN * 2 * 1024;
    constexpr int NSTEP = (NCH + 1) / 2;
    constexpr bool SWZ = CPP == 8;                     // 128-byte pixels: XOR swizzle; 80-byte pixels are conflict-free as they lie.
    // 32-byte pixels (the space-to-depth stem) are not -- a ds_read_b128's 16-lane service group {0-3, 12-15, 20-27} lands on the 8 even
    // (or odd) 16-byte slots: two-way conflicts, the 0.121 LDS conflicts per wave-cycle of the r03 PMC summary -- and they stay that way:
    // chunk ^ (halo column & 1) removes every conflict (tools/probe/swizzle_search.py --stem) and was measured on one box in round 4
    // (gpurun_out/r04/pl_swz*.json vs pl_ref*.json): 243 / 229 us against 225 / 222 us.  The pooled stem is bound by vector-instruction
    // issue, not by LDS cycles; the second base register per pixel block cost more than the conflicts.
    static_assert(NWV <= 8 && (BH == 16 || BH == 8), "at most eight waves over 16 x 16 or 8 x 16 output blocks");
    extern __shared__ __attribute__((aligned(16))) char hsm[];
    char *zones = hsm + NSLOT * SLOT;                  // [2][NWV][ZW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int mw = wave / NW, nw = wave - mw * NW;
    const int n0 = nw * TN * 32;                       // this wave's first output channel

    // ---- the blocks of this workgroup: b = blockIdx.x, + gridDim.x, ...
    const int tyn = POOL ? (p.pool_h + 6) / 7 : p.Ho / BH, txn = POOL ? (p.pool_w + 6) / 7 : p.Wo >> 4, per_img = tyn * txn, nblk = p.N * per_img;
    const int ntl = nblk > (int)blockIdx.x ? (nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (ntl == 0) return;

    const _Float16 *zero16 = reinterpret_cast<const _Float16 *>(p.zero);
    _Float16 *trash = reinterpret_cast<_Float16 *>(const_cast<float *>(p.zero) + 64 + 4 * lane);
    const _Float16 *Ain = reinterpret_cast<const _Float16 *>(p.in);
    const _Float16 *Rin = reinterpret_cast<const _Float16 *>(p.res);
    _Float16 *Out = reinterpret_cast<_Float16 *>(p.out);

    // ---- weights -> registers, once (issuing them behind the first halo requests instead measured the same: the launch's fixed
    // ~9 us are not their latency): block b, step Q: the 8 halfs k = 16 Q + 8 kh .. of row
    // swap23(l31) (conv_igemm's transposed-output convention: registers 8j .. 8j+7 are eight consecutive channels)
    const int wl31 = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    hf16x8 wreg[TN][NSTEP];
    {
        const _Float16 *wb = reinterpret_cast<const _Float16 *>(p.wgt) + (size_t)(n0 + wl31) * p.ldw + 8 * kh;
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int Q = 0; Q < NSTEP; ++Q) wreg[b][Q] = *reinterpret_cast<const hf16x8 *>(wb + (size_t)(32 * b) * p.ldw + 16 * Q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- halo DMA roles, block-invariant: pass i moves 16-byte unit L = 512 i + tid of the image = chunk L % CPP of halo pixel
    // L / CPP (with 128-byte pixels the chunk it FETCHES is XOR-swizzled by the pixel's COLUMN: the 16 lanes of a ds_read_b128 group
    // read 16 consecutive columns of two image rows -- every (column parity, chunk ^ key) pair once, no bank conflict -- and the key of
    // a tap's fragment depends on the tap's column offset only, so a fragment address is one block-invariant base per (pixel block,
    // column offset), one XOR for the chunk pair and an immediate for the tap's row)
    // row << 24 | column << 16 | element offset of (column, source chunk) inside an image row (lda == Cin == 8 CPP: conv_hs_supported);
    // units past the halo get row 200: always out of the image -> zero page
    int hyx[HP];
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const int L = NT * i + tid, hp = L / CPP, ch = L - hp * CPP;
        const int hx = hp - (hp / HW) * HW;
        static_assert(8 * (HW * CPP + CPP) < 65536, "column offset field");
        hyx[i] = ((hp < HROWS ? hp / HW : 200) << 24) | (hx << 16) | (8 * (hx * CPP + (SWZ ? (ch ^ ((hx >> 1) & 7)) : ch)));
    }

    auto block_origin = [&](int tt, int &n, int &by, int &bx) {
        const int g = (int)blockIdx.x + tt * (int)gridDim.x;
        n = g / per_img;
        const int rem = g - n * per_img;
        by = rem / txn;
        bx = rem - by * txn;
    };
    auto issue_halo = [&](int tt) {
        int n, by, bx;
        block_origin(tt, n, by, bx);
        const bool live = tt >= 0 && tt < ntl;
        char *dst = hsm + (((tt % NSLOT) + NSLOT) % NSLOT) * SLOT + wave * 1024;
#pragma unroll
        for (int i = 0; i < HP; ++i) {
            int h = hyx[i];
            asm volatile("" : "+v"(h));   // its fields are this block's arithmetic: hoisted, each would hold a register for the launch
            const int iy = (POOL ? by * 14 - 1 : by * BH) - p.pad_h + (int)((unsigned)h >> 24), ix0 = (POOL ? bx * 14 - 1 : bx * 16) - p.pad_w, ix = ix0 + ((h >> 16) & 255);
            const bool ok = live && ((unsigned)h >> 24) < 200 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            // (32-bit element index: conv_hs_supported keeps N H W lda below 2^31.  A 64-bit form lets the compiler keep one
            // block-invariant base POINTER per pass: ten registers the 80-channel variant does not have)
            // (unsigned: a wrapping sum is extended to 64 bits as a whole, not term by term)
            const _Float16 *src = ok ? Ain + ((unsigned)(((n * p.H + iy) * p.W + ix0) * (8 * CPP)) + ((unsigned)h & 0xffffu)) : zero16;
            asm volatile("" : "+v"(src));   // ONE DMA instruction per schedule entry
            HMV_HGLDS16(src, dst + i * (NT * 16));
        }
    };
    // pixel of lane l31 in 32-pixel block a of this wave: output row 2 (mw TM + a) + (l31 >> 4), column l31 & 15
    auto issue_R = [&](int tt) {
        if constexpr (HAS_RES) {
            int n, by, bx;
            block_origin(tt, n, by, bx);
            const bool live = tt >= 0 && tt < ntl;
            char *z = zones + ((tt & 1) * NWV + wave) * ZW;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const size_t pix = (size_t)(n * p.Ho + by * BH + 2 * (mw * TM + a) + (l31 >> 4)) * p.Wo + bx * 16 + (l31 & 15);
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int col = n0 + 32 * b + 16 * j + 8 * kh;
                        const _Float16 *src = (live && col < p.Cout) ? Rin + pix * p.ldr + col : zero16;
                        asm volatile("" : "+v"(src));
                        HMV_HGLDS16(src, z + ((a * TN + b) * 2 + j) * 1024);
                    }
            }
        }
    };

    // ---- prologue: the D blocks "before the first" replay the schedule (real DMAs where they belong to block 0 .., dummies else)
    for (int ft = -D; ft < 0; ++ft) {
        issue_halo(ft + D);
        issue_R(ft + 1);
#pragma unroll
        for (int i = 0; i < OS; ++i)
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(trash), "v"(hf32x4{0.f, 0.f, 0.f, 0.f}) : "memory");
    }

    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    const __attribute__((address_space(4))) float *bp0 =
        (const __attribute__((address_space(4))) float *)(p.bias + __builtin_amdgcn_readfirstlane(n0));
    // halo pixel index of the lane's output pixel at tap (0, 0), per block a
    int hb[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) hb[a] = (2 * (mw * TM + a) + (l31 >> 4)) * HW + (l31 & 15);
    hf32x16 acc[TM][TN];

    // diagnostic launches only (hmv_bench_conv with HMV_BENCH_CLOCK=1): shader cycles workgroup 0's first wave spends per phase, summed
    // over its blocks -- {halo wait + barrier, main loop, epilogue, blocks, total}
    unsigned long long ph_wait = 0, ph_main = 0, ph_epi = 0, ph_t0 = 0;
    if (p.dbg) ph_t0 = __builtin_amdgcn_s_memtime();
    for (int tt = 0; tt < ntl; ++tt) {
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
        if (p.dbg) ts0 = __builtin_amdgcn_s_memtime();
        int bz;   // bias re-read per block behind an opaque zero offset (scalar cache): hoisted it would cost 16 TN VGPRs
        asm volatile("s_mov_b32 %0, 0" : "=s"(bz));
        const __attribute__((address_space(4))) float *bp = bp0 + bz;
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = 32 * b + 16 * (e >> 3) + (e & 7);
                const float v0 = bp[c], v1 = bp[c + 8];
                const float v = kh ? v1 : v0;
#pragma unroll
                for (int a = 0; a < TM; ++a) acc[a][b][e] = v;
            }
        hs_wait_vm<hs_after_halo(D, HP, RB, OS)>();                           // my share of halo(tt) has landed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // everyone's has; everyone is done with halo(tt - 1)
        if (p.dbg) ts1 = __builtin_amdgcn_s_memtime();
        issue_halo(tt + D);                                                   // ... whose slot takes halo(tt + D)
        issue_R(tt + 1);
        const char *himg = hsm + (tt % NSLOT) * SLOT;
        // the fragment addresses are block-invariant: left visible, the compiler hoists all of them out of the block loop and
        // spills them (the weights hold up to 144 registers).  An opaque zero makes them this block's own arithmetic.
        int hz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(hz));
        // pixel fragments of step Q (chunk of lanes 0-31 / 32-63; a chunk past the reduction's end multiplies zero weights: any
        // finite pixel will do)
        auto load_px = [&](int Q, hf16x8 (&px)[TM]) {
            constexpr int dummy = 0;
            const int gA = 2 * Q < NCH ? 2 * Q : dummy, gB = 2 * Q + 1 < NCH ? 2 * Q + 1 : dummy;
            const int tA = gA / CPP, cA = gA - tA * CPP, tB = gB / CPP, cB = gB - tB * CPP;
            const int oA = (tA / S) * HW + tA % S, oB = (tB / S) * HW + tB % S;   // halo pixel offset of the tap
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                static_assert(!SWZ || CPP % 2 == 0, "swizzled pixels take the pipelined loop below");
                const int u = (hb[a] + hz) * CPP + (kh ? oB * CPP + cB : oA * CPP + cA);
                px[a] = *reinterpret_cast<const hf16x8 *>(himg + u * 16);
            }
        };
        // One wave per SIMD (the three-wave variant) has nobody to hide an LDS round trip behind.  Its fragments of step Q + 1 are
        // requested BEFORE the MFMAs of step Q, into the other of two register sets, and waited for with a COUNTED lgkmcnt: the
        // reads are inline asm (the compiler, left to track them, reuses one register set and drains the LDS queue before every MFMA
        // group -- 3x the MFMA time), the wait names the registers it releases so that no MFMA can move above it.
        constexpr bool PIPE = true;
        hf16x8 pxs[2][TM];
        if constexpr (PIPE) {
            // an even CPP keeps the two chunks of a k16 step inside one tap: lanes 32-63 read 16 bytes behind lanes 0-31, and the
            // step's own displacement is an instruction immediate -- four address registers per block, no per-step arithmetic
            // An odd CPP (HRNet-w40's 40-channel pixels) puts the two chunks of some k16 steps into two taps.  Inside an image row the next
            // tap is the next pixel, i.e. the next 16 bytes, as inside a tap; only where the tap sequence changes to the next image row do
            // lanes 32-63 read DROW bytes behind lanes 0-31 instead of 16: those steps (one, for 3x3 x 5 chunks) take a second base
            constexpr bool ODD = CPP % 2 == 1;
            constexpr int DROW = (HW * CPP - ((S - 1) * CPP + CPP - 1)) * 16;
            static_assert(!ODD || !SWZ, "odd chunk counts: unswizzled pixels");
            // unswizzled: one base per pixel block, the whole displacement an immediate.  Swizzled (128-byte pixels): one base per (pixel
            // block, tap column s) holding the lane's key (kh ^ column key) in its chunk field; chunk pair q XORs bit 5 / 6, the tap row is the immediate
            constexpr int NB = SWZ ? S : (ODD ? 2 : 1);
            unsigned vb[NB][TM];
#pragma unroll
            for (int sx = 0; sx < NB; ++sx)
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    if constexpr (SWZ)
                        vb[sx][a] = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char *)himg +
                                    (unsigned)(((hb[a] + hz + sx) * 8 + (kh ^ ((((l31 & 15) + sx) >> 1) & 7))) * 16);
                    else
                        vb[sx][a] = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char *)himg +
                                    (unsigned)((hb[a] + hz) * CPP * 16 + kh * (sx ? DROW : 16));
                }
            auto load_asm = [](auto Qc, hf16x8 (&px)[TM], const unsigned (&vbr)[NB][TM]) {
                constexpr int Q = decltype(Qc)::value, tA = (2 * Q) / CPP, cA = 2 * Q - tA * CPP;
                if constexpr (SWZ) {
                    constexpr int off = (tA / S) * HW * 128;
                    static_assert(off < 65536 && CPP == 8, "ds_read offset field");
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        const unsigned ad = vbr[tA % S][a] ^ (unsigned)(cA * 16);
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(px[a]) : "v"(ad), "n"(off));
                    }
                } else {
                    constexpr int off = (((tA / S) * HW + tA % S) * CPP + cA) * 16;
                    // lanes 32-63: chunk 2 Q + 1 (past the reduction's end it multiplies zero weights: the 16 bytes behind the last chunk,
                    // inside the slot's zero-filled padding, do)
                    constexpr bool strad = ODD && cA == CPP - 1 && tA % S == S - 1 && 2 * Q + 1 < NCH;
                    static_assert(off + (strad ? DROW : 16) + 16 <= SLOT && off < 65536, "inside the halo slot; ds_read offset field");
#pragma unroll
                    for (int a = 0; a < TM; ++a) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(px[a]) : "v"(vbr[strad ? 1 : 0][a]), "n"(off));
                }
            };
            load_asm(std::integral_constant<int, 0>{}, pxs[0], vb);
            hs_static_for<NSTEP>([&](auto Qc) {
                constexpr int Q = decltype(Qc)::value;
                if constexpr (Q + 1 < NSTEP) {
                    load_asm(std::integral_constant<int, Q + 1>{}, pxs[(Q + 1) & 1], vb);
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(pxs[Q & 1][0]) : "n"(TM));   // the TM reads just issued may fly
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pxs[Q & 1][0]));
                }
#pragma unroll
                for (int a = 1; a < TM; ++a) asm volatile("" : "+v"(pxs[Q & 1][a]));
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[b][Q], pxs[Q & 1][a], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
#pragma unroll
            for (int Q = 0; Q < NSTEP; ++Q) {
                load_px(Q, pxs[0]);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[b][Q], pxs[0][a], acc[a][b], 0, 0, 0);
                // keep the scheduler from hoisting every step's address arithmetic and fragment reads to the top of the block
                if ((Q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue
        if (p.dbg) { asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[TM - 1][TN - 1][15])); ts2 = __builtin_amdgcn_s_memtime(); }
        if constexpr (HAS_RES) hs_wait_vm<hs_after_residual(HP, RB, OS)>();
        int n, by, bx;
        block_origin(tt, n, by, bx);
        const char *z = zones + ((tt & 1) * NWV + wave) * ZW + lane * 16;
        if constexpr (POOL) {
            // the block's outputs -> LDS tile [256 pixels][64 channels], 16-byte chunks XOR-swizzled by the pixel (inline asm: a C++ store
            // to LDS makes the compiler wait for every LDS-DMA in flight)
            const unsigned tile = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char *)zones;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int pxl = 32 * (mw * TM + a) + l31;
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        // relu(acc * 1 + 0) -> fp16 as round-then-clamp on packed halves: the same bits for every input (a pooled launch
                        // has ReLU and unscaled weights: conv_hs_supported), 1 instead of 2.5 vector instructions per output -- this kernel
                        // is bound by instruction issue, not by the matrix pipe
                        hf16x8 hv;
#pragma unroll
                        for (int u = 0; u < 8; ++u) hv[u] = (_Float16)acc[a][b][8 * j + u];
                        {
                            typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));
                            hu32x4 w = __builtin_bit_cast(hu32x4, hv);
#pragma unroll
                            for (int u = 0; u < 4; ++u) asm("v_pk_max_f16 %0, %1, 0" : "=v"(w[u]) : "v"(w[u]));
                            hv = __builtin_bit_cast(hf16x8, w);
                        }
                        const unsigned ad = tile + (unsigned)(pxl * 128 + ((((n0 >> 3) + 4 * b + 2 * j + kh) ^ (pxl & 7)) * 16));
                        asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(hv) : "memory");
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // pooled pixel (py, px) of the block x 8 channels per task (512 / NT per thread); window rows 2 py .. 2 py + 2 of the tile = conv rows 14 by - 1 + ..
#pragma unroll
            for (int task0 = 0; task0 < 512; task0 += NT) {
            const int task = task0 + tid;
            const int pp = task >> 3, c8 = task & 7, py = pp / 7, pxx = pp - 7 * py;
            const int gy = 7 * by + py, gx = 7 * bx + pxx;
            hf16x8 m;
            {   // all nine window vectors are requested at once (every window lies inside the tile; threads past the 49 pooled pixels
                // read pixel 0 and store to the trash page).  A window position outside the conv map does not count (the pool kernel
                // skips it): with even map sizes (conv_hs_supported) the only such positions of a pooled pixel that is KEPT are conv row
                // / column -1 -- the first row / column of the first block's tile -- and reading the window's middle row / column a second
                // time instead leaves the maximum unchanged.  v_pk_max_f16 directly: the builtin canonicalises both operands first
                const int py_ = pp < 49 ? py : 0, px_ = pp < 49 ? pxx : 0;
                const int r0 = 2 * py_ + ((by == 0 && py_ == 0) ? 1 : 0), c0 = 2 * px_ + ((bx == 0 && px_ == 0) ? 1 : 0);
                typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));
                hu32x4 wv[3][3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int pl = (r == 0 ? r0 : 2 * py_ + r) * 16 + (q == 0 ? c0 : 2 * px_ + q);
                        wv[r][q] = *reinterpret_cast<const hu32x4 *>(zones + pl * 128 + ((c8 ^ (pl & 7)) * 16));
                    }
                hu32x4 mw_ = wv[0][0];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (r | q) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) asm("v_pk_max_f16 %0, %1, %2" : "=v"(mw_[u]) : "v"(mw_[u]), "v"(wv[r][q][u]));
                        }
                m = __builtin_bit_cast(hf16x8, mw_);
            }
            _Float16 *dst = (pp < 49 && gy < p.pool_h && gx < p.pool_w) ? Out + ((size_t)(n * p.pool_h + gy) * p.pool_w + gx) * p.ldc + 8 * c8 : trash;
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(m) : "memory");
            }
        } else {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const size_t pix = (size_t)(n * p.Ho + by * BH + 2 * (mw * TM + a) + (l31 >> 4)) * p.Wo + bx * 16 + (l31 & 15);
            _Float16 *orow = Out + pix * p.ldc + n0 + 8 * kh;
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    hf16x8 rr = {0, 0, 0, 0, 0, 0, 0, 0};
                    if constexpr (HAS_RES) rr = *reinterpret_cast<const hf16x8 *>(z + ((a * TN + b) * 2 + j) * 1024);
                    hf16x8 hv;
#pragma unroll
                    for (int u = 0; u < 8; ++u) hv[u] = (_Float16)fmaxf(acc[a][b][8 * j + u] * p.acc_scale + (float)rr[u], lo);
                    // (every block of a launch is whole: H and W are multiples of 16; channel groups past Cout -- 40-channel
                    // layers on 64 weight rows -- go to the trash page: every lane stores, the counts above rely on it)
                    _Float16 *dst = (n0 + 32 * b + 16 * j + 8 * kh) < p.Cout ? orow + 32 * b + 16 * j : trash;
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(hv) : "memory");
                }
            // (the 80-channel variant holds 180 weight registers: finish one pixel block before the next one's residual vectors,
            // accumulator copies and results become live)
            if constexpr (CPP == 10) __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (p.dbg) { const unsigned long long ts3 = __builtin_amdgcn_s_memtime(); ph_wait += ts1 - ts0; ph_main += ts2 - ts1; ph_epi += ts3 - ts2; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (p.dbg && blockIdx.x == 0 && tid == 0) {
        p.dbg[0] = ph_wait; p.dbg[1] = ph_main; p.dbg[2] = ph_epi; p.dbg[3] = (unsigned long long)ntl; p.dbg[4] = __builtin_amdgcn_s_memtime() - ph_t0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_hs_stem_f32: the same structure for the fp32 path's stem (conv1 7x7 s2 as a 4x4 stride-1 convolution over 2x2 space-to-depth
// frames of 12 channels, 48-byte pixels; conv_igemm: `128x64,dense`, six k-steps per tile, 0.53 of the fp32 MFMA peak).  A 16-byte chunk
// is 4 channels and feeds four v_mfma_f32_32x32x2_f32 steps (k = 8 q + 4 kh + e: conv_igemm's pairing and order); the weights of a wave's
// 32 channels are K / 2 = 96 registers; the accumulator starts at zero and the bias is added in the epilogue, (acc + bias) + 0 -> relu,
// as conv_igemm's fp32 kernels do: bit-identical.  No residual.  4 waves along the pixels x 2 along the 64 channels.
template <int NSLOT>
__global__ __launch_bounds__(512, 2) void conv_hs_stem_f32(const ConvParams p) {
    constexpr int R = 4, S = 4, CPP = 3, TM = 2, MW = 4, NW = 2, NT = 512;
    constexpr int HH = 16 + R - 1, HW = 16 + S - 1, HROWS = HH * HW;
    constexpr int NCH = R * S * CPP, NSTEP = NCH / 2;     // 48 chunks of 4 channels, 24 chunk pairs
    constexpr int HP = (HROWS * CPP + NT - 1) / NT, SLOT = HP * NT * 16;
    constexpr int D = NSLOT - 1, OS = TM * 4;
    extern __shared__ __attribute__((aligned(16))) char hsm[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int mw = wave / NW, nw = wave - mw * NW;
    const int n0 = nw * 32;
    const int tyn = p.Ho >> 4, txn = p.Wo >> 4, per_img = tyn * txn, nblk = p.N * per_img;
    const int ntl = nblk > (int)blockIdx.x ? (nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (ntl == 0) return;

    const float *zero32 = p.zero;
    float *trash = const_cast<float *>(p.zero) + 64 + 4 * lane;
    const float *Ain = reinterpret_cast<const float *>(p.in);
    float *Out = reinterpret_cast<float *>(p.out);

    // weights and bias -> registers, once: row n0 + l31, chunk pair Q: W[row][8 Q + 4 kh .. + 3]
    hf32x4 wreg[NSTEP], bias[4];
    {
        const float *wb = reinterpret_cast<const float *>(p.wgt) + (size_t)(n0 + l31) * p.ldw + 4 * kh;
#pragma unroll
        for (int Q = 0; Q < NSTEP; ++Q) wreg[Q] = *reinterpret_cast<const hf32x4 *>(wb + 8 * Q);
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[g] = *reinterpret_cast<const hf32x4 *>(p.bias + n0 + 8 * g + 4 * kh);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // halo DMA roles: unit L = 512 i + tid = chunk L % 3 of halo pixel L / 3; row << 24 | column << 16 | element offset inside an image row
    int hyx[HP];
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const int L = NT * i + tid, hp = L / CPP, ch = L - hp * CPP, hx = hp - (hp / HW) * HW;
        hyx[i] = ((hp < HROWS ? hp / HW : 200) << 24) | (hx << 16) | (4 * (hx * CPP + ch));
    }
    auto block_origin = [&](int tt, int &n, int &by, int &bx) {
        const int g = (int)blockIdx.x + tt * (int)gridDim.x;
        n = g / per_img;
        const int rem = g - n * per_img;
        by = rem / txn;
        bx = rem - by * txn;
    };
    auto issue_halo = [&](int tt) {
        int n, by, bx;
        block_origin(tt, n, by, bx);
        const bool live = tt >= 0 && tt < ntl;
        char *dst = hsm + (((tt % NSLOT) + NSLOT) % NSLOT) * SLOT + wave * 1024;
#pragma unroll
        for (int i = 0; i < HP; ++i) {
            int h = hyx[i];
            asm volatile("" : "+v"(h));
            const int iy = by * 16 - p.pad_h + (int)((unsigned)h >> 24), ix0 = bx * 16 - p.pad_w, ix = ix0 + ((h >> 16) & 255);
            const bool ok = live && ((unsigned)h >> 24) < 200 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float *src = ok ? Ain + ((unsigned)(((n * p.H + iy) * p.W + ix0) * (4 * CPP)) + ((unsigned)h & 0xffffu)) : zero32;
            asm volatile("" : "+v"(src));   // ONE DMA instruction per schedule entry
            HMV_HGLDS16(src, dst + i * (NT * 16));
        }
    };
    for (int ft = -D; ft < 0; ++ft) {
        issue_halo(ft + D);
#pragma unroll
        for (int i = 0; i < OS; ++i)
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(trash), "v"(hf32x4{0.f, 0.f, 0.f, 0.f}) : "memory");
    }

    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    int hb[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) hb[a] = (2 * (mw * TM + a) + (l31 >> 4)) * HW + (l31 & 15);
    hf32x16 acc[TM];
    for (int tt = 0; tt < ntl; ++tt) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
        hs_wait_vm<hs_after_halo(D, HP, 0, OS)>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue_halo(tt + D);
        const char *himg = hsm + (tt % NSLOT) * SLOT;
        int hz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(hz));
#pragma unroll
        for (int Q = 0; Q < NSTEP; ++Q) {
            const int gA = 2 * Q, gB = 2 * Q + 1;
            const int tA = gA / CPP, cA = gA - tA * CPP, tB = gB / CPP, cB = gB - tB * CPP;
            const int oA = (tA / S) * HW + tA % S, oB = (tB / S) * HW + tB % S;
            hf32x4 px[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int u = (hb[a] + hz) * CPP + (kh ? oB * CPP + cB : oA * CPP + cA);
                px[a] = *reinterpret_cast<const hf32x4 *>(himg + u * 16);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < TM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[Q][e], px[a][e], acc[a], 0, 0, 0);
            if ((Q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        int n, by, bx;
        block_origin(tt, n, by, bx);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const size_t pix = (size_t)(n * p.Ho + by * 16 + 2 * (mw * TM + a) + (l31 >> 4)) * p.Wo + bx * 16 + (l31 & 15);
            float *orow = Out + pix * p.ldc + n0 + 4 * kh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                hf32x4 t;
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = fmaxf((acc[a][4 * g + u] + bias[g][u]) + 0.f, lo);
                float *dst = orow + 8 * g;
                asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(t) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// ====================================================================== host side
static int g_hs_mode = -1;   // -1: the launcher's rule (HMV_NO_HS=1 disables it); 0 never; 1 whenever supported (op-level tests)
void conv_hs_set_mode(int mode) { g_hs_mode = mode; }

// 0: no instantiation; 1: 3x3 pad 1, 64 -> 64 channels; 2: the 4x4 space-to-depth stem, 16 -> 64 channels; 3 / 4: 3x3 pad 1, 40 -> 40 /
// 80 -> 80 channels in the plain (r, s, c) K order (HRNet-w40's two highest-resolution branches, hrnet.py:96-221; kind 4 on 8 x 16
// blocks under three waves)
static bool hs_wave8() { static const bool v = HMV_DEV_ENV("HMV_HS_8WAVE") != nullptr; return v; }   // development knob (A/B runs)
static int hs_bh(int kind) { return kind == 4 ? 8 : 16; }   // (the size rule counts 16-row blocks for kinds 1 - 3 whatever form runs)
static int hs_kind(const ConvParams &p) {
    if (p.R == 3 && p.S == 3 && p.pad_h == 1 && p.pad_w == 1 && p.Cin == 64 && p.Cout == 64 && p.Kpad == 576) return 1;
    if (p.R == 4 && p.S == 4 && p.pad_h == 2 && p.pad_w == 2 && p.Cin == 16 && p.Cout == 64 && p.Kpad == 256 && !p.res) return 2;
    if (p.R == 3 && p.S == 3 && p.pad_h == 1 && p.pad_w == 1 && p.Cin == 40 && p.Cout == 40 && p.Kpad == 384 && !p.rd_cout) return 3;
    if (p.R == 3 && p.S == 3 && p.pad_h == 1 && p.pad_w == 1 && p.Cin == 80 && p.Cout == 80 && p.Kpad == 768 && !p.rd_cout) return 4;
    return 0;
}
// fp32: the space-to-depth stem, 4x4 pad 2, 12 -> 64 channels (conv_hs_stem_f32)
static bool hs_stem32(const ConvParams &p) {
    return !p.in_f16 && !p.out_f16 && !p.res && p.R == 4 && p.S == 4 && p.pad_h == 2 && p.pad_w == 2 && p.Cin == 12 && p.Cout == 64 && p.Kpad == 192 && p.K == 192;
}

bool conv_hs_supported(const ConvParams &p) {
    static int off = -1;   // development knob: HMV_NO_HS=1 keeps these convs on conv_igemm (A/B runs)
    if (off < 0) off = HMV_DEV_ENV("HMV_NO_HS") ? 1 : 0;
    if (g_hs_mode == 0 || (g_hs_mode < 0 && off)) return false;
    if (p.pool) {   // conv + ReLU + MaxPool2d(3, 2, 1) in one launch: the fp16 stem only, at least four 7 x 7 pooled blocks per workgroup
        static const bool nopool = HMV_DEV_ENV("HMV_NO_STEMPOOL") != nullptr;   // development knob (A/B runs)
        if (nopool || hs_kind(p) != 2 || !p.in_f16 || !p.out_f16 || p.res || p.fill || p.act != ACT_RELU) return false;
        if (p.pool_h != (p.Ho + 2 - 3) / 2 + 1 || p.pool_w != (p.Wo + 2 - 3) / 2 + 1 || p.ldc != 64 || (p.Ho & 1) || (p.Wo & 1) || p.acc_shift) return false;
        if (g_hs_mode <= 0 && (long long)p.N * ((p.pool_h + 6) / 7) * ((p.pool_w + 6) / 7) < 4 * 256) return false;
    }
    if (hs_stem32(p)) {
        static const bool off32 = HMV_DEV_ENV("HMV_NO_HS32") != nullptr;   // development knob (A/B runs)
        if (off32 && g_hs_mode <= 0) return false;
        if (p.stride != 1 || p.up || p.in2 || p.ksl > 1 || p.phases > 1 || p.Ho != p.H || p.Wo != p.W || (p.H & 15) || (p.W & 15)) return false;
        if (p.cwrap || p.x3_plane || p.res_split || p.out_split || p.acc_shift || p.rd_cout || p.scatter || p.rg_out || p.fill) return false;
        if (p.act != ACT_NONE && p.act != ACT_RELU) return false;
        if ((p.lda ? p.lda : p.Cin) != 12 || ((p.ldw ? p.ldw : p.Kpad) & 3) || (p.ldc & 3) || p.ldc < 64) return false;
        if ((long long)p.N * p.H * p.W * 12 >= (1ll << 31)) return false;
        if (g_hs_mode > 0) return true;
        return (long long)p.N * (p.H >> 4) * (p.W >> 4) >= 4 * 256;
    }
    if (!hs_kind(p) || !p.in_f16 || !p.out_f16 || (p.res && !p.res_f16)) return false;
    if (p.stride != 1 || p.up || p.in2 || p.ksl > 1 || p.phases > 1 || p.Ho != p.H || p.Wo != p.W || p.H % hs_bh(hs_kind(p)) || (p.W & 15)) return false;
    if (p.cwrap || p.x3_plane || p.res_split || p.out_split || p.acc_shift || p.rd_cout || p.scatter || p.rg_out) return false;
    if (p.fill && p.ldc != p.Cout) return false;   // (pad columns to clear: conv_igemm's epilogue does that)
    if (p.act != ACT_NONE && p.act != ACT_RELU) return false;
    if ((p.lda ? p.lda : p.Cin) != p.Cin || ((p.ldw ? p.ldw : p.Kpad) & 7) || (p.ldc & 7) || (p.res && (p.ldr & 7))) return false;
    if ((long long)p.N * p.H * p.W * p.Cin >= (1ll << 31)) return false;   // 32-bit element offsets of the halo pixels
    if (g_hs_mode > 0) return true;
    return (long long)p.N * (p.H / hs_bh(hs_kind(p))) * (p.W >> 4) >= 4 * 256;   // at least four blocks per workgroup
}

template <int R, int S, int CPP, int TM, int TN, int MW, int NW, int NSLOT, bool HAS_RES, bool POOL = false>
static hipError_t launch_hs_one(const ConvParams &p, hipStream_t s) {
    constexpr int NWV = MW * NW, NT = 64 * NWV, BH = 2 * MW * TM;
    constexpr int HROWS = (BH + R - 1) * (16 + S - 1), HP = (HROWS * CPP + NT - 1) / NT;
    constexpr size_t lds = (size_t)NSLOT * HP * NT * 16 + (HAS_RES ? (size_t)2 * NWV * TM * TN * 2 * 1024 : 0) + (POOL ? 256 * 128 : 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool configured[64] = {};
    auto kern = conv_hs_f16<R, S, CPP, TM, TN, MW, NW, NSLOT, HAS_RES, POOL>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    const int nblk = POOL ? p.N * ((p.pool_h + 6) / 7) * ((p.pool_w + 6) / 7) : p.N * (p.H / BH) * (p.W >> 4);
    const int cap = NT == 256 ? 512 : 256;   // four-wave workgroups: two per CU
    hipLaunchKernelGGL(kern, dim3(nblk < cap ? nblk : cap), dim3(NT), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_hs(const ConvParams &p, hipStream_t s, const char **name) {
    if (hs_stem32(p)) {
        constexpr int NSLOT = 4, HP = (19 * 19 * 3 + 511) / 512;
        constexpr size_t lds = (size_t)NSLOT * HP * 512 * 16;
        static bool configured[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!configured[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_hs_stem_f32<NSLOT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            configured[dev] = true;
        }
        if (name) *name = "conv_hs_stem_f32<4x4,12->64>";
        const int nblk = p.N * (p.H >> 4) * (p.W >> 4);
        hipLaunchKernelGGL(conv_hs_stem_f32<NSLOT>, dim3(nblk < 256 ? nblk : 256), dim3(512), lds, s, p);
        return hipGetLastError();
    }
    switch (hs_kind(p)) {
        case 1:
            // two FOUR-wave workgroups per CU on 8 x 16 blocks (2 x 2 waves; 24 KB halo images): they drift apart, and one's halo wait
            // and epilogue run under the other's MFMAs (the pooled stem gained 24 % that way).  HMV_HS_8WAVE=1: the eight-wave 16 x 16 form
            if (p.res) {
                if (name) *name = "conv_hs_f16<3x3,64->64,res>";
                if (hs_wave8()) return launch_hs_one<3, 3, 8, 2, 1, 4, 2, 2, true>(p, s);
                return launch_hs_one<3, 3, 8, 2, 1, 2, 2, 2, true>(p, s);
            }
            if (name) *name = "conv_hs_f16<3x3,64->64>";
            if (hs_wave8()) return launch_hs_one<3, 3, 8, 2, 1, 4, 2, 3, false>(p, s);   // no landing zones: three halo images (two in flight)
            return launch_hs_one<3, 3, 8, 2, 1, 2, 2, 3, false>(p, s);
        case 2:
            // eight waves along the pixels, both 32-channel blocks per wave (128 weight registers): every pixel fragment read from LDS
            // feeds TWO MFMAs.  With one block per wave (4 x 2 waves) a k16 step is one 1 KB LDS read per 32-cycle MFMA on every SIMD --
            // exactly the LDS peak of the CU (128 B / clk), and the kernel ran at 0.34 MFMA-busy
            {
                static const bool old_split = HMV_DEV_ENV("HMV_STEM_4x2") != nullptr;   // development knob (A/B runs): the round-3 wave split
                if (old_split && !p.pool) {
                    if (name) *name = "conv_hs_f16<4x4,16->64>";
                    return launch_hs_one<4, 4, 2, 2, 1, 4, 2, 4, false>(p, s);
                }
            }
            if (p.pool) {
                if (name) *name = "conv_hs_f16<4x4,16->64,+maxpool>";
                // two FOUR-wave workgroups per CU (each wave four pixel rows x 64 channels; 68 KB of LDS each): they drift apart, and one's
                // epilogue + pooling (vector instructions) runs under the other's MFMAs.  HMV_STEM_POOL8=1: one eight-wave workgroup (A/B runs)
                static const bool pool8 = HMV_DEV_ENV("HMV_STEM_POOL8") != nullptr;
                if (pool8) return launch_hs_one<4, 4, 2, 1, 2, 8, 1, 4, false, true>(p, s);
                return launch_hs_one<4, 4, 2, 2, 2, 4, 1, 3, false, true>(p, s);
            }
            if (name) *name = "conv_hs_f16<4x4,16->64>";
            return launch_hs_one<4, 4, 2, 1, 2, 8, 1, 4, false>(p, s);
        case 3:   // 40-channel pixels: 26 KB halo images (16 KB for the four-wave form's 8 x 16 blocks)
            if (p.res) {
                if (name) *name = "conv_hs_f16<3x3,40->40,res>";
                if (hs_wave8()) return launch_hs_one<3, 3, 5, 2, 1, 4, 2, 3, true>(p, s);
                return launch_hs_one<3, 3, 5, 2, 1, 2, 2, 3, true>(p, s);
            }
            if (name) *name = "conv_hs_f16<3x3,40->40>";
            if (hs_wave8()) return launch_hs_one<3, 3, 5, 2, 1, 4, 2, 4, false>(p, s);
            return launch_hs_one<3, 3, 5, 2, 1, 2, 2, 4, false>(p, s);
        case 4:   // 80-channel pixels, 96 weight rows on three channel waves (one per SIMD, up to 512 registers each): 8 x 16 blocks, 30 KB halo images
            if (p.res) {
                if (name) *name = "conv_hs_f16<3x3,80->80,res>";
                return launch_hs_one<3, 3, 10, 2, 1, 2, 3, 3, true>(p, s);
            }
            if (name) *name = "conv_hs_f16<3x3,80->80>";
            return launch_hs_one<3, 3, 10, 2, 1, 2, 3, 4, false>(p, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace hmv
