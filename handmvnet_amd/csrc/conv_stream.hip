// conv_stream.hip -- persistent, weight-stationary 1x1 convolution for the fp16 path's HBM-bound layers.
//
// Replaces, for large pixel counts, the 1x1 convs of the Bottleneck whose reduction is short
// (/root/reference/src/models/backbones/resnet.py:124-144: conv3 + bn3 + `out += residual` + relu with K = planes <= 256, and the
// squeezing conv1 + bn1 + relu of layer1 / layer2):
//
//   out[m][n] = relu( sum_k A[m][k] * Wt[n][k] + bias[n] + res[m][n] )        fp16 rows in, fp32 accumulation, fp16 rows out
//
// Why a second kernel.  conv_igemm gives every 256 x 256 output tile its own workgroup: per tile it streams 128 KB of weights and
// 128 KB of pixels through the CU's vector-memory pipe for 128 KB of residual and 128 KB of output, the three phases (operands,
// residual burst, stores) run one after the other, and the next workgroup starts cold (~4 us).  These layers do 4-16 k-steps of
// MFMA work per tile and move at ~25 GB/s per CU under chip-wide load (HBM's share per CU: DESIGN.md section 8): 3.2 TB/s of
// algorithmic bytes.
//
// MI355X mapping
//   * ONE workgroup per CU for the whole launch (grid = 256).  A workgroup owns an N-slice (BN output channels) and walks the
//     pixel tiles of its stream; the workgroups that own the other slices of the same pixel tiles sit on the same XCD and run in
//     step, so a pixel tile leaves HBM once and is re-read from that XCD's L2.
//   * the WEIGHTS LIVE IN REGISTERS for the whole launch: a wave keeps the MFMA A-operand fragments of its 32*TN channels for
//     the full reduction (TN * K / 4 VGPRs: 128 for K = 256) -- no weight bytes move after the prologue, and no LDS is spent
//     on them.  The 160 KB of LDS are all in-flight memory: a ring of pixel pieces ([BM pixels][64 channels] each, LDS-DMA,
//     XOR-swizzled on the source side like conv_igemm) and two wave-private landing zones for the residual tiles.
//   * one static instruction schedule per pixel tile, identical for every wave and every tile:
//         piece step j:  wait(piece j landed) . barrier . DMA piece j + D . [j = 0: DMA the NEXT tile's residual] . MFMAs
//         epilogue:      wait(residual landed) . acc + residual -> relu -> fp16 . 16-byte row stores
//     so every wait is a COUNTED `s_waitcnt vmcnt(N)` whose N is a compile-time constant (vm_after_* below count the
//     instructions issued since the awaited one); nothing in the loop ever drains the queue.  The next tile's pixels and
//     residual (~90 KB) are in flight while this tile computes, drains and stores.  Rows past M and tiles past the stream's end
//     read the zero page and store to the trash page, which keeps the counts exact; the prologue replays the schedule of the
//     tiles "before the first" with the same dummies.
//   * operand roles, accumulation order and epilogue arithmetic are those of conv_igemm's transposed-output path (weights as
//     the MFMA A operand, bias as the accumulator's initial value, k ascending in 16-element blocks), so the results are
//     BIT-IDENTICAL to conv_igemm's -- which keeps a sample's bits independent of the batch it is in, whichever kernel the
//     launcher picks (tests/test_gpu_parity.py::test_stream_kernel_is_bit_identical).
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "kernels.h"

namespace hmv {

typedef float sf32x16 __attribute__((ext_vector_type(16)));
typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 sf16x8 __attribute__((ext_vector_type(8)));

#define HMV_SGLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)
#define HMV_SGLDS16_NT(gptr, lptr)   /* aux = 2: non-temporal */                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 2)

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// ---- the static schedule.  A pixel tile is NP + 1 slots: slot j < NP = piece step j (PA pixel DMAs, and at j = 0 the RB residual
// DMAs of the next tile behind them), slot NP = the epilogue (OS stores).  vector-memory instructions retire in issue order, so
// "X has landed" == "at most (instructions issued after X) are still out".
// RS = residual DMAs issued at EVERY piece step (the spread schedule), 0 = all RB of them at step 0
constexpr int sched_r(int RB, int RS, int j) { return RS ? RS : (j == 0 ? RB : 0); }
constexpr int sched_ops(int NP, int PA, int RB, int RS, int OS, int slot) { return slot < NP ? PA + sched_r(RB, RS, slot) : OS; }
// instructions issued after the pixel DMAs of piece (t, j) when step (t, j) begins; they went out at step index g - D
constexpr int sched_after_piece(int NP, int D, int PA, int RB, int RS, int OS, int j) {
    const int T0 = 8, g = T0 * NP + j - D, tq = g / NP, jq = g - tq * NP;
    int n = sched_r(RB, RS, jq);
    for (int s = tq * (NP + 1) + jq + 1; s < T0 * (NP + 1) + j; ++s) n += sched_ops(NP, PA, RB, RS, OS, s % (NP + 1));
    return n;
}
// instructions issued after the LAST residual DMA of tile t (step (t - 1, 0), or (t - 1, NP - 1) when spread) when the epilogue of
// tile t begins
constexpr int sched_after_residual(int NP, int PA, int RB, int RS, int OS) {
    const int T0 = 8, last = RS ? NP - 1 : 0;
    int n = 0;
    for (int s = (T0 - 1) * (NP + 1) + last + 1; s < T0 * (NP + 1) + NP; ++s) n += sched_ops(NP, PA, RB, RS, OS, s % (NP + 1));
    return n;
}
static_assert(sched_after_piece(4, 3, 1, 8, 0, 8, 0) == 10 && sched_after_piece(4, 3, 1, 8, 0, 8, 1) == 18 && sched_after_piece(4, 3, 1, 8, 0, 8, 2) == 18 &&
              sched_after_piece(4, 3, 1, 8, 0, 8, 3) == 10 && sched_after_residual(4, 1, 8, 0, 8) == 23, "the K = 256 schedule, counted by hand");
static_assert(sched_after_piece(4, 3, 1, 8, 2, 8, 0) == 2 + 3 + 3 + 8 && sched_after_residual(4, 1, 8, 2, 8) == 8 + 4 * 3, "the spread K = 256 schedule");

// TM x TN 32x32 blocks per wave, MW x NW waves (pixels x channels), NP 64-channel pieces of reduction, NSLOT ring slots.
// SPREAD: the next tile's residual DMAs go out RB / NP per piece step instead of all at step 0 (a smoother request stream);
// NT: 0 default cache policy, 1 residual rows non-temporal (read once), 2 pixel pieces too
// DUAL: the reduction is the concatenation [first source | second source] (conv3 + downsample of a layer's first Bottleneck as one
// GEMM, ConvParams::in2): pieces past ksplit come from the second tensor at pixel (ho * stride2, wo * stride2)
// N2 > 0 ("chain"): the launch also computes a FOLLOWING 1x1 conv with N2 output channels over its own output pixels (Bottleneck i's
// conv3 -> Bottleneck i + 1's conv1, resnet.py:124-130).  The workgroup owns ALL BN = Cout channels of its pixel tile; after the
// epilogue every wave writes its fp16 output block into an LDS image of the tile laid out like K2 = BN / 64 pixel pieces (for
// residual-bearing launches: over the landing zone it has just consumed, which is exactly its [64-channel piece][rows] region), one
// barrier, then wave (pixel block group, 32-channel block) runs the second GEMM's full reduction k ascending against weights it
// holds in registers for the launch (BN / 4 VGPRs) and stores its [32 pixels][32 channels] blocks.  Operand roles, bias-as-initial-
// accumulator, k16 order and epilogue arithmetic are those of the kernel that would have read the tensor back from HBM, so the
// chained result is BIT-IDENTICAL to the two launches it replaces; its stores join the epilogue slot of the static schedule.
template <int TM, int TN, int MW, int NW, int NP, int NSLOT, bool HAS_RES, bool SPREAD = false, int NT_ = 0, bool DUAL = false, int N2 = 0>
__global__ __launch_bounds__(64 * MW *NW, (MW * NW) / 4) void conv_stream_f16(const ConvParams p) {
    constexpr int NWV = MW * NW, NT = 64 * NWV;
    constexpr int BM = 32 * TM * MW, BN = 32 * TN * NW;
    constexpr int D = NSLOT - 1;                 // pieces in flight ahead of the one being consumed
    constexpr int PA = BM * 8 / NT;              // pixel DMA instructions per piece and thread (a piece row = 8 x 16 bytes)
    constexpr int NB2 = N2 / 32;                                   // chain: 32-channel output blocks of the second conv,
    constexpr int TM2 = N2 ? (BM / 32) * NB2 / NWV : 0;            //   pixel blocks per wave (they share the wave's weight block),
    constexpr int K2Q = BN / 16;                                   //   k16 steps of its reduction over this launch's BN channels
    static_assert(N2 == 0 || (N2 % 32 == 0 && TN == 2 && TM2 >= 1 && NWV % NB2 == 0 && (BM / 32) == (NWV / NB2) * TM2),
                  "chain: a wave's 64 output channels are one pixel piece of the second reduction; whole blocks per wave");
    constexpr int RB = HAS_RES ? TM * TN * 2 : 0, OS = TM * TN * 2 + 2 * TM2;
    constexpr int RS = (SPREAD && HAS_RES) ? RB / NP : 0;
    static_assert(!SPREAD || RB % NP == 0, "spread schedule: whole residual DMAs per step");
    constexpr int ZW = TM * TN * 2 * 1024;       // bytes of one wave's residual landing zone
    constexpr int NFAKE = (D + NP - 1) / NP;     // tiles "before the first" whose schedule the prologue replays
    static_assert(NT == 512 && BM % 64 == 0 && PA >= 1, "eight waves; whole DMA passes per piece");
    static_assert((NSLOT & (NSLOT - 1)) == 0 && NSLOT >= 2, "a power of two");
    extern __shared__ __attribute__((aligned(16))) char ssm[];
    _Float16 *sA = reinterpret_cast<_Float16 *>(ssm);            // [NSLOT][BM][64]
    char *zones = ssm + NSLOT * BM * 128;                         // [2][NWV][ZW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int mw = wave / NW, nw = wave - mw * NW;
    // landing zone of this wave; chain: zone (nw, mw) IS the region [piece nw][rows mw * TM * 32 ..] of the output tile's LDS image
    const int zi = N2 ? nw * MW + mw : wave;
    static_assert(N2 == 0 || !HAS_RES || ZW == TM * 32 * 128, "chain: a landing zone is the wave's region of the tile image");

    // ---- stream assignment: the nsl slices of one pixel stream are workgroups of ONE XCD (blockIdx & 7), dispatched together
    const int nsl = p.ntiles;                       // N-slices (Cout / BN), a divisor of 32
    const int loc = (int)blockIdx.x >> 3, per_xcd = 32 / nsl;
    const int slice = loc % nsl, stream = ((int)blockIdx.x & 7) * per_xcd + loc / nsl, nstreams = 8 * per_xcd;
    const int ntl = p.mtiles > stream ? (p.mtiles - stream + nstreams - 1) / nstreams : 0;   // tiles of this workgroup
    if (ntl == 0) return;
    const int n0 = slice * BN + nw * TN * 32;       // this wave's first output channel

    const _Float16 *zero16 = reinterpret_cast<const _Float16 *>(p.zero);
    _Float16 *trash = reinterpret_cast<_Float16 *>(const_cast<float *>(p.zero) + 64 + 4 * lane);   // 16 bytes per lane, never read
    const _Float16 *Ain = reinterpret_cast<const _Float16 *>(p.in);
    const _Float16 *Rin = reinterpret_cast<const _Float16 *>(p.res);
    _Float16 *Out = reinterpret_cast<_Float16 *>(p.out);

    // ---- weights -> registers, once.  MFMA A operand of block b, k16 step Q: 8 halfs k = 16 Q + 8 kh .. + 7 of weight row
    // swap23(l31) of the block (the row swap makes a lane's registers 8j .. 8j+7 eight CONSECUTIVE channels: conv_igemm TOUT)
    const int wl31 = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    sf16x8 wreg[TN][NP * 4];
    {
        const _Float16 *wb = reinterpret_cast<const _Float16 *>(p.wgt) + (size_t)(n0 + wl31) * p.ldw + 8 * kh;
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int Q = 0; Q < NP * 4; ++Q) wreg[b][Q] = *reinterpret_cast<const sf16x8 *>(wb + (size_t)(32 * b) * p.ldw + 16 * Q);
    }
    // chain: wave -> (pixel block group pg2, 32-channel block ob2) of the second conv; its weight block, the whole reduction
    const int ob2 = N2 ? wave % (N2 ? NB2 : 1) : 0, pg2 = N2 ? wave / (N2 ? NB2 : 1) : 0;
    sf16x8 w2reg[N2 ? K2Q : 1];
    if constexpr (N2 > 0) {
        const _Float16 *wb2 = reinterpret_cast<const _Float16 *>(p.nx_wgt) + (size_t)(32 * ob2 + wl31) * p.nx_ldw + 8 * kh;
#pragma unroll
        for (int Q = 0; Q < K2Q; ++Q) w2reg[Q] = *reinterpret_cast<const sf16x8 *>(wb2 + 16 * Q);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the counted waits below see only the schedule's own instructions

    // pixel DMA role: thread -> (row arow of each 64-row pass, physical chunk tid & 7) fetching logical chunk kqs (XOR swizzle)
    const int arow = tid >> 3, kqs = (tid & 7) ^ ((tid >> 4) & 7);
    const int fsw = (l31 >> 1) & 7;

    // issue the pixel DMAs of piece G (in this workgroup's own piece sequence; outside [0, ntl * NP) -> zero page)
    // DUAL: element offset of this thread's pixel rows in the second source, computed at the FIRST piece of a tile that reads it and kept
    // for the tile's later ones (pieces go out in order; two integer divisions per row and piece otherwise: the k384 launch was bound by them)
    size_t off2[DUAL ? PA : 1] = {};
    auto issue_A = [&](int G) {
        const int tt = G >= 0 ? G / NP : -1, jj = G >= 0 ? G - tt * NP : 0, slot = G & (NSLOT - 1);
        const int mt = stream + tt * nstreams;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = mt * BM + i * 64 + arow;
            const bool ok = tt >= 0 && tt < ntl && m < p.M;
            const _Float16 *src = zero16;
            if constexpr (DUAL) {
                if (jj * 64 == p.ksplit) {   // (wave-uniform)
                    const int hw = p.Ho * p.Wo, n = m / hw, rem = m - n * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    off2[i] = ((size_t)(n * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.lda2 + 8 * kqs;
                }
                if (ok && jj * 64 >= p.ksplit) {
                    src = reinterpret_cast<const _Float16 *>(p.in2) + off2[i] + (jj * 64 - p.ksplit);
                } else if (ok) {
                    src = Ain + (size_t)m * p.lda + jj * 64 + 8 * kqs;
                }
            } else if (ok) {
                src = Ain + (size_t)m * p.lda + jj * 64 + 8 * kqs;
            }
            asm volatile("" : "+v"(src));   // ONE DMA instruction per schedule entry: keep the select out of the control flow
            if constexpr (NT_ >= 2) HMV_SGLDS16_NT(src, sA + ((slot * BM + i * 64 + wave * 8) * 64));
            else HMV_SGLDS16(src, sA + ((slot * BM + i * 64 + wave * 8) * 64));
        }
    };
    // issue the residual DMAs of tile tt: every lane fetches ITS OWN 8 channels of its own pixel, lane-linear landing zone
    // (instruction idx = (a * TN + b) * 2 + j; `first` .. `first + count` go out in this call)
    auto issue_R = [&](int tt, auto first_c, auto count_c) {
        if constexpr (HAS_RES) {
            constexpr int first = decltype(first_c)::value, count = decltype(count_c)::value;
            const int mt = stream + tt * nstreams;
            char *z = zones + ((tt & 1) * NWV + zi) * ZW;
#pragma unroll
            for (int idx = first; idx < first + count; ++idx) {
                const int a = idx / (2 * TN), b = (idx / 2) % TN, j = idx & 1;
                const int m = mt * BM + (mw * TM + a) * 32 + l31;
                const bool ok = tt >= 0 && tt < ntl && m < p.M;
                const _Float16 *src = ok ? Rin + (size_t)m * p.ldr + n0 + 32 * b + 16 * j + 8 * kh : zero16;
                asm volatile("" : "+v"(src));
                if constexpr (NT_ >= 1) HMV_SGLDS16_NT(src, z + idx * 1024);
                else HMV_SGLDS16(src, z + idx * 1024);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using IRB = std::integral_constant<int, RB>;
    using IRS = std::integral_constant<int, RS>;

    // ---- prologue: the schedule of the NFAKE tiles before the first, with every instruction it would have issued
    // (real ones where they belong to tile 0 .., dummies otherwise), so that the counted waits hold from tile 0 on
#pragma unroll
    for (int ft = -NFAKE; ft < 0; ++ft) {
        static_for<NP>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            issue_A(ft * NP + j + D);
            if constexpr (RS > 0) issue_R(ft + 1, std::integral_constant<int, j * RS>{}, IRS{});
            else if constexpr (j == 0) issue_R(ft + 1, I0{}, IRB{});
        });
#pragma unroll
        for (int i = 0; i < OS; ++i)   // inline asm: identical stores to one address must not be merged away
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(trash), "v"(sf32x4{0.f, 0.f, 0.f, 0.f}) : "memory");
    }

    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    // wave-uniform address in the CONSTANT address space: the bias reads below are scalar (s_load) instructions -- a vector load
    // here would enter the vmcnt queue and make the compiler drain it at every tile
    const __attribute__((address_space(4))) float *bp0 =
        (const __attribute__((address_space(4))) float *)(p.bias + __builtin_amdgcn_readfirstlane(n0));
    sf32x16 acc[TM][TN];

    for (int tt = 0; tt < ntl; ++tt) {
        // bias = initial accumulator value: register 8j + u holds channel 16j + 8kh + u of the block.  Re-read (scalar cache) per
        // tile behind an opaque zero offset: hoisted out of the loop the 16 * TN values would cost that many VGPRs for the launch
        int bz;
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
        static_for<NP>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            wait_vm<sched_after_piece(NP, D, PA, RB, RS, OS, j)>();                                   // my share of piece (tt, j) has landed
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // everyone's has; everyone is done with piece g - 1
            issue_A(tt * NP + j + D);                                            // ... whose slot takes piece g + D
            if constexpr (RS > 0) issue_R(tt + 1, std::integral_constant<int, j * RS>{}, IRS{});
            else if constexpr (j == 0) issue_R(tt + 1, I0{}, IRB{});
            const _Float16 *pa = sA + ((((tt * NP + j) & (NSLOT - 1)) * BM + mw * TM * 32 + l31) * 64);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sf16x8 px[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a) px[a] = *reinterpret_cast<const sf16x8 *>(pa + a * 32 * 64 + (((2 * q + kh) ^ fsw) * 8));
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[b][4 * j + q], px[a], acc[a][b], 0, 0, 0);
            }
        });
        // ---- epilogue: acc (+ residual) -> relu -> fp16, 16-byte stores of 8 consecutive channels
        if constexpr (HAS_RES) wait_vm<sched_after_residual(NP, PA, RB, RS, OS)>();
        const int mt = stream + tt * nstreams;
        const char *z = zones + ((tt & 1) * NWV + zi) * ZW + lane * 16;
        if constexpr (N2 > 0) {
            // the output tile's LDS image: [BN / 64 pieces][BM rows][64 channels], 16-byte chunks XOR-swizzled by the row like a pixel piece
            char *yt = HAS_RES ? zones + (tt & 1) * NWV * ZW : zones;
            sf16x8 rr[TM][TN][2], hv[TM][TN][2];
            if constexpr (HAS_RES) {   // ALL residual vectors leave the zone before the first output vector is written over it
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int j = 0; j < 2; ++j) rr[a][b][j] = *reinterpret_cast<const sf16x8 *>(z + ((a * TN + b) * 2 + j) * 1024);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int m = mt * BM + (mw * TM + a) * 32 + l31;
                _Float16 *orow = Out + (size_t)m * p.ldc + n0 + 8 * kh;
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        sf16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
                        if constexpr (HAS_RES) r = rr[a][b][j];
#pragma unroll
                        for (int u = 0; u < 8; ++u) hv[a][b][j][u] = (_Float16)fmaxf(acc[a][b][8 * j + u] * p.acc_scale + (float)r[u], lo);
                        _Float16 *dst = m < p.M ? orow + 32 * b + 16 * j : trash;
                        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(hv[a][b][j]) : "memory");
                        if constexpr (!HAS_RES) {   // no landing zone under the image: the vector can go to LDS at once
                            const unsigned ya = (unsigned)(size_t)(__attribute__((address_space(3))) char *)yt +
                                                (unsigned)((nw * BM + (mw * TM + a) * 32 + l31) * 128 + (((4 * b + 2 * j + kh) ^ fsw) * 16));
                            asm volatile("ds_write_b128 %0, %1" ::"v"(ya), "v"(hv[a][b][j]) : "memory");
                        }
                    }
            }
            // Inline asm: a C++ store to LDS makes the compiler wait for every LDS-DMA in flight (it cannot tell the zones apart).
            // Residual-bearing: the writes go over the zone the residual vectors were read from -- every hv depends on its rr, so all
            // reads have RETURNED before the first write is issued
            if constexpr (HAS_RES) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned ya = (unsigned)(size_t)(__attribute__((address_space(3))) char *)yt +
                                            (unsigned)((nw * BM + (mw * TM + a) * 32 + l31) * 128 + (((4 * b + 2 * j + kh) ^ fsw) * 16));
                        asm volatile("ds_write_b128 %0, %1" ::"v"(ya), "v"(hv[a][b][j]) : "memory");
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the tile image is complete
            // ---- second conv: bias as the initial accumulator, k ascending in 16-element blocks (conv_stream<..,kBN> / conv_igemm's order)
            const __attribute__((address_space(4))) float *bq =
                (const __attribute__((address_space(4))) float *)(p.nx_bias + __builtin_amdgcn_readfirstlane(32 * ob2)) + bz;
            sf32x16 acc2[TM2];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = 16 * (e >> 3) + (e & 7);
                const float v0 = bq[c], v1 = bq[c + 8];
                const float v = kh ? v1 : v0;
#pragma unroll
                for (int a = 0; a < TM2; ++a) acc2[a][e] = v;
            }
            const unsigned yr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)yt + (unsigned)((pg2 * TM2 * 32 + l31) * 128);
            // fragment reads in groups of four, group g + 1 requested before group g's MFMAs; the counted lgkmcnt names the
            // registers it releases (the compiler does not see the asm reads' latency: the "+v" operands order the MFMAs behind the wait)
            constexpr int G = 4 / TM2, NG = K2Q / G;
            static_assert(K2Q % G == 0 && G * TM2 == 4, "four fragment reads per group");
            sf16x8 px[2][G][TM2];
#define HMV_CHAIN_READ_GROUP(g)                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < G; ++i) _Pragma("unroll") for (int a = 0; a < TM2; ++a) {                                \
        const int Q = (g) * G + i;                                                                                                  \
        const unsigned ad = yr + (unsigned)(((Q >> 2) * BM + a * 32) * 128) + (unsigned)((((2 * (Q & 3) + kh) ^ fsw)) * 16);        \
        asm volatile("ds_read_b128 %0, %1" : "=v"(px[(g) & 1][i][a]) : "v"(ad) : "memory");                                         \
    }
            HMV_CHAIN_READ_GROUP(0)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                sf16x8(&pg)[G][TM2] = px[g & 1];
                if (g + 1 < NG) {
                    HMV_CHAIN_READ_GROUP(g + 1)
                    if constexpr (TM2 == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(pg[0][0]), "+v"(pg[1][0]), "+v"(pg[2][0]), "+v"(pg[3][0]) :: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(pg[0][0]), "+v"(pg[0][1]), "+v"(pg[1][0]), "+v"(pg[1][1]) :: "memory");
                } else {
                    if constexpr (TM2 == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pg[0][0]), "+v"(pg[1][0]), "+v"(pg[2][0]), "+v"(pg[3][0]) :: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pg[0][0]), "+v"(pg[0][1]), "+v"(pg[1][0]), "+v"(pg[1][1]) :: "memory");
                }
#pragma unroll
                for (int i = 0; i < G; ++i)
#pragma unroll
                    for (int a = 0; a < TM2; ++a) acc2[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2reg[g * G + i], pg[i][a], acc2[a], 0, 0, 0);
            }
#undef HMV_CHAIN_READ_GROUP
            const float lo2 = (p.nx_act == ACT_RELU) ? 0.f : -INFINITY;
            _Float16 *Out2 = reinterpret_cast<_Float16 *>(p.nx_out);
#pragma unroll
            for (int a = 0; a < TM2; ++a) {
                const int m = mt * BM + (pg2 * TM2 + a) * 32 + l31;
                _Float16 *orow = Out2 + (size_t)m * p.nx_ldc + 32 * ob2 + 8 * kh;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const sf16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
                    sf16x8 h2;
#pragma unroll
                    for (int u = 0; u < 8; ++u) h2[u] = (_Float16)fmaxf(acc2[a][8 * j + u] * p.acc_scale + (float)r[u], lo2);
                    _Float16 *dst = m < p.M ? orow + 16 * j : trash;
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(h2) : "memory");
                }
            }
        } else {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int m = mt * BM + (mw * TM + a) * 32 + l31;
            _Float16 *orow = Out + (size_t)m * p.ldc + n0 + 8 * kh;
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    sf16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
                    if constexpr (HAS_RES) r = *reinterpret_cast<const sf16x8 *>(z + ((a * TN + b) * 2 + j) * 1024);
                    sf16x8 hv;
#pragma unroll
                    for (int u = 0; u < 8; ++u) hv[u] = (_Float16)fmaxf(acc[a][b][8 * j + u] * p.acc_scale + (float)r[u], lo);
                    // every lane stores (the counts above rely on it): rows past M go to the trash page
                    // one global store per schedule entry (a select spelled in C++ may become two exec-masked stores, and a pointer
                    // made opaque loses its address space: flat stores retire out of order).  s_nop 1: the store must have read hv
                    _Float16 *dst = m < p.M ? orow + 32 * b + 16 * j : trash;
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(hv) : "memory");
                }
        }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_stream_f32: the same kernel for the fp32 path's residual-bearing 1x1 convs with short reductions (Bottleneck conv3 of layer1 /
// layer2 / layer3, K = 64 / 128 / 256: conv_igemm's `128x128,k16` and `256x128,k16,w8` families, 0.53 - 0.76 of the fp32 MFMA peak).
// A piece is [BM pixels][32 channels] (the same 128-byte rows), an MFMA step is v_mfma_f32_32x32x2_f32 on one float per lane: a
// 16-byte fragment feeds four steps (k = 8 q + 4 kh + e, e = 0 .. 3: conv_igemm's pairing and order), a wave's weights are
// K / 2 registers per 32-channel block, an accumulator register group 4 q .. 4 q + 3 is four consecutive channels 8 q + 4 kh .. (no
// row swap needed for 16-byte fp32 stores).  conv_igemm's fp32 kernels add the bias in the epilogue, (acc + bias) + residual: so does
// this one (the bias vectors live in registers for the launch), which keeps the results bit-identical.
// HALF: a FOUR-wave workgroup (two per CU, grid 512) with ONE wave-private residual zone, filled for the tile's own epilogue at its first
// piece step: the MFMA-bound K = 256 case, where the two workgroups of a CU drift apart and one's epilogue runs under the other's MFMAs
template <int TM, int TN, int MW, int NW, int NP, int NSLOT, bool HAS_RES, bool DUAL = false, bool HALF = false>
__global__ __launch_bounds__(64 * MW *NW, 2) void conv_stream_f32(const ConvParams p) {
    constexpr int NWV = MW * NW, NT = 64 * NWV;
    constexpr int BM = 32 * TM * MW, BN = 32 * TN * NW;
    constexpr int D = NSLOT - 1;
    constexpr int PA = BM * 8 / NT;
    constexpr int RB = HAS_RES ? TM * TN * 4 : 0, OS = TM * TN * 4, RS = 0;
    constexpr int ZW = TM * TN * 4 * 1024;
    constexpr int NFAKE = (D + NP - 1) / NP;
    constexpr int RPP = NT / 8;                  // piece rows per DMA pass
    static_assert((NT == 512 || (HALF && NT == 256)) && BM % RPP == 0 && PA >= 1 && (!HALF || (HAS_RES && !DUAL)), "whole DMA passes per piece");
    static_assert((NSLOT & (NSLOT - 1)) == 0 && NSLOT >= 2, "a power of two");
    extern __shared__ __attribute__((aligned(16))) char ssm[];
    float *sA = reinterpret_cast<float *>(ssm);                  // [NSLOT][BM][32]
    char *zones = ssm + NSLOT * BM * 128;                         // [2][NWV][ZW]  (HALF: [NWV][ZW])

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int mw = wave / NW, nw = wave - mw * NW;
    const int nsl = p.ntiles;
    const int loc = (int)blockIdx.x >> 3, per_xcd = ((int)gridDim.x >> 3) / nsl;
    const int slice = loc % nsl, stream = ((int)blockIdx.x & 7) * per_xcd + loc / nsl, nstreams = 8 * per_xcd;
    const int ntl = p.mtiles > stream ? (p.mtiles - stream + nstreams - 1) / nstreams : 0;
    if (ntl == 0) return;
    const int n0 = slice * BN + nw * TN * 32;

    const float *zero32 = p.zero;
    float *trash = const_cast<float *>(p.zero) + 64 + 4 * lane;
    const float *Ain = reinterpret_cast<const float *>(p.in);
    const float *Rin = reinterpret_cast<const float *>(p.res);
    float *Out = reinterpret_cast<float *>(p.out);

    // ---- weights and bias -> registers, once: block b, piece j, group q: W[n0 + 32 b + l31][32 j + 8 q + 4 kh .. + 3]
    sf32x4 wreg[TN][NP * 4], bias[TN][4];
    {
        const float *wb = reinterpret_cast<const float *>(p.wgt) + (size_t)(n0 + l31) * p.ldw + 4 * kh;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
            for (int Q = 0; Q < NP * 4; ++Q) wreg[b][Q] = *reinterpret_cast<const sf32x4 *>(wb + (size_t)(32 * b) * p.ldw + 8 * Q);
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[b][g] = *reinterpret_cast<const sf32x4 *>(p.bias + n0 + 32 * b + 8 * g + 4 * kh);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int arow = tid >> 3, kqs = (tid & 7) ^ ((tid >> 4) & 7);
    const int fsw = (l31 >> 1) & 7;
    auto issue_A = [&](int G) {
        const int tt = G >= 0 ? G / NP : -1, jj = G >= 0 ? G - tt * NP : 0, slot = G & (NSLOT - 1);
        const int mt = stream + tt * nstreams;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int m = mt * BM + i * RPP + arow;
            const bool ok = tt >= 0 && tt < ntl && m < p.M;
            const float *src = zero32;
            if constexpr (DUAL) {   // pieces past ksplit come from the second tensor at pixel (ho * stride2, wo * stride2)
                if (ok && jj * 32 >= p.ksplit) {
                    const int hw = p.Ho * p.Wo, n = m / hw, rem = m - n * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    src = reinterpret_cast<const float *>(p.in2) + ((size_t)(n * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.lda2 +
                          (jj * 32 - p.ksplit) + 4 * kqs;
                } else if (ok) {
                    src = Ain + (size_t)m * p.lda + jj * 32 + 4 * kqs;
                }
            } else if (ok) {
                src = Ain + (size_t)m * p.lda + jj * 32 + 4 * kqs;
            }
            asm volatile("" : "+v"(src));
            HMV_SGLDS16(src, sA + ((slot * BM + i * RPP + wave * 8) * 32));
        }
    };
    auto issue_R = [&](int tt) {
        if constexpr (HAS_RES) {
            const int mt = stream + tt * nstreams;
            char *z = zones + ((HALF ? 0 : (tt & 1)) * NWV + wave) * ZW;
#pragma unroll
            for (int idx = 0; idx < RB; ++idx) {
                const int a = idx / (4 * TN), b = (idx / 4) % TN, g = idx & 3;
                const int m = mt * BM + (mw * TM + a) * 32 + l31;
                const bool ok = tt >= 0 && tt < ntl && m < p.M;
                const float *src = ok ? Rin + (size_t)m * p.ldr + n0 + 32 * b + 8 * g + 4 * kh : zero32;
                asm volatile("" : "+v"(src));
                HMV_SGLDS16(src, z + idx * 1024);
            }
        }
    };
#pragma unroll
    for (int ft = -NFAKE; ft < 0; ++ft) {
        static_for<NP>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            issue_A(ft * NP + j + D);
            if constexpr (j == 0) issue_R(HALF ? ft : ft + 1);
        });
#pragma unroll
        for (int i = 0; i < OS; ++i)
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(trash), "v"(sf32x4{0.f, 0.f, 0.f, 0.f}) : "memory");
    }

    const float lo = (p.act == ACT_RELU) ? 0.f : -INFINITY;
    sf32x16 acc[TM][TN];
    for (int tt = 0; tt < ntl; ++tt) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        static_for<NP>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            wait_vm<sched_after_piece(NP, D, PA, RB, RS, OS, j)>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue_A(tt * NP + j + D);
            if constexpr (j == 0) issue_R(HALF ? tt : tt + 1);   // (HALF: this wave's epilogue reads of the previous tile are behind the lgkmcnt(0) above)
            const float *pa = sA + ((((tt * NP + j) & (NSLOT - 1)) * BM + mw * TM * 32 + l31) * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sf32x4 px[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a) px[a] = *reinterpret_cast<const sf32x4 *>(pa + a * 32 * 32 + (((2 * q + kh) ^ fsw) * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[b][4 * j + q][e], px[a][e], acc[a][b], 0, 0, 0);
            }
        });
        if constexpr (HAS_RES) wait_vm<HALF ? (NP - 1) * PA : sched_after_residual(NP, PA, RB, RS, OS)>();   // HALF: the piece DMAs of steps 1 .. NP - 1 followed it
        const int mt = stream + tt * nstreams;
        const char *z = zones + ((HALF ? 0 : (tt & 1)) * NWV + wave) * ZW + lane * 16;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int m = mt * BM + (mw * TM + a) * 32 + l31;
            float *orow = Out + (size_t)m * p.ldc + n0 + 4 * kh;
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    sf32x4 r = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (HAS_RES) r = *reinterpret_cast<const sf32x4 *>(z + ((a * TN + b) * 4 + g) * 1024);
                    sf32x4 t;
#pragma unroll
                    for (int u = 0; u < 4; ++u) t[u] = fmaxf((acc[a][b][4 * g + u] + bias[b][g][u]) + r[u], lo);
                    float *dst = m < p.M ? orow + 32 * b + 8 * g : trash;
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(t) : "memory");
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// ====================================================================== host side
template <int TM, int TN, int MW, int NW, int NP, int NSLOT, bool HAS_RES, bool SPREAD = false, int NT_ = 0, bool DUAL = false, int N2 = 0>
static hipError_t launch_stream_one(ConvParams p, hipStream_t s) {
    constexpr int BM = 32 * TM * MW, BN = 32 * TN * NW, NWV = MW * NW;
    // chain without landing zones: the output tile's LDS image has its own BM x BN x 2 bytes behind the ring
    constexpr size_t lds = (size_t)NSLOT * BM * 128 + (HAS_RES ? (size_t)2 * NWV * TM * TN * 2 * 1024 : (N2 ? (size_t)BM * BN * 2 : 0));
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool configured[64] = {};
    auto kern = conv_stream_f16<TM, TN, MW, NW, NP, NSLOT, HAS_RES, SPREAD, NT_, DUAL, N2>;
    if (N2 && (p.Cout != BN || p.nx_cout != N2 || !p.nx_wgt || !p.nx_bias || !p.nx_out)) return hipErrorInvalidValue;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = p.Cout / BN;
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * NWV), lds, s, p);
    return hipGetLastError();
}

// shapes with an instantiation.  Residual-bearing (Bottleneck conv3): (K, BN) = (256, 512), (128, 512), (64, 256).  Without a residual
// (the squeezing conv1 of layer1 / layer2, resnet.py:128-130): (K, Cout) = (256, 64), (512, 128), (256, 128), (64, 64), all of Cout in one slice.
// Returns the channel-slice width BN (0: none) and the pixel-tile height through *bm.
static int stream_bn(const ConvParams &p, int *bm = nullptr) {
    int bn = 0, m = 0;
    if (p.in2) {   // conv3 + downsample of layer1.0 (64 + 64 -> 256) and layer2.0 (128 + 256 -> 512, strided second source)
        if (!p.res && p.Kpad == 128 && p.ksplit == 64 && p.Cout == 256) { bn = 256; m = 128; }
        else if (!p.res && p.Kpad == 384 && p.ksplit == 128 && p.Cout == 512) { bn = 256; m = 64; }
    } else if (p.res) {
        if (p.Kpad == 256 || p.Kpad == 128) { bn = 512; m = 64; }
        else if (p.Kpad == 64) { bn = 256; m = 128; }
    } else {
        if (p.Kpad == 256 && p.Cout == 64) { bn = 64; m = 256; }
        else if (p.Kpad == 512 && p.Cout == 128) { bn = 128; m = 128; }
        else if (p.Kpad == 256 && p.Cout == 128) { bn = 128; m = 128; }
        else if (p.Kpad == 64 && p.Cout == 64) { bn = 64; m = 256; }   // layer1.0's conv1 (resnet.py:128-130 on the pooled stem output)
    }
    if (bm) *bm = m;
    return bn;
}

// -1: the launcher's rule (HMV_NO_STREAM=1 in the environment disables the kernel for A/B runs); 0: never; 1: whenever the shape
// has an instantiation, whatever the pixel count (op-level tests: hmv_op_conv2d_f16)
static int g_stream_mode = -1;
void conv_stream_set_mode(int mode) { g_stream_mode = mode; }

// fp32: residual-bearing 1x1 convs with K = 64 / 128 / 256 and Cout a multiple of 256 (64 x 256 tiles, one 32-channel block per wave)
static int stream32_shape(const ConvParams &p) {   // 0 none; 1 residual-bearing (64 x 256 tiles); 2 the squeezing conv1 256 -> 64 (128 x 64 tiles);
    if (p.in_f16 || p.out_f16) return 0;            // 3 conv3 + downsample of layer1.0 (64 + 64 -> 256) as one GEMM over two sources
    if (p.in2) return (!p.res && p.Kpad == 128 && p.ksplit == 64 && p.Cout == 256 && p.lda == p.ksplit && p.lda2 % 4 == 0) ? 3 : 0;
    if (p.res) return (!p.res_f16 && (p.Kpad == 64 || p.Kpad == 128 || p.Kpad == 256) && p.Cout % 256 == 0 && 32 % (p.Cout / 256) == 0) ? 1 : 0;
    return (p.Kpad == 256 && p.Cout == 64) ? 2 : ((p.Kpad == 64 && p.Cout == 64) ? 4 : 0);   // 4: layer1.0's conv1 64 -> 64
}
// ... all of which the launcher takes.  K = 64 and 128 (layer1 / layer2 conv3) are HBM-bound: 566 -> 489 us and 377 -> 359 us.  K = 256
// (layer3 conv3) is MFMA-bound: as eight waves that reach their epilogue together it is 2.5 % SLOWER than conv_igemm's two paired
// 256 x 128 workgroups per CU (1 172 vs 1 143 us); as two four-wave workgroups per CU on 64 x 128 tiles (HALF) the epilogues overlap the
// other workgroup's MFMAs again and it is 1.8 % faster (1 125 us).
static bool stream32_rule(const ConvParams &p) {
    static const bool no256 = HMV_DEV_ENV("HMV_NO_STREAM32_K256") != nullptr;   // development knob (A/B runs)
    return p.res ? (p.Kpad <= 128 || (!no256 && p.Cout % 128 == 0 && 64 % (p.Cout / 128) == 0)) : true;
}

bool conv_stream_supported(const ConvParams &p) {
    static int off = -1;   // development knob: HMV_NO_STREAM=1 keeps every conv on conv_igemm (A/B runs)
    if (off < 0) off = HMV_DEV_ENV("HMV_NO_STREAM") ? 1 : 0;
    static int min_env = -1;   // tiles per workgroup below which the per-tile workgroups of conv_igemm fill the chip better
    if (min_env < 0) { const char *e = HMV_DEV_ENV("HMV_STREAM_MIN_TILES"); min_env = e ? atoi(e) : 4; }
    if (g_stream_mode == 0 || (g_stream_mode < 0 && off)) return false;
    const int min_tiles = g_stream_mode > 0 ? 0 : min_env;
    if (stream32_shape(p)) {
        static const bool off32 = HMV_DEV_ENV("HMV_NO_STREAM32") != nullptr;   // development knob (A/B runs)
        if (off32 && g_stream_mode <= 0) return false;
        if (p.R != 1 || p.S != 1 || p.stride != 1 || p.pad_h || p.pad_w || p.up || p.ksl > 1 || p.phases > 1) return false;
        if (p.cwrap || p.x3_plane || p.res_split || p.out_split || p.acc_shift || p.rd_cout || p.scatter || p.rg_out || p.fill) return false;
        if (p.act != ACT_NONE && p.act != ACT_RELU) return false;
        if ((!p.in2 && p.Cin != p.Kpad) || p.K != p.Kpad) return false;
        if ((p.lda ? p.lda : p.Cin) % 4 || (p.ldw ? p.ldw : p.Kpad) % 4 || p.ldc % 4 || (p.res && p.ldr % 4)) return false;
        if (g_stream_mode > 0) return true;
        const int bm32 = (p.res || p.in2) ? 64 : 128, streams32 = p.res ? 256 / (p.Cout / 256) : 256;
        return stream32_rule(p) && (long long)(p.M + bm32 - 1) / bm32 >= (long long)min_tiles * streams32;
    }
    int bm = 0;
    const int bn = stream_bn(p, &bm);
    if (!bn || !p.in_f16 || !p.out_f16 || (p.res && !p.res_f16)) return false;
    if (p.R != 1 || p.S != 1 || p.stride != 1 || p.pad_h || p.pad_w || p.up || p.ksl > 1 || p.phases > 1) return false;
    if (p.in2 && ((p.lda2 & 7) || (p.ksplit & 63) || p.lda != p.ksplit)) return false;
    if (p.cwrap || p.x3_plane || p.res_split || p.out_split || p.acc_shift || p.rd_cout || p.scatter || p.rg_out || p.fill) return false;
    if (p.act != ACT_NONE && p.act != ACT_RELU) return false;
    if ((!p.in2 && p.Cin != p.Kpad) || p.K != p.Kpad || p.Cout % bn != 0 || 32 % (p.Cout / bn) != 0) return false;
    if ((p.lda ? p.lda : p.Cin) % 8 || (p.ldw ? p.ldw : p.Kpad) % 8 || p.ldc % 8 || (p.res && p.ldr % 8)) return false;
    const int streams = 256 / (p.Cout / bn);
    return (long long)(p.M + bm - 1) / bm >= (long long)min_tiles * streams;
}

// chain (ConvParams::nx_*): the fp16 launches whose workgroup owns all output channels of its pixel tile -- Bottleneck conv3 of layer1
// (64 -> 256 + residual; conv3 + downsample of layer1.0 as two sources) -- followed by a 1x1 conv 256 -> 64 / 128 + ReLU
bool conv_stream_chain_ok(const ConvParams &q, int nx_cout) {
    static const bool off = HMV_DEV_ENV("HMV_NO_CHAIN") != nullptr;   // development knob (A/B runs)
    if (off || (nx_cout != 64 && nx_cout != 128)) return false;
    ConvParams p = q;
    p.nx_wgt = nullptr;
    if (!p.lda) p.lda = p.Cin;
    if (!p.ldw) p.ldw = p.Kpad;
    if (!p.in_f16 || !p.out_f16 || p.acc_shift || p.Cout != 256 || (p.ldc & 7)) return false;
    if (!(p.in2 ? (!p.res && p.Kpad == 128 && p.ksplit == 64 && nx_cout == 64) : (p.res && p.Kpad == 64))) return false;
    return conv_stream_supported(p);
}

template <int TM, int TN, int MW, int NW, int NP, int NSLOT, bool HAS_RES, bool DUAL = false, bool HALF = false>
static hipError_t launch_stream32(ConvParams p, hipStream_t s) {
    constexpr int BM = 32 * TM * MW, BN = 32 * TN * NW;
    constexpr size_t lds = (size_t)NSLOT * BM * 128 + (HAS_RES ? (size_t)(HALF ? 1 : 2) * MW * NW * TM * TN * 4 * 1024 : 0);
    static_assert(lds * (HALF ? 2 : 1) <= 160 * 1024, "LDS budget");
    static bool configured[64] = {};
    auto kern = conv_stream_f32<TM, TN, MW, NW, NP, NSLOT, HAS_RES, DUAL, HALF>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = p.Cout / BN;
    hipLaunchKernelGGL(kern, dim3(HALF ? 512 : 256), dim3(64 * MW * NW), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_stream(const ConvParams &p, hipStream_t s, const char **name) {
    if (const int k32 = stream32_shape(p)) {
        if (k32 == 2) { if (name) *name = "conv_stream_f32<128x64,k256>"; return launch_stream32<1, 1, 4, 2, 8, 8, false>(p, s); }
        if (k32 == 4) { if (name) *name = "conv_stream_f32<128x64,k64>"; return launch_stream32<1, 1, 4, 2, 2, 8, false>(p, s); }
        if (k32 == 3) { if (name) *name = "conv_stream_f32<64x256,k128,dual>"; return launch_stream32<2, 1, 1, 8, 4, 8, false, true>(p, s); }
        if (p.Kpad == 256 && p.Cout % 128 == 0 && 64 % (p.Cout / 128) == 0) {   // four-wave workgroups, two per CU, 64 x 128 tiles
            if (name) *name = "conv_stream_f32<64x128,k256,res>";
            return launch_stream32<2, 1, 1, 4, 8, 4, true, false, true>(p, s);
        }
        if (p.Kpad == 256) { if (name) *name = "conv_stream_f32<64x256,k256,res>"; return launch_stream32<2, 1, 1, 8, 8, 4, true>(p, s); }
        if (p.Kpad == 128) { if (name) *name = "conv_stream_f32<64x256,k128,res>"; return launch_stream32<2, 1, 1, 8, 4, 4, true>(p, s); }
        if (name) *name = "conv_stream_f32<64x256,k64,res>";
        return launch_stream32<2, 1, 1, 8, 2, 4, true>(p, s);
    }
    // development knob (A/B runs): HMV_STREAM_VARIANT = 0 residual DMAs at step 0, 1 spread over the piece steps, 2 / 3 the same with
    // non-temporal residual loads, 4 non-temporal pixel pieces too.  Measured (profiles/r03_probe_stream_variants.txt): spreading
    // gains 3-4 % at K = 256 (four piece steps) and loses 4 % at K = 128; non-temporal loads lose 15-25 % everywhere (the next
    // launch finds less of its input in the Infinity Cache).  Default: spread at K = 256 only.
    static int variant = -1;
    if (variant < 0) { const char *e = HMV_DEV_ENV("HMV_STREAM_VARIANT"); variant = e ? atoi(e) : -1; if (variant < 0) variant = 100; }
#define HMV_STREAM_VARIANTS(...)                                                                            \
    switch (variant) {                                                                                      \
        case 1: return launch_stream_one<__VA_ARGS__, true, true, 0>(p, s);                                 \
        case 2: return launch_stream_one<__VA_ARGS__, true, false, 1>(p, s);                                \
        case 3: return launch_stream_one<__VA_ARGS__, true, true, 1>(p, s);                                 \
        case 4: return launch_stream_one<__VA_ARGS__, true, true, 2>(p, s);                                 \
        case 100: if (p.Kpad == 256) return launch_stream_one<__VA_ARGS__, true, true, 0>(p, s);            \
                  return launch_stream_one<__VA_ARGS__, true, false, 0>(p, s);                              \
        default: return launch_stream_one<__VA_ARGS__, true, false, 0>(p, s);                               \
    }
    if (p.nx_wgt) {   // chained launches (conv_stream_chain_ok): pixel ring of four pieces + the output tile's image
        if (p.nx_ldw < p.Cout || (p.nx_ldw & 7) || (p.nx_ldc & 7) || (p.nx_act != ACT_NONE && p.nx_act != ACT_RELU) || p.acc_shift) return hipErrorInvalidValue;
        if (p.in2 && p.Kpad == 128 && p.nx_cout == 64) {
            if (name) *name = "conv_stream_f16<128x256,k128,dual,+1x1:64>";
            return launch_stream_one<2, 2, 2, 4, 2, 4, false, false, 0, true, 64>(p, s);
        }
        if (p.res && p.Kpad == 64 && p.nx_cout == 64) {
            if (name) *name = "conv_stream_f16<128x256,k64,res,+1x1:64>";
            return launch_stream_one<2, 2, 2, 4, 1, 2, true, false, 0, false, 64>(p, s);
        }
        if (p.res && p.Kpad == 64 && p.nx_cout == 128) {
            if (name) *name = "conv_stream_f16<128x256,k64,res,+1x1:128>";
            return launch_stream_one<2, 2, 2, 4, 1, 2, true, false, 0, false, 128>(p, s);
        }
        return hipErrorInvalidValue;
    }
    if (p.in2) {
        if (p.Kpad == 128) {
            if (name) *name = "conv_stream_f16<128x256,k128,dual>";
            return launch_stream_one<2, 2, 2, 4, 2, 8, false, false, 0, true>(p, s);
        }
        if (name) *name = "conv_stream_f16<64x256,k384,dual>";
        return launch_stream_one<2, 1, 1, 8, 6, 8, false, false, 0, true>(p, s);
    }
    if (!p.res) {   // the squeezing 1x1 convs: no landing zones, the whole LDS is the pixel ring (128 KB, seven pieces ahead)
        if (p.Kpad == 256 && p.Cout == 64) {
            if (name) *name = "conv_stream_f16<256x64,k256>";
            return launch_stream_one<1, 2, 8, 1, 4, 4, false>(p, s);
        }
        if (p.Kpad == 64 && p.Cout == 64) {
            if (name) *name = "conv_stream_f16<256x64,k64>";
            return launch_stream_one<1, 2, 8, 1, 1, 4, false>(p, s);
        }
        if (p.Kpad == 512) {
            if (name) *name = "conv_stream_f16<128x128,k512>";
            return launch_stream_one<2, 1, 2, 4, 8, 8, false>(p, s);
        }
        if (name) *name = "conv_stream_f16<128x128,k256>";
        return launch_stream_one<1, 2, 4, 2, 4, 8, false>(p, s);
    }
    if (p.Kpad == 256) {
        if (name) *name = "conv_stream_f16<64x512,k256,res>";
        HMV_STREAM_VARIANTS(2, 2, 1, 8, 4, 4)
    }
    if (p.Kpad == 128) {
        if (name) *name = "conv_stream_f16<64x512,k128,res>";
        HMV_STREAM_VARIANTS(2, 2, 1, 8, 2, 4)
    }
    if (p.Kpad == 64) {
        if (name) *name = "conv_stream_f16<128x256,k64,res>";
        HMV_STREAM_VARIANTS(2, 2, 2, 4, 1, 2)
    }
#undef HMV_STREAM_VARIANTS
    return hipErrorInvalidValue;
}

}  // namespace hmv
