// fusion_kernels.hip -- the launch-bound tail of the forward as a few fused kernels (fused FeedForward, K9).
//
//   ff_block_kernel      everything of a fusion block behind its `to_out` GEMM in ONE launch per 16-token tile:
//                          t  = sum of the split-K partial products + bias + residual           (layers.py:224-228, `out + _q`)
//                          n1 = LayerNorm1(t)                                                    (layers.py:229; absent in the
//                                                                                                 learnable-query blocks)
//                          f0 = LayerNorm_ff(n1) ; h = GELU(f0 W1^T + b1) ; f2 = h W2^T + b2 + n1   (FeedForward, layers.py:161-174)
//                          out = LayerNorm2(f2)                                                  (layers.py:232-233; absent ...)
//                        before: split-K reduction + LayerNorm, ff1 GEMM, ff2 GEMM, LayerNorm = 4 launches whose GEMMs are one
//                        wave's dependent MFMA chain long (7 us for K = 544 whatever the grid).
//   cheb_layer1_kernel   JointsDecoderGCN layer 1 (nets.py:133-135, layers.py:387-403): X W_k for the three Chebyshev orders AND
//                        the T_k mix + bias + LeakyReLU, per (sample, 16 output channels): the [B*21][768] product never exists
//   cheb_tail_kernel     layers 2 and 3 (256 -> 64 -> 3) per sample in one launch
//
// All three multiply on v_mfma_f32_16x16x4_f32 (exact fp32; 32 cycles): a 16-row tile needs no padding of the 21-joint /
// 16-token blocks to 32 rows, and a wave's dependent chain is K/4 x 32 cycles -- 1.8 us for K = 544 -- with eight waves
// owning eight column blocks.  Operands: the activation tile sits in LDS (row stride + 4 floats), the weights stream from
// L2 as 16-byte vectors (lane (n, g) fetches W[n][16 s + 4 g .. + 3] and feeds the four MFMAs of step s; the k order inside
// a 16-wide step is permuted identically for both operands).  Row-wise arithmetic only: nothing depends on the batch.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace hmv {

typedef float ff32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ff16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fwave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

constexpr int FF_WAVES = 8;          // a workgroup takes 16 RB token rows, RB = 1 | 2 (round 4: 32-row tiles when 16-row ones would need
                                     // more than one round of workgroups: half the weight bytes per token, one round at cfg-3's 5 376 rows)
// (the kernel runs ONCE per wave: its straight-line code is kept short -- 48 KB of unrolled code cost more in cold instruction
// fetches than the arithmetic it held)

// One 16 x 16 output block = A[16][K] (LDS, row stride lda) . W[16 rows of n][K]^T (global, row stride ldw), K = 16 nv.
// The weight vectors of a block do not depend on the activations, so they are requested in ONE burst (wload) as early as the
// registers allow -- ahead of the LayerNorm phase for the first GEMM -- and the block is then 4 nv back-to-back MFMAs (wmma): a
// launch-bound kernel must not pay an L2 round trip per 64 columns.  Two accumulators alternate (issue-bound, not latency-bound).
template <int NV>
__device__ __forceinline__ void wload(ff32x4 (&b)[NV], const float *w, int ldw, int nv, int lane, int first = 0, int last = NV) {
    const float *wp = w + (size_t)(lane & 15) * ldw + 4 * (lane >> 4);
#pragma unroll
    for (int s = 0; s < NV; ++s)
        if (s >= first && s < last && s < nv) b[s] = *reinterpret_cast<const ff32x4 *>(wp + 16 * s);
}
template <int NV>
__device__ __forceinline__ ff32x4 wmma(const float *sa, int lda, const ff32x4 (&b)[NV], int nv, int lane) {
    const float *ap = sa + (lane & 15) * lda + 4 * (lane >> 4);
    ff32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NV; ++s)
        if (s < nv) {
            const ff32x4 a = *reinterpret_cast<const ff32x4 *>(ap + 16 * s);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[s][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[s][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[s][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[s][3], acc1, 0, 0, 0);
        }
    return acc0 + acc1;   // lane holds D[4 g + e][r], e = 0..3
}
constexpr int FF_NV1 = 36;   // ld <= 576: the 16-wide steps of the first GEMM's reduction
constexpr int FF_W1A = 8;    // ... of which this many are requested at kernel entry, the rest behind the first row loads of phase 0
constexpr int FF_NV2 = 16;   // hid <= 256

// A lane owns the 16-byte column groups c = 4 lane + 256 i (+ 0..3), i < FF_VEC: rows move as 16-byte vectors (a third of the
// load instructions of a column-per-lane layout: the kernel's phase 0 is bound by how many loads it has to issue and retire).
constexpr int FF_VEC = 3;            // ld <= 768
// LayerNorm of the wave's TWO rows (zeros past d), the arithmetic of layernorm_kernel (misc_kernels.hip); gamma / beta arrive in
// registers (requested at kernel entry: a load behind the reductions would add a round trip per LayerNorm), the two rows'
// reductions interleave
__device__ __forceinline__ void ln_rows2(ff32x4 (&v)[2][FF_VEC], int d, int lane, const float *sg, const float *sb) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < FF_VEC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0 += v[0][i][e]; s1 += v[1][i][e]; }
    const float inv_d = 1.f / (float)d;
    const float m0 = fwave_sum(s0) * inv_d, m1 = fwave_sum(s1) * inv_d;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < FF_VEC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = (4 * lane + 256 * i + e) < d;
            const float t0 = in ? v[0][i][e] - m0 : 0.f, t1 = in ? v[1][i][e] - m1 : 0.f;
            q0 += t0 * t0;
            q1 += t1 * t1;
        }
    const float r0 = 1.f / sqrtf(fwave_sum(q0) * inv_d + 1e-5f), r1 = 1.f / sqrtf(fwave_sum(q1) * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < FF_VEC; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c >= d) continue;
        // gamma / beta from the workgroup's LDS copy (staged up to ld >= round4(d), zeros past d)
        const ff32x4 g = *reinterpret_cast<const ff32x4 *>(sg + c), b = *reinterpret_cast<const ff32x4 *>(sb + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = (c + e) < d;
            v[0][i][e] = in ? (v[0][i][e] - m0) * r0 * g[e] + b[e] : 0.f;
            v[1][i][e] = in ? (v[1][i][e] - m1) * r1 * g[e] + b[e] : 0.f;
        }
    }
}
// barrier for LDS hand-offs only: __syncthreads() also drains the vector-memory queue, i.e. it would wait for the weight vectors
// that are deliberately left in flight across it
#define FF_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// HID = hidden width of the FeedForward: 128 (fusion.py:7-30) or 256 (the learnable-query blocks); RB = 16-row blocks per workgroup.
// Row-wise arithmetic, the same MFMA sequence per row whatever RB: a row's bits do not depend on the tile height.
template <int HID, int RB>
__global__ __launch_bounds__(64 * FF_WAVES) void ff_block_kernel(const FfBlockParams p) {
    constexpr int FF_ROWS = 16 * RB;         // token rows per workgroup: a wave owns 2 RB of them in the row phases
    constexpr int NV2 = HID / 16;            // 16-wide reduction steps of the second GEMM
    constexpr int NB2 = 5;                   // column blocks of the second GEMM per wave: ceil(36 / 8)
    constexpr bool ALL2 = HID == 128;        // 128-wide hidden layer: all of a wave's W2 vectors fit the registers W1 vacates
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const int LD = p.ld + 4, LH = HID + 4;
    float *sN = fsm;                   // [FF_ROWS][LD]  n1 (the FeedForward's residual)
    float *sF = sN + FF_ROWS * LD;     // [FF_ROWS][LD]  f0, later f2
    float *sH = sF + FF_ROWS * LD;     // [FF_ROWS][LH]  hidden activations (phases 1-2); before them: LayerNorm1's and the FeedForward
                                       // LayerNorm's gamma / beta, [4][ld] (phase 0 only -- the 32-row tile has no LDS to spare)
    const int hreg = FF_ROWS * LH > 4 * p.ld ? FF_ROWS * LH : 4 * p.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * FF_ROWS;
    unsigned long long stamp[6] = {};
    if (p.dbg) stamp[0] = __builtin_amdgcn_s_memrealtime();
    // nothing below depends on the activations: requested first, in flight while the rows are assembled and normalised
    ff32x4 wv1[FF_NV1];
    const int nv1 = p.ld >> 4, nblk2 = p.ld >> 4;
    // (round 4: only the first FF_W1A steps here -- vector-memory returns are in order per wave, and the rows' own loads of phase 0 would
    // otherwise wait behind all 35 KB of this wave's weight vectors; the rest goes out once the first row pair's loads are on their way)
    wload(wv1, p.w1 + (size_t)wave * 16 * p.ldw1, p.ldw1, nv1, lane, 0, FF_W1A);
    // the three LayerNorms' gamma / beta, staged once per workgroup: [4][ld] over the sH region, LayerNorm2's [2][ld] behind it
    float *sP = sH, *sP2 = sH + hreg;
    {
        const float *src[6] = {p.n1g, p.n1b, p.fg, p.fb, p.n2g, p.n2b};
        float t[6][2];   // ld <= 768 < 2 x 512 threads: every load is issued before the first store
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = tid + 64 * FF_WAVES * u;
                t[a][u] = (src[a] && c < p.d) ? src[a][c] : 0.f;
            }
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = tid + 64 * FF_WAVES * u;
                if (c < p.ld) (a < 4 ? sP + a * p.ld : sP2 + (a - 4) * p.ld)[c] = t[a][u];
            }
    }
    FF_BARRIER();

    // ---- phase 0: assemble the pre-norm rows, LayerNorm1, the FeedForward's LayerNorm (wave w: rows 2 RB w .. + 2 RB - 1, two at a time)
#pragma unroll
    for (int rp = 0; rp < RB; ++rp) {
        const int lr0 = 2 * RB * wave + 2 * rp;   // first of the pair's two tile rows
        ff32x4 v[2][FF_VEC];
        const int dv = (p.d + 3) & ~3;   // rows are readable up to round4(d): every row stride is a multiple of 4
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = min(row0 + lr0 + r, p.rows - 1);
            const float *res = nullptr;
            if (p.res) res = p.res + (size_t)(p.rg_out ? (row / p.rg_out) * p.rg_in + (row % p.rg_out) : row) * p.ldr;
#pragma unroll
            for (int i = 0; i < FF_VEC; ++i) {
                const int c = 4 * lane + 256 * i;
                v[r][i] = ff32x4{0.f, 0.f, 0.f, 0.f};
                if (256 * i >= p.d) continue;   // wave-uniform
                const int cc = c < dv ? c : 0;
                ff32x4 t;
                if (p.slab) {   // slices added in index order, then bias, then the residual: the arithmetic of splitk_reduce_kernel
                    // (every load is issued before anything is added: a runtime-bounded loop here costs S dependent round trips)
                    ff32x4 part[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) part[k] = *reinterpret_cast<const ff32x4 *>(p.slab + (k < p.S ? k : 0) * p.slice + (size_t)row * p.lds + cc);
                    const ff32x4 bb = *reinterpret_cast<const ff32x4 *>(p.bias0 + cc);
                    ff32x4 rv = {0.f, 0.f, 0.f, 0.f};
                    if (res) rv = *reinterpret_cast<const ff32x4 *>(res + cc);
                    t = ff32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < p.S) t += part[k];
                    t += bb;
                    if (res) t += rv;
                } else {
                    t = *reinterpret_cast<const ff32x4 *>(p.x + (size_t)row * p.ldx + cc);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[r][i][e] = (c + e) < p.d ? t[e] : 0.f;
            }
        }
        if (rp == 0) wload(wv1, p.w1 + (size_t)wave * 16 * p.ldw1, p.ldw1, nv1, lane, FF_W1A, FF_NV1);   // (behind the first row pair's loads)
        if (p.n1g) ln_rows2(v, p.d, lane, sP, sP + p.ld);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < FF_VEC; ++i) {
                const int c = 4 * lane + 256 * i;
                if (c < p.ld) *reinterpret_cast<ff32x4 *>(&sN[(lr0 + r) * LD + c]) = v[r][i];
            }
        ln_rows2(v, p.d, lane, sP + 2 * p.ld, sP + 3 * p.ld);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < FF_VEC; ++i) {
                const int c = 4 * lane + 256 * i;
                if (c < p.ld) *reinterpret_cast<ff32x4 *>(&sF[(lr0 + r) * LD + c]) = v[r][i];
            }
    }
    FF_BARRIER();
    if (p.dbg) stamp[1] = __builtin_amdgcn_s_memrealtime();

    // ---- phase 1: h = GELU(f0 W1^T + b1), 16-column blocks dealt to the waves
    const int r16 = lane & 15, g4 = lane >> 4;
    ff32x4 w2all[ALL2 ? NB2 : 1][NV2];
#pragma unroll
    for (int u = 0; u < HID / 16 / FF_WAVES; ++u) {
        const int nb = wave + FF_WAVES * u;
        if (u > 0) wload(wv1, p.w1 + (size_t)nb * 16 * p.ldw1, p.ldw1, nv1, lane);   // (256-wide hidden layer: a second block)
        const float bb = p.b1[nb * 16 + r16];
        ff32x4 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = wmma(sF + rb * 16 * LD, LD, wv1, nv1, lane);   // the same weight registers for every row block
        if constexpr (ALL2) {   // the second GEMM's weights, all of this wave's blocks, go out behind the last MFMA of the first
#pragma unroll
            for (int v2 = 0; v2 < NB2; ++v2)
                if (wave + FF_WAVES * v2 < nblk2) wload(w2all[v2], p.w2 + (size_t)(wave + FF_WAVES * v2) * 16 * p.ldw2, p.ldw2, NV2, lane);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int e = 0; e < 4; ++e) sH[(16 * rb + 4 * g4 + e) * LH + nb * 16 + r16] = gelu_erf(acc[rb][e] + bb);
    }
    FF_BARRIER();
    if (p.dbg) stamp[2] = __builtin_amdgcn_s_memrealtime();

    // ---- phase 2: f2 = h W2^T + b2 + n1 (into sF; columns past d: zero weights, zero bias, zero n1)
    auto finish2 = [&](int nbx, int rb, const ff32x4 &acc) {
        const int col = nbx * 16 + r16;
        const float bb = p.b2[col];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int lr = 16 * rb + 4 * g4 + e;
            sF[lr * LD + col] = acc[e] + bb + sN[lr * LD + col];
        }
    };
    if constexpr (ALL2) {
#pragma unroll
        for (int v2 = 0; v2 < NB2; ++v2) {
            const int nbx = wave + FF_WAVES * v2;
            if (nbx < nblk2) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) finish2(nbx, rb, wmma(sH + rb * 16 * LH, LH, w2all[v2], NV2, lane));
            }
        }
    } else {
        // this wave's column blocks two at a time: both blocks' weights are requested before either is multiplied
        for (int nb = wave; nb < nblk2; nb += 2 * FF_WAVES) {
            ff32x4 wa[NV2], wb[NV2];
            const int nb2 = nb + FF_WAVES;
            wload(wa, p.w2 + (size_t)nb * 16 * p.ldw2, p.ldw2, NV2, lane);
            if (nb2 < nblk2) wload(wb, p.w2 + (size_t)nb2 * 16 * p.ldw2, p.ldw2, NV2, lane);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                finish2(nb, rb, wmma(sH + rb * 16 * LH, LH, wa, NV2, lane));
                if (nb2 < nblk2) finish2(nb2, rb, wmma(sH + rb * 16 * LH, LH, wb, NV2, lane));
            }
        }
    }
    FF_BARRIER();
    if (p.dbg) stamp[3] = __builtin_amdgcn_s_memrealtime();

    // ---- phase 3: LayerNorm2 (or none) and the output rows, pad columns written as zeros
#pragma unroll
    for (int rp = 0; rp < RB; ++rp) {
        const int lr0 = 2 * RB * wave + 2 * rp;
        ff32x4 v[2][FF_VEC];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < FF_VEC; ++i) {
                const int c = 4 * lane + 256 * i;
                v[r][i] = ff32x4{0.f, 0.f, 0.f, 0.f};
                if (c < p.ld) v[r][i] = *reinterpret_cast<const ff32x4 *>(&sF[(lr0 + r) * LD + c]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[r][i][e] = (c + e) < p.d ? v[r][i][e] : 0.f;
            }
        if (p.n2g) ln_rows2(v, p.d, lane, sP2, sP2 + p.ld);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = row0 + lr0 + r;
#pragma unroll
            for (int i = 0; i < FF_VEC; ++i) {
                const int c = 4 * lane + 256 * i;
                if (row < p.rows && c < p.ldo) {
                    *reinterpret_cast<ff32x4 *>(p.out + (size_t)row * p.ldo + c) = v[r][i];
                    if (p.out_pairs) {   // hi = fp16(clamp(v)), lo = fp16(clamp(v) - hi): split_f16 of misc_kernels.hip
                        ff16x4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float cv = fminf(fmaxf(v[r][i][e], -65504.f), 65504.f);
                            hi[e] = (_Float16)cv;
                            lo[e] = (_Float16)(cv - (float)hi[e]);
                        }
                        _Float16 *pr = reinterpret_cast<_Float16 *>(p.out_pairs) + (size_t)row * 2 * p.ldo + c;
                        *reinterpret_cast<ff16x4 *>(pr) = hi;
                        *reinterpret_cast<ff16x4 *>(pr + p.ldo) = lo;
                    }
                }
            }
        }
    }
    if (p.dbg) {
        stamp[4] = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) for (int i = 0; i < 5; ++i) p.dbg[(size_t)blockIdx.x * 8 + i] = stamp[i];
    }
}

hipError_t launch_ff_block(const FfBlockParams &p, hipStream_t s) {
    if (p.rows <= 0) return hipSuccess;
    if (p.d > 256 * FF_VEC || p.ld > 256 * FF_VEC || p.ldo != p.ld || p.ld % 16 || p.ld < p.d || (p.lds & 3) || (p.ldr & 3) || (p.ldx & 3) ||
        p.ld > 16 * FF_NV1 || (p.hid != 128 && p.hid != 256) || p.ldw1 < p.ld || p.ldw2 < p.hid || (p.ldw1 & 3) || (p.ldw2 & 3) || (!p.slab && !p.x) ||
        (p.slab && (p.S < 1 || p.S > 4 || !p.bias0)))
        return hipErrorInvalidValue;
    auto lds_of = [&](int rb) {
        const size_t rows = 16 * rb, hreg = rows * (p.hid + 4) > (size_t)4 * p.ld ? rows * (p.hid + 4) : (size_t)4 * p.ld;
        return ((size_t)2 * rows * (p.ld + 4) + hreg + (size_t)2 * p.ld) * sizeof(float);
    };
    // 32-row tiles once 16-row ones would not fit one round of workgroups (one per CU: 90 KB of LDS, 240 registers), if they fit LDS
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    static int ncu[64] = {};
    if (!ncu[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        ncu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int rb = ((p.rows + 15) / 16 > ncu[dev] && lds_of(2) <= 160 * 1024) ? 2 : 1;
    const size_t lds = lds_of(rb);
    static bool configured[64] = {};
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ff_block_kernel<128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(ff_block_kernel<256, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(ff_block_kernel<128, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(ff_block_kernel<256, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    const dim3 grid((p.rows + 16 * rb - 1) / (16 * rb)), block(64 * FF_WAVES);
    if (p.hid == 128) {
        if (rb == 2) hipLaunchKernelGGL((ff_block_kernel<128, 2>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((ff_block_kernel<128, 1>), grid, block, lds, s, p);
    } else {
        if (rb == 2) hipLaunchKernelGGL((ff_block_kernel<256, 2>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((ff_block_kernel<256, 1>), grid, block, lds, s, p);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ JointsDecoderGCN, layer 1
// out[b][i][o] = leaky( sum_k sum_j T_k[i][j] * (X_b W_k)[j][o] + bias[o] ),  X_b = the sample's 21 token rows [21][K],
// W packed as ONE [3 * co][K] matrix (row k * co + o), the layout the unfused GEMM uses.  One workgroup per (sample, 16 output
// channels): 6 waves = 2 row blocks (joints 0-15, 16-20 + padding) x 3 Chebyshev orders, each one 16 x 16 MFMA block.
constexpr int CH_LDX_PAD = 4;
__global__ __launch_bounds__(384) void cheb_layer1_kernel(const float *__restrict__ x, int ldx, int K, const float *__restrict__ w, int ldw,
                                                          int co, const float *__restrict__ tk, const float *__restrict__ bias, int leaky,
                                                          float *__restrict__ out, int ldo) {
    extern __shared__ __attribute__((aligned(16))) float csm[];
    const int LD = K + CH_LDX_PAD;
    float *sX = csm;                 // [32][LD], rows 21..31 zero
    float *sY = sX + 32 * LD;        // [3][32][16]
    float *sT = sY + 3 * 32 * 16;    // [3][21][21]
    const int b = blockIdx.x, o0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = wave & 1, kord = wave >> 1;   // row block, Chebyshev order of this wave's 16 x 16 MFMA block
    ff32x4 wv[FF_NV1];
    wload(wv, w + (size_t)(kord * co + o0) * ldw, ldw, K >> 4, lane);   // in flight while the token rows are staged
    for (int i = tid; i < 3 * 21 * 21; i += 384) sT[i] = tk[i];
    const int k4 = K >> 2;
    {   // the sample's 21 token rows: every load is issued before the first store (K <= 576: at most 8 vectors per thread)
        ff32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 384 * u, r = i / k4, c = i - r * k4;
            v[u] = ff32x4{0.f, 0.f, 0.f, 0.f};
            if (i < 21 * k4) v[u] = *reinterpret_cast<const ff32x4 *>(x + ((size_t)b * 21 + r) * ldx + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 384 * u, r = i / k4, c = i - r * k4;
            if (i < 21 * k4) *reinterpret_cast<ff32x4 *>(sX + r * LD + 4 * c) = v[u];
        }
        for (int i = 21 * k4 + tid; i < 32 * k4; i += 384) {   // padding rows 21 .. 31
            const int r = i / k4, c = i - r * k4;
            *reinterpret_cast<ff32x4 *>(sX + r * LD + 4 * c) = ff32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    {
        const ff32x4 acc = wmma(sX + rb * 16 * LD, LD, wv, K >> 4, lane);
        const int r16 = lane & 15, g4 = lane >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) sY[(kord * 32 + rb * 16 + 4 * g4 + e) * 16 + r16] = acc[e];
    }
    __syncthreads();
    if (tid < 21 * 16) {
        const int i = tid >> 4, o = tid & 15;
        if (o0 + o < co) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float part = 0.f;
#pragma unroll
                for (int j = 0; j < 21; ++j) part += sT[(k * 21 + i) * 21 + j] * sY[(k * 32 + j) * 16 + o];
                acc += part;
            }
            float v = acc + bias[o0 + o];
            if (leaky) v = v > 0.f ? v : 0.01f * v;
            out[((size_t)b * 21 + i) * ldo + o0 + o] = v;
        }
    }
}

// ------------------------------------------------------------------ JointsDecoderGCN, layers 2 and 3 (c1 -> c2 -> c3 = 256 -> 64 -> 3)
// One workgroup (8 waves) per sample.  Layer 2: 2 row blocks x (3 c2 / 16) column blocks of K = c1; mix + LeakyReLU into LDS;
// layer 3: 2 row blocks x 1 column block (3 * c3 = 9 <= 16 columns) of K = c2; mix -> out [B*21][3].
__global__ __launch_bounds__(512) void cheb_tail_kernel(const float *__restrict__ x, int ldx, int c1, const float *__restrict__ w2, int ldw2,
                                                        int c2, const float *__restrict__ bias2, const float *__restrict__ w3, int ldw3,
                                                        int c3, const float *__restrict__ bias3, const float *__restrict__ tk,
                                                        float *__restrict__ out, int ldo) {
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    const int LD1 = c1 + 4, LY = 3 * c2 + 4, LD2 = c2 + 4;
    float *sX = tsm;                  // [32][LD1]
    float *sY = sX + 32 * LD1;        // [32][LY]   X W2 for the three orders (columns k * c2 + o)
    float *sZ = sY + 32 * LY;         // [32][LD2]  layer-2 output (rows 21.. zero)
    float *sT = sZ + 32 * LD2;        // [3][21][21]
    float *sY3 = sT + 3 * 21 * 21;    // [32][16]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // every weight vector this wave will need is requested before anything else: blocks t = wave, wave + 8, wave + 16 of layer 2
    // (row block t & 1, column block t >> 1; c1 <= 256) and layer 3's single column block (c2 <= 64)
    const int nblk = 2 * (3 * c2 / 16);
    ff32x4 w2v[3][16], w3v[4];
#pragma unroll
    for (int u = 0; u < 3; ++u)
        if (wave + 8 * u < nblk) wload(w2v[u], w2 + (size_t)((wave + 8 * u) >> 1) * 16 * ldw2, ldw2, c1 >> 4, lane);
    if (wave < 2) wload(w3v, w3, ldw3, c2 >> 4, lane);
    for (int i = tid; i < 3 * 21 * 21; i += 512) sT[i] = tk[i];
    const int k4 = c1 >> 2;
    {   // layer-1 output rows of the sample (c1 <= 256: at most 3 vectors per thread), loads before stores
        ff32x4 v[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = tid + 512 * u, r = i / k4, c = i - r * k4;
            v[u] = ff32x4{0.f, 0.f, 0.f, 0.f};
            if (i < 21 * k4) v[u] = *reinterpret_cast<const ff32x4 *>(x + ((size_t)b * 21 + r) * ldx + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = tid + 512 * u, r = i / k4, c = i - r * k4;
            if (i < 21 * k4) *reinterpret_cast<ff32x4 *>(sX + r * LD1 + 4 * c) = v[u];
        }
        for (int i = 21 * k4 + tid; i < 32 * k4; i += 512) {
            const int r = i / k4, c = i - r * k4;
            *reinterpret_cast<ff32x4 *>(sX + r * LD1 + 4 * c) = ff32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    for (int i = tid; i < 32 * LD2; i += 512) sZ[i] = 0.f;
    __syncthreads();
    const int r16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = wave + 8 * u;
        if (t < nblk) {
            const int rb = t & 1, nb = t >> 1;
            const ff32x4 acc = wmma(sX + rb * 16 * LD1, LD1, w2v[u], c1 >> 4, lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) sY[(rb * 16 + 4 * g4 + e) * LY + nb * 16 + r16] = acc[e];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 21 * c2; idx += 512) {
        const int i = idx / c2, o = idx - i * c2;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 21; ++j) part += sT[(k * 21 + i) * 21 + j] * sY[j * LY + k * c2 + o];
            acc += part;
        }
        const float v = acc + bias2[o];
        sZ[i * LD2 + o] = v > 0.f ? v : 0.01f * v;
    }
    __syncthreads();
    if (wave < 2) {   // layer 3: 3 * c3 <= 16 columns (the packed weights are zero-padded to 16 rows and beyond)
        const ff32x4 acc = wmma(sZ + wave * 16 * LD2, LD2, w3v, c2 >> 4, lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) sY3[(wave * 16 + 4 * g4 + e) * 16 + r16] = acc[e];
    }
    __syncthreads();
    if (tid < 21 * c3) {
        const int i = tid / c3, o = tid - i * c3;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 21; ++j) part += sT[(k * 21 + i) * 21 + j] * sY3[j * 16 + k * c3 + o];
            acc += part;
        }
        out[((size_t)b * 21 + i) * ldo + o] = acc + bias3[o];
    }
}

// Host predicate: the shapes the two fused ChebConv launches take (the engine asks it BEFORE choosing the fused branch, so that a
// shape outside it -- a wider token matrix, another decoder width -- takes the launch-per-op branch instead of failing the forward).
bool cheb_fusable(int K, int ldx, int ldw1, int c1, int ldw2, int c2, int ldw3, int c3) {
    if (K > 16 * FF_NV1 || c1 > 256 || c2 > 64 || 2 * (3 * c2 / 16) > 24 || K % 16 || ldx < K || ldw1 < K || c1 % 16 || (3 * c2) % 16 || c2 % 16 || 3 * c3 > 16 ||
        ldw2 < c1 || ldw3 < c2 || (ldx & 3) || (ldw1 & 3) || (ldw2 & 3) || (ldw3 & 3))
        return false;
    const size_t lds1 = ((size_t)32 * (K + CH_LDX_PAD) + 3 * 32 * 16 + 3 * 21 * 21) * sizeof(float);
    const size_t lds2 = ((size_t)32 * (c1 + 4) + 32 * (3 * c2 + 4) + 32 * (c2 + 4) + 3 * 21 * 21 + 32 * 16) * sizeof(float);
    return lds1 <= 160 * 1024 && lds2 <= 160 * 1024;
}

hipError_t launch_cheb_fused(const ChebFusedParams &p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    if (!cheb_fusable(p.K, p.ldx, p.ldw1, p.c1, p.ldw2, p.c2, p.ldw3, p.c3) || !p.scratch) return hipErrorInvalidValue;
    const size_t lds1 = ((size_t)32 * (p.K + CH_LDX_PAD) + 3 * 32 * 16 + 3 * 21 * 21) * sizeof(float);
    const size_t lds2 = ((size_t)32 * (p.c1 + 4) + 32 * (3 * p.c2 + 4) + 32 * (p.c2 + 4) + 3 * 21 * 21 + 32 * 16) * sizeof(float);
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cheb_layer1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(cheb_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    hipLaunchKernelGGL(cheb_layer1_kernel, dim3(p.B, p.c1 / 16), dim3(384), lds1, s, p.x, p.ldx, p.K, p.w1, p.ldw1, p.c1, p.tk, p.bias1, 1,
                       p.scratch, p.c1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cheb_tail_kernel, dim3(p.B), dim3(512), lds2, s, p.scratch, p.c1, p.c1, p.w2, p.ldw2, p.c2, p.bias2, p.w3, p.ldw3, p.c3,
                       p.bias3, p.tk, p.out, p.ldo);
    return hipGetLastError();
}

}  // namespace hmv
