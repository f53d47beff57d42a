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

__device__ __forceinline__ float fwave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

constexpr int FF_ROWS = 16;          // token rows per workgroup
constexpr int FF_WAVES = 8;
constexpr int FF_MAX_PER_LANE = 16;  // d <= 1024, as launch_layernorm

// one 16 x 16 output block: A[16][K] (LDS, row stride lda) . W[16 rows of n][K]^T (global, row stride ldw), K % 16 == 0.
// The weight vectors of the NEXT 64-wide chunk are in flight while this chunk's 16 MFMAs run (an L2 round trip per chunk would
// otherwise cost more than the MFMAs); two accumulators alternate so that the chain is issue-bound, not latency-bound.
__device__ __forceinline__ ff32x4 mfma_block_16(const float *sa, int lda, const float *w, int ldw, int K, int lane) {
    const int r = lane & 15, g = lane >> 4;
    const float *ap = sa + r * lda + 4 * g;
    const float *wp = w + (size_t)r * ldw + 4 * g;
    ff32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int nfull = K >> 6;
    ff32x4 b0[4], b1[4];
    if (nfull > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) b0[u] = *reinterpret_cast<const ff32x4 *>(wp + 16 * u);
    }
    for (int c = 0; c < nfull; ++c) {
        const int s = c << 6;
        if (c + 1 < nfull) {
#pragma unroll
            for (int u = 0; u < 4; ++u) b1[u] = *reinterpret_cast<const ff32x4 *>(wp + s + 64 + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const ff32x4 a = *reinterpret_cast<const ff32x4 *>(ap + s + 16 * u);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0[u][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b0[u][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b0[u][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b0[u][3], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) b0[u] = b1[u];
    }
    for (int s = nfull << 6; s < K; s += 16) {
        const ff32x4 a = *reinterpret_cast<const ff32x4 *>(ap + s);
        const ff32x4 b = *reinterpret_cast<const ff32x4 *>(wp + s);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
    }
    return acc0 + acc1;   // lane holds D[4 g + e][r], e = 0..3
}

// LayerNorm of one row held as v[i] = x[lane + 64 i] (zeros past d), the arithmetic of layernorm_kernel (misc_kernels.hip)
__device__ __forceinline__ void ln_row(float (&v)[FF_MAX_PER_LANE], int d, int lane, const float *g, const float *b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FF_MAX_PER_LANE; ++i) s += v[i];
    const float inv_d = 1.f / (float)d, mean = fwave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < FF_MAX_PER_LANE; ++i) {
        const float t = (lane + 64 * i) < d ? v[i] - mean : 0.f;
        q += t * t;
    }
    const float rstd = 1.f / sqrtf(fwave_sum(q) * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < FF_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < d ? (v[i] - mean) * rstd * g[c] + b[c] : 0.f;
    }
}

__global__ __launch_bounds__(64 * FF_WAVES) void ff_block_kernel(const FfBlockParams p) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const int LD = p.ld + 4, LH = p.hid + 4;
    float *sN = fsm;                   // [16][LD]  n1 (the FeedForward's residual)
    float *sF = sN + FF_ROWS * LD;     // [16][LD]  f0, later f2
    float *sH = sF + FF_ROWS * LD;     // [16][LH]  hidden activations
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * FF_ROWS;

    // ---- phase 0: assemble the pre-norm rows, LayerNorm1, the FeedForward's LayerNorm (wave w: rows 2w, 2w + 1)
#pragma unroll
    for (int rr = 0; rr < FF_ROWS / FF_WAVES; ++rr) {
        const int lr = wave * (FF_ROWS / FF_WAVES) + rr;
        const int row = min(row0 + lr, p.rows - 1);
        const float *res = nullptr;
        if (p.res) res = p.res + (size_t)(p.rg_out ? (row / p.rg_out) * p.rg_in + (row % p.rg_out) : row) * p.ldr;
        float v[FF_MAX_PER_LANE];
#pragma unroll
        for (int i = 0; i < FF_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (64 * i >= p.d) { v[i] = 0.f; continue; }   // wave-uniform
            const int cc = c < p.d ? c : 0;
            float t;
            if (p.slab) {   // slices added in index order, then bias, then the residual: the arithmetic of splitk_reduce_kernel
                t = 0.f;
                for (int k = 0; k < p.S; ++k) t += p.slab[k * p.slice + (size_t)row * p.lds + cc];
                t += p.bias0[cc];
                if (res) t += res[cc];
            } else {
                t = p.x[(size_t)row * p.ldx + cc];
            }
            v[i] = c < p.d ? t : 0.f;
        }
        if (p.n1g) ln_row(v, p.d, lane, p.n1g, p.n1b);
#pragma unroll
        for (int i = 0; i < FF_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (c < p.ld) sN[lr * LD + c] = v[i];
        }
        ln_row(v, p.d, lane, p.fg, p.fb);
#pragma unroll
        for (int i = 0; i < FF_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (c < p.ld) sF[lr * LD + c] = v[i];
        }
    }
    __syncthreads();

    // ---- phase 1: h = GELU(f0 W1^T + b1), 16-column blocks dealt to the waves
    const int r16 = lane & 15, g4 = lane >> 4;
    for (int nb = wave; nb < p.hid / 16; nb += FF_WAVES) {
        const ff32x4 acc = mfma_block_16(sF, LD, p.w1 + (size_t)nb * 16 * p.ldw1, p.ldw1, p.ld, lane);
        const int col = nb * 16 + r16;
        const float bb = p.b1[col];
#pragma unroll
        for (int e = 0; e < 4; ++e) sH[(4 * g4 + e) * LH + col] = gelu_erf(acc[e] + bb);
    }
    __syncthreads();

    // ---- phase 2: f2 = h W2^T + b2 + n1 (into sF; columns past d: zero weights, zero bias, zero n1)
    for (int nb = wave; nb < p.ld / 16; nb += FF_WAVES) {
        const ff32x4 acc = mfma_block_16(sH, LH, p.w2 + (size_t)nb * 16 * p.ldw2, p.ldw2, p.hid, lane);
        const int col = nb * 16 + r16;
        const float bb = p.b2[col];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int lr = 4 * g4 + e;
            sF[lr * LD + col] = acc[e] + bb + sN[lr * LD + col];
        }
    }
    __syncthreads();

    // ---- phase 3: LayerNorm2 (or none) and the output rows, pad columns written as zeros
#pragma unroll
    for (int rr = 0; rr < FF_ROWS / FF_WAVES; ++rr) {
        const int lr = wave * (FF_ROWS / FF_WAVES) + rr, row = row0 + lr;
        if (row >= p.rows) continue;
        float v[FF_MAX_PER_LANE];
#pragma unroll
        for (int i = 0; i < FF_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < p.d ? sF[lr * LD + c] : 0.f;
        }
        if (p.n2g) ln_row(v, p.d, lane, p.n2g, p.n2b);
#pragma unroll
        for (int i = 0; i < FF_MAX_PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (c < p.ldo) p.out[(size_t)row * p.ldo + c] = v[i];
        }
    }
}

hipError_t launch_ff_block(const FfBlockParams &p, hipStream_t s) {
    if (p.rows <= 0) return hipSuccess;
    if (p.d > 64 * FF_MAX_PER_LANE || p.ld > 64 * FF_MAX_PER_LANE || p.ldo > 64 * FF_MAX_PER_LANE || p.ld % 16 || p.ld < p.d ||
        (p.hid != 128 && p.hid != 256) || p.ldw1 < p.ld || p.ldw2 < p.hid || (p.ldw1 & 3) || (p.ldw2 & 3) || (!p.slab && !p.x) ||
        (p.slab && (p.S < 1 || !p.bias0)))
        return hipErrorInvalidValue;
    const size_t lds = ((size_t)2 * FF_ROWS * (p.ld + 4) + (size_t)FF_ROWS * (p.hid + 4)) * sizeof(float);
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ff_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    hipLaunchKernelGGL(ff_block_kernel, dim3((p.rows + FF_ROWS - 1) / FF_ROWS), dim3(64 * FF_WAVES), lds, s, p);
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
    for (int i = tid; i < 3 * 21 * 21; i += 384) sT[i] = tk[i];
    const int k4 = K >> 2;
    for (int i = tid; i < 32 * k4; i += 384) {
        const int r = i / k4, c = i - r * k4;
        ff32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < 21) v = *reinterpret_cast<const ff32x4 *>(x + ((size_t)b * 21 + r) * ldx + 4 * c);
        *reinterpret_cast<ff32x4 *>(sX + r * LD + 4 * c) = v;
    }
    __syncthreads();
    {
        const int rb = wave & 1, k = wave >> 1;   // row block, Chebyshev order
        const ff32x4 acc = mfma_block_16(sX + rb * 16 * LD, LD, w + (size_t)(k * co + o0) * ldw, ldw, K, lane);
        const int r16 = lane & 15, g4 = lane >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) sY[(k * 32 + rb * 16 + 4 * g4 + e) * 16 + r16] = acc[e];
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
    for (int i = tid; i < 3 * 21 * 21; i += 512) sT[i] = tk[i];
    const int k4 = c1 >> 2;
    for (int i = tid; i < 32 * k4; i += 512) {
        const int r = i / k4, c = i - r * k4;
        ff32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < 21) v = *reinterpret_cast<const ff32x4 *>(x + ((size_t)b * 21 + r) * ldx + 4 * c);
        *reinterpret_cast<ff32x4 *>(sX + r * LD1 + 4 * c) = v;
    }
    for (int i = tid; i < 32 * LD2; i += 512) sZ[i] = 0.f;
    __syncthreads();
    const int r16 = lane & 15, g4 = lane >> 4;
    const int nblk = 2 * (3 * c2 / 16);
    for (int t = wave; t < nblk; t += 8) {
        const int rb = t & 1, nb = t >> 1;
        const ff32x4 acc = mfma_block_16(sX + rb * 16 * LD1, LD1, w2 + (size_t)nb * 16 * ldw2, ldw2, c1, lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) sY[(rb * 16 + 4 * g4 + e) * LY + nb * 16 + r16] = acc[e];
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
        const ff32x4 acc = mfma_block_16(sZ + wave * 16 * LD2, LD2, w3, ldw3, c2, lane);
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

hipError_t launch_cheb_fused(const ChebFusedParams &p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    if (p.K % 16 || p.ldx < p.K || p.ldw1 < p.K || p.c1 % 16 || (3 * p.c2) % 16 || p.c2 % 16 || 3 * p.c3 > 16 || p.ldw2 < p.c1 || p.ldw3 < p.c2 ||
        (p.ldx & 3) || (p.ldw1 & 3) || (p.ldw2 & 3) || (p.ldw3 & 3) || !p.scratch)
        return hipErrorInvalidValue;
    const size_t lds1 = ((size_t)32 * (p.K + CH_LDX_PAD) + 3 * 32 * 16 + 3 * 21 * 21) * sizeof(float);
    const size_t lds2 = ((size_t)32 * (p.c1 + 4) + 32 * (3 * p.c2 + 4) + 32 * (p.c2 + 4) + 3 * 21 * 21 + 32 * 16) * sizeof(float);
    if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return hipErrorInvalidValue;
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
