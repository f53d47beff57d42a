// hr_fuse.hip -- the up-sampling terms of an HRNet fuse layer as ONE launch per output branch (round 4).
//
// HighResolutionModule.forward (/root/reference/src/models/backbones/hrnet.py:194-212):
//     y_i = relu( sum_j f_ij(x_j) ),   f_ii = identity,   f_ij (j > i) = Upsample_{2^(j-i), nearest}( BN( Conv1x1_{C_j -> C_i}(x_j) ) )
// summed in ascending j.  The engine ran every non-identity term as its own conv launch whose epilogue added the running sum (the
// 1x1 convs on the up-sampled index map): for branch 0 of a four-branch module three launches that each read AND re-write the
// 1 M-pixel x 40-channel map (3 x 200 MB, 70 us each at ~3 TB/s) for 3 GFLOP.  The up-sampling terms are always the LAST terms of
// the sum (j > i), so they fuse behind whatever came before:
//     out = act( ((base + G_1) + G_2) + G_3 ),   G_s = W_s x_s + b_s at x_s's resolution, read at (y >> shift_s, x >> shift_s)
// with base = x_i (i = 0) or the running sum after the stride-2 terms (i >= 1): the map is read once and written once.
//
// MI355X mapping: one workgroup (four waves) per TH x 32 output pixels.
//   phase A  the source tiles ((TH >> s) x (32 >> s) pixels x C_s channels, s = shift) -> LDS in the activation's own type
//   phase B  G_s = A_s W_s^T on v_mfma_f32_16x16x4_f32 (exact fp32 products and sums in all arithmetic modes: the reduction is 80 .. 512
//            long and the launch is bound by the map's bytes, not by these 3 GFLOP).  An item = (source, 16-channel block): its weight
//            rows stream from L2 in bursts of <= 16 sixteen-wide k-steps and stay in registers for all of the source's pixel blocks
//            (fusion_kernels.hip's operand scheme: the k order inside a 16-wide step is permuted identically for both operands);
//            items are dealt to the waves by the host, longest first.  G_s + bias -> LDS (fp32)
//   phase C  every output pixel: base row + the G rows of its three coarser pixels, in the reference's order, activation, store.
// Row-wise arithmetic: a pixel's result does not depend on the tile, the grid or the batch.  fp32 mode: G_s and the sums are what the
// per-term launches computed up to the summation order inside a dot product; fp16 mode: the per-term launches rounded the running sum
// to fp16 after every term, this kernel rounds once.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace hmv {

typedef float hf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 hf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));

constexpr int HRF_WAVES = 4, HRF_TW = 32, HRF_NVB = 8;   // waves; tile width; 16-wide k-steps per weight burst

constexpr int HRF_UB = 6, HRF_UC = 6;   // 16-byte units a thread moves per batch in the load / apply phase
// 16-byte units: 4 fp32 or 8 fp16 channels of a pixel
template <typename T> struct hrf_unit;
template <> struct hrf_unit<float> {
    static constexpr int E = 4;
    hf32x4 v;
    __device__ __forceinline__ void zero() { v = hf32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct hrf_unit<_Float16> {
    static constexpr int E = 8;
    hf16x8 v;
    __device__ __forceinline__ void zero() { for (int i = 0; i < 8; ++i) v[i] = (_Float16)0; }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (_Float16)x; }
};
template <typename T> __device__ __forceinline__ hf32x4 hrf_load4(const T *p);
template <> __device__ __forceinline__ hf32x4 hrf_load4<float>(const float *p) { return *reinterpret_cast<const hf32x4 *>(p); }
template <> __device__ __forceinline__ hf32x4 hrf_load4<_Float16>(const _Float16 *p) {
    const hf16x4 h = *reinterpret_cast<const hf16x4 *>(p);
    return hf32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

// One item: G[P pixels][16 channels] = A[P][K] . W[16][K]^T + bias for MBN blocks of 16 pixels.  The weight vectors stream from L2 in
// chunks of HRF_NVB sixteen-wide k-steps, chunk c + 1 requested before chunk c multiplies (clamped to the last step: every lane issues
// the same loads), and stay in registers for all MBN blocks.
template <typename T, int MBN>
__device__ __forceinline__ void hrf_item(const T *arow, int lda, const float *wrow, int nsteps, float *sgcol, int ldg, int P, float bv, int g) {
    hf32x4 acc[MBN];
#pragma unroll
    for (int mb = 0; mb < MBN; ++mb) acc[mb] = hf32x4{0.f, 0.f, 0.f, 0.f};
    hf32x4 bcur[HRF_NVB], bnxt[HRF_NVB];
#pragma unroll
    for (int t = 0; t < HRF_NVB; ++t) bcur[t] = *reinterpret_cast<const hf32x4 *>(wrow + 16 * min(t, nsteps - 1));
    for (int t0 = 0; t0 < nsteps; t0 += HRF_NVB) {
        if (t0 + HRF_NVB < nsteps) {
#pragma unroll
            for (int t = 0; t < HRF_NVB; ++t) bnxt[t] = *reinterpret_cast<const hf32x4 *>(wrow + 16 * min(t0 + HRF_NVB + t, nsteps - 1));
        }
#pragma unroll
        for (int t = 0; t < HRF_NVB; ++t) {
            if (t0 + t < nsteps) {   // (uniform)
                hf32x4 a[MBN];
#pragma unroll
                for (int mb = 0; mb < MBN; ++mb) a[mb] = hrf_load4<T>(arow + mb * 16 * lda + 16 * (t0 + t));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mb = 0; mb < MBN; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][e], bcur[t][e], acc[mb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < HRF_NVB; ++t) bcur[t] = bnxt[t];
    }
    // lane holds D[pixel 4 g + e][channel l15] of every pixel block; + bias -> G
#pragma unroll
    for (int mb = 0; mb < MBN; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int px = 16 * mb + 4 * g + e;
            if (px < P) sgcol[px * ldg] = acc[mb][e] + bv;
        }
}

template <typename T>
__global__ __launch_bounds__(64 * HRF_WAVES) void hr_fuse_up_kernel(const HrFuseParams p) {
    using U = hrf_unit<T>;
    constexpr int E = U::E;
    extern __shared__ __attribute__((aligned(16))) char hsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int tyn = (p.H + p.th - 1) / p.th, txn = (p.W + HRF_TW - 1) / HRF_TW;
    const int n = blockIdx.x / (tyn * txn), trem = blockIdx.x - n * tyn * txn, ty = trem / txn, tx = trem - ty * txn;
    const int y0 = ty * p.th, x0 = tx * HRF_TW;

    // ---- phase A: the source tiles -> LDS ([P_s][C_s + E] of T at a_off; rows past P_s are never written: their MFMA rows are never read).
    // One unit = 16 bytes of a pixel; a thread takes HRF_UB units of a source per batch and has all of a batch's loads in flight before the
    // first LDS write waits for one.  Index arithmetic without divisions: the tile width is a power of two, the unit count per pixel
    // divides by a host-computed reciprocal (units < 2^16)
    for (int s = 0; s < p.nsrc; ++s) {
        const HrFuseSrc &S = p.src[s];
        T *sa = reinterpret_cast<T *>(hsm + S.a_off);
        const int lpw = 5 - S.shift, cun = S.C / E, lda = S.C + E, total = ((p.th >> S.shift) << lpw) * cun;
        const int sy0 = y0 >> S.shift, sx0 = x0 >> S.shift;
        const T *xs = reinterpret_cast<const T *>(S.x) + (size_t)n * S.H * S.W * S.ld;
        for (int u0 = 0; u0 < total; u0 += HRF_UB * 64 * HRF_WAVES) {
            U v[HRF_UB];
            int dsto[HRF_UB];
#pragma unroll
            for (int k = 0; k < HRF_UB; ++k) {
                const int u = u0 + k * 64 * HRF_WAVES + tid;
                const int px = (int)__umulhi((unsigned)u, S.rcpu), cu = u - px * cun, r = px >> lpw, c = px & ((1 << lpw) - 1);
                const int yy = sy0 + r, xx = sx0 + c;
                dsto[k] = u < total ? px * lda + E * cu : -1;
                v[k].zero();
                if (u < total && yy < S.H && xx < S.W) v[k] = *reinterpret_cast<const U *>(xs + (yy * S.W + xx) * S.ld + E * cu);
            }
#pragma unroll
            for (int k = 0; k < HRF_UB; ++k)
                if (dsto[k] >= 0) *reinterpret_cast<U *>(sa + dsto[k]) = v[k];
        }
    }
    __syncthreads();

    // ---- phase B: this wave's items (source, 16-channel block)
    for (int it = 0; it < p.nitems[wave]; ++it) {
        const int code = p.items[wave][it], s = code >> 4, nb = code & 15;
        const HrFuseSrc &S = p.src[s];
        const int lda = S.C + E, P = (p.th >> S.shift) * (HRF_TW >> S.shift), mbn = (P + 15) >> 4;
        const T *arow = reinterpret_cast<const T *>(hsm + S.a_off) + l15 * lda + 4 * g;
        float *sgcol = reinterpret_cast<float *>(hsm + S.g_off) + 16 * nb + l15;
        const float *wrow = S.w + (size_t)(16 * nb + l15) * S.ldw + 4 * g;
        const float bv = S.bias[16 * nb + l15];
        switch (mbn) {   // pixel blocks of the source tile: 8 / 2 / 1 (16-row tiles), 4 / 1 / 1 (8-row tiles)
            case 8: hrf_item<T, 8>(arow, lda, wrow, S.C >> 4, sgcol, p.ldg, P, bv, g); break;
            case 4: hrf_item<T, 4>(arow, lda, wrow, S.C >> 4, sgcol, p.ldg, P, bv, g); break;
            case 2: hrf_item<T, 2>(arow, lda, wrow, S.C >> 4, sgcol, p.ldg, P, bv, g); break;
            default: hrf_item<T, 1>(arow, lda, wrow, S.C >> 4, sgcol, p.ldg, P, bv, g); break;
        }
    }
    __syncthreads();

    // ---- phase C (batched like phase A)
    {
        const int cun = p.C / E, total = p.th * HRF_TW * cun;
        const T *bs = reinterpret_cast<const T *>(p.base) + (size_t)n * p.H * p.W * p.ldc;
        T *os = reinterpret_cast<T *>(p.out) + (size_t)n * p.H * p.W * p.ldc;
        for (int u0 = 0; u0 < total; u0 += HRF_UC * 64 * HRF_WAVES) {
            U bv[HRF_UC];
            int go[HRF_UC], gof[HRF_UC];   // global element offset (-1: no pixel); (tile pixel << 8) | unit of the pixel
#pragma unroll
            for (int k = 0; k < HRF_UC; ++k) {
                const int u = u0 + k * 64 * HRF_WAVES + tid;
                const int px = (int)__umulhi((unsigned)u, p.rcpu), cu = u - px * cun, r = px >> 5, c = px & 31;
                const int yy = y0 + r, xx = x0 + c;
                const bool ok = u < total && yy < p.H && xx < p.W;
                go[k] = ok ? (yy * p.W + xx) * p.ldc + E * cu : -1;
                gof[k] = (px << 8) | cu;
                bv[k].zero();
                if (ok) bv[k] = *reinterpret_cast<const U *>(bs + go[k]);
            }
#pragma unroll
            for (int k = 0; k < HRF_UC; ++k) {
                if (go[k] < 0) continue;
                const int px = gof[k] >> 8, cu = gof[k] & 255, r = px >> 5, c = px & 31;
                float v[E];
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = bv[k].get(e);
                for (int s = 0; s < p.nsrc; ++s) {
                    const HrFuseSrc &S = p.src[s];
                    const float *sg = reinterpret_cast<const float *>(hsm + S.g_off) + ((r >> S.shift) * (HRF_TW >> S.shift) + (c >> S.shift)) * p.ldg + E * cu;
#pragma unroll
                    for (int q = 0; q < E / 4; ++q) {
                        const hf32x4 gq = *reinterpret_cast<const hf32x4 *>(sg + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q + e] += gq[e];
                    }
                }
                U o;
#pragma unroll
                for (int e = 0; e < E; ++e) o.set(e, p.relu ? fmaxf(v[e], 0.f) : v[e]);
                *reinterpret_cast<U *>(os + go[k]) = o;
            }
        }
    }
}

// ====================================================================== host side
// Fills the tile height, the LDS layout and the item lists; false: the shape has no fused form (the caller runs the per-term launches)
bool hr_fuse_up_plan(HrFuseParams &p) {
    p.th = 0;
    p.lds = 0;
    if (p.nsrc < 1 || p.nsrc > 3 || p.C % (p.f16 ? 8 : 4) || p.C > 256 || p.ldc % 4 || p.N <= 0 || p.H <= 0 || p.W <= 0) return false;
    if ((long long)p.H * p.W * p.ldc >= (1ll << 31) || p.C < 8) return false;   // 32-bit element offsets inside an image
    const int esz = p.f16 ? 2 : 4;
    for (int s = 0; s < p.nsrc; ++s) {
        const HrFuseSrc &S = p.src[s];
        if (S.shift < 1 || S.shift > 3 || S.C % 16 || S.C > 512 || S.ld % 4 || S.ldw % 4 || S.ldw < S.C) return false;
        if (S.H != (p.H >> S.shift) || S.W != (p.W >> S.shift) || (S.H << S.shift) != p.H || (S.W << S.shift) != p.W) return false;
    }
    const int epu = p.f16 ? 8 : 4;   // elements per 16-byte unit
    p.rcpu = (unsigned)(((1ull << 32) + (unsigned)(p.C / epu) - 1) / (unsigned)(p.C / epu));   // ceil(2^32 / units per pixel): exact quotients for units < 2^16
    for (int s = 0; s < p.nsrc; ++s) p.src[s].rcpu = (unsigned)(((1ull << 32) + (unsigned)(p.src[s].C / epu) - 1) / (unsigned)(p.src[s].C / epu));
    const int nbn = (p.C + 15) / 16;   // 16-channel blocks (the weight / bias rows past C are zeros: Layer's Cout_pad)
    p.ldg = 16 * nbn + 4;
    for (int th : {16, 8}) {
        size_t off = 0;
        bool ok = true;
        // the A tiles, the one with the fewest pixels first (its MFMA rows past P read on into the next tile: inside the allocation)
        int order[3] = {0, 1, 2};
        for (int a = 0; a < p.nsrc; ++a)
            for (int b = a + 1; b < p.nsrc; ++b)
                if (p.src[order[b]].shift > p.src[order[a]].shift) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
        for (int a = 0; a < p.nsrc; ++a) {
            HrFuseSrc &S = p.src[order[a]];
            const int P = (th >> S.shift) * (HRF_TW >> S.shift);
            if ((th >> S.shift) < 1) { ok = false; break; }
            S.a_off = (int)off;
            off += ((size_t)P * (S.C + epu) * esz + 15) / 16 * 16;
        }
        if (!ok) continue;
        // a pixel block of 16 MFMA rows may start up to 15 rows before a tile's end: keep that much readable behind the last A tile
        const HrFuseSrc &L = p.src[order[p.nsrc - 1]];
        size_t need = (size_t)L.a_off + (size_t)(((th >> L.shift) * (HRF_TW >> L.shift) + 15) / 16 * 16) * (L.C + epu) * esz;
        for (int a = 0; a < p.nsrc; ++a) {   // (every A tile's padded extent)
            const HrFuseSrc &S = p.src[order[a]];
            const size_t ext = (size_t)S.a_off + (size_t)(((th >> S.shift) * (HRF_TW >> S.shift) + 15) / 16 * 16) * (S.C + epu) * esz;
            if (ext > need) need = ext;
        }
        if (off < need) off = (need + 15) / 16 * 16;
        for (int s = 0; s < p.nsrc; ++s) {
            HrFuseSrc &S = p.src[s];
            const int P = (th >> S.shift) * (HRF_TW >> S.shift);
            S.g_off = (int)off;
            off += (size_t)P * p.ldg * 4;
        }
        if (off > 160 * 1024) continue;
        p.th = th;
        p.lds = (int)off;
        if (off <= 80 * 1024 || th == 8) break;   // two workgroups per CU where a tile height allows it
    }
    if (!p.th) return false;
    // items (source, 16-channel block), longest first to the least loaded wave; cost = MFMAs
    struct It { int code, cost; } items[48];
    int ni = 0;
    for (int s = 0; s < p.nsrc; ++s) {
        const HrFuseSrc &S = p.src[s];
        const int P = (p.th >> S.shift) * (HRF_TW >> S.shift), mbn = (P + 15) / 16;
        for (int nb = 0; nb < nbn; ++nb) items[ni++] = {s * 16 + nb, mbn * (S.C / 4)};
    }
    for (int a = 0; a < ni; ++a)
        for (int b = a + 1; b < ni; ++b)
            if (items[b].cost > items[a].cost) { const It t = items[a]; items[a] = items[b]; items[b] = t; }
    int load[HRF_WAVES] = {0, 0, 0, 0};
    for (int w = 0; w < HRF_WAVES; ++w) p.nitems[w] = 0;
    for (int a = 0; a < ni; ++a) {
        int w = 0;
        for (int q = 1; q < HRF_WAVES; ++q)
            if (load[q] < load[w]) w = q;
        if (p.nitems[w] >= 12) return false;
        p.items[w][p.nitems[w]++] = (unsigned char)items[a].code;
        load[w] += items[a].cost;
    }
    return true;
}

hipError_t launch_hr_fuse_up(const HrFuseParams &p, hipStream_t s) {
    if (!p.th || p.lds <= 0) return hipErrorInvalidValue;
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(hr_fuse_up_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(hr_fuse_up_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    const int tyn = (p.H + p.th - 1) / p.th, txn = (p.W + HRF_TW - 1) / HRF_TW;
    const dim3 grid((unsigned)(p.N * tyn * txn)), block(64 * HRF_WAVES);
    if (p.f16) hipLaunchKernelGGL(hr_fuse_up_kernel<_Float16>, grid, block, p.lds, s, p);
    else hipLaunchKernelGGL(hr_fuse_up_kernel<float>, grid, block, p.lds, s, p);
    return hipGetLastError();
}

}  // namespace hmv
