// misc_kernels.hip -- the memory-bound / small kernels around the conv-GEMM family.
// All are HBM- or latency-bound; they are written for coalesced 16-byte accesses and
// 64-lane wave reductions (gfx950 wavefront = 64).
#include <cstdlib>

#include "kernels.h"

namespace hmv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------ frames: NCHW fp32 -> NHWC4
// x.view(-1, c, h, w) of handmvnet.py:163 ; the 4th channel is a zero so that one conv tap
// is one aligned 16-byte vector for the stem's implicit GEMM.
__global__ void nchw_to_nhwc4_kernel(const float *__restrict__ x, f32x4 *__restrict__ out, int HW, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const size_t n = i / HW, pix = i - n * HW;
        const float *src = x + n * 3 * (size_t)HW + pix;
        out[i] = f32x4{src[0], src[HW], src[2 * (size_t)HW], 0.f};
    }
}
hipError_t launch_nchw_to_nhwc4(const float *x, float *out, int N, int H, int W, hipStream_t s) {
    const size_t total = (size_t)N * H * W;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(grid), dim3(256), 0, s, x, reinterpret_cast<f32x4 *>(out), H * W, total);
    return hipGetLastError();
}

// fp16 path: NCHW fp32 frames -> NHWC8 fp16 (3 real channels + 5 zeros = one 16-byte tap)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__global__ void nchw_to_nhwc8_f16_kernel(const float *__restrict__ x, f16x8 *__restrict__ out, int HW, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const size_t n = i / HW, pix = i - n * HW;
        const float *src = x + n * 3 * (size_t)HW + pix;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        v[0] = (_Float16)src[0]; v[1] = (_Float16)src[HW]; v[2] = (_Float16)src[2 * (size_t)HW];
        out[i] = v;
    }
}
hipError_t launch_nchw_to_nhwc8_f16(const float *x, void *out, int N, int H, int W, hipStream_t s) {
    const size_t total = (size_t)N * H * W;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(nchw_to_nhwc8_f16_kernel, dim3(grid), dim3(256), 0, s, x, reinterpret_cast<f16x8 *>(out), H * W, total);
    return hipGetLastError();
}

__global__ void maxpool3s2_f16_kernel(const f16x8 *__restrict__ in, f16x8 *__restrict__ out, int H, int W, int C8, int Ho,
                                      int Wo, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % C8);
        size_t t = i / C8;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const size_t n = t / Ho;
        f16x8 m;
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = (_Float16)(-65504.f);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = ho * 2 - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int wi = wo * 2 - 1 + q;
                if ((unsigned)wi >= (unsigned)W) continue;
                m = __builtin_elementwise_max(m, in[((n * H + hi) * W + wi) * C8 + c]);
            }
        }
        out[i] = m;
    }
}
hipError_t launch_maxpool3s2_f16(const void *in, void *out, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool3s2_f16_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const f16x8 *>(in),
                       reinterpret_cast<f16x8 *>(out), H, W, C / 8, Ho, Wo, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------ HMV_F32X3: fp32 values as (hi, lo) fp16 pairs
// A tensor row is [hi plane (C halfs) | lo plane (C halfs)]: hi = fp16(v), lo = fp16(v - hi); hi + lo == v to ~2^-22.
__device__ __forceinline__ void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    hi = (_Float16)c;
    lo = (_Float16)(c - (float)hi);
}
// frames NCHW fp32 -> per pixel [hi: r g b 0 0 0 0 0 | lo: r g b 0 0 0 0 0] (32 bytes)
__global__ void nchw_to_nhwc_split_kernel(const float *__restrict__ x, f16x8 *__restrict__ out, int HW, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const size_t n = i / HW, pix = i - n * HW;
        const float *src = x + n * 3 * (size_t)HW + pix;
        f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            _Float16 a, b;
            split_f16(src[c * (size_t)HW], a, b);
            h[c] = a; l[c] = b;
        }
        out[2 * i] = h;
        out[2 * i + 1] = l;
    }
}
hipError_t launch_nchw_to_nhwc_split(const float *x, void *out, int N, int H, int W, hipStream_t s) {
    const size_t total = (size_t)N * H * W;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(nchw_to_nhwc_split_kernel, dim3(grid), dim3(256), 0, s, x, reinterpret_cast<f16x8 *>(out), H * W, total);
    return hipGetLastError();
}
// ------------------------------------------------------------------ space-to-depth stem input (ResNet backbones, round 2)
// conv1 7x7 stride 2 pad 3 over 3 channels == a 4x4 stride-1 pad-2 conv over the 2x2 space-to-depth image with 12 channels
// whose extra (8th) row / column of weights is zero: input row 2y - 3 + r is row pair y - 2 + (r + 1) / 2, sub-row (r + 1) % 2.
// The reduction shrinks from 7 x 7 x 4 (pad channel) = 196 -> 224 executed to 192 (fp32), from 7 x 7 x 8 = 392 -> 448 to
// 4 x 4 x 16 = 256 (fp16), and every 16-byte vector the dense-K gather moves is 75-100 % real data instead of 37-75 %.
// Channel order inside an s2d pixel: (dy, dx, c).  MODE 0: fp32 [12]; 1: fp16 [12 + 4 zeros]; 2: split [hi 16 | lo 16].
// Rows / columns beyond an odd H / W are zeros.
template <int MODE>
__device__ __forceinline__ void s2d_store(void *__restrict__ out, size_t q, const float (&v)[12]) {
    if (MODE == 0) {
        f32x4 *o = reinterpret_cast<f32x4 *>(out) + 3 * q;
#pragma unroll
        for (int j = 0; j < 3; ++j) o[j] = f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
    } else if (MODE == 1) {
        f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (_Float16)v[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (_Float16)v[8 + j];
        f16x8 *o = reinterpret_cast<f16x8 *>(out) + 2 * q;
        o[0] = a; o[1] = b;
    } else {
        f16x8 h0 = {0, 0, 0, 0, 0, 0, 0, 0}, h1 = h0, l0 = h0, l1 = h0;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            _Float16 a, b;
            split_f16(v[j], a, b);
            if (j < 8) { h0[j] = a; l0[j] = b; } else { h1[j - 8] = a; l1[j - 8] = b; }
        }
        f16x8 *o = reinterpret_cast<f16x8 *>(out) + 4 * q;
        o[0] = h0; o[1] = h1; o[2] = l0; o[3] = l1;
    }
}
template <int MODE>
__global__ void nchw_to_s2d_kernel(const float *__restrict__ x, void *__restrict__ out, int H, int W, int Hs, int Ws, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int xs = (int)(i % Ws);
        size_t t = i / Ws;
        const int ys = (int)(t % Hs);
        const size_t n = t / Hs;
        float v[12];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = 2 * ys + dy, xx = 2 * xs + dx;
                const bool in = yy < H && xx < W;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[(dy * 2 + dx) * 3 + c] = in ? x[((n * 3 + c) * H + yy) * (size_t)W + xx] : 0.f;
            }
        s2d_store<MODE>(out, i, v);
    }
}
hipError_t launch_nchw_to_s2d(const float *x, void *out, int N, int H, int W, int mode, hipStream_t s) {
    const int Hs = (H + 1) / 2, Ws = (W + 1) / 2;
    const size_t total = (size_t)N * Hs * Ws;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (mode == 2) hipLaunchKernelGGL(nchw_to_s2d_kernel<2>, dim3(grid), dim3(256), 0, s, x, out, H, W, Hs, Ws, total);
    else if (mode == 1) hipLaunchKernelGGL(nchw_to_s2d_kernel<1>, dim3(grid), dim3(256), 0, s, x, out, H, W, Hs, Ws, total);
    else hipLaunchKernelGGL(nchw_to_s2d_kernel<0>, dim3(grid), dim3(256), 0, s, x, out, H, W, Hs, Ws, total);
    return hipGetLastError();
}
// fp32 rows [rows][C] -> fp16 rows [rows][C] (mode 1) or split rows [rows][hi C | lo C] (mode 2): op-level tests
__global__ void rows_f32_to_half_kernel(const float *__restrict__ in, _Float16 *__restrict__ out, int C, int mode, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const size_t r = i / C;
        const int c = (int)(i - r * C);
        if (mode == 2) {
            _Float16 a, b;
            split_f16(in[i], a, b);
            out[r * 2 * C + c] = a;
            out[r * 2 * C + C + c] = b;
        } else {
            out[i] = (_Float16)in[i];
        }
    }
}
hipError_t launch_rows_f32_to_half(const float *in, void *out, size_t rows, int C, int mode, hipStream_t s) {
    const size_t total = rows * (size_t)C;
    if (!total) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(rows_f32_to_half_kernel, dim3(grid), dim3(256), 0, s, in, reinterpret_cast<_Float16 *>(out), C, mode, total);
    return hipGetLastError();
}
// MaxPool2d(3, 2, 1) on split tensors: the maximum of the reconstructed values, re-split
__global__ void maxpool3s2_split_kernel(const f16x8 *__restrict__ in, f16x8 *__restrict__ out, int H, int W, int C8, int Ho,
                                        int Wo, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % C8);
        size_t t = i / C8;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const size_t n = t / Ho;
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = ho * 2 - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int wi = wo * 2 - 1 + q;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f16x8 *px = in + ((n * H + hi) * W + wi) * (size_t)(2 * C8);
                const f16x8 h = px[c], l = px[C8 + c];
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)h[j] + (float)l[j]);
            }
        }
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 a, b;
            split_f16(m[j], a, b);
            h[j] = a; l[j] = b;
        }
        f16x8 *o = out + (i / C8) * (size_t)(2 * C8);
        o[c] = h;
        o[C8 + c] = l;
    }
}
hipError_t launch_maxpool3s2_split(const void *in, void *out, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool3s2_split_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const f16x8 *>(in),
                       reinterpret_cast<f16x8 *>(out), H, W, C / 8, Ho, Wo, total);
    return hipGetLastError();
}
// split NHWC -> NCHW fp32 (stage capture)
__global__ void nhwc_split_to_nchw_kernel(const _Float16 *__restrict__ in, float *__restrict__ out, int HW, int C, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int p = (int)(i % HW);
        const size_t t = i / HW;
        const int c = (int)(t % C);
        const size_t n = t / C;
        const _Float16 *px = in + (n * HW + p) * (size_t)(2 * C);
        out[i] = (float)px[c] + (float)px[C + c];
    }
}
hipError_t launch_nhwc_split_to_nchw(const void *in, float *out, int N, int H, int W, int C, hipStream_t s) {
    const size_t total = (size_t)N * H * W * C;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nhwc_split_to_nchw_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const _Float16 *>(in), out, H * W, C, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------ MaxPool2d(3, stride 2, pad 1), NHWC
// resnet.py:165,221
__global__ void maxpool3s2_kernel(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, int H, int W, int C4, int Ho,
                                  int Wo, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const size_t n = t / Ho;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = ho * 2 - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int wi = wo * 2 - 1 + q;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = in[((n * H + hi) * W + wi) * C4 + c];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        out[i] = m;
    }
}
hipError_t launch_maxpool3s2(const float *in, float *out, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const f32x4 *>(in),
                       reinterpret_cast<f32x4 *>(out), H, W, C / 4, Ho, Wo, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------ soft_argmax_2d (temperature 1000)
// models/utils.py:35-62.  One wave per (image, joint); logits stay fp32 end to end.
__global__ void soft_argmax_kernel(const float *__restrict__ hm, int ld, int h, int w, int total, float *coords,
                                   float *crop_img, float image_size, float heatmap_size, float *hm_nchw) {
    const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wid >= total) return;
    const int n = wid / 21, j = wid - n * 21, hw = h * w;
    const float *src = hm + (size_t)n * hw * ld + j;
    float mx = -INFINITY;
    for (int p = lane; p < hw; p += 64) {
        const float raw = src[(size_t)p * ld];
        if (hm_nchw) hm_nchw[((size_t)n * 21 + j) * hw + p] = raw;
        mx = fmaxf(mx, raw * 1000.0f);
    }
    mx = wave_max(mx);
    float se = 0.f, sx = 0.f, sy = 0.f;
    for (int p = lane; p < hw; p += 64) {
        const float e = expf(src[(size_t)p * ld] * 1000.0f - mx);
        const int y = p / w, x = p - y * w;
        se += e;
        sx += e * (float)x;
        sy += e * (float)y;
    }
    se = wave_sum(se);
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    if (lane == 0) {
        const float cx = sx / se, cy = sy / se;
        coords[2 * wid] = cx;
        coords[2 * wid + 1] = cy;
        crop_img[2 * wid] = cx * image_size / heatmap_size;  // handmvnet.py:252
        crop_img[2 * wid + 1] = cy * image_size / heatmap_size;
    }
}
// The same op, one workgroup per FRAME (heat maps of up to 32 x 32 pixels: the ResNet50-paper head).  The channels-last logits
// [h * w][32] are read as whole 128-byte pixel rows -- 8 lanes x 16 bytes per pixel, 32 pixels per pass -- instead of one
// 4-byte column per lane and pixel: the wave-per-(frame, joint) kernel above moved 402 MB per cfg-3 launch for 34 MB of logits
// (every wave pulls whole cache lines for one float each).  Two passes (channel maxima, then the three sums), the second from
// L2; per-thread partials over its pixel slot, then a fixed-order reduction over the 32 slots through LDS.  Which kernel runs
// depends on the heat-map size alone, never on the batch.
__global__ __launch_bounds__(256) void soft_argmax_frame_kernel(const float *__restrict__ hm, int h, int w, float *coords, float *crop_img,
                                                                float image_size, float heatmap_size, float *hm_nchw) {
    __shared__ float red[4][32][33];   // [max | se | sx | sy][pixel slot][channel]
    const int n = blockIdx.x, tid = threadIdx.x, slot = tid >> 3, c4 = tid & 7, hw = h * w;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(hm + (size_t)n * hw * 32) + c4;
    const bool act = c4 < 6;   // channels 0 .. 23 (21 joints); the last two vectors of a pixel row are padding
    f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    // (eight pixels per trip, every load issued before the first use: a load-use-load loop pays a memory round trip per pixel)
    for (int p0 = slot; p0 < hw; p0 += 256) {
        f32x4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 32 * u;
            raw[u] = (act && p < hw) ? src[(size_t)p * 8] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 32 * u;
            if (hm_nchw && act && p < hw) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * c4 + e < 21) hm_nchw[((size_t)n * 21 + 4 * c4 + e) * hw + p] = raw[u][e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], raw[u][e] * 1000.0f);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[0][slot][4 * c4 + e] = mx[e];
    __syncthreads();
    if (tid < 32) {
        float m = red[0][0][tid];
        for (int q = 1; q < 32; ++q) m = fmaxf(m, red[0][q][tid]);
        red[0][0][tid] = m;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) mx[e] = red[0][0][4 * c4 + e];
    f32x4 se = {0.f, 0.f, 0.f, 0.f}, sx = se, sy = se;
    for (int p0 = slot; p0 < hw; p0 += 256) {
        f32x4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 32 * u;
            raw[u] = (act && p < hw) ? src[(size_t)p * 8] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // exp(-inf) = 0
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 32 * u, y = p / w, x = p - y * w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ex = act ? expf(raw[u][e] * 1000.0f - mx[e]) : 0.f;
                se[e] += ex;
                sx[e] += ex * (float)x;
                sy[e] += ex * (float)y;
            }
        }
    }
    __syncthreads();   // red[0][0][*] has been read by everyone
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[1][slot][4 * c4 + e] = se[e];
        red[2][slot][4 * c4 + e] = sx[e];
        red[3][slot][4 * c4 + e] = sy[e];
    }
    __syncthreads();
    if (tid < 21) {
        float a = 0.f, bx = 0.f, by = 0.f;
        for (int q = 0; q < 32; ++q) { a += red[1][q][tid]; bx += red[2][q][tid]; by += red[3][q][tid]; }
        const float cx = bx / a, cy = by / a;
        const int wid = n * 21 + tid;
        coords[2 * wid] = cx;
        coords[2 * wid + 1] = cy;
        crop_img[2 * wid] = cx * image_size / heatmap_size;  // handmvnet.py:252
        crop_img[2 * wid + 1] = cy * image_size / heatmap_size;
    }
}
hipError_t launch_soft_argmax(const float *hm, int ld, int N, int h, int w, float *coords, float *crop_img,
                              float image_size, float heatmap_size, float *hm_nchw, hipStream_t s) {
    static const bool no_frame = HMV_DEV_ENV("HMV_SOFTARGMAX_WAVE") != nullptr;   // development knob (A/B runs)
    if (!no_frame && ld == 32 && h * w <= 1024) {
        hipLaunchKernelGGL(soft_argmax_frame_kernel, dim3(N), dim3(256), 0, s, hm, h, w, coords, crop_img, image_size, heatmap_size, hm_nchw);
        return hipGetLastError();
    }
    const int total = N * 21;
    hipLaunchKernelGGL(soft_argmax_kernel, dim3((total + 3) / 4), dim3(256), 0, s, hm, ld, h, w, total, coords, crop_img,
                       image_size, heatmap_size, hm_nchw);
    return hipGetLastError();
}

// ------------------------------------------------------------------ SampleNet: gather-then-conv
// nets.py:46-63.  conv1x1+BN+ReLU is pointwise, so conv(map) sampled at 4 taps == conv applied to
// the 4 gathered input pixels; the bilinear blend (with zero padding) follows.  The unnormalise
// arithmetic replays F.grid_sample(align_corners=True) in fp32.
struct Taps {
    int x0, y0;
    float w[4];   // nw, ne, sw, se
    bool v[4];
};
__device__ __forceinline__ Taps bilinear_taps(float jx, float jy, int H, int W) {
    const float gx = jx / (float)(W - 1) * 2.f - 1.f, gy = jy / (float)(H - 1) * 2.f - 1.f;
    const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1), iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    Taps t;
    // clamp before the int conversion so that absurd coordinates cannot overflow; they are invalid anyway
    t.x0 = (int)fminf(fmaxf(fx0, -4.f), (float)W + 4.f);
    t.y0 = (int)fminf(fmaxf(fy0, -4.f), (float)H + 4.f);
    const float x1 = fx0 + 1.f, y1 = fy0 + 1.f;
    t.w[0] = (x1 - ix) * (y1 - iy);
    t.w[1] = (ix - fx0) * (y1 - iy);
    t.w[2] = (x1 - ix) * (iy - fy0);
    t.w[3] = (ix - fx0) * (iy - fy0);
    const bool vx0 = t.x0 >= 0 && t.x0 < W, vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    const bool vy0 = t.y0 >= 0 && t.y0 < H, vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    const bool fin = (fx0 == fx0) && (fy0 == fy0);  // NaN coordinates sample nothing
    t.v[0] = fin && vy0 && vx0;
    t.v[1] = fin && vy0 && vx1;
    t.v[2] = fin && vy1 && vx0;
    t.v[3] = fin && vy1 && vx1;
    return t;
}

__global__ void sample_gather_kernel(const f32x4 *__restrict__ feat, int H, int W, int C4, const float *__restrict__ coords,
                                     f32x4 *__restrict__ out) {
    const int row = blockIdx.x;  // n*21 + j
    const int n = row / 21;
    const Taps t = bilinear_taps(coords[2 * row], coords[2 * row + 1], H, W);
    // (no t.v[tap]: a run-time index would put the struct into scratch memory -- 29 us for this 5 us copy)
    const unsigned vmask = (t.v[0] ? 1u : 0u) | (t.v[1] ? 2u : 0u) | (t.v[2] ? 4u : 0u) | (t.v[3] ? 8u : 0u);
    for (int i = threadIdx.x; i < 4 * C4; i += blockDim.x) {
        const int tap = i / C4, c = i - tap * C4;
        const int x = t.x0 + (tap & 1), y = t.y0 + (tap >> 1);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((vmask >> tap) & 1u) v = feat[(((size_t)n * H + y) * W + x) * C4 + c];
        out[((size_t)row * 4 + tap) * C4 + c] = v;
    }
}
hipError_t launch_sample_gather(const float *feat, int N, int H, int W, int C, const float *coords, float *out,
                                hipStream_t s, int elem_bytes) {
    // the kernel copies 16-byte chunks, so it is dtype-agnostic: C * elem_bytes / 16 chunks per pixel
    hipLaunchKernelGGL(sample_gather_kernel, dim3(N * 21), dim3(256), 0, s, reinterpret_cast<const f32x4 *>(feat), H, W,
                       C * elem_bytes / 16, coords, reinterpret_cast<f32x4 *>(out));
    return hipGetLastError();
}

__global__ void sample_blend_kernel(const float *__restrict__ s4, int lds4, int C, int H, int W,
                                    const float *__restrict__ coords, float *__restrict__ tokens, int ldt, int col0) {
    const int row = blockIdx.x;
    const Taps t = bilinear_taps(coords[2 * row], coords[2 * row + 1], H, W);
    const float *b = s4 + (size_t)row * 4 * lds4;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v = 0.f;  // same accumulation order as grid_sample: nw, ne, sw, se
        if (t.v[0]) v += b[c] * t.w[0];
        if (t.v[1]) v += b[lds4 + c] * t.w[1];
        if (t.v[2]) v += b[2 * lds4 + c] * t.w[2];
        if (t.v[3]) v += b[3 * lds4 + c] * t.w[3];
        tokens[(size_t)row * ldt + col0 + c] = v;
    }
}
hipError_t launch_sample_blend(const float *s4, int lds4, int C, int N, int H, int W, const float *coords, float *tokens,
                               int ldt, int col0, hipStream_t s) {
    hipLaunchKernelGGL(sample_blend_kernel, dim3(N * 21), dim3(128), 0, s, s4, lds4, C, H, W, coords, tokens, ldt, col0);
    return hipGetLastError();
}

// ------------------------------------------------------------------ token tail
// pos2d (handmvnet.py:189-191), crop FoV (handmvnet.py:205-222, utils.py:134-171), zero pad,
// optional capture of the raw tokens, then + sinusoidal PE (layers.py:152-158; table built on host).
// pairs (optional): the finished rows once more as (hi, lo) fp16 pairs [row][hi ldt | lo ldt] -- what the first fusion block's q/k/v
// projection reads in the fp16-kernel modes (rows_f32_to_half_kernel's arithmetic: same bits, one launch less)
__global__ void tokens_finalize_kernel(float *tokens, int ldt, int d, int fdim, int V, const float *coords,
                                       const float *bbox, const float *intr, int pos_mask, const float *pe,
                                       float *raw_copy, _Float16 *pairs) {
    const int row = blockIdx.x;           // n*21 + j, n = b*V + v
    const int n = row / 21, j = row - n * 21;
    float *t = tokens + (size_t)row * ldt;
    int col = fdim;
    if (threadIdx.x == 0) {
        if (pos_mask & 1) { t[col] = coords[2 * row]; t[col + 1] = coords[2 * row + 1]; }
    }
    if (pos_mask & 1) col += 2;
    if ((pos_mask & 2) && threadIdx.x < 10) {
        const float *bb = bbox + 4 * n, *in = intr + 4 * n;
        const int pt = threadIdx.x >> 1, isy = threadIdx.x & 1;
        float px, py;
        if (pt == 0) { px = bb[0]; py = bb[1]; }
        else if (pt == 1) { px = bb[0]; py = bb[3]; }
        else if (pt == 2) { px = bb[2]; py = bb[1]; }
        else if (pt == 3) { px = bb[2]; py = bb[3]; }
        else { px = (bb[0] + bb[2]) / 2.f; py = (bb[1] + bb[3]) / 2.f; }
        t[col + threadIdx.x] = isy ? atanf((py - in[3]) / in[1]) : atanf((px - in[2]) / in[0]);
    }
    for (int c = d + threadIdx.x; c < ldt; c += blockDim.x) t[c] = 0.f;
    __syncthreads();
    const int pos = (n % V) * 21 + j;  // token index inside its sample, view-major
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float v = t[c];
        if (raw_copy) raw_copy[(size_t)row * d + c] = v;
        if (pe) { v = v + pe[(size_t)pos * d + c]; t[c] = v; }
        if (pairs) {
            _Float16 a, b;
            split_f16(v, a, b);
            pairs[(size_t)row * 2 * ldt + c] = a;
            pairs[(size_t)row * 2 * ldt + ldt + c] = b;
        }
    }
    if (pairs)
        for (int c = d + threadIdx.x; c < ldt; c += blockDim.x) { pairs[(size_t)row * 2 * ldt + c] = (_Float16)0.f; pairs[(size_t)row * 2 * ldt + ldt + c] = (_Float16)0.f; }
}
hipError_t launch_tokens_finalize(float *tokens, int ldt, int d, int fdim, int N, int V, const float *coords,
                                  const float *bbox, const float *intr, int pos_mask, const float *pe, float *raw_copy,
                                  hipStream_t s, void *pairs) {
    hipLaunchKernelGGL(tokens_finalize_kernel, dim3(N * 21), dim3(128), 0, s, tokens, ldt, d, fdim, V, coords, bbox, intr,
                       pos_mask, pe, raw_copy, reinterpret_cast<_Float16 *>(pairs));
    return hipGetLastError();
}

// ------------------------------------------------------------------ LayerNorm (+ optional chained LayerNorm)
// layers.py:194-195 (norm1/norm2), layers.py:165 (FeedForward's leading LayerNorm); eps 1e-5.
// One wave per row, two-pass (mean, then centred variance) in registers.
constexpr int LN_MAX_PER_LANE = 16;  // d <= 1024
// SK: the row is not read from x but assembled from the S partial products of a split-K GEMM (slices added in index order,
// then bias, then the residual row -- the arithmetic of splitk_reduce_kernel, so both forms give the same bits)
struct LnSplitK { const float *slab; int S; size_t slice; int lds; const float *bias; const float *res; int ldr, rg_out, rg_in; };
template <int SK>   // 0: plain rows; otherwise the number of split-K slices
__global__ void layernorm_kernel(const float *__restrict__ x, int ldx, int rows, int d, const float *__restrict__ g1,
                                 const float *__restrict__ b1, float *__restrict__ y, int ldy,
                                 const float *__restrict__ g2, const float *__restrict__ b2, float *__restrict__ y2, LnSplitK sk) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = SK ? nullptr : x + (size_t)row * ldx;
    const float *rr = nullptr;
    if (SK && sk.res) rr = sk.res + (size_t)(sk.rg_out ? (row / sk.rg_out) * sk.rg_in + (row % sk.rg_out) : row) * sk.ldr;
    float v[LN_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (64 * i >= d) { v[i] = 0.f; continue; }   // wave-uniform
        const int cc = c < d ? c : 0;   // clamped: the loads are unconditional, so all of a row's loads are in flight together
        if (SK) {
            float part[SK ? SK : 1];
#pragma unroll
            for (int k = 0; k < SK; ++k) part[k] = sk.slab[k * sk.slice + (size_t)row * sk.lds + cc];
            const float bb = sk.bias[cc], rv = rr ? rr[cc] : 0.f;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < SK; ++k) t += part[k];
            t += bb;
            if (rr) t += rv;
            v[i] = c < d ? t : 0.f;
        } else {
            const float t = xr[cc];
            v[i] = c < d ? t : 0.f;
        }
        s += v[i];
    }
    const float inv_d = 1.f / (float)d;
    float mean = wave_sum(s) * inv_d, q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float t = c < d ? v[i] - mean : 0.f;
        q += t * t;
    }
    float rstd = 1.f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
    s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < d) {
            v[i] = (v[i] - mean) * rstd * g1[c] + b1[c];
            y[(size_t)row * ldy + c] = v[i];
            s += v[i];
        } else if (c < ldy) {
            y[(size_t)row * ldy + c] = 0.f;
        }
    }
    if (!y2) return;
    mean = wave_sum(s) * inv_d;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        const float t = c < d ? v[i] - mean : 0.f;
        q += t * t;
    }
    rstd = 1.f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
        const int c = lane + 64 * i;
        if (c < d) y2[(size_t)row * ldy + c] = (v[i] - mean) * rstd * g2[c] + b2[c];
        else if (c < ldy) y2[(size_t)row * ldy + c] = 0.f;
    }
}
hipError_t launch_layernorm(const float *x, int ldx, int rows, int d, const float *g1, const float *b1, float *y, int ldy,
                            const float *g2, const float *b2, float *y2, hipStream_t s) {
    if (d > 64 * LN_MAX_PER_LANE || ldy > 64 * LN_MAX_PER_LANE) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_kernel<0>, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, rows, d, g1, b1, y, ldy, g2, b2, y2, LnSplitK{});
    return hipGetLastError();
}
hipError_t launch_splitk_layernorm(const float *slab, int S, int rows, int lds, int d, const float *bias, const float *res, int ldr,
                                   int rg_out, int rg_in, const float *g1, const float *b1, float *y, int ldy, const float *g2,
                                   const float *b2, float *y2, hipStream_t s) {
    if (d > 64 * LN_MAX_PER_LANE || ldy > 64 * LN_MAX_PER_LANE || S != 4 || !slab || !bias) return hipErrorInvalidValue;
    if (!rows) return hipSuccess;
    const LnSplitK sk{slab, S, (size_t)rows * lds, lds, bias, res, ldr, rg_out, rg_in};
    hipLaunchKernelGGL(layernorm_kernel<4>, dim3((rows + 3) / 4), dim3(256), 0, s, nullptr, 0, rows, d, g1, b1, y, ldy, g2, b2, y2, sk);
    return hipGetLastError();
}

// ------------------------------------------------------------------ attention on the fp32 matrix cores
// layers.py:216-221: dots = q k^T * 128^-0.5 ; softmax over keys ; out = attn v      (8 heads x 128)
// One workgroup (4 waves) per (sample, head, 32-row query block); the 32-key chunks of the key range are dealt round-robin
// to the 4 waves (wave w: chunks w, w + 4, ...), each wave runs the flash recurrence over ITS chunks, and the four partial
// (max, sum, O) triples are merged through LDS at the end.  Nothing depends on the batch.
// Everything is computed TRANSPOSED, so that a query row lives on a lane instead of across lanes:
//   S^T = K_chunk Q_blk^T   64 x v_mfma_f32_32x32x2_f32 (k = 128): A = the lane's own key row straight from global memory,
//                           B = Q block from LDS (staged once per workgroup).  Lane (q, half) then holds the logits of query q
//                           against 16 of the chunk's 32 keys in its 16 accumulator registers
//   online softmax          row max = max over the lane's 16 registers + ONE cross-half exchange (the round-1 form, key on the
//                           lane, needed 80 dependent wave shuffles per chunk and a P round trip through LDS: a 13 us floor
//                           per launch); running max / sum / rescale factor are one register each
//   O^T += V_chunk^T P^T    64 x MFMA: A = V rows from a per-wave LDS buffer (channel on the lane), B = P^T -- which is exactly
//                           the accumulator layout S^T came out in, so P never leaves the registers
// The k order inside every 8-wide group is permuted identically for both operands (16-byte reads).
typedef float f32x16 __attribute__((ext_vector_type(16)));
// ATT_WAVES: waves per workgroup; the 32-key chunks of the key range are dealt round-robin to them (4 everywhere: launch_attention_any)
template <int D, int ATT_WAVES> struct AttShape {
    static constexpr int LDK = D + 4;                           // Q block row stride (floats): conflict-free ds_read_b128 of the B operand
    static constexpr int NC = D / 32;                           // 32-channel blocks of the head
    static constexpr int Q_FLOATS = 32 * LDK;
    static constexpr int V_FLOATS = 8 * D;                      // per wave: a quarter of a chunk
    static constexpr int LOOP_FLOATS = Q_FLOATS + ATT_WAVES * V_FLOATS;
    static constexpr int MERGE_FLOATS = ATT_WAVES * NC * 16 * 64 + 2 * ATT_WAVES * 32;   // O partials [w][c][e][lane], then max and sum [w][query]
    static constexpr int LDS_FLOATS = LOOP_FLOATS > MERGE_FLOATS ? LOOP_FLOATS : MERGE_FLOATS;
};

#ifndef HMV_ATT_OCC
#define HMV_ATT_OCC 2   // workgroups per CU the register budget is set for (128-wide heads; the 256-wide ones take the CU alone)
#endif
// q rows: q + (b * q_bstride + i) * q_ld + h * D;  k / v rows: k + (b * T + j) * kv_ld + h * D for keys j < Tk (the caller offsets
// k / v to the first key);  out rows [B * Tq][8 * D]
template <int D, int ATT_WAVES>
__global__ __launch_bounds__(64 * ATT_WAVES, D == 128 ? HMV_ATT_OCC : 1) void attention_mfma_kernel(const float *__restrict__ q, int q_ld, int q_bstride,
        const float *__restrict__ k, const float *__restrict__ v, int kv_ld, int T, int Tq, int Tk, int nqb, float *__restrict__ out, int pairs) {
    using SH = AttShape<D, ATT_WAVES>;
    constexpr int NC = SH::NC, NU = D / 8, NV = D / 32;   // K vectors per lane, V vectors per lane and quarter chunk
    extern __shared__ __attribute__((aligned(16))) float att_smem[];
    const int qblk = blockIdx.x % nqb, bh = blockIdx.x / nqb;
    const int b = bh >> 3, h = bh & 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const size_t ld = (size_t)kv_ld;
    const float *qb = q + (size_t)b * q_bstride * q_ld + h * D;
    const float *kb = k + (size_t)b * T * ld + h * D, *vb = v + (size_t)b * T * ld + h * D;
    const int nkc = (Tk + 31) >> 5;
    const float scale = D == 128 ? 0.08838834764831845f : 0.0625f;  // D ** -0.5
    float *sQ = att_smem;                                                   // [32][LDK]
    float *sVw = att_smem + SH::Q_FLOATS + wave * SH::V_FLOATS;             // [8][D], this wave's

    // every global load of a wave's first chunk is issued before anything waits
    constexpr bool PREFETCH = D == 128;
    f32x4 kf[NU], va[NV], vb_[NV];
    int kc = wave;
#define ATT_LOAD_K(KC)                                                                                  \
    do {                                                                                                \
        const int key_ = (KC) * 32 + l31;                                                               \
        const bool kv_ = key_ < Tk;                                                                     \
        const float *krow_ = kb + (size_t)(kv_ ? key_ : 0) * ld + 4 * kh;                               \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                \
            kf[u] = *reinterpret_cast<const f32x4 *>(krow_ + 8 * u);                                    \
            if (!kv_) kf[u] = f32x4{0.f, 0.f, 0.f, 0.f};                                                \
        }                                                                                               \
    } while (0)
    // 8 keys x D channels: NV coalesced 16-byte vectors per lane (keys >= Tk are zeros)
#define ATT_LOAD_V(KEY0, VR)                                                                            \
    do {                                                                                                \
        _Pragma("unroll") for (int it = 0; it < NV; ++it) {                                             \
            const int idx_ = it * 64 + lane, k2_ = (KEY0) + idx_ / (D / 4);                             \
            VR[it] = f32x4{0.f, 0.f, 0.f, 0.f};                                                         \
            if (k2_ < Tk) VR[it] = *reinterpret_cast<const f32x4 *>(vb + (size_t)k2_ * ld + 4 * (idx_ % (D / 4))); \
        }                                                                                               \
    } while (0)
#define ATT_STORE_V(VR)                                                                                 \
    do {                                                                                                \
        __builtin_amdgcn_wave_barrier();                                                                \
        _Pragma("unroll") for (int it = 0; it < NV; ++it) {                                             \
            const int idx_ = it * 64 + lane;                                                            \
            *reinterpret_cast<f32x4 *>(&sVw[(idx_ / (D / 4)) * D + 4 * (idx_ % (D / 4))]) = VR[it];     \
        }                                                                                               \
        __builtin_amdgcn_wave_barrier();                                                                \
    } while (0)
    // O^T += V^T P^T over the 8 keys of quarter G: the k-pair of step e2 is (key 8G + e2, key 8G + e2 + 4) = P register 4G + e2
#define ATT_PV(G)                                                                                       \
    do {                                                                                                \
        _Pragma("unroll") for (int e2 = 0; e2 < 4; ++e2) {                                              \
            const float *vrow = &sVw[(4 * kh + e2) * D + l31];                                          \
            _Pragma("unroll") for (int c = 0; c < NC; ++c)                                              \
                o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * c], sacc[4 * (G) + e2], o[c], 0, 0, 0); \
        }                                                                                               \
    } while (0)
    if (kc < nkc) {
        ATT_LOAD_K(kc);
        ATT_LOAD_V(kc * 32, va);
    }
    // Q block (rows >= Tq are zeros), shared by the waves
#pragma unroll
    for (int it = 0; it < (8 * D) / (64 * ATT_WAVES); ++it) {
        const int idx = it * (64 * ATT_WAVES) + tid, r = idx / (D / 4), c4 = idx % (D / 4), row = qblk * 32 + r;
        f32x4 qv = {0.f, 0.f, 0.f, 0.f};
        if (row < Tq) qv = *reinterpret_cast<const f32x4 *>(qb + (size_t)row * q_ld + 4 * c4);
        *reinterpret_cast<f32x4 *>(&sQ[r * SH::LDK + 4 * c4]) = qv;
    }
    __syncthreads();

    f32x16 o[NC];   // o[c][e] on lane (q, half): O[q][32c + (e&3) + 8(e>>2) + 4 half]
    float m_run = -INFINITY, l_run = 0.f;   // of query l31, over the keys this lane has seen (its half of every chunk)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[c][e] = 0.f;

#define ATT_CHUNK(KC)                                                                                   \
    do {                                                                                                \
        f32x16 sacc;                                                                                    \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) sacc[e] = 0.f;                                   \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                \
            const f32x4 qf = *reinterpret_cast<const f32x4 *>(&sQ[l31 * SH::LDK + 8 * u + 4 * kh]);     \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                               \
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[u][e], qf[e], sacc, 0, 0, 0);            \
        }                                                                                               \
        /* the key rows of this wave's NEXT chunk fly during the softmax and the P V products of this one (their registers are free) */ \
        if (PREFETCH && (KC) + ATT_WAVES < nkc) ATT_LOAD_K((KC) + ATT_WAVES);                           \
        ATT_LOAD_V((KC) * 32 + 8, vb_);   /* the next 8 keys fly during the softmax */                  \
        /* register e = key (e&3) + 8(e>>2) + 4 half of the chunk, for query l31 */                     \
        float mx = -INFINITY;                                                                           \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                \
            const bool kv_ = (KC) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh < Tk;                          \
            sacc[e] = kv_ ? sacc[e] * scale : -INFINITY;                                                \
            mx = fmaxf(mx, sacc[e]);                                                                    \
        }                                                                                               \
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));          /* finite: the chunk has >= 1 valid key */     \
        const float m_new = fmaxf(m_run, mx);                                                           \
        const float alpha = expf(m_run - m_new);         /* exp(-inf) = 0 on the first chunk */         \
        float psum = 0.f;                                                                               \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                \
            sacc[e] = expf(sacc[e] - m_new);             /* exp(-inf) = 0 for keys >= Tk */             \
            psum += sacc[e];                                                                            \
        }                                                                                               \
        l_run = l_run * alpha + psum;                                                                   \
        m_run = m_new;                                                                                  \
        _Pragma("unroll") for (int c = 0; c < NC; ++c)                                                  \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) o[c][e] *= alpha;                            \
        /* 8 keys at a time through the wave's LDS buffer; the loads run two quarters ahead */          \
        ATT_STORE_V(va);                                                                                \
        ATT_LOAD_V((KC) * 32 + 16, va);                                                                 \
        ATT_PV(0);                                                                                      \
        ATT_STORE_V(vb_);                                                                               \
        ATT_LOAD_V((KC) * 32 + 24, vb_);                                                                \
        ATT_PV(1);                                                                                      \
        ATT_STORE_V(va);                                                                                \
        if (PREFETCH && (KC) + ATT_WAVES < nkc) ATT_LOAD_V(((KC) + ATT_WAVES) * 32, va);   /* ... and its first 8 value rows */ \
        ATT_PV(2);                                                                                      \
        ATT_STORE_V(vb_);                                                                               \
        ATT_PV(3);                                                                                      \
        __builtin_amdgcn_wave_barrier();                                                                \
    } while (0)
    if (kc < nkc) {
        // 128-wide heads: a chunk requests the operands of this wave's next one (round 4); the 256-wide ones have no registers for that
        if constexpr (PREFETCH) {
            for (; kc < nkc; kc += ATT_WAVES) ATT_CHUNK(kc);
        } else {
            ATT_CHUNK(kc);
            for (kc += ATT_WAVES; kc < nkc; kc += ATT_WAVES) {
                ATT_LOAD_K(kc);
                ATT_LOAD_V(kc * 32, va);
                ATT_CHUNK(kc);
            }
        }
    }
#undef ATT_CHUNK
#undef ATT_LOAD_K
#undef ATT_LOAD_V
#undef ATT_STORE_V
#undef ATT_PV

    // ---- merge the 4 waves' partials in wave order: out = sum_w O_w e^(m_w - M) / sum_w l_w e^(m_w - M)
    __syncthreads();   // the merge area aliases the loop's buffers
    float *sO = att_smem, *sM = att_smem + ATT_WAVES * NC * 16 * 64, *sL = sM + ATT_WAVES * 32;
    l_run += __shfl_xor(l_run, 32, 64);
    if (kh == 0) { sM[wave * 32 + l31] = m_run; sL[wave * 32 + l31] = l_run; }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) sO[((wave * NC + c) * 16 + e) * 64 + lane] = o[c][e];
    __syncthreads();
    {
        float M = sM[l31];
#pragma unroll
        for (int w = 1; w < ATT_WAVES; ++w) M = fmaxf(M, sM[w * 32 + l31]);
        float a[ATT_WAVES], den = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) {
            a[w] = expf(sM[w * 32 + l31] - M);   // exp(-inf) = 0 for a wave that had no chunk
            den += sL[w * 32 + l31] * a[w];
        }
        const float inv = 1.f / den;
        const int row = qblk * 32 + l31;
#pragma unroll
        for (int c = wave; c < NC; c += ATT_WAVES) {   // this wave finishes channel blocks wave, wave + 4, ..
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 r4;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    float num = 0.f;
#pragma unroll
                    for (int w = 0; w < ATT_WAVES; ++w) num += sO[((w * NC + c) * 16 + 4 * g + e2) * 64 + lane] * a[w];
                    r4[e2] = num * inv;
                }
                if (row < Tq) {
                    if (pairs) {   // the rows as (hi, lo) fp16 pairs [hi 8 D | lo 8 D] for a split-pair to_out GEMM (gemm_x3.hip): split_f16's arithmetic
                        f16x4 hi4, lo4;
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            _Float16 a_, b_;
                            split_f16(r4[e2], a_, b_);
                            hi4[e2] = a_; lo4[e2] = b_;
                        }
                        _Float16 *pr = reinterpret_cast<_Float16 *>(out) + ((size_t)b * Tq + row) * (16 * D) + h * D + 32 * c + 8 * g + 4 * kh;
                        *reinterpret_cast<f16x4 *>(pr) = hi4;
                        *reinterpret_cast<f16x4 *>(pr + 8 * D) = lo4;
                    } else {
                        *reinterpret_cast<f32x4 *>(out + ((size_t)b * Tq + row) * (8 * D) + h * D + 32 * c + 8 * g + 4 * kh) = r4;
                    }
                }
            }
        }
    }
}
// ------------------------------------------------------------------ attention on the fp16 matrix cores over (hi, lo) pairs (round 4)
// The fp16-kernel modes' form of attention_mfma_kernel<128>: same decomposition (one workgroup per (sample, head, 32-query block), the
// 32-key chunks dealt to four waves, S^T = K Q^T so that a query row is a lane, P stays in registers, partials merged through LDS in
// wave order), but both products run on v_mfma_f32_32x32x16_f16 with every operand split into fp16 (hi, lo) pairs, hi = fp16(x),
// lo = fp16(x - hi), and three MFMAs per k16 step -- hi.hi + lo.hi + hi.lo, fp32 accumulation: fp32-equivalent (2^-22 per operand) like
// the q / k / v projections in front of it (gemm_x3.hip) at 8 + 8 instead of 64 + 64 matrix-pipe cycles per 16 channels / keys.  The
// exact-fp32 kernel spends 41 of its 88 us at cfg-3 in the matrix pipe (tools/att_probe.py).
//   S^T   A = K rows (the lane's own key row from global memory), B = Q block from LDS (staged once per workgroup)
//   P V   B = P^T straight from the accumulator registers (slot j of step s = register 8 s + j = key 16 s + 8 (j >> 2) + 4 kh + (j & 3)),
//         A = V^T: the 16 keys of a step sit row-major [key][channel] in a per-wave LDS buffer and come back
//         column-major through ds_read_b64_tr_b16 -- lane 4 q + p of a 16-lane group supplies the address of row q, channels 4 p .. 4 p + 3,
//         lane i receives channel i of the four rows -- two reads per operand (keys 4 kh + 0 .. 3 and 8 + 4 kh + 0 .. 3).  Row stride
//         320 bytes: the eight 32-byte row pieces of a 32-lane half fall on distinct banks.
typedef __fp16 atr4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
constexpr int AX_QLD = 136, AX_VLD = 160;                      // halfs per Q row (272 B) / V row (320 B) in LDS
constexpr int AX_Q_HALFS = 2 * 32 * AX_QLD, AX_V_HALFS = 2 * 16 * AX_VLD;   // [plane][32 queries] ; per wave [plane][16 keys]
constexpr int AX_WAVES = 4;
constexpr int AX_LOOP_BYTES = (AX_Q_HALFS + AX_WAVES * AX_V_HALFS) * 2;
constexpr int AX_MERGE_BYTES = (AX_WAVES * 4 * 16 * 64 + 2 * AX_WAVES * 32) * 4;
constexpr int AX_LDS_BYTES = AX_LOOP_BYTES > AX_MERGE_BYTES ? AX_LOOP_BYTES : AX_MERGE_BYTES;

// split_f16 on four values (the clamp as ONE v_med3_f32: fminf / fmaxf canonicalise their operands first; same result for finite inputs)
__device__ __forceinline__ void ax_split4(const f32x4 v, f16x4 &hi, f16x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float c = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
        hi[e] = (_Float16)c;
        lo[e] = (_Float16)(c - (float)hi[e]);
    }
}
__device__ __forceinline__ f16x8 ax_cat(const f16x4 a, const f16x4 b) { return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// q / k / v arrive as rows of (hi, lo) fp16 pairs [hi | lo] -- what the split-pair projection GEMMs write with their pair epilogue (conv_igemm
// out_split, gemm_x3k16) -- so every operand is loaded as the MFMAs take it (splitting fp32 rows here cost ~550 of a chunk's ~1 050 vector
// instructions, six times per key row at cfg-3).  Pointers address the hi plane in halfs, lo_off halfs further the lo plane; a lane's eight
// slots of step u are the eight consecutive channels 16 u + 8 kh + j for both operands.
__global__ __launch_bounds__(64 * AX_WAVES, 2) void attention_x3_kernel(const _Float16 *__restrict__ q, int q_ld, int q_bstride,
        const _Float16 *__restrict__ k, const _Float16 *__restrict__ v, int kv_ld, int lo_off, int T, int Tq, int Tk, int nqb, float *__restrict__ out, int pairs) {
    constexpr int D = 128, NC = 4;
    extern __shared__ __attribute__((aligned(16))) char ax_smem[];
    const int qblk = blockIdx.x % nqb, bh = blockIdx.x / nqb;
    const int b = bh >> 3, h = bh & 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const size_t ld = (size_t)kv_ld;
    const _Float16 *qb = q + (size_t)b * q_bstride * q_ld + h * D;
    const _Float16 *kb = k + (size_t)b * T * ld + h * D, *vb = v + (size_t)b * T * ld + h * D;
    const int nkc = (Tk + 31) >> 5;
    const float scale = 0.08838834764831845f;   // 128 ** -0.5
    _Float16 *sQ = reinterpret_cast<_Float16 *>(ax_smem);                                   // [2][32][AX_QLD]
    _Float16 *sVw = sQ + AX_Q_HALFS + wave * AX_V_HALFS;                                    // [2][16][AX_VLD], this wave's

    f16x8 kh8[8], kl8[8], vh8[4], vl8[4];   // the chunk's key rows (8 steps x (hi, lo)); 16 of its value rows (16 keys x 128 channels = 4 + 4 vectors per lane)
    int kc = wave;
    // (keys >= Tk read the last valid row: their logits are set to -inf and their P to exactly 0 below, so only finiteness matters)
#define AX_LOAD_K(KC)                                                                                   \
    do {                                                                                                \
        const _Float16 *krow_ = kb + (size_t)min((KC) * 32 + l31, Tk - 1) * ld + 8 * kh;                \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                 \
            kh8[u] = *reinterpret_cast<const f16x8 *>(krow_ + 16 * u);                                  \
            kl8[u] = *reinterpret_cast<const f16x8 *>(krow_ + lo_off + 16 * u);                         \
        }                                                                                               \
    } while (0)
#define AX_LOAD_V(KEY0)                                                                                 \
    do {                                                                                                \
        _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                              \
            const int k2_ = min((KEY0) + 4 * it + (lane >> 4), Tk - 1);   /* unit it * 64 + lane = key 4 it + (lane >> 4), channels 8 (lane & 15) .. */ \
            vh8[it] = *reinterpret_cast<const f16x8 *>(vb + (size_t)k2_ * ld + 8 * (lane & 15));       \
            vl8[it] = *reinterpret_cast<const f16x8 *>(vb + (size_t)k2_ * ld + lo_off + 8 * (lane & 15)); \
        }                                                                                               \
    } while (0)
    // 16 keys -> the wave's LDS buffer, row-major
#define AX_STORE_V()                                                                                    \
    do {                                                                                                \
        __builtin_amdgcn_wave_barrier();                                                                \
        _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                              \
            *reinterpret_cast<f16x8 *>(&sVw[(4 * it + (lane >> 4)) * AX_VLD + 8 * (lane & 15)]) = vh8[it]; \
            *reinterpret_cast<f16x8 *>(&sVw[(16 + 4 * it + (lane >> 4)) * AX_VLD + 8 * (lane & 15)]) = vl8[it]; \
        }                                                                                               \
        __builtin_amdgcn_wave_barrier();                                                                \
    } while (0)
    // O^T += V^T P^T over the 16 keys of step S_: per 32-channel block two transposed reads per plane, three MFMAs
#define AX_PV(S_)                                                                                       \
    do {                                                                                                \
        _Pragma("unroll") for (int c = 0; c < NC; ++c) {                                                \
            const _Float16 *t0 = sVw + (4 * kh + ((lane & 15) >> 2)) * AX_VLD + 32 * c + 16 * ((lane >> 4) & 1) + 4 * (lane & 3); \
            const f16x4 h0 = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) atr4 *)(t0))); \
            const f16x4 h1 = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) atr4 *)(t0 + 8 * AX_VLD))); \
            const f16x4 l0 = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) atr4 *)(t0 + 16 * AX_VLD))); \
            const f16x4 l1 = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) atr4 *)(t0 + 24 * AX_VLD))); \
            const f16x8 ah_ = ax_cat(h0, h1), al_ = ax_cat(l0, l1);                                     \
            o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, ph[S_], o[c], 0, 0, 0);                  \
            o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_, ph[S_], o[c], 0, 0, 0);                  \
            o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_, pl[S_], o[c], 0, 0, 0);                  \
            if (c & 1) __builtin_amdgcn_sched_barrier(0);   /* (two blocks' operands in flight at a time: registers) */ \
        }                                                                                               \
    } while (0)
    if (kc < nkc) {
        AX_LOAD_K(kc);
        AX_LOAD_V(kc * 32);
    }
    // Q block (rows >= Tq are zeros) -> LDS, shared by the waves: the rows as they are (slot j of step u, half kh = channel 16 u + 8 kh + j)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = it * 256 + tid, r = idx >> 4, c8 = 8 * (idx & 15), row = qblk * 32 + r;
        f16x8 qh = {0, 0, 0, 0, 0, 0, 0, 0}, ql = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < Tq) {
            qh = *reinterpret_cast<const f16x8 *>(qb + (size_t)row * q_ld + c8);
            ql = *reinterpret_cast<const f16x8 *>(qb + (size_t)row * q_ld + lo_off + c8);
        }
        *reinterpret_cast<f16x8 *>(&sQ[r * AX_QLD + c8]) = qh;
        *reinterpret_cast<f16x8 *>(&sQ[(32 + r) * AX_QLD + c8]) = ql;
    }
    __syncthreads();

    f32x16 o[NC];   // o[c][e] on lane (q, half): O[q][32c + (e&3) + 8(e>>2) + 4 half]
    float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[c][e] = 0.f;

    for (; kc < nkc; kc += AX_WAVES) {
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f16x8 ah = kh8[u], al = kl8[u];
            const f16x8 bh_ = *reinterpret_cast<const f16x8 *>(&sQ[l31 * AX_QLD + u * 16 + kh * 8]);
            const f16x8 bl_ = *reinterpret_cast<const f16x8 *>(&sQ[(32 + l31) * AX_QLD + u * 16 + kh * 8]);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh_, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh_, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl_, sacc, 0, 0, 0);
            if (u & 1) __builtin_amdgcn_sched_barrier(0);   // (two steps' operands at a time: the splits of all eight would not fit the registers)
        }
        // the key rows of this wave's NEXT chunk fly during the softmax and the P V products of this one (their registers are free)
        if (kc + AX_WAVES < nkc) AX_LOAD_K(kc + AX_WAVES);
        // register e = key (e&3) + 8(e>>2) + 4 half of the chunk, for query l31
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool kv_ = kc * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh < Tk;
            sacc[e] = kv_ ? sacc[e] * scale : -INFINITY;
            mx = fmaxf(mx, sacc[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));          // finite: the chunk has >= 1 valid key
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);         // exp(-inf) = 0 on the first chunk
        float psum = 0.f;
        f16x8 ph[2], pl[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float pe = expf(sacc[e] - m_new);      // exp(-inf) = 0 for keys >= Tk
            psum += pe;
            const _Float16 a = (_Float16)pe;               // 0 <= pe <= 1: no clamp
            ph[e >> 3][e & 7] = a;
            pl[e >> 3][e & 7] = (_Float16)(pe - (float)a);
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[c][e] *= alpha;
        AX_STORE_V();
        AX_LOAD_V(kc * 32 + 16);              // the second 16 value rows fly under the first 16 keys' products
        AX_PV(0);
        AX_STORE_V();
        if (kc + AX_WAVES < nkc) AX_LOAD_V((kc + AX_WAVES) * 32);   // ... and the next chunk's first 16 value rows under the second
        AX_PV(1);
        __builtin_amdgcn_wave_barrier();
    }
#undef AX_LOAD_K
#undef AX_LOAD_V
#undef AX_STORE_V
#undef AX_PV

    // ---- merge the 4 waves' partials in wave order (attention_mfma_kernel's): out = sum_w O_w e^(m_w - M) / sum_w l_w e^(m_w - M)
    __syncthreads();   // the merge area aliases the loop's buffers
    float *att_smem = reinterpret_cast<float *>(ax_smem);
    float *sO = att_smem, *sM = att_smem + AX_WAVES * NC * 16 * 64, *sL = sM + AX_WAVES * 32;
    l_run += __shfl_xor(l_run, 32, 64);
    if (kh == 0) { sM[wave * 32 + l31] = m_run; sL[wave * 32 + l31] = l_run; }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) sO[((wave * NC + c) * 16 + e) * 64 + lane] = o[c][e];
    __syncthreads();
    {
        float M = sM[l31];
#pragma unroll
        for (int w = 1; w < AX_WAVES; ++w) M = fmaxf(M, sM[w * 32 + l31]);
        float a[AX_WAVES], den = 0.f;
#pragma unroll
        for (int w = 0; w < AX_WAVES; ++w) {
            a[w] = expf(sM[w * 32 + l31] - M);   // exp(-inf) = 0 for a wave that had no chunk
            den += sL[w * 32 + l31] * a[w];
        }
        const float inv = 1.f / den;
        const int row = qblk * 32 + l31;
        const int c = wave;   // this wave finishes channel block `wave`
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 r4;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                float num = 0.f;
#pragma unroll
                for (int w = 0; w < AX_WAVES; ++w) num += sO[((w * NC + c) * 16 + 4 * g + e2) * 64 + lane] * a[w];
                r4[e2] = num * inv;
            }
            if (row < Tq) {
                if (pairs) {
                    f16x4 hi4, lo4;
                    ax_split4(r4, hi4, lo4);
                    _Float16 *pr = reinterpret_cast<_Float16 *>(out) + ((size_t)b * Tq + row) * (16 * D) + h * D + 32 * c + 8 * g + 4 * kh;
                    *reinterpret_cast<f16x4 *>(pr) = hi4;
                    *reinterpret_cast<f16x4 *>(pr + 8 * D) = lo4;
                } else {
                    *reinterpret_cast<f32x4 *>(out + ((size_t)b * Tq + row) * (8 * D) + h * D + 32 * c + 8 * g + 4 * kh) = r4;
                }
            }
        }
    }
}

template <int D, int W>
static hipError_t launch_attention_w(const float *q, int q_ld, int q_bstride, const float *k, const float *v, int kv_ld, int B, int T, int Tq,
                                     int Tk, float *out, hipStream_t s, int pairs, int nqb) {
    static bool configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const int lds = AttShape<D, W>::LDS_FLOATS * (int)sizeof(float);
    if (!configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attention_mfma_kernel<D, W>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        configured[dev] = true;
    }
    hipLaunchKernelGGL((attention_mfma_kernel<D, W>), dim3((unsigned)B * 8 * nqb), dim3(64 * W), lds, s, q, q_ld, q_bstride, k, v, kv_ld, T, Tq,
                       Tk, nqb, out, pairs);
    return hipGetLastError();
}
template <int D>
static hipError_t launch_attention_any(const float *q, int q_ld, int q_bstride, const float *k, const float *v, int kv_ld, int B, int T, int Tq,
                                       int Tk, float *out, hipStream_t s, int pairs = 0) {
    if (Tk <= 0 || Tq <= 0 || B <= 0) return hipErrorInvalidValue;
    const int nqb = (Tq + 31) >> 5;
    // (measured in round 4, tools/att_probe.py: TWO waves per workgroup -- six chunks as 3 : 3 instead of 2 : 2 : 1 : 1, four workgroups per CU --
    // take 86.4 us where four take 88.7 at B = 32 x 168 tokens and 22.7 against 18.4 us at B = 1: the launch is bound by its fp32 MFMA time
    // plus the latencies two waves per SIMD cannot hide, not by the split)
    return launch_attention_w<D, 4>(q, q_ld, q_bstride, k, v, kv_ld, B, T, Tq, Tk, out, s, pairs, nqb);
}
hipError_t launch_attention(const float *qkv, int B, int T, int Tq, int koff, int Tk, float *out, hipStream_t s, int pairs, int x3) {
    if (x3) {   // the fp16-kernel modes: qkv holds rows of (hi, lo) fp16 pairs [hi 3072 | lo 3072] (the projection GEMMs' pair epilogue) and both
                // products run on the fp16 matrix cores (attention_x3_kernel); by the arithmetic mode alone, never by a size
        if (Tk <= 0 || Tq <= 0 || B <= 0) return hipErrorInvalidValue;
        static bool configured[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!configured[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(attention_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, AX_LDS_BYTES);
            if (e != hipSuccess) return e;
            configured[dev] = true;
        }
        const int nqb = (Tq + 31) >> 5;
        const _Float16 *ph = reinterpret_cast<const _Float16 *>(qkv);
        hipLaunchKernelGGL(attention_x3_kernel, dim3((unsigned)B * 8 * nqb), dim3(64 * AX_WAVES), AX_LDS_BYTES, s, ph, 6 * 1024, T,
                           ph + (size_t)koff * 6144 + 1024, ph + (size_t)koff * 6144 + 2048, 6 * 1024, 3 * 1024, T, Tq, Tk, nqb, out, pairs);
        return hipGetLastError();
    }
    return launch_attention_any<128>(qkv, 3 * 1024, T, qkv + (size_t)koff * 3072 + 1024, qkv + (size_t)koff * 3072 + 2048, 3 * 1024, B, T, Tq, Tk, out, s, pairs);
}

// ------------------------------------------------------------------ learnable-query fusion
// MultiHeadAttentionLearnableQuery.forward, layers.py:273-301: x = pos_embed(x) at the top of EVERY block.
__global__ void add_pe_kernel(const float *__restrict__ x, int ldx, int T, int d, const float *__restrict__ pe, float *__restrict__ y,
                              int ldy, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const size_t r = i / ldy;
        const int c = (int)(i - r * ldy);
        y[i] = c < d ? x[r * ldx + c] + pe[(r % T) * (size_t)d + c] : 0.f;
    }
}
hipError_t launch_add_pe(const float *x, int ldx, int rows, int T, int d, const float *pe, float *y, int ldy, hipStream_t s) {
    const size_t total = (size_t)rows * ldy;
    if (!total) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_pe_kernel, dim3(grid), dim3(256), 0, s, x, ldx, T, d, pe, y, ldy, total);
    return hipGetLastError();
}

// Attention with 256-wide heads (layers.py:241, 284-291): the round-2 kernel, kept as the A/B reference of the MFMA one.  The
// plain wave-per-query-row form: a lane owns 4 of the 256 head dimensions (one 16-byte vector), a score is a wave
// reduction, softmax runs online (running max / sum are wave-uniform), K and V rows stream from L2 as 1 KiB wave loads.
// Four query rows share each K / V row load.
constexpr int LQ_QB = 4;
__global__ __launch_bounds__(256) void attention_d256_kernel(const float *__restrict__ q, int q_ld, int q_bstride,
                                                             const float *__restrict__ k, const float *__restrict__ v, int kv_ld, int T,
                                                             int Tq, float *__restrict__ out) {
    const int b = blockIdx.x >> 3, h = blockIdx.x & 7;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float scale = 0.0625f;   // 256 ** -0.5
    const float *kb = k + (size_t)b * T * kv_ld + h * 256 + 4 * lane, *vb = v + (size_t)b * T * kv_ld + h * 256 + 4 * lane;
    for (int i0 = wave * LQ_QB; i0 < Tq; i0 += 4 * LQ_QB) {
        f32x4 qv[LQ_QB], o[LQ_QB];
        float m[LQ_QB], l[LQ_QB];
#pragma unroll
        for (int r = 0; r < LQ_QB; ++r) {
            const int i = i0 + r < Tq ? i0 + r : Tq - 1;   // rows past the end repeat the last one and are not stored
            qv[r] = *reinterpret_cast<const f32x4 *>(q + ((size_t)b * q_bstride + i) * q_ld + h * 256 + 4 * lane);
            o[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            m[r] = -INFINITY;
            l[r] = 0.f;
        }
        for (int j = 0; j < T; ++j) {
            const f32x4 kv = *reinterpret_cast<const f32x4 *>(kb + (size_t)j * kv_ld);
            const f32x4 vv = *reinterpret_cast<const f32x4 *>(vb + (size_t)j * kv_ld);
#pragma unroll
            for (int r = 0; r < LQ_QB; ++r) {
                const float sc = wave_sum(qv[r].x * kv.x + qv[r].y * kv.y + qv[r].z * kv.z + qv[r].w * kv.w) * scale;
                const float mn = fmaxf(m[r], sc), alpha = expf(m[r] - mn), pj = expf(sc - mn);   // exp(-inf) = 0 on the first key
                l[r] = l[r] * alpha + pj;
                o[r] = o[r] * alpha + vv * pj;
                m[r] = mn;
            }
        }
#pragma unroll
        for (int r = 0; r < LQ_QB; ++r)
            if (i0 + r < Tq) *reinterpret_cast<f32x4 *>(out + ((size_t)b * Tq + i0 + r) * 2048 + h * 256 + 4 * lane) = o[r] * (1.f / l[r]);
    }
}
hipError_t launch_attention_d256(const float *q, int q_ld, int q_bstride, const float *k, const float *v, int kv_ld, int B, int T,
                                 int Tq, float *out, hipStream_t s, int pairs) {
    if (T <= 0 || Tq <= 0) return hipErrorInvalidValue;
    // QK^T and PV on the fp32 matrix cores like the 128-wide heads (the same kernel template, D = 256); HMV_LQ_SCALAR_ATT=1
    // keeps the wave-per-query-row form below (A/B runs, read per launch)
    if (pairs || !HMV_DEV_ENV("HMV_LQ_SCALAR_ATT")) return launch_attention_any<256>(q, q_ld, q_bstride, k, v, kv_ld, B, T, Tq, T, out, s, pairs);
    hipLaunchKernelGGL(attention_d256_kernel, dim3(B * 8), dim3(256), 0, s, q, q_ld, q_bstride, k, v, kv_ld, T, Tq, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ split-K reduction
// The token GEMMs of a small batch (to_out: M = B*V*21 rows, N = d, K = 1024) tile into a few dozen workgroups with a long
// serial reduction.  The engine cuts K into S slices (S times the workgroups, 1/S the k-steps), each slice writes a plain
// partial product, and this kernel adds the slices in index order (deterministic) and applies the GEMM's epilogue.
__global__ void splitk_reduce_kernel(const float *__restrict__ slab, int S, size_t slice, int lds, int N, const float *__restrict__ bias,
                                     const float *__restrict__ res, int ldr, int rg_out, int rg_in, int act, float *__restrict__ out,
                                     int ldc, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const size_t r = i / N;
        const int c = (int)(i - r * N);
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += slab[s * slice + r * lds + c];
        v += bias[c];
        if (res) {
            const size_t rr = rg_out ? (r / rg_out) * rg_in + (r % rg_out) : r;
            v += res[rr * ldr + c];
        }
        if (act == ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        else if (act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
        out[r * ldc + c] = v;
    }
}
hipError_t launch_splitk_reduce(const float *slab, int S, int rows, int lds, int N, const float *bias, const float *res, int ldr,
                                int rg_out, int rg_in, int act, float *out, int ldc, hipStream_t s) {
    const size_t total = (size_t)rows * N;
    if (!total) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, slab, S, (size_t)rows * lds, lds, N, bias, res, ldr, rg_out,
                       rg_in, act, out, ldc, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------ ChebConv mix
// layers.py:387-403: sum_k T_k (X W_k) + b, the X W_k products come from one GEMM with N = 3*co.
// One workgroup per (sample, 32-channel slice): 8 joints x 32 channels per pass (the previous one-workgroup-per-sample form
// left a small batch on B CUs, 21 serial outputs per thread).  Same sums in the same order.
__global__ void cheb_mix_kernel(const float *__restrict__ y, int ldy, int co, const float *__restrict__ tk,
                                const float *__restrict__ bias, int leaky, float *__restrict__ out, int ldo) {
    __shared__ float sT[3 * 21 * 21];
    for (int i = threadIdx.x; i < 3 * 21 * 21; i += blockDim.x) sT[i] = tk[i];
    __syncthreads();
    const int b = blockIdx.x, o = blockIdx.y * 32 + (threadIdx.x & 31);
    if (o >= co) return;
    const float *yb = y + (size_t)b * 21 * ldy;
    for (int i = threadIdx.x >> 5; i < 21; i += 8) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 21; ++j) part += sT[(k * 21 + i) * 21 + j] * yb[(size_t)j * ldy + k * co + o];
            acc += part;
        }
        float v = acc + bias[o];
        if (leaky) v = v > 0.f ? v : 0.01f * v;
        out[((size_t)b * 21 + i) * ldo + o] = v;
    }
}
hipError_t launch_cheb_mix(const float *y, int ldy, int B, int co, const float *tk, const float *bias, int leaky,
                           float *out, int ldo, hipStream_t s) {
    hipLaunchKernelGGL(cheb_mix_kernel, dim3(B, (co + 31) / 32), dim3(256), 0, s, y, ldy, co, tk, bias, leaky, out, ldo);
    return hipGetLastError();
}

// ------------------------------------------------------------------ capture helpers
__global__ void nhwc_to_nchw_kernel(const float *__restrict__ in, float *__restrict__ out, int HW, int C, int ld, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {  // i indexes the NCHW output
        const int p = (int)(i % HW);
        const size_t t = i / HW;
        const int c = (int)(t % C);
        const size_t n = t / C;
        out[i] = in[(n * HW + p) * ld + c];
    }
}
__global__ void nhwc_f16_to_nchw_kernel(const _Float16 *__restrict__ in, float *__restrict__ out, int HW, int C, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int p = (int)(i % HW);
        const size_t t = i / HW;
        const int c = (int)(t % C);
        const size_t n = t / C;
        out[i] = (float)in[(n * HW + p) * C + c];
    }
}
hipError_t launch_nhwc_f16_to_nchw(const void *in, float *out, int N, int H, int W, int C, hipStream_t s) {
    const size_t total = (size_t)N * H * W * C;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nhwc_f16_to_nchw_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const _Float16 *>(in), out, H * W, C, total);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw(const float *in, float *out, int N, int H, int W, int C, hipStream_t s, int ld) {
    const size_t total = (size_t)N * H * W * C;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid), dim3(256), 0, s, in, out, H * W, C, ld ? ld : C, total);
    return hipGetLastError();
}

__global__ void copy_rows_kernel(const float *__restrict__ in, int ldi, float *__restrict__ out, int ldo, int cols,
                                 size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const size_t r = i / cols;
        const int c = (int)(i - r * cols);
        out[r * ldo + c] = in[r * ldi + c];
    }
}
hipError_t launch_copy_rows(const float *in, int ldi, float *out, int ldo, int rows, int cols, hipStream_t s) {
    const size_t total = (size_t)rows * cols;
    if (!total) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid), dim3(256), 0, s, in, ldi, out, ldo, cols, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------ camera frames: uint8 HWC -> model input
// The reference prepares every view on the host (datasets/ho3d.py:35-40, 136-149; datasets/utils.py:40-77):
//   crop_and_pad_image(frame, box) -> ToTensor (/255) -> Resize((S, S), antialias=True) -> Normalize(mean, std).
// Here it is one kernel from the raw frames to the stem conv's channels-last input (NHWC4 fp32 / NHWC8 fp16): a
// lane owns one output pixel, walks the separable triangle filter of torch's antialiased bilinear resize
// (aten `_compute_indices_min_size_weights_aa`: support = max(scale, 1), weights 1 - |x| / max(scale, 1),
// renormalised over the taps that fall inside the window) and reads window pixels that leave the frame as 0.
// An empty box (x2 <= x1 or y2 <= y1; also a window wider than 65536 px) is the reference's "no visible joint" black view.
struct FrameNorm { float mean[3], inv_std[3]; };

__device__ __forceinline__ void aa_span(int o, int in_size, float scale, float &center, float &invscale, int &first, int &count) {
    const float support = scale >= 1.f ? scale : 1.f;
    invscale = scale >= 1.f ? 1.f / scale : 1.f;
    center = scale * ((float)o + 0.5f);
    first = max((int)(center - support + 0.5f), 0);
    count = min((int)(center + support + 0.5f), in_size) - first;
}

// one output pixel of the antialiased crop-resize + normalisation (datasets' transform, see hmv_forward_frames in handmv.h)
__device__ __forceinline__ void frame_pixel(const uint8_t *__restrict__ frames, const int *__restrict__ boxes, size_t n, int oy, int ox,
                                            int Hf, int Wf, int S_h, int S_w, const FrameNorm &nm, float &pr, float &pg, float &pb) {
        const int x1 = boxes[n * 4], y1 = boxes[n * 4 + 1], x2 = boxes[n * 4 + 2], y2 = boxes[n * 4 + 3];
        float acc[3] = {0.f, 0.f, 0.f};
        // windows wider than kMaxWindow px are treated like empty ones: a garbage box must not turn into an unbounded loop
        constexpr int kMaxWindow = 1 << 16;
        const long long cwl = (long long)x2 - x1, chl = (long long)y2 - y1;
        if (cwl > 0 && chl > 0 && cwl <= kMaxWindow && chl <= kMaxWindow) {
            const int cw = (int)cwl, ch = (int)chl;
            float cx, ix, cy, iy;
            int fx, nx, fy, ny;
            aa_span(ox, cw, (float)cw / (float)S_w, cx, ix, fx, nx);
            aa_span(oy, ch, (float)ch / (float)S_h, cy, iy, fy, ny);
            const uint8_t *img = frames + n * (size_t)Hf * Wf * 3;
            // weight sums run over ALL taps of the window; pixel reads only over the taps that fall inside the frame
            float wysum = 0.f, wxsum = 0.f;
            for (int a = 0; a < nx; ++a) wxsum += fmaxf(0.f, 1.f - fabsf((float)(a + fx) - cx + 0.5f) * ix);
            for (int b = 0; b < ny; ++b) wysum += fmaxf(0.f, 1.f - fabsf((float)(b + fy) - cy + 0.5f) * iy);
            const long long ox0 = (long long)fx + x1, oy0 = (long long)fy + y1;   // frame coordinates of tap 0
            const int a_lo = (int)(ox0 < 0 ? (-ox0 < nx ? -ox0 : nx) : 0), a_hi = (int)(ox0 + nx > Wf ? (Wf - ox0 > 0 ? Wf - ox0 : 0) : nx);
            const int b_lo = (int)(oy0 < 0 ? (-oy0 < ny ? -oy0 : ny) : 0), b_hi = (int)(oy0 + ny > Hf ? (Hf - oy0 > 0 ? Hf - oy0 : 0) : ny);
            for (int b = b_lo; b < b_hi; ++b) {
                const float wy = fmaxf(0.f, 1.f - fabsf((float)(b + fy) - cy + 0.5f) * iy);
                if (wy == 0.f) continue;
                const uint8_t *line = img + ((size_t)(oy0 + b) * Wf + (size_t)(ox0 + a_lo)) * 3;
                float row[3] = {0.f, 0.f, 0.f};
                for (int a = a_lo; a < a_hi; ++a, line += 3) {
                    const float wx = fmaxf(0.f, 1.f - fabsf((float)(a + fx) - cx + 0.5f) * ix);
                    row[0] += wx * (float)line[0];
                    row[1] += wx * (float)line[1];
                    row[2] += wx * (float)line[2];
                }
                acc[0] += wy * row[0];
                acc[1] += wy * row[1];
                acc[2] += wy * row[2];
            }
            const float k = 1.f / (wxsum * wysum * 255.f);
            acc[0] *= k; acc[1] *= k; acc[2] *= k;
        }
        pr = (acc[0] - nm.mean[0]) * nm.inv_std[0];
        pg = (acc[1] - nm.mean[1]) * nm.inv_std[1];
        pb = (acc[2] - nm.mean[2]) * nm.inv_std[2];
}

// OUT: 0 = NHWC4 fp32, 1 = NHWC8 fp16, 2 = split [hi8 | lo8] fp16 pairs (HMV_F32X3)
template <int OUT>
__global__ void frames_to_input_kernel(const uint8_t *__restrict__ frames, const int *__restrict__ boxes, int Hf, int Wf, int S_h,
                                       int S_w, FrameNorm nm, void *__restrict__ out, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int ox = (int)(i % S_w);
        size_t t = i / S_w;
        const int oy = (int)(t % S_h);
        const size_t n = t / S_h;
        float r, g, b;
        frame_pixel(frames, boxes, n, oy, ox, Hf, Wf, S_h, S_w, nm, r, g, b);
        if (OUT == 2) {
            f16x8 hv = {0, 0, 0, 0, 0, 0, 0, 0}, lv = {0, 0, 0, 0, 0, 0, 0, 0};
            _Float16 a, c;
            split_f16(r, a, c); hv[0] = a; lv[0] = c;
            split_f16(g, a, c); hv[1] = a; lv[1] = c;
            split_f16(b, a, c); hv[2] = a; lv[2] = c;
            reinterpret_cast<f16x8 *>(out)[2 * i] = hv;
            reinterpret_cast<f16x8 *>(out)[2 * i + 1] = lv;
        } else if (OUT == 1) {
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            v[0] = (_Float16)r; v[1] = (_Float16)g; v[2] = (_Float16)b;
            reinterpret_cast<f16x8 *>(out)[i] = v;
        } else {
            reinterpret_cast<f32x4 *>(out)[i] = f32x4{r, g, b, 0.f};
        }
    }
}
// the same into the space-to-depth stem layout (nchw_to_s2d_kernel above): one thread per s2d pixel = 2 x 2 output pixels
template <int MODE>
__global__ void frames_to_s2d_kernel(const uint8_t *__restrict__ frames, const int *__restrict__ boxes, int Hf, int Wf, int S_h,
                                     int S_w, int Hs, int Ws, FrameNorm nm, void *__restrict__ out, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int xs = (int)(i % Ws);
        size_t t = i / Ws;
        const int ys = (int)(t % Hs);
        const size_t n = t / Hs;
        float v[12];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int oy = 2 * ys + dy, ox = 2 * xs + dx, j = (dy * 2 + dx) * 3;
                v[j] = v[j + 1] = v[j + 2] = 0.f;
                if (oy < S_h && ox < S_w) frame_pixel(frames, boxes, n, oy, ox, Hf, Wf, S_h, S_w, nm, v[j], v[j + 1], v[j + 2]);
            }
        s2d_store<MODE>(out, i, v);
    }
}
hipError_t launch_frames_to_input(const uint8_t *frames, const int *boxes, int N, int Hf, int Wf, int S_h, int S_w, const float *mean,
                                  const float *std, int out_mode, void *out, hipStream_t s, bool s2d) {
    FrameNorm nm;
    for (int c = 0; c < 3; ++c) { nm.mean[c] = mean[c]; nm.inv_std[c] = 1.f / std[c]; }
    if (s2d) {
        const int Hs = (S_h + 1) / 2, Ws = (S_w + 1) / 2;
        const size_t total = (size_t)N * Hs * Ws;
        const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        if (out_mode == 2) hipLaunchKernelGGL(frames_to_s2d_kernel<2>, dim3(grid), dim3(256), 0, s, frames, boxes, Hf, Wf, S_h, S_w, Hs, Ws, nm, out, total);
        else if (out_mode == 1) hipLaunchKernelGGL(frames_to_s2d_kernel<1>, dim3(grid), dim3(256), 0, s, frames, boxes, Hf, Wf, S_h, S_w, Hs, Ws, nm, out, total);
        else hipLaunchKernelGGL(frames_to_s2d_kernel<0>, dim3(grid), dim3(256), 0, s, frames, boxes, Hf, Wf, S_h, S_w, Hs, Ws, nm, out, total);
        return hipGetLastError();
    }
    const size_t total = (size_t)N * S_h * S_w;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (out_mode == 2) hipLaunchKernelGGL(frames_to_input_kernel<2>, dim3(grid), dim3(256), 0, s, frames, boxes, Hf, Wf, S_h, S_w, nm, out, total);
    else if (out_mode == 1) hipLaunchKernelGGL(frames_to_input_kernel<1>, dim3(grid), dim3(256), 0, s, frames, boxes, Hf, Wf, S_h, S_w, nm, out, total);
    else hipLaunchKernelGGL(frames_to_input_kernel<0>, dim3(grid), dim3(256), 0, s, frames, boxes, Hf, Wf, S_h, S_w, nm, out, total);
    return hipGetLastError();
}

}  // namespace hmv
