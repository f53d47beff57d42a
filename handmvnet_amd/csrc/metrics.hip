// Evaluation metrics on the device: MPJPE, Procrustes-aligned MPJPE and the PCK curve / AUC the
// reference logs from test_step (handmvnet.py:352-368 -> models/metrics.py:6-24, 64-123, 128-176).
//
// The inputs are tiny ([b,21,3] per step), so this is one single-workgroup launch: deterministic
// block-level reductions (fixed-order LDS tree, no float atomics), one lane per pose for the 3x3
// Procrustes problem.  The arithmetic that decides a comparison (joint distance vs threshold) is fp32 like
// the reference's; sums and the 3x3 SVD run in fp64.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/handmv.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kMaxSteps = 256;

// Fixed-order tree reduction over the workgroup; every lane gets the total.
__device__ double block_sum(double v, double* red) {
    const int t = threadIdx.x;
    __syncthreads();
    red[t] = v;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    return red[0];
}

// One-sided Jacobi (Hestenes) SVD of a 3x3 matrix: a = u * diag(s) * v^T, s descending.
// A column of u that belongs to a zero singular value is completed with the cross product of the others.
__device__ void svd3(const double a[3][3], double u[3][3], double s[3], double v[3][3]) {
    double w[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            w[i][j] = a[i][j];
            v[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) {
                    al += w[i][p] * w[i][p];
                    be += w[i][q] * w[i][q];
                    ga += w[i][p] * w[i][q];
                }
                if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                off = fmax(off, fabs(ga) / sqrt(al * be));
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double wp = w[i][p], wq = w[i][q];
                    w[i][p] = c * wp - sn * wq;
                    w[i][q] = sn * wp + c * wq;
                    const double vp = v[i][p], vq = v[i][q];
                    v[i][p] = c * vp - sn * vq;
                    v[i][q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    double n[3];
    for (int j = 0; j < 3; ++j) n[j] = sqrt(w[0][j] * w[0][j] + w[1][j] * w[1][j] + w[2][j] * w[2][j]);
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (n[ord[j]] < n[ord[j + 1]]) {
                const int tmp = ord[j];
                ord[j] = ord[j + 1];
                ord[j + 1] = tmp;
            }
    double vs[3][3];
    for (int j = 0; j < 3; ++j) {
        const int o = ord[j];
        s[j] = n[o];
        for (int i = 0; i < 3; ++i) {
            vs[i][j] = v[i][o];
            u[i][j] = n[o] > 0 ? w[i][o] / n[o] : 0.0;
        }
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = vs[i][j];
    const double tiny = 1e-14 * s[0];
    if (s[2] <= tiny) {
        if (s[1] <= tiny) {   // rank <= 1: any orthonormal completion
            int k = 0;
            if (s[0] > 0) {
                for (int i = 1; i < 3; ++i)
                    if (fabs(u[i][0]) < fabs(u[k][0])) k = i;
            } else {
                u[0][0] = 1; u[1][0] = 0; u[2][0] = 0;
                k = 1;
            }
            double e[3] = {0, 0, 0};
            e[k] = 1.0;
            const double d = e[0] * u[0][0] + e[1] * u[1][0] + e[2] * u[2][0];
            double nn = 0;
            for (int i = 0; i < 3; ++i) {
                u[i][1] = e[i] - d * u[i][0];
                nn += u[i][1] * u[i][1];
            }
            nn = sqrt(nn);
            for (int i = 0; i < 3; ++i) u[i][1] /= nn;
        }
        u[0][2] = u[1][0] * u[2][1] - u[2][0] * u[1][1];
        u[1][2] = u[2][0] * u[0][1] - u[0][0] * u[2][1];
        u[2][2] = u[0][0] * u[1][1] - u[1][0] * u[0][1];
    }
}

__device__ double det3(const double m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

// result: [0] mean distance, [1] mean distance after similarity alignment (NaN when not requested),
//         [2] auc, [3] norm_auc, [4 .. 4+steps) pck values, [4+steps .. 4+2*steps) thresholds.
__global__ __launch_bounds__(kThreads) void pose_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                               int n_sets, int n_pts, int dim, float tmin, float tmax,
                                                               int steps, int procrustes, float* __restrict__ aligned,
                                                               float* __restrict__ result) {
    __shared__ double red[kThreads];
    __shared__ float thr[kMaxSteps];
    __shared__ int hist[kMaxSteps + 1];
    const int t = threadIdx.x;
    // torch.linspace (fp32): symmetric fill from both ends, one fma per value (metrics.py:105)
    if (t < steps) {
        const float step = steps > 1 ? (tmax - tmin) / (float)(steps - 1) : 0.f;
        thr[t] = t < steps / 2 ? fmaf(step, (float)t, tmin) : fmaf(-step, (float)(steps - 1 - t), tmax);
        if (steps == 1) thr[t] = tmin;
    }
    if (t <= steps) hist[t] = 0;
    __syncthreads();

    // ---- MPJPE (metrics.py:12) and the PCK histogram (metrics.py:77-85)
    const long rows = (long)n_sets * n_pts;
    double dsum = 0.0;
    for (long r = t; r < rows; r += kThreads) {
        float acc = 0.f;
        for (int c = 0; c < dim; ++c) {
            const float d = pred[r * dim + c] - gt[r * dim + c];
            acc += d * d;
        }
        const float dist = sqrtf(acc);
        dsum += (double)dist;
        int b = 0;   // first threshold with dist <= thr; thresholds ascend
        while (b < steps && !(dist <= thr[b])) ++b;
        atomicAdd(&hist[b], 1);
    }
    const double mean_dist = block_sum(dsum, red) / (double)rows;

    // ---- PA-MPJPE: similarity transform of each predicted pose onto its target (metrics.py:128-176)
    double pasum = 0.0;
    if (procrustes) {
        for (int sidx = t; sidx < n_sets; sidx += kThreads) {
            const float* p = pred + (long)sidx * n_pts * 3;
            const float* g = gt + (long)sidx * n_pts * 3;
            double mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
            for (int j = 0; j < n_pts; ++j)
                for (int c = 0; c < 3; ++c) {
                    mu1[c] += p[j * 3 + c];
                    mu2[c] += g[j * 3 + c];
                }
            for (int c = 0; c < 3; ++c) {
                mu1[c] /= n_pts;
                mu2[c] /= n_pts;
            }
            double var1 = 0.0, K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (int j = 0; j < n_pts; ++j) {
                double x1[3], x2[3];
                for (int c = 0; c < 3; ++c) {
                    x1[c] = p[j * 3 + c] - mu1[c];
                    x2[c] = g[j * 3 + c] - mu2[c];
                    var1 += x1[c] * x1[c];
                }
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) K[a][b] += x1[a] * x2[b];
            }
            double U[3][3], S[3], V[3][3];
            svd3(K, U, S, V);
            // Z = diag(1, 1, sign(det(U V^T)));  R = V Z U^T
            double UVt[3][3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) UVt[a][b] = U[a][0] * V[b][0] + U[a][1] * V[b][1] + U[a][2] * V[b][2];
            const double dd = det3(UVt);
            const double z = dd > 0 ? 1.0 : (dd < 0 ? -1.0 : 0.0);
            double R[3][3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) R[a][b] = V[a][0] * U[b][0] + V[a][1] * U[b][1] + z * V[a][2] * U[b][2];
            double trace = 0.0;   // trace(R K)
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) trace += R[a][b] * K[b][a];
            const double scale = trace / var1;
            double tr[3];
            for (int a = 0; a < 3; ++a)
                tr[a] = mu2[a] - scale * (R[a][0] * mu1[0] + R[a][1] * mu1[1] + R[a][2] * mu1[2]);
            for (int j = 0; j < n_pts; ++j) {
                double e2 = 0.0;
                for (int a = 0; a < 3; ++a) {
                    const double y = scale * (R[a][0] * p[j * 3] + R[a][1] * p[j * 3 + 1] + R[a][2] * p[j * 3 + 2]) + tr[a];
                    if (aligned) aligned[((long)sidx * n_pts + j) * 3 + a] = (float)y;
                    const double e = y - (double)g[j * 3 + a];
                    e2 += e * e;
                }
                pasum += sqrt(e2);
            }
        }
    }
    const double pa_mean = block_sum(pasum, red) / (double)rows;

    if (t == 0) {
        result[0] = (float)mean_dist;
        result[1] = procrustes ? (float)pa_mean : nanf("");
        // PCK curve: cumulative histogram; AUC by the trapezoidal rule in fp32 (metrics.py:114-121)
        int cum = 0;
        float auc = 0.f, one = 0.f, prev = 0.f;
        for (int i = 0; i < steps; ++i) {
            cum += hist[i];
            const float pck = (float)cum / (float)rows;
            result[4 + i] = pck;
            result[4 + steps + i] = thr[i];
            if (i > 0) {
                const float dx = thr[i] - thr[i - 1];
                auc += dx * (pck + prev) * 0.5f;
                one += dx;
            }
            prev = pck;
        }
        result[2] = auc;
        result[3] = auc / one;   // 0/0 = NaN for a single threshold, like the reference
    }
}

}  // namespace

extern "C" int hmv_pose_metrics(int32_t device, const float* pred, const float* target, int32_t n_sets, int32_t n_pts,
                                int32_t dim, float thr_min, float thr_max, int32_t steps, int32_t procrustes,
                                float* aligned, float* result, void* stream) {
    if (!pred || !target || !result || n_sets <= 0 || n_pts <= 0 || dim < 1 || dim > 4 || steps < 1 || steps > kMaxSteps)
        return HMV_ERR_ARG;
    if ((procrustes || aligned) && dim != 3) return HMV_ERR_ARG;
    if (aligned && !procrustes) return HMV_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return HMV_ERR_HIP;
    hipLaunchKernelGGL(pose_metrics_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, pred, target, n_sets, n_pts, dim,
                       thr_min, thr_max, steps, procrustes, aligned, result);
    return hipGetLastError() == hipSuccess ? HMV_OK : HMV_ERR_HIP;
}
