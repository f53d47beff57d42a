/*
 * hmv_oracle.c -- CPU restatement of the HandMvNet inference forward pass.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP engine is compared with.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (handmvnet_amd/csrc -> libhandmv.so) never links, loads or calls it.
 *
 * It follows the reference op by op, in the reference's own layouts (NCHW activations,
 * OIHW weights, un-folded BatchNorm), so that it is an independent check of the engine's
 * NHWC / folded-BN / fused formulation.  Every function cites the reference lines it
 * restates (paths relative to /root/reference/src).
 *
 * Pinning: tests/golden/*.npz hold outputs of the REAL reference (imported in the build
 * container by tests/golden/make_fixtures.py); tests/test_oracle_golden.py checks this
 * file against them.
 *
 * Build: see oracle/Makefile.  -DHMVO_ACC_DOUBLE accumulates every contraction, softmax
 * and normalisation in double (the "f64" checker); default accumulates in float like the
 * reference does (the "f32" port, also used as bench.py's cpu_baseline, kind "port").
 */
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef HMVO_ACC_DOUBLE
typedef double acc_t;
#define VL 4
typedef double vacc __attribute__((vector_size(32)));
#else
typedef float acc_t;
#define VL 8
typedef float vacc __attribute__((vector_size(32)));
#endif
#define MR 6
#define NR (2 * VL)

#define NJ 21
#define HEADS 8
#define DHEAD 128

/* ------------------------------------------------------------------ tensor registry */
typedef struct {
    char key[128];
    float *data;
    int64_t shape[4];
    int ndim;
    int64_t numel;
} otensor;

static otensor *g_tab = NULL;
static int g_ntab = 0, g_captab = 0;
static char g_err[512] = "";

const char *hmvo_last_error(void) { return g_err; }

int hmvo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void hmvo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int hmvo_acc_is_double(void) { return (int)(sizeof(acc_t) == 8); }

void hmvo_clear(void) {
    for (int i = 0; i < g_ntab; ++i) free(g_tab[i].data);
    free(g_tab);
    g_tab = NULL;
    g_ntab = g_captab = 0;
}

int hmvo_set_tensor(const char *key, const float *data, const int64_t *shape, int ndim) {
    if (ndim > 4 || strlen(key) >= 128) { snprintf(g_err, sizeof g_err, "bad tensor %s", key); return 1; }
    if (g_ntab == g_captab) {
        g_captab = g_captab ? 2 * g_captab : 512;
        g_tab = (otensor *)realloc(g_tab, (size_t)g_captab * sizeof(otensor));
    }
    otensor *t = &g_tab[g_ntab++];
    memset(t, 0, sizeof *t);
    strcpy(t->key, key);
    t->ndim = ndim;
    t->numel = 1;
    for (int i = 0; i < ndim; ++i) { t->shape[i] = shape[i]; t->numel *= shape[i]; }
    t->data = (float *)malloc((size_t)t->numel * sizeof(float));
    memcpy(t->data, data, (size_t)t->numel * sizeof(float));
    return 0;
}

static int g_missing = 0;
static const otensor *T(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static const otensor *T(const char *fmt, ...) {
    char key[128];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(key, sizeof key, fmt, ap);
    va_end(ap);
    for (int i = 0; i < g_ntab; ++i)
        if (!strcmp(g_tab[i].key, key)) return &g_tab[i];
    snprintf(g_err, sizeof g_err, "missing tensor %s", key);
    g_missing = 1;
    static float zero[4096];
    static otensor dummy;
    dummy.data = zero;
    return &dummy;
}

static float *falloc(size_t n) {
    float *p = (float *)aligned_alloc(64, ((n * sizeof(float) + 63) / 64) * 64);
    if (!p) { fprintf(stderr, "hmv_oracle: out of memory (%zu floats)\n", n); abort(); }
    return p;
}

/* ------------------------------------------------------------------ GEMM core
 * C[M][N] = A[M][K] * B[K][N]   (row-major, leading dimensions given).
 * 6 x 2VL register tile, B panel packed per thread; accumulation in acc_t. */
static void gemm_nn(int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc) {
    int npanels = (N + NR - 1) / NR;
#pragma omp parallel
    {
        float *bp = falloc((size_t)K * NR);
#pragma omp for schedule(dynamic, 1)
        for (int p = 0; p < npanels; ++p) {
            int n0 = p * NR, nw = N - n0 < NR ? N - n0 : NR;
            for (int k = 0; k < K; ++k) {
                const float *src = B + (size_t)k * ldb + n0;
                float *dst = bp + (size_t)k * NR;
                int j = 0;
                for (; j < nw; ++j) dst[j] = src[j];
                for (; j < NR; ++j) dst[j] = 0.f;
            }
            for (int m0 = 0; m0 < M; m0 += MR) {
                int mh = M - m0 < MR ? M - m0 : MR;
                vacc acc[MR][2];
                for (int i = 0; i < MR; ++i) { acc[i][0] = (vacc){0}; acc[i][1] = (vacc){0}; }
                const float *a[MR];
                for (int i = 0; i < MR; ++i) a[i] = A + (size_t)(m0 + (i < mh ? i : 0)) * lda;
                for (int k = 0; k < K; ++k) {
                    const float *b = bp + (size_t)k * NR;
                    vacc b0, b1;
                    for (int j = 0; j < VL; ++j) { b0[j] = (acc_t)b[j]; b1[j] = (acc_t)b[VL + j]; }
                    for (int i = 0; i < MR; ++i) {
                        acc_t av = (acc_t)a[i][k];
                        acc[i][0] += av * b0;
                        acc[i][1] += av * b1;
                    }
                }
                for (int i = 0; i < mh; ++i) {
                    float *c = C + (size_t)(m0 + i) * ldc + n0;
                    for (int j = 0; j < nw; ++j) c[j] = (float)(j < VL ? acc[i][0][j] : acc[i][1][j - VL]);
                }
            }
        }
        free(bp);
    }
}

/* y[T][out] = x[T][in] * W[out][in]^T + bias   -- torch.nn.Linear */
static void linear(const float *x, int Tn, int in, const float *W, const float *bias, int out, float *y) {
    float *wt = falloc((size_t)in * out);
#pragma omp parallel for
    for (int i = 0; i < in; ++i)
        for (int o = 0; o < out; ++o) wt[(size_t)i * out + o] = W[(size_t)o * in + i];
    gemm_nn(Tn, out, in, x, in, wt, out, y, out);
    if (bias) {
#pragma omp parallel for
        for (int t = 0; t < Tn; ++t)
            for (int o = 0; o < out; ++o) y[(size_t)t * out + o] += bias[o];
    }
    free(wt);
}

/* ------------------------------------------------------------------ conv / bn / pool (NCHW)
 * nn.Conv2d: models/backbones/resnet.py:21-28,114-118,162,193; models/layers.py:318-334 */
static void conv2d(const float *in, int N, int C, int H, int W, const float *w, const float *bias, int O, int kh,
                   int kw, int stride, int pad, float *out, int *Ho_, int *Wo_) {
    int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    *Ho_ = Ho;
    *Wo_ = Wo;
    int K = C * kh * kw, P = Ho * Wo;
    int direct = (kh == 1 && kw == 1 && stride == 1 && pad == 0);
    float *col = direct ? NULL : falloc((size_t)K * P);
    for (int n = 0; n < N; ++n) {
        const float *src = in + (size_t)n * C * H * W;
        if (!direct) {
#pragma omp parallel for
            for (int k = 0; k < K; ++k) {
                int c = k / (kh * kw), r = (k / kw) % kh, s = k % kw;
                float *dst = col + (size_t)k * P;
                for (int ho = 0; ho < Ho; ++ho) {
                    int hi = ho * stride - pad + r;
                    for (int wo = 0; wo < Wo; ++wo) {
                        int wi = wo * stride - pad + s;
                        dst[ho * Wo + wo] =
                            (hi >= 0 && hi < H && wi >= 0 && wi < W) ? src[((size_t)c * H + hi) * W + wi] : 0.f;
                    }
                }
            }
        }
        float *dst = out + (size_t)n * O * P;
        gemm_nn(O, P, K, w, K, direct ? src : col, P, dst, P);
        if (bias) {
#pragma omp parallel for
            for (int o = 0; o < O; ++o)
                for (int p = 0; p < P; ++p) dst[(size_t)o * P + p] += bias[o];
        }
    }
    free(col);
}

/* nn.ConvTranspose2d(k=4, s=2, p=1), weight [Cin][Cout][4][4]: models/handmvnet.py:75 */
static void conv_transpose_4s2p1(const float *in, int N, int C, int H, int W, const float *w, const float *bias,
                                 int O, float *out) {
    int Ho = 2 * H, Wo = 2 * W, P = H * W;
    float *wk = falloc((size_t)O * C), *tmp = falloc((size_t)O * P);
    for (int n = 0; n < N; ++n) {
        float *dst = out + (size_t)n * O * Ho * Wo;
#pragma omp parallel for
        for (int o = 0; o < O; ++o)
            for (int p = 0; p < Ho * Wo; ++p) dst[(size_t)o * Ho * Wo + p] = bias ? bias[o] : 0.f;
        for (int ky = 0; ky < 4; ++ky)
            for (int kx = 0; kx < 4; ++kx) {
                for (int o = 0; o < O; ++o)
                    for (int c = 0; c < C; ++c) wk[(size_t)o * C + c] = w[(((size_t)c * O + o) * 4 + ky) * 4 + kx];
                gemm_nn(O, P, C, wk, C, in + (size_t)n * C * P, P, tmp, P);
#pragma omp parallel for
                for (int o = 0; o < O; ++o)
                    for (int iy = 0; iy < H; ++iy) {
                        int y = iy * 2 - 1 + ky;
                        if (y < 0 || y >= Ho) continue;
                        for (int ix = 0; ix < W; ++ix) {
                            int x = ix * 2 - 1 + kx;
                            if (x < 0 || x >= Wo) continue;
                            dst[((size_t)o * Ho + y) * Wo + x] += tmp[(size_t)o * P + iy * W + ix];
                        }
                    }
            }
    }
    free(wk);
    free(tmp);
}

/* BatchNorm2d in eval mode / FrozenBatchNorm2d (eps 1e-5): resnet.py:59-74; in place.
 * relu: nn.ReLU fused here for brevity (the reference applies it right after). */
static void bn_eval(float *x, int N, int C, int P, const char *prefix, int relu) {
    const float *g = T("%s.weight", prefix)->data, *b = T("%s.bias", prefix)->data;
    const float *rm = T("%s.running_mean", prefix)->data, *rv = T("%s.running_var", prefix)->data;
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            acc_t scale = (acc_t)g[c] / (acc_t)sqrt((double)((acc_t)rv[c] + (acc_t)1e-5));
            acc_t shift = (acc_t)b[c] - (acc_t)rm[c] * scale;
            float *p = x + ((size_t)n * C + c) * P;
            for (int i = 0; i < P; ++i) {
                float v = (float)((acc_t)p[i] * scale + shift);
                p[i] = relu && v < 0.f ? 0.f : v;
            }
        }
}

static void relu_(float *x, size_t n) {
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) x[i] = x[i] < 0.f ? 0.f : x[i];
}

/* nn.MaxPool2d(3, stride 2, padding 1): resnet.py:165,221 */
static void maxpool_3s2p1(const float *in, int N, int C, int H, int W, float *out, int *Ho_, int *Wo_) {
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    *Ho_ = Ho;
    *Wo_ = Wo;
#pragma omp parallel for
    for (int nc = 0; nc < N * C; ++nc) {
        const float *s = in + (size_t)nc * H * W;
        float *d = out + (size_t)nc * Ho * Wo;
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
                float m = -INFINITY;
                for (int r = 0; r < 3; ++r)
                    for (int q = 0; q < 3; ++q) {
                        int hi = ho * 2 - 1 + r, wi = wo * 2 - 1 + q;
                        if (hi >= 0 && hi < H && wi >= 0 && wi < W && s[hi * W + wi] > m) m = s[hi * W + wi];
                    }
                d[ho * Wo + wo] = m;
            }
    }
}

/* ------------------------------------------------------------------ ResNet blocks
 * Bottleneck.forward resnet.py:124-144 ; BasicBlock.forward resnet.py:90-106 */
static float *res_block(float *x, int N, int *C, int *H, int *W, const char *prefix, int planes, int stride,
                        int bottleneck) {
    int Cin = *C, Hin = *H, Win = *W, Ho, Wo, h1, w1;
    char name[160];
    int outC = bottleneck ? planes * 4 : planes;
    float *o1, *o2, *o3 = NULL, *res = x;
    if (bottleneck) {
        o1 = falloc((size_t)N * planes * Hin * Win);
        conv2d(x, N, Cin, Hin, Win, T("%s.conv1.weight", prefix)->data, NULL, planes, 1, 1, 1, 0, o1, &h1, &w1);
        snprintf(name, sizeof name, "%s.bn1", prefix);
        bn_eval(o1, N, planes, h1 * w1, name, 1);
        Ho = (Hin + 2 - 3) / stride + 1;
        Wo = (Win + 2 - 3) / stride + 1;
        o2 = falloc((size_t)N * planes * Ho * Wo);
        conv2d(o1, N, planes, h1, w1, T("%s.conv2.weight", prefix)->data, NULL, planes, 3, 3, stride, 1, o2, &Ho, &Wo);
        snprintf(name, sizeof name, "%s.bn2", prefix);
        bn_eval(o2, N, planes, Ho * Wo, name, 1);
        o3 = falloc((size_t)N * outC * Ho * Wo);
        conv2d(o2, N, planes, Ho, Wo, T("%s.conv3.weight", prefix)->data, NULL, outC, 1, 1, 1, 0, o3, &h1, &w1);
        snprintf(name, sizeof name, "%s.bn3", prefix);
        bn_eval(o3, N, outC, Ho * Wo, name, 0);
        free(o1);
        free(o2);
    } else {
        Ho = (Hin + 2 - 3) / stride + 1;
        Wo = (Win + 2 - 3) / stride + 1;
        o1 = falloc((size_t)N * planes * Ho * Wo);
        conv2d(x, N, Cin, Hin, Win, T("%s.conv1.weight", prefix)->data, NULL, planes, 3, 3, stride, 1, o1, &Ho, &Wo);
        snprintf(name, sizeof name, "%s.bn1", prefix);
        bn_eval(o1, N, planes, Ho * Wo, name, 1);
        o3 = falloc((size_t)N * planes * Ho * Wo);
        conv2d(o1, N, planes, Ho, Wo, T("%s.conv2.weight", prefix)->data, NULL, planes, 3, 3, 1, 1, o3, &h1, &w1);
        snprintf(name, sizeof name, "%s.bn2", prefix);
        bn_eval(o3, N, planes, Ho * Wo, name, 0);
        free(o1);
    }
    if (stride != 1 || Cin != outC) { /* _make_layer: resnet.py:189-195 */
        res = falloc((size_t)N * outC * Ho * Wo);
        conv2d(x, N, Cin, Hin, Win, T("%s.downsample.0.weight", prefix)->data, NULL, outC, 1, 1, stride, 0, res, &h1,
               &w1);
        snprintf(name, sizeof name, "%s.downsample.1", prefix);
        bn_eval(res, N, outC, Ho * Wo, name, 0);
    }
    size_t tot = (size_t)N * outC * Ho * Wo;
#pragma omp parallel for
    for (size_t i = 0; i < tot; ++i) { /* out += residual; relu */
        float v = o3[i] + res[i];
        o3[i] = v < 0.f ? 0.f : v;
    }
    if (res != x) free(res);
    *C = outC;
    *H = Ho;
    *W = Wo;
    return o3;
}

/* ------------------------------------------------------------------ HRNet (backbones/hrnet.py) */
/* conv (no bias) + BatchNorm2d (+ReLU): the building block of every HRNet Sequential */
static float *conv_bn(const float *x, int N, int Cin, int H, int W, const char *wkey, const char *bnprefix, int Cout,
                      int k, int stride, int pad, int relu, int *Ho, int *Wo) {
    int ho = (H + 2 * pad - k) / stride + 1, wo = (W + 2 * pad - k) / stride + 1;
    float *o = falloc((size_t)N * Cout * ho * wo);
    conv2d(x, N, Cin, H, W, T("%s", wkey)->data, NULL, Cout, k, k, stride, pad, o, &ho, &wo);
    bn_eval(o, N, Cout, ho * wo, bnprefix, relu);
    *Ho = ho;
    *Wo = wo;
    return o;
}

/* nn.Upsample(scale_factor=2**s, mode='nearest'): hrnet.py:165 */
static float *upsample_nearest(const float *x, int NC, int H, int W, int s) {
    int f = 1 << s, Ho = H * f, Wo = W * f;
    float *o = falloc((size_t)NC * Ho * Wo);
#pragma omp parallel for
    for (int i = 0; i < NC; ++i)
        for (int y = 0; y < Ho; ++y)
            for (int xx = 0; xx < Wo; ++xx) o[((size_t)i * Ho + y) * Wo + xx] = x[((size_t)i * H + (y >> s)) * W + (xx >> s)];
    return o;
}

/* HighResolutionModule.forward: hrnet.py:194-212.  x[] (nbr branches) is consumed and replaced. */
static void hr_module(float **x, const int *ch, int *h, int *w, int nbr, int N, const char *mp) {
    char key[192], bnk[192];
    for (int b = 0; b < nbr; ++b)
        for (int blk = 0; blk < 4; ++blk) { /* _make_one_branch: 4 BasicBlocks, no downsample */
            int C = ch[b], hh = h[b], ww = w[b];
            snprintf(key, sizeof key, "%s.branches.%d.%d", mp, b, blk);
            float *nx = res_block(x[b], N, &C, &hh, &ww, key, ch[b], 1, 0);
            free(x[b]);
            x[b] = nx;
        }
    float *out[4];
    for (int i = 0; i < nbr; ++i) {
        float *y = NULL;
        size_t tot = (size_t)N * ch[i] * h[i] * w[i];
        for (int j = 0; j < nbr; ++j) {
            float *term;
            int owned = 1;
            if (j == i) {
                term = x[j];
                owned = 0;
            } else if (j > i) { /* 1x1 conv + BN + nearest upsample */
                int ho, wo;
                snprintf(key, sizeof key, "%s.fuse_layers.%d.%d.0.weight", mp, i, j);
                snprintf(bnk, sizeof bnk, "%s.fuse_layers.%d.%d.1", mp, i, j);
                float *c = conv_bn(x[j], N, ch[j], h[j], w[j], key, bnk, ch[i], 1, 1, 0, 0, &ho, &wo);
                term = upsample_nearest(c, N * ch[i], ho, wo, j - i);
                free(c);
            } else { /* i - j stride-2 3x3 convs; ReLU on all but the last */
                const float *cur = x[j];
                float *tmp = NULL;
                int hc = h[j], wc = w[j];
                for (int q = 0; q < i - j; ++q) {
                    int last = q == i - j - 1, outc = last ? ch[i] : ch[j];
                    snprintf(key, sizeof key, "%s.fuse_layers.%d.%d.%d.0.weight", mp, i, j, q);
                    snprintf(bnk, sizeof bnk, "%s.fuse_layers.%d.%d.%d.1", mp, i, j, q);
                    float *nx = conv_bn(cur, N, ch[j], hc, wc, key, bnk, outc, 3, 2, 1, !last, &hc, &wc);
                    free(tmp);
                    tmp = nx;
                    cur = nx;
                }
                term = tmp;
            }
            if (!y) { /* y = x[0] if i == 0 else fuse_layers[i][0](x[0]) */
                y = falloc(tot);
                memcpy(y, term, tot * sizeof(float));
            } else {
                for (size_t e = 0; e < tot; ++e) y[e] = y[e] + term[e];
            }
            if (owned) free(term);
        }
        relu_(y, tot);
        out[i] = y;
    }
    for (int b = 0; b < nbr; ++b) { free(x[b]); x[b] = out[b]; }
}

/* HighResolutionNet.forward: hrnet.py:357-393.  Returns 4 feature maps (highest resolution first). */
static void hrnet_forward(int w64, const float *x, int N, int H, int W, float **feats, int *fc, int *fh, int *fw) {
    static const int CH[2][4] = {{40, 80, 160, 320}, {64, 128, 256, 512}};
    const int *ch = CH[w64];
    int h1, w1, h2, w2;
    char key[192], bnk[192];
    float *c1 = conv_bn(x, N, 3, H, W, "backbone.conv1.weight", "backbone.bn1", 64, 3, 2, 1, 1, &h1, &w1);
    float *cur = conv_bn(c1, N, 64, h1, w1, "backbone.conv2.weight", "backbone.bn2", 64, 3, 2, 1, 1, &h2, &w2);
    free(c1);
    int C = 64, hc = h2, wc = w2;
    for (int bi = 0; bi < 4; ++bi) { /* layer1: 4 Bottlenecks */
        snprintf(key, sizeof key, "backbone.layer1.%d", bi);
        float *nx = res_block(cur, N, &C, &hc, &wc, key, 64, 1, 1);
        free(cur);
        cur = nx;
    }
    float *xs[4] = {0, 0, 0, 0};
    int hs[4], ws[4];
    int npre = 1, prec[4] = {256, 0, 0, 0};
    float *pre[4] = {cur, 0, 0, 0};
    int preh[4] = {hc, 0, 0, 0}, prew[4] = {wc, 0, 0, 0};
    static const int NMOD[3] = {1, 4, 3};
    for (int st = 0; st < 3; ++st) {
        int nbr = st + 2;
        for (int i = 0; i < nbr; ++i) { /* transition layers: hrnet.py:287-311, 368-390 */
            if (i < npre) {
                if (ch[i] != prec[i]) {
                    snprintf(key, sizeof key, "backbone.transition%d.%d.0.weight", st + 1, i);
                    snprintf(bnk, sizeof bnk, "backbone.transition%d.%d.1", st + 1, i);
                    xs[i] = conv_bn(pre[i], N, prec[i], preh[i], prew[i], key, bnk, ch[i], 3, 1, 1, 1, &hs[i], &ws[i]);
                } else {
                    size_t tot = (size_t)N * prec[i] * preh[i] * prew[i];
                    xs[i] = falloc(tot);
                    memcpy(xs[i], pre[i], tot * sizeof(float));
                    hs[i] = preh[i];
                    ws[i] = prew[i];
                }
            } else { /* new branch from the LAST previous branch through (i + 1 - npre) stride-2 convs */
                const float *src = pre[npre - 1];
                float *tmp = NULL;
                int hh = preh[npre - 1], ww = prew[npre - 1], cin = prec[npre - 1];
                for (int j = 0; j < i + 1 - npre; ++j) {
                    int outc = (j == i - npre) ? ch[i] : cin;
                    snprintf(key, sizeof key, "backbone.transition%d.%d.%d.0.weight", st + 1, i, j);
                    snprintf(bnk, sizeof bnk, "backbone.transition%d.%d.%d.1", st + 1, i, j);
                    float *nx = conv_bn(src, N, cin, hh, ww, key, bnk, outc, 3, 2, 1, 1, &hh, &ww);
                    free(tmp);
                    tmp = nx;
                    src = nx;
                    cin = outc;
                }
                xs[i] = tmp;
                hs[i] = hh;
                ws[i] = ww;
            }
        }
        for (int i = 0; i < npre; ++i) free(pre[i]);
        for (int m = 0; m < NMOD[st]; ++m) {
            snprintf(key, sizeof key, "backbone.stage%d.%d", st + 2, m);
            hr_module(xs, ch, hs, ws, nbr, N, key);
        }
        npre = nbr;
        for (int i = 0; i < nbr; ++i) { pre[i] = xs[i]; prec[i] = ch[i]; preh[i] = hs[i]; prew[i] = ws[i]; xs[i] = NULL; }
    }
    for (int i = 0; i < 4; ++i) { feats[i] = pre[i]; fc[i] = ch[i]; fh[i] = preh[i]; fw[i] = prew[i]; }
}

/* ------------------------------------------------------------------ heads */
/* soft_argmax_2d(heatmap, temperature=1000): models/utils.py:35-62 */
static void soft_argmax_2d(const float *hm, int NC, int H, int W, float *coords) {
#pragma omp parallel for
    for (int i = 0; i < NC; ++i) {
        const float *p = hm + (size_t)i * H * W;
        acc_t mx = -INFINITY;
        for (int k = 0; k < H * W; ++k) {
            acc_t v = (acc_t)(p[k] * 1000.0f);
            if (v > mx) mx = v;
        }
        acc_t *e = (acc_t *)malloc(sizeof(acc_t) * H * W), sum = 0;
        for (int k = 0; k < H * W; ++k) {
            e[k] = (acc_t)exp((double)((acc_t)(p[k] * 1000.0f) - mx));
            sum += e[k];
        }
        acc_t ex = 0, ey = 0;
        for (int w = 0; w < W; ++w) { /* accu_x = heatmap.sum(dim=2) ; * x_indices ; sum */
            acc_t a = 0;
            for (int h = 0; h < H; ++h) a += e[h * W + w] / sum;
            ex += a * (acc_t)w;
        }
        for (int h = 0; h < H; ++h) {
            acc_t a = 0;
            for (int w = 0; w < W; ++w) a += e[h * W + w] / sum;
            ey += a * (acc_t)h;
        }
        coords[2 * i] = (float)ex;
        coords[2 * i + 1] = (float)ey;
        free(e);
    }
}

/* SampleNet._sample_joint_features: models/nets.py:46-53 (F.grid_sample bilinear,
 * align_corners=True, padding_mode zeros).  feat [N][C][H][W], xy [N][21][2] -> out [N][21][ldo] at col0 */
static void sample_joint_features(const float *feat, int N, int C, int H, int W, const float *xy, float *out,
                                  int ldo, int col0) {
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int j = 0; j < NJ; ++j) {
            float jx = xy[((size_t)n * NJ + j) * 2], jy = xy[((size_t)n * NJ + j) * 2 + 1];
            float gx = jx / (float)(W - 1) * 2.f - 1.f, gy = jy / (float)(H - 1) * 2.f - 1.f;
            float ix = ((gx + 1.f) / 2.f) * (float)(W - 1), iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
            float fx0 = floorf(ix), fy0 = floorf(iy);
            int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
            float wnw = ((float)x1 - ix) * ((float)y1 - iy), wne = (ix - (float)x0) * ((float)y1 - iy);
            float wsw = ((float)x1 - ix) * (iy - (float)y0), wse = (ix - (float)x0) * (iy - (float)y0);
            int vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
            float *dst = out + ((size_t)n * NJ + j) * ldo + col0;
            for (int c = 0; c < C; ++c) {
                const float *f = feat + ((size_t)n * C + c) * H * W;
                acc_t v = 0;
                if (vy0 && vx0) v += (acc_t)f[y0 * W + x0] * (acc_t)wnw;
                if (vy0 && vx1) v += (acc_t)f[y0 * W + x1] * (acc_t)wne;
                if (vy1 && vx0) v += (acc_t)f[y1 * W + x0] * (acc_t)wsw;
                if (vy1 && vx1) v += (acc_t)f[y1 * W + x1] * (acc_t)wse;
                dst[c] = (float)v;
            }
        }
}

/* nn.LayerNorm(d), eps 1e-5, biased variance: models/layers.py:194-195,165 */
static void layer_norm(const float *x, int Tn, int d, const float *g, const float *b, float *y) {
#pragma omp parallel for
    for (int t = 0; t < Tn; ++t) {
        const float *r = x + (size_t)t * d;
        acc_t m = 0, v = 0;
        for (int i = 0; i < d; ++i) m += (acc_t)r[i];
        m /= (acc_t)d;
        for (int i = 0; i < d; ++i) v += ((acc_t)r[i] - m) * ((acc_t)r[i] - m);
        v /= (acc_t)d;
        acc_t rs = (acc_t)1 / (acc_t)sqrt((double)(v + (acc_t)1e-5));
        for (int i = 0; i < d; ++i) y[(size_t)t * d + i] = (float)(((acc_t)r[i] - m) * rs * (acc_t)g[i] + (acc_t)b[i]);
    }
}

/* MultiHeadAttention.forward: models/layers.py:202-237 ; FeedForward: layers.py:161-174.
 * x [B][Tn][d]; qlen>0 => cross-attention (Q = first qlen tokens, K/V = the rest). Returns new [B][Tq][d]. */
static float *mha_block(const float *x, int B, int Tn, int d, int layer, int qlen, int *Tq_out) {
    int Tq = qlen > 0 ? qlen : Tn, Tk = qlen > 0 ? Tn - qlen : Tn, koff = qlen > 0 ? qlen : 0;
    int inner = HEADS * DHEAD;
    char pfx[96];
    snprintf(pfx, sizeof pfx, "joints_late_fusion.attn_fusion.%d", layer);
    float *xq = falloc((size_t)B * Tq * d), *xk = falloc((size_t)B * Tk * d);
    for (int b = 0; b < B; ++b) {
        memcpy(xq + (size_t)b * Tq * d, x + (size_t)b * Tn * d, sizeof(float) * Tq * d);
        memcpy(xk + (size_t)b * Tk * d, x + ((size_t)b * Tn + koff) * d, sizeof(float) * Tk * d);
    }
    float *q = falloc((size_t)B * Tq * inner), *k = falloc((size_t)B * Tk * inner), *v = falloc((size_t)B * Tk * inner);
    linear(xq, B * Tq, d, T("%s.to_q.weight", pfx)->data, NULL, inner, q);
    linear(xk, B * Tk, d, T("%s.to_k.weight", pfx)->data, NULL, inner, k);
    linear(xk, B * Tk, d, T("%s.to_v.weight", pfx)->data, NULL, inner, v);
    float *att = falloc((size_t)B * Tq * inner);
    const acc_t scale = (acc_t)(1.0 / sqrt((double)DHEAD)); /* dim_head ** -0.5 */
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < HEADS; ++h) {
            acc_t *dots = (acc_t *)malloc(sizeof(acc_t) * Tk);
            for (int i = 0; i < Tq; ++i) {
                const float *qi = q + ((size_t)b * Tq + i) * inner + h * DHEAD;
                acc_t mx = -INFINITY;
                for (int j = 0; j < Tk; ++j) {
                    const float *kj = k + ((size_t)b * Tk + j) * inner + h * DHEAD;
                    acc_t s = 0;
                    for (int c = 0; c < DHEAD; ++c) s += (acc_t)qi[c] * (acc_t)kj[c];
                    dots[j] = s * scale;
                    if (dots[j] > mx) mx = dots[j];
                }
                acc_t sum = 0;
                for (int j = 0; j < Tk; ++j) { dots[j] = (acc_t)exp((double)(dots[j] - mx)); sum += dots[j]; }
                float *o = att + ((size_t)b * Tq + i) * inner + h * DHEAD;
                for (int c = 0; c < DHEAD; ++c) {
                    acc_t s = 0;
                    for (int j = 0; j < Tk; ++j)
                        s += (dots[j] / sum) * (acc_t)v[((size_t)b * Tk + j) * inner + h * DHEAD + c];
                    o[c] = (float)s;
                }
            }
            free(dots);
        }
    float *out = falloc((size_t)B * Tq * d);
    linear(att, B * Tq, inner, T("%s.to_out.weight", pfx)->data, T("%s.to_out.bias", pfx)->data, d, out);
    size_t tot = (size_t)B * Tq * d;
    for (size_t i = 0; i < tot; ++i) out[i] += xq[i]; /* out + _q */
    float *n1 = falloc(tot);
    layer_norm(out, B * Tq, d, T("%s.norm1.weight", pfx)->data, T("%s.norm1.bias", pfx)->data, n1);
    /* ff: LayerNorm -> Linear(d,128) -> GELU(erf) -> Linear(128,d) */
    float *f0 = falloc(tot), *f1 = falloc((size_t)B * Tq * DHEAD), *f2 = falloc(tot);
    layer_norm(n1, B * Tq, d, T("%s.ff.net.0.weight", pfx)->data, T("%s.ff.net.0.bias", pfx)->data, f0);
    linear(f0, B * Tq, d, T("%s.ff.net.1.weight", pfx)->data, T("%s.ff.net.1.bias", pfx)->data, DHEAD, f1);
    for (size_t i = 0; i < (size_t)B * Tq * DHEAD; ++i) {
        acc_t z = (acc_t)f1[i];
        f1[i] = (float)((acc_t)0.5 * z * ((acc_t)1 + (acc_t)erf((double)z * 0.70710678118654752440)));
    }
    linear(f1, B * Tq, DHEAD, T("%s.ff.net.4.weight", pfx)->data, T("%s.ff.net.4.bias", pfx)->data, d, f2);
    for (size_t i = 0; i < tot; ++i) f2[i] += n1[i];
    layer_norm(f2, B * Tq, d, T("%s.norm2.weight", pfx)->data, T("%s.norm2.bias", pfx)->data, out);
    free(xq); free(xk); free(q); free(k); free(v); free(att); free(n1); free(f0); free(f1); free(f2);
    *Tq_out = Tq;
    return out;
}

/* PositionalEncoding table value pe[p][c]: models/layers.py:134-158 (fp32 arithmetic like torch's) */
static float pe_value(int p, int c, int d) {
    int k2 = c & ~1;
    float div = expf((float)k2 * (float)(-log(10000.0) / (double)d));
    float ang = (float)p * div;
    return (c & 1) ? cosf(ang) : sinf(ang);
}

/* MultiHeadAttentionLearnableQuery.forward: models/layers.py:273-301 (heads 8 x 256, FeedForward hidden 256, no
 * LayerNorm around the attention).  x [B][Tn][d].  cross != 0: the queries are the learnable probe (+ PE) and the block
 * returns [B][21][d] = ff(out) + out; else self-attention over x + PE with out = to_out(att) + (x + PE), then ff(out) + out. */
#define DHEAD_LQ 256
static float *mha_lq_block(const float *x, int B, int Tn, int d, int layer, int cross, int *T_out) {
    const int inner = HEADS * DHEAD_LQ, Tq = cross ? NJ : Tn;
    char pfx[96];
    snprintf(pfx, sizeof pfx, "joints_late_fusion.attn_fusion.%d", layer);
    float *xp = falloc((size_t)B * Tn * d);                 /* x = self.pos_embed(x) */
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < Tn; ++t)
            for (int c = 0; c < d; ++c) xp[((size_t)b * Tn + t) * d + c] = x[((size_t)b * Tn + t) * d + c] + pe_value(t, c, d);
    float *qin = xp;
    if (cross) {                                            /* probe.repeat(batch) then pos_embed(probe) */
        const float *probe = T("%s.probe", pfx)->data;
        qin = falloc((size_t)B * NJ * d);
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < NJ; ++t)
                for (int c = 0; c < d; ++c) qin[((size_t)b * NJ + t) * d + c] = probe[(size_t)t * d + c] + pe_value(t, c, d);
    }
    float *q = falloc((size_t)B * Tq * inner), *k = falloc((size_t)B * Tn * inner), *v = falloc((size_t)B * Tn * inner);
    linear(qin, B * Tq, d, T("%s.to_q.weight", pfx)->data, NULL, inner, q);
    linear(xp, B * Tn, d, T("%s.to_k.weight", pfx)->data, NULL, inner, k);
    linear(xp, B * Tn, d, T("%s.to_v.weight", pfx)->data, NULL, inner, v);
    float *att = falloc((size_t)B * Tq * inner);
    const acc_t scale = (acc_t)(1.0 / sqrt((double)DHEAD_LQ));
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < HEADS; ++h) {
            acc_t *dots = (acc_t *)malloc(sizeof(acc_t) * Tn);
            for (int i = 0; i < Tq; ++i) {
                const float *qi = q + ((size_t)b * Tq + i) * inner + h * DHEAD_LQ;
                acc_t mx = -INFINITY;
                for (int j = 0; j < Tn; ++j) {
                    const float *kj = k + ((size_t)b * Tn + j) * inner + h * DHEAD_LQ;
                    acc_t s = 0;
                    for (int c = 0; c < DHEAD_LQ; ++c) s += (acc_t)qi[c] * (acc_t)kj[c];
                    dots[j] = s * scale;
                    if (dots[j] > mx) mx = dots[j];
                }
                acc_t sum = 0;
                for (int j = 0; j < Tn; ++j) { dots[j] = (acc_t)exp((double)(dots[j] - mx)); sum += dots[j]; }
                float *o = att + ((size_t)b * Tq + i) * inner + h * DHEAD_LQ;
                for (int c = 0; c < DHEAD_LQ; ++c) {
                    acc_t s = 0;
                    for (int j = 0; j < Tn; ++j)
                        s += (dots[j] / sum) * (acc_t)v[((size_t)b * Tn + j) * inner + h * DHEAD_LQ + c];
                    o[c] = (float)s;
                }
            }
            free(dots);
        }
    size_t tot = (size_t)B * Tq * d;
    float *out = falloc(tot);
    linear(att, B * Tq, inner, T("%s.to_out.0.weight", pfx)->data, T("%s.to_out.0.bias", pfx)->data, d, out);
    if (!cross)
        for (size_t i = 0; i < tot; ++i) out[i] += xp[i];   /* out = out + x (x already carries the PE) */
    /* ff: LayerNorm -> Linear(d,256) -> GELU(erf) -> Linear(256,d); out = ff(out) + out */
    float *f0 = falloc(tot), *f1 = falloc((size_t)B * Tq * DHEAD_LQ), *f2 = falloc(tot);
    layer_norm(out, B * Tq, d, T("%s.ff.net.0.weight", pfx)->data, T("%s.ff.net.0.bias", pfx)->data, f0);
    linear(f0, B * Tq, d, T("%s.ff.net.1.weight", pfx)->data, T("%s.ff.net.1.bias", pfx)->data, DHEAD_LQ, f1);
    for (size_t i = 0; i < (size_t)B * Tq * DHEAD_LQ; ++i) {
        acc_t z = (acc_t)f1[i];
        f1[i] = (float)((acc_t)0.5 * z * ((acc_t)1 + (acc_t)erf((double)z * 0.70710678118654752440)));
    }
    linear(f1, B * Tq, DHEAD_LQ, T("%s.ff.net.4.weight", pfx)->data, T("%s.ff.net.4.bias", pfx)->data, d, f2);
    for (size_t i = 0; i < tot; ++i) f2[i] += out[i];
    if (qin != xp) free(qin);
    free(xp); free(q); free(k); free(v); free(att); free(out); free(f0); free(f1);
    *T_out = Tq;
    return f2;
}

/* hand graph: models/utils.py:108-120 (adj_mx_from_edges) + constants.py:37-41 (HAND_EDGES) */
static void hand_adjacency(float adj[NJ][NJ]) {
    static const int E[20][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {0, 5}, {5, 6}, {6, 7}, {7, 8}, {0, 9}, {9, 10},
                                 {10, 11}, {11, 12}, {0, 13}, {13, 14}, {14, 15}, {15, 16}, {0, 17}, {17, 18},
                                 {18, 19}, {19, 20}};
    memset(adj, 0, sizeof(float) * NJ * NJ);
    for (int e = 0; e < 20; ++e) { adj[E[e][0]][E[e][1]] = 1.f; adj[E[e][1]][E[e][0]] = 1.f; }
    for (int i = 0; i < NJ; ++i) adj[i][i] += 1.f; /* + sp.eye */
    for (int i = 0; i < NJ; ++i) {                 /* normalize_sparse_matrix: row-normalise */
        float rs = 0.f;
        for (int j = 0; j < NJ; ++j) rs += adj[i][j];
        float inv = rs != 0.f ? 1.f / rs : 0.f;
        for (int j = 0; j < NJ; ++j) adj[i][j] *= inv;
    }
}

/* ChebConv.forward (K=2 -> 3 terms, normalize=True): models/layers.py:387-445 */
static void cheb_conv(const float *x, int B, int in, int out, const char *name, int leaky, float *y) {
    float adj[NJ][NJ], L[NJ][NJ], Tk[3][NJ][NJ];
    hand_adjacency(adj);
    float dsq[NJ];
    for (int i = 0; i < NJ; ++i) { /* D = diag(rowsum ** -1/2) */
        float rs = 0.f;
        for (int j = 0; j < NJ; ++j) rs += adj[i][j];
        dsq[i] = 1.f / sqrtf(rs);
    }
    for (int i = 0; i < NJ; ++i)
        for (int j = 0; j < NJ; ++j) L[i][j] = (i == j ? 1.f : 0.f) - dsq[i] * adj[i][j] * dsq[j];
    for (int i = 0; i < NJ; ++i)
        for (int j = 0; j < NJ; ++j) {
            Tk[0][i][j] = i == j ? 1.f : 0.f;
            Tk[1][i][j] = L[i][j];
        }
    for (int i = 0; i < NJ; ++i)
        for (int j = 0; j < NJ; ++j) { /* 2 * L @ T1 - T0 */
            float s = 0.f;
            for (int m = 0; m < NJ; ++m) s += L[i][m] * Tk[1][m][j];
            Tk[2][i][j] = 2.f * s - Tk[0][i][j];
        }
    const float *Wt = T("%s.weight", name)->data, *bias = T("%s.bias", name)->data;
    float *tx = falloc((size_t)B * NJ * in), *part = falloc((size_t)B * NJ * out);
    memset(y, 0, sizeof(float) * B * NJ * out);
    for (int k = 0; k < 3; ++k) {
#pragma omp parallel for collapse(2)
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < NJ; ++i)
                for (int c = 0; c < in; ++c) {
                    acc_t s = 0;
                    for (int j = 0; j < NJ; ++j) s += (acc_t)Tk[k][i][j] * (acc_t)x[((size_t)b * NJ + j) * in + c];
                    tx[((size_t)b * NJ + i) * in + c] = (float)s;
                }
        gemm_nn(B * NJ, out, in, tx, in, Wt + (size_t)k * in * out, out, part, out);
        for (size_t i = 0; i < (size_t)B * NJ * out; ++i) y[i] += part[i];
    }
    for (size_t i = 0; i < (size_t)B * NJ * out; ++i) {
        float v = y[i] + bias[i % out];
        y[i] = leaky && v < 0.f ? 0.01f * v : v; /* nn.LeakyReLU() default slope */
    }
    free(tx);
    free(part);
}

/* ------------------------------------------------------------------ the forward */
typedef struct {
    int backbone;    /* 0 = resnet18, 1 = resnet34, 2 = resnet50_paper */
    int n_levels;    /* len(backbone_channels) */
    int channels[4]; /* backbone_channels */
    int num_views;
    int image_size, heatmap_size; /* data.image_size / data.heatmap_size (config constants) */
    int pos_mask;                 /* 1 pos2d | 2 crop | 4 sin */
    int fusion_layers;
    int use_gcn;
    int fusion;                   /* 0 = cross_attn (fusion.py:7-30), 1 = cross_attn_learnable_query (fusion.py:33-49) */
} hmvo_config;

/* The part of HandMvNet.forward behind the token matrix (handmvnet.py:225-229): joints_late_fusion, then joints_decoder.
 * tok [B][V*21][d] (before the positional encoding) is CONSUMED (freed).  Shared by hmvo_forward and hmvo_fuse_tokens. */
static void fuse_and_decode(const hmvo_config *cfg, int B, int d, float *tok, float *fused_out, float *joints_cam) {
    int V = cfg->num_views;
    /* ---- CrossAttentionFusion.forward: fusion.py:26-30 ; PositionalEncoding: layers.py:134-158 */
    int Tn = V * NJ;
    if (cfg->fusion == 1) {
        /* CrossAttentionFusionLearnableQuery.forward (fusion.py:47-49): 5 blocks, the middle one with the probe queries;
         * every block adds its own positional embedding, so nothing is added here whatever pos_enc says */
        float *f = tok;
        int Tcur = Tn;
        for (int l = 0; l < 5; ++l) {
            float *nf = mha_lq_block(f, B, Tcur, d, l, l == 2, &Tcur);
            free(f);
            f = nf;
        }
        tok = f;
    } else if (cfg->pos_mask & 4) {
        for (int p = 0; p < Tn; ++p)
            for (int c = 0; c < d; ++c) {
                int k2 = c & ~1;
                float div = expf((float)k2 * (float)(-log(10000.0) / (double)d));
                float ang = (float)p * div;
                float pe = (c & 1) ? cosf(ang) : sinf(ang);
                for (int b = 0; b < B; ++b) tok[((size_t)b * Tn + p) * d + c] += pe;
            }
    }
    int half = (cfg->fusion_layers - 1) / 2, Tcur = Tn;
    float *f = tok;
    for (int l = 0; l < (cfg->fusion == 1 ? 0 : cfg->fusion_layers); ++l) {
        float *nf = mha_block(f, B, Tcur, d, l, l == half ? NJ : 0, &Tcur);
        free(f);
        f = nf;
    }
    if (fused_out) memcpy(fused_out, f, sizeof(float) * (size_t)B * NJ * d);
    /* ---- decoder: nets.py:133-139 / 150-154 */
    if (cfg->use_gcn) {
        float *g1 = falloc((size_t)B * NJ * 256), *g2 = falloc((size_t)B * NJ * 64);
        cheb_conv(f, B, d, 256, "joints_decoder.joints_gcn1", 1, g1);
        cheb_conv(g1, B, 256, 64, "joints_decoder.joints_gcn2", 1, g2);
        cheb_conv(g2, B, 64, 3, "joints_decoder.joints_gcn3", 0, joints_cam);
        free(g1);
        free(g2);
    } else {
        float *g1 = falloc((size_t)B * NJ * 64);
        linear(f, B * NJ, d, T("joints_decoder.joints_fc1.weight")->data, T("joints_decoder.joints_fc1.bias")->data, 64, g1);
        for (size_t i = 0; i < (size_t)B * NJ * 64; ++i) g1[i] = g1[i] < 0.f ? 0.01f * g1[i] : g1[i];
        linear(g1, B * NJ, 64, T("joints_decoder.joints_fc2.weight")->data, T("joints_decoder.joints_fc2.bias")->data, 3,
               joints_cam);
        free(g1);
    }
    free(f);
}

/* Test entry: the fusion + decoder tail alone, on a token matrix supplied by the caller (e.g. the one an implementation
 * under test captured), so that an implementation's tail can be checked apart from the conditioning of what precedes it.
 * tokens [B][V*21][d] is not modified. */
int hmvo_fuse_tokens(const hmvo_config *cfg, int B, const float *tokens, float *fused_out, float *joints_cam) {
    g_missing = 0;
    g_err[0] = 0;
    int fdim = 0;
    for (int i = 0; i < cfg->n_levels; ++i) fdim += cfg->channels[i] / 2;
    int d = fdim + ((cfg->pos_mask & 1) ? 2 : 0) + ((cfg->pos_mask & 2) ? 10 : 0);
    size_t n = (size_t)B * cfg->num_views * NJ * d;
    float *tok = falloc(n);
    memcpy(tok, tokens, sizeof(float) * n);
    fuse_and_decode(cfg, B, d, tok, fused_out, joints_cam);
    return g_missing ? 3 : 0;
}

/* HandMvNet.forward: models/handmvnet.py:158-266.
 * x [B][V][3][H][W]; bbox [B][V][4]; intr [B][V][4].
 * Optional stage dumps (may be NULL): feat0 = feats[0] NCHW, coords_hm [B*V][21][2] (heat-map units),
 * tokens [B][V*21][d] (before PE), fused [B][21][d]. */
int hmvo_forward(const hmvo_config *cfg, int B, int H, int W, const float *x, const float *bbox, const float *intr,
                 float *joints_crop_img, float *joints_cam, float *heatmap, float *feat0_out, float *coords_out,
                 float *tokens_out, float *fused_out) {
    g_missing = 0;
    g_err[0] = 0;
    static const int blocks[3][4] = {{2, 2, 2, 2}, {3, 4, 6, 3}, {3, 4, 6, 3}};
    int V = cfg->num_views, N = B * V, paper = cfg->backbone == 2, hrnet = cfg->backbone >= 3;
    int C, Hc, Wc, h1, w1;
    const float *feats[4];
    int fc[4], fh[4], fw[4], nfe;
    float *lv[4] = {0, 0, 0, 0};
    float *hm;
    int hh, hw;
    if (hrnet) {
        /* handmvnet.py:46-57: HRNet returns a LIST (highest resolution first); pose_net = Conv2d(C0, 21, 3, 2, 1) */
        hrnet_forward(cfg->backbone == 4, x, N, H, W, lv, fc, fh, fw);
        nfe = 4;
        for (int i = 0; i < 4; ++i) feats[i] = lv[i];
        if (cfg->n_levels > 4 || fc[0] != cfg->channels[0]) { snprintf(g_err, sizeof g_err, "backbone_channels do not match the backbone"); return 2; }
        if (feat0_out) memcpy(feat0_out, feats[0], sizeof(float) * (size_t)N * fc[0] * fh[0] * fw[0]);
        hh = (fh[0] + 2 - 3) / 2 + 1; hw = (fw[0] + 2 - 3) / 2 + 1;
        hm = falloc((size_t)N * NJ * hh * hw);
        conv2d(feats[0], N, fc[0], fh[0], fw[0], T("pose_net.weight")->data, T("pose_net.bias")->data, NJ, 3, 3, 2, 1, hm, &hh, &hw);
    } else {
    /* ---- ResNet.forward: resnet.py:216-254 */
    float *c1 = falloc((size_t)N * 64 * (H / 2 + 1) * (W / 2 + 1));
    conv2d(x, N, 3, H, W, T("backbone.conv1.weight")->data, NULL, 64, 7, 7, 2, 3, c1, &h1, &w1);
    bn_eval(c1, N, 64, h1 * w1, "backbone.bn1", 1);
    float *cur = falloc((size_t)N * 64 * (h1 / 2 + 1) * (w1 / 2 + 1));
    maxpool_3s2p1(c1, N, 64, h1, w1, cur, &Hc, &Wc);
    free(c1);
    C = 64;
    int lc[3], lh[3], lw[3];
    for (int li = 0; li < 3; ++li) {
        int planes = 64 << li, stride = li == 0 ? 1 : 2;
        if (paper && li == 2) stride = 1; /* resnet.py:176-177 */
        for (int bi = 0; bi < blocks[cfg->backbone][li]; ++bi) {
            char pfx[64];
            snprintf(pfx, sizeof pfx, "backbone.layer%d.%d", li + 1, bi);
            float *nx = res_block(cur, N, &C, &Hc, &Wc, pfx, planes, bi == 0 ? stride : 1, paper);
            int keep = 0;
            for (int q = 0; q < 3; ++q) keep |= (lv[q] == cur);
            if (!keep) free(cur);
            cur = nx;
        }
        lv[li] = cur; lc[li] = C; lh[li] = Hc; lw[li] = Wc;
    }
    /* handmvnet.py:165-177: r18/34 -> [layer3, layer2, layer1]; paper -> [layer3] */
    nfe = paper ? 1 : 3;
    float *lvr[3] = {lv[0], lv[1], lv[2]};
    for (int i = 0; i < nfe; ++i) { feats[i] = lvr[2 - i]; fc[i] = lc[2 - i]; fh[i] = lh[2 - i]; fw[i] = lw[2 - i]; }
    if (cfg->n_levels > nfe || fc[0] != cfg->channels[0]) {
        snprintf(g_err, sizeof g_err, "backbone_channels do not match the backbone");
        return 2;
    }
    if (feat0_out) memcpy(feat0_out, feats[0], sizeof(float) * (size_t)N * fc[0] * fh[0] * fw[0]);
    /* ---- pose_net: handmvnet.py:70-86,180 */
    if (paper) {
        float *p0 = falloc((size_t)N * 512 * fh[0] * fw[0]);
        conv2d(feats[0], N, fc[0], fh[0], fw[0], T("pose_net.0.weight")->data, T("pose_net.0.bias")->data, 512, 1, 1, 1,
               0, p0, &hh, &hw);
        bn_eval(p0, N, 512, hh * hw, "pose_net.1", 1);
        hm = falloc((size_t)N * NJ * hh * hw);
        conv2d(p0, N, 512, hh, hw, T("pose_net.3.weight")->data, T("pose_net.3.bias")->data, NJ, 1, 1, 1, 0, hm, &hh, &hw);
        free(p0);
    } else {
        hh = 2 * fh[0]; hw = 2 * fw[0];
        float *p0 = falloc((size_t)N * 128 * hh * hw);
        conv_transpose_4s2p1(feats[0], N, fc[0], fh[0], fw[0], T("pose_net.0.weight")->data, T("pose_net.0.bias")->data,
                             128, p0);
        bn_eval(p0, N, 128, hh * hw, "pose_net.1", 1);
        float *p1 = falloc((size_t)N * 64 * hh * hw);
        conv2d(p0, N, 128, hh, hw, T("pose_net.3.weight")->data, T("pose_net.3.bias")->data, 64, 3, 3, 1, 1, p1, &h1, &w1);
        bn_eval(p1, N, 64, hh * hw, "pose_net.4", 1);
        hm = falloc((size_t)N * NJ * hh * hw);
        conv2d(p1, N, 64, hh, hw, T("pose_net.6.weight")->data, T("pose_net.6.bias")->data, NJ, 3, 3, 1, 1, hm, &h1, &w1);
        free(p0);
        free(p1);
    }
    } /* resnet */
    if (heatmap) memcpy(heatmap, hm, sizeof(float) * (size_t)N * NJ * hh * hw);
    /* ---- soft-argmax: handmvnet.py:182 */
    float *coords = falloc((size_t)N * NJ * 2);
    soft_argmax_2d(hm, N * NJ, hh, hw, coords);
    free(hm);
    if (coords_out) memcpy(coords_out, coords, sizeof(float) * (size_t)N * NJ * 2);
    /* ---- sample nets + token assembly: handmvnet.py:185-225 */
    int fdim = 0;
    for (int i = 0; i < cfg->n_levels; ++i) fdim += cfg->channels[i] / 2;
    int d = fdim + ((cfg->pos_mask & 1) ? 2 : 0) + ((cfg->pos_mask & 2) ? 10 : 0);
    float *tok = falloc((size_t)N * NJ * d);
    int col = 0;
    for (int i = 0; i < cfg->n_levels; ++i) {
        int ci = fc[i], co = ci / 2;
        if (ci != cfg->channels[i]) { snprintf(g_err, sizeof g_err, "backbone_channels[%d] mismatch", i); return 2; }
        float *sf = falloc((size_t)N * co * fh[i] * fw[i]);
        char nm[64];
        snprintf(nm, sizeof nm, "sample_nets.%d.conv", i);
        conv2d(feats[i], N, ci, fh[i], fw[i], T("%s.0.weight", nm)->data, T("%s.0.bias", nm)->data, co, 1, 1, 1, 0, sf,
               &h1, &w1); /* nets.py:60: conv over the WHOLE map, then sample */
        snprintf(nm, sizeof nm, "sample_nets.%d.conv.1", i);
        bn_eval(sf, N, co, h1 * w1, nm, 1);
        sample_joint_features(sf, N, co, fh[i], fw[i], coords, tok, d, col);
        col += co;
        free(sf);
    }
    if (cfg->pos_mask & 1) { /* pos2d: handmvnet.py:189-191 */
        for (int r = 0; r < N * NJ; ++r) { tok[(size_t)r * d + col] = coords[2 * r]; tok[(size_t)r * d + col + 1] = coords[2 * r + 1]; }
        col += 2;
    }
    if (cfg->pos_mask & 2) { /* crop FoV: handmvnet.py:205-222; utils.py:134-171 */
        for (int n = 0; n < N; ++n) {
            const float *bb = bbox + 4 * n, *in = intr + 4 * n;
            float px[5] = {bb[0], bb[0], bb[2], bb[2], (bb[0] + bb[2]) / 2.f};
            float py[5] = {bb[1], bb[3], bb[1], bb[3], (bb[1] + bb[3]) / 2.f};
            float fov[10];
            for (int p = 0; p < 5; ++p) {
                fov[2 * p] = atanf((px[p] - in[2]) / in[0]);
                fov[2 * p + 1] = atanf((py[p] - in[3]) / in[1]);
            }
            for (int j = 0; j < NJ; ++j) memcpy(tok + ((size_t)n * NJ + j) * d + col, fov, sizeof fov);
        }
        col += 10;
    }
    for (int l = 0; l < 4; ++l) free(lv[l]);
    if (tokens_out) memcpy(tokens_out, tok, sizeof(float) * (size_t)N * NJ * d);
    fuse_and_decode(cfg, B, d, tok, fused_out, joints_cam);
    /* ---- handmvnet.py:252: joint_coords * image_size / heatmap_size */
    for (size_t i = 0; i < (size_t)N * NJ * 2; ++i)
        joints_crop_img[i] = coords[i] * (float)cfg->image_size / (float)cfg->heatmap_size;
    free(coords);
    return g_missing ? 3 : 0;
}
