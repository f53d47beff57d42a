// engine.hip -- host side of libhandmv.so: weight ingestion (BN folding, K-major repack),
// workspace planning and the forward orchestration behind the C ABI of include/handmv.h.
//
// The forward mirrors HandMvNet.forward (/root/reference/src/models/handmvnet.py:158-266)
// stage by stage, but NHWC end to end and with every conv/linear routed to the one
// implicit-GEMM MFMA kernel family (conv_igemm.hip).  There is no CPU fallback: every
// entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/handmv.h"
#include "kernels.h"

using namespace hmv;

namespace {

constexpr int NJ = 21, HEADS = 8, DHEAD = 128, INNER = HEADS * DHEAD;
constexpr int DHEAD_LQ = 256, INNER_LQ = HEADS * DHEAD_LQ;   // MultiHeadAttentionLearnableQuery, layers.py:241
const int kBlocks[3][4] = {{2, 2, 2, 2}, {3, 4, 6, 3}, {3, 4, 6, 3}};
const int kHrChannels[2][4] = {{40, 80, 160, 320}, {64, 128, 256, 512}};   // hrnet.py:430-447
inline int cpad(int c) { return (c + 3) / 4 * 4; }   // NHWC channel stride: 16-byte pixels are all the conv kernel needs

std::string g_create_err;

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

// One GEMM-shaped layer on the device: Wt [Cout_pad][Kpad] + bias [Cout_pad].
struct Layer {
    float *w = nullptr, *bias = nullptr;
    int Cin = 0, Cout = 0, R = 1, S = 1, K = 0, Kpad = 0, Cout_pad = 0;
    int Kreal = 0;      // reduction length without channel padding (FLOP accounting)
    int in_real = 0;    // real input elements per input pixel when that is not Kreal / (R * S) (the space-to-depth stem)
    int acc_shift = 0;  // HMV_F32X3: the packed weights are W * 2^acc_shift, the epilogue scales the accumulator back
    bool x3n = false;   // HMV_F32X3: fused split reduction (one tile load per three products; conv_igemm.hip), else the cwrap scheme
    int plane = 0;      // HMV_F32X3 (split operands): physical channels per (hi | lo) plane; Cin is then the virtual 3 * plane
    int rd_cout = 0;    // row-decomposed 3x3 (conv_igemm.hip, RD): the real Cout; Cout / R / S then describe the 3x1 GEMM
    bool f16 = false;   // operands (activations + packed weights) are fp16; bias stays fp32
    bool tall = false;  // fp16 3x3 packed in conv_ht.hip's K order (32-channel chunk, r, s, c % 32): runs on that kernel at every batch
    std::string label;
};

struct Block {
    Layer c1, c2, c3, ds;
    bool has_ds = false;
    int stride = 1;
    // Bottleneck with a downsample branch: conv3 + BN3 and downsample conv + BN as ONE GEMM over the concatenated reduction
    // [t2 | x] . [s3 W3 ; sds Wds] + (b3 + bds) -- the downsample output is never written or re-read (resnet.py:124-144)
    Layer c3ds;
    bool fused_ds = false;
};

// HRNet (backbones/hrnet.py): one HighResolutionModule = per-branch BasicBlocks + the fuse layers
struct HrModule {
    Layer br[4][4][2];             // [branch][block][conv1 | conv2]
    std::vector<Layer> fuse[4][4]; // [i][j]: j > i one 1x1 conv; j < i a chain of (i - j) stride-2 3x3 convs
    Layer fup[4][4];               // fp16 mode: the j > i 1x1 convs once more as fp32 [Cout][Cin] for hr_fuse.hip (the fp32 mode reads fuse[i][j][0])
};
struct HrNet {
    Layer conv1, conv2;            // stem: two 3x3 stride-2 convs
    std::vector<Block> layer1;     // 4 Bottlenecks (planes 64)
    std::vector<Layer> trans[3][4];
    std::vector<HrModule> stage[3];
    int ch[4] = {0, 0, 0, 0};
};

struct AttnLayer {
    Layer qkv, out, ff1, ff2;
    Layer out_x3;   // fp16-kernel modes: to_out once more as a split-pair GEMM (Loader::linear_x3; gemm_x3.hip), fed by attention rows
                    // written as (hi, lo) pairs; its bias stays with `out` (ff_block_kernel adds it)
    float *n1g = nullptr, *n1b = nullptr, *n2g = nullptr, *n2b = nullptr, *fg = nullptr, *fb = nullptr;
    // learnable-query fusion (layers.py:240-301): the probe block projects only K and V from the tokens; its queries
    // to_q(probe + PE) do not depend on the input and are computed once at load time ([21][2048])
    Layer kv;
    float *qprobe = nullptr;
};

// First-fit planner over one contiguous arena.  Run once "dry" to size the workspace and
// once for real; both runs issue the same alloc/free sequence so offsets agree.
struct Arena {
    struct Seg { size_t off, size; };
    std::vector<Seg> free_list;  // sorted by offset
    std::map<size_t, size_t> live;
    size_t high = 0, top = 0;
    char *base = nullptr;
    void reset(char *b) { free_list.clear(); live.clear(); high = top = 0; base = b; }
    float *alloc(size_t floats) {
        size_t bytes = (floats * sizeof(float) + 255) / 256 * 256;
        if (bytes == 0) bytes = 256;
        for (size_t i = 0; i < free_list.size(); ++i) {
            if (free_list[i].size >= bytes) {
                size_t off = free_list[i].off;
                if (free_list[i].size == bytes) free_list.erase(free_list.begin() + i);
                else { free_list[i].off += bytes; free_list[i].size -= bytes; }
                live[off] = bytes;
                return reinterpret_cast<float *>(base + off);
            }
        }
        size_t off = top;
        top += bytes;
        if (top > high) high = top;
        live[off] = bytes;
        return reinterpret_cast<float *>(base + off);
    }
    void release(float *p) {
        if (!p) return;
        size_t off = (size_t)(reinterpret_cast<char *>(p) - base);
        auto it = live.find(off);
        if (it == live.end()) return;
        Seg s{off, it->second};
        live.erase(it);
        size_t i = 0;
        while (i < free_list.size() && free_list[i].off < s.off) ++i;
        free_list.insert(free_list.begin() + i, s);
        // coalesce neighbours
        if (i + 1 < free_list.size() && free_list[i].off + free_list[i].size == free_list[i + 1].off) {
            free_list[i].size += free_list[i + 1].size;
            free_list.erase(free_list.begin() + i + 1);
        }
        if (i > 0 && free_list[i - 1].off + free_list[i - 1].size == free_list[i].off) {
            free_list[i - 1].size += free_list[i].size;
            free_list.erase(free_list.begin() + i);
            --i;
        }
        if (!free_list.empty() && free_list.back().off + free_list.back().size == top) {
            top = free_list.back().off;
            free_list.pop_back();
        }
    }
};

struct ProfRec {
    const char *name;
    std::string label;
    double flops;
    double bytes;   // algorithmic HBM bytes of the launch: every operand / result element moved once
    hipEvent_t e0, e1;
};

}  // namespace

struct hmv_engine {
    hmv_config cfg{};
    std::string err;
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    std::vector<void *> dev_allocs;

    // derived shape facts
    int d = 0, ldt = 0, fdim = 0;
    bool paper = false;
    bool lq = false;   // model.fusion == cross_attn_learnable_query

    Layer stem;
    HrNet hr;
    bool hrnet = false;
    std::vector<Block> blocks[3];
    Layer pose0, pose1, pose2;   // r50: pose0 (1x1 1024->512), pose1 (1x1 512->21); r18/34: pose1 (3x3 128->64), pose2 (3x3 64->21)
    Layer deconv[4];             // r18/34 ConvTranspose2d as 4 sub-pixel 2x2 convs (phase a*2+b)
    Layer deconv_all;            // ... the same four with their weight blocks back to back (one launch); w == nullptr: not used
    size_t deconv_stride = 0;    // elements between the phases' weight blocks
    Layer sample[4];
    std::vector<AttnLayer> attn;
    Layer gcn[3];
    float *gcn_bias[3] = {nullptr, nullptr, nullptr};
    float *cheb_t = nullptr;
    Layer fc1, fc2;
    float *pe = nullptr;
    float *zero_bias = nullptr;   // 4096 zeros: bias of split-K slices

    char *arena = nullptr;
    size_t arena_bytes = 0;
    int reserved_batch = 0;
    Arena plan;

    bool capture = false;
    int cap_batch = 0;
    float *cap_feat0 = nullptr, *cap_coords = nullptr, *cap_tokens = nullptr, *cap_fused = nullptr;
    size_t cap_feat0_n = 0, cap_coords_n = 0, cap_tokens_n = 0, cap_fused_n = 0;

    // set only for the duration of hmv_forward_frames: raw camera frames instead of prepared NCHW input
    struct FrameSrc {
        const uint8_t *frames = nullptr;
        const int *boxes = nullptr;
        int fh = 0, fw = 0;
        float mean[3] = {0, 0, 0}, std[3] = {1, 1, 1};
    } fsrc;

    // hipGraph replay of repeated forwards (same batch and the same caller buffers): the ~100 launches of a forward
    // become one graph launch.  Opt-in: measured on MI355X it does not change throughput (eager enqueue already runs
    // ahead of the GPU); what it saves is host time per forward.
    // A key is first run eagerly, captured on its second use, replayed from then on.
    typedef std::array<uintptr_t, 12> GraphKey;
    struct GraphEntry { GraphKey key; hipGraphExec_t exec; unsigned long long stamp; };
    bool graphs = false;
    // fused tail kernels (fusion_kernels.hip); HMV_NO_FFFUSE=1 / HMV_NO_CHEBFUSE=1 in the environment or hmv_set_tail_fusion(h, 0)
    // select the launch-per-op path (A/B runs, the equivalence test).  Part of the workspace plan: changing them re-plans.
    bool ff_fuse = true, cheb_fuse = true;
    bool hr_fuse = true;      // HRNet: the up-sampling terms of a fuse layer as one launch (hr_fuse.hip); hmv_set_hr_fusion(h, 0): one launch per term
    bool hr_overlap = true;   // HRNet, fp16-kernel modes: a module's lowest-resolution branch runs on a second stream beside the branch above it
                              // (hmv_set_hr_fusion bit 1; measured hr40 fp16 -1.2 %, f32x3 -0.9 %, fp32 +0.3 %: off in the fp32 mode)
    hipStream_t aux = nullptr;            // the second stream (forked from / joined into the caller's stream by events: also under graph capture)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool chain_fuse = true;   // conv3 -> next block's conv1 in one launch (conv_stream.hip chain); hmv_set_chain_fusion(h, 0) / HMV_NO_CHAIN=1
    std::vector<GraphEntry> gcache;
    std::vector<GraphKey> gseen;     // buffer sets run eagerly once and not captured yet (callers often alternate between a few)
    hipStream_t gstream = nullptr;   // capture happens here (the caller's stream may be the NULL stream, which cannot capture)
    unsigned long long gclock = 0, greplays = 0;
    void drop_graphs() {
        for (auto &e : gcache) (void)hipGraphExecDestroy(e.exec);
        gcache.clear();
        gseen.clear();
    }

    bool profiling = false;
    std::vector<ProfRec> prof;
    size_t prof_used = 0;
    int launches = 0;      // device operations (kernels, memsets, copies) enqueued by the last eager / captured forward

    int fail(int code, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return (h)->fail(HMV_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

// ------------------------------------------------------------------ weight ingestion helpers
struct Loader {
    hmv_engine *h;
    int rc = HMV_OK;

    const HostTensor *get(const std::string &key, std::initializer_list<int64_t> shape) {
        auto it = h->host.find(key);
        if (it == h->host.end()) {
            if (rc == HMV_OK) rc = h->fail(HMV_ERR_MISSING_TENSOR, "Missing key(s) in state_dict: \"%s\"", key.c_str());
            return nullptr;
        }
        std::vector<int64_t> want(shape);
        if (it->second.shape != want) {
            if (rc == HMV_OK) {
                std::string got, exp;
                for (auto v : it->second.shape) got += std::to_string(v) + ",";
                for (auto v : want) exp += std::to_string(v) + ",";
                rc = h->fail(HMV_ERR_SHAPE, "size mismatch for %s: got [%s] expected [%s]", key.c_str(), got.c_str(), exp.c_str());
            }
            return nullptr;
        }
        return &it->second;
    }

    float *upload(const std::vector<float> &v) {
        if (rc != HMV_OK) return nullptr;
        float *p = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), v.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            rc = h->fail(HMV_ERR_HIP, "weight upload failed: %s", hipGetErrorString(e));
            return nullptr;
        }
        h->dev_allocs.push_back(p);
        return p;
    }

    void *upload_bytes(const void *src, size_t bytes) {
        if (rc != HMV_OK) return nullptr;
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            rc = h->fail(HMV_ERR_HIP, "weight upload failed: %s", hipGetErrorString(e));
            return nullptr;
        }
        h->dev_allocs.push_back(p);
        return p;
    }

    // BatchNorm (eval) folded to scale/shift in double: resnet.py:59-74
    bool bn_fold(const std::string &prefix, int C, std::vector<double> &scale, std::vector<double> &shift) {
        const HostTensor *g = get(prefix + ".weight", {C}), *b = get(prefix + ".bias", {C});
        const HostTensor *rm = get(prefix + ".running_mean", {C}), *rv = get(prefix + ".running_var", {C});
        if (!g || !b || !rm || !rv) return false;
        scale.resize(C);
        shift.resize(C);
        for (int c = 0; c < C; ++c) {
            scale[c] = (double)g->data[c] / std::sqrt((double)rv->data[c] + 1e-5);
            shift[c] = (double)b->data[c] - (double)rm->data[c] * scale[c];
        }
        return true;
    }

    bool split = false;   // HMV_F32X3: fp16 layers carry [W_hi | W_hi | W_lo] against the activation sequence hi, lo, hi

    // Generic finish: `wt(o, k)` supplies the un-scaled weight for output o, packed index k.
    template <typename F>
    void finish(Layer &L, const std::string &label, int Cin, int Cout, int R, int S, int K, F wt,
                const std::vector<double> *scale, const std::vector<double> *shift, const float *conv_bias, bool f16 = false,
                const std::vector<unsigned char> *lo_plane = nullptr) {
        L.label = label;
        L.Cin = Cin; L.Cout = Cout; L.R = R; L.S = S; L.K = K;
        L.f16 = f16;
        L.Kpad = round_up(K, f16 ? 64 : 32);
        L.Cout_pad = round_up(Cout, 256);
        if (rc != HMV_OK) return;
        std::vector<float> w((size_t)L.Cout_pad * L.Kpad, 0.f), b((size_t)L.Cout_pad, 0.f);
        for (int o = 0; o < Cout; ++o) {
            const double sc = scale ? (*scale)[o] : 1.0;
            for (int k = 0; k < K; ++k) w[(size_t)o * L.Kpad + k] = (float)((double)wt(o, k) * sc);
            double bb = shift ? (*shift)[o] : 0.0;
            if (conv_bias) bb += (double)conv_bias[o] * sc;
            b[o] = (float)bb;
        }
        if (f16) {   // BN scale is folded in fp32/double first, THEN rounded once to fp16
            std::vector<_Float16> wh(w.size());
            if (lo_plane) {   // split layers: scale by a power of two so that max |W| ~ 2^14 and W_lo stays a NORMAL fp16
                float mx = 0.f;
                for (float v : w) mx = std::max(mx, std::fabs(v));
                int sh = 0;
                while (sh < 24 && mx > 0.f && mx * 2.f <= 16384.f) { mx *= 2.f; ++sh; }
                while (sh > -24 && mx > 16384.f) { mx *= 0.5f; --sh; }   // huge folded weights (tiny running_var): scale down instead
                L.acc_shift = sh;
                const float f = std::ldexp(1.f, sh);
                for (float &v : w) v *= f;
            }
            for (size_t i = 0; i < w.size(); ++i) {
                const _Float16 hi = (_Float16)w[i];
                // split layers: packed index k of the third plane carries the residue W - fp16(W)
                wh[i] = (lo_plane && (*lo_plane)[i % (size_t)L.Kpad]) ? (_Float16)(w[i] - (float)hi) : hi;
            }
            L.w = static_cast<float *>(upload_bytes(wh.data(), wh.size() * sizeof(_Float16)));
        } else {
            L.w = upload(w);
        }
        L.bias = upload(b);
    }

    // nn.Conv2d weight OIHW (+ optional bias key) followed by an optional BN
    // (wsrc: an already re-arranged OIHW weight instead of the state_dict tensor `wkey` -- the space-to-depth stem)
    void conv(Layer &L, const std::string &label, const std::string &wkey, const std::string &bkey, const std::string &bn,
              int Cout, int Cin, int R, int S, int cin_pad = 0, bool f16 = false, bool rd = false, const float *wsrc = nullptr, bool tall = false) {
        const HostTensor *w = wsrc ? nullptr : get(wkey, {Cout, Cin, R, S});
        const HostTensor *cb = bkey.empty() ? nullptr : get(bkey, {Cout});
        std::vector<double> sc, sh;
        const bool has_bn = !bn.empty();
        if (has_bn && !bn_fold(bn, Cout, sc, sh)) return;
        if ((!w && !wsrc) || (!bkey.empty() && !cb)) return;
        const int plane = cin_pad ? cin_pad : Cin;            // physical channels (per plane when split)
        const bool sp = split && f16;
        // fused split reduction: a k-step = 32 channels as [32 hi | 32 lo] halfs, K order (32-channel chunk, r, s, plane, c % 32)
        const bool x3n = sp && plane % 8 == 0 && !HMV_DEV_ENV("HMV_NO_X3N");
        const int cp = sp ? (x3n ? 2 * plane : 3 * plane) : plane;   // the channel count the kernel's K order walks
        const float *wd = wsrc ? wsrc : w->data.data();
        const bool chunked = cp % (f16 ? 64 : 32) == 0 && (!sp || (x3n && plane % 32 == 0) || (!x3n && plane % 64 == 0));
        tall = tall && f16 && !sp && !cin_pad && !rd && !wsrc;
        auto wt = [=](int o, int k) -> float {
            int c, tap;
            const int CH = (f16 && !tall) ? 64 : 32;
            if (x3n && chunked) {
                const int step = k / 64, c32 = k % 32;
                tap = step % (R * S);
                c = (step / (R * S)) * 32 + c32;
            } else if (x3n) {   // dense fused: 4 (tap, 8-channel) vectors per step, K order (r, s, c) over the real channels
                const int g = (k / 64) * 4 + (k % 32) / 8, cpt = plane / 8;
                tap = g / cpt;
                c = (g % cpt) * 8 + k % 8;
                if (tap >= R * S) return 0.f;
            } else if (chunked) {   // K order (chunk, r, s, c % CH), CH = 32 (fp32) / 64 (fp16): conv_igemm.hip
                const int chunk = k / (CH * R * S), rem = k % (CH * R * S);
                tap = rem / CH;
                c = chunk * CH + rem % CH;
            } else {          // dense K order (r, s, c) over the real channels: the stem, HRNet's 40 / 80-channel tensors
                c = k % cp;
                tap = k / cp;
            }
            if (sp && !x3n) c %= plane;   // virtual channel -> channel; which plane it is only decides hi / lo (lo_plane below)
            if (c >= Cin) return 0.f;
            return wd[(((size_t)o * Cin + c) * R + tap / S) * S + tap % S];
        };
        std::vector<unsigned char> lo_plane;
        // packed reduction length: the dense fused scheme packs 4 (tap, 8-channel) vectors per 64-wide step
        const int Kpack = (x3n && !chunked) ? (R * S * (plane / 8) + 3) / 4 * 64 : R * S * cp;
        if (sp) {   // packed indices whose virtual channel lies in the third plane
            const int K = Kpack, Kp = round_up(K, 64), CH = 64;
            lo_plane.assign(Kp, 0);
            for (int k = 0; k < K; ++k) {
                int c;
                if (x3n) { lo_plane[k] = (k % 64) >= 32; continue; }
                if (chunked) { const int chunk = k / (CH * R * S), rem = k % (CH * R * S); c = chunk * CH + rem % CH; }
                else c = k % cp;
                lo_plane[k] = c / plane == 2;
            }
        }
        if (rd && R == 3 && S == 3 && !cb && !sp) {
            // row-decomposed packing: GEMM output o' = (s, n), reduction k = (r, c); bias / BN shift live in group s = 0
            const int CH = f16 ? 64 : 32;
            auto wt3 = [=](int o2, int k) -> float {
                const int sx = o2 / Cout, n = o2 % Cout;
                int c, r;
                if (cp % CH == 0) {
                    const int chunk = k / (CH * 3), rem = k % (CH * 3);
                    r = rem / CH;
                    c = chunk * CH + rem % CH;
                } else {
                    c = k % cp;
                    r = k / cp;
                }
                if (c >= Cin) return 0.f;
                return wd[(((size_t)n * Cin + c) * 3 + r) * 3 + sx];
            };
            std::vector<double> sc3(3 * Cout, 1.0), sh3(3 * Cout, 0.0);
            for (int o2 = 0; o2 < 3 * Cout; ++o2) {
                if (has_bn) sc3[o2] = sc[o2 % Cout];
                if (has_bn && o2 < Cout) sh3[o2] = sh[o2];
            }
            finish(L, label, cp, 3 * Cout, 3, 1, 3 * cp, wt3, &sc3, &sh3, nullptr, f16);
            L.rd_cout = Cout;
            L.Kreal = 3 * Cin;   // (3 * Cout) x (3 * Cin) = Cout x 9 * Cin: the FLOP accounting sees the real convolution
            return;
        }
        finish(L, label, cp, Cout, R, S, Kpack, wt, has_bn ? &sc : nullptr, has_bn ? &sh : nullptr,
               cb ? cb->data.data() : nullptr, f16, sp ? &lo_plane : nullptr);
        L.Kreal = R * S * Cin;
        L.tall = tall;
        if (sp) { L.plane = plane; L.x3n = x3n; }
    }

    // nn.Linear weight [out][in] (+ optional bias)
    void linear(Layer &L, const std::string &label, const std::string &wkey, const std::string &bkey, int out, int in) {
        const HostTensor *w = get(wkey, {out, in});
        const HostTensor *b = bkey.empty() ? nullptr : get(bkey, {out});
        if (!w || (!bkey.empty() && !b)) return;
        const float *wd = w->data.data();
        auto wt = [=](int o, int k) -> float { return wd[(size_t)o * in + k]; };
        finish(L, label, round_up(in, 32), out, 1, 1, in, wt, nullptr, nullptr, b ? b->data.data() : nullptr);
    }

    // A bias-free fp32 linear on the fused split kernels (the fp16 / f32x3 modes' qkv projections): token rows arrive as
    // [hi plane | lo plane] halfs, a k-step is 32 columns as [32 W_hi | 32 W_lo]; fp32-equivalent, 3 fp16 MFMAs per fp32 one
    template <typename F>
    void linear_x3(Layer &L, const std::string &label, int plane, int out, int in, F wt) {
        auto wp = [=](int o, int k) -> float {
            const int c = (k / 64) * 32 + k % 32;
            return c < in ? wt(o, c) : 0.f;
        };
        std::vector<unsigned char> lo((size_t)2 * plane);
        for (int k = 0; k < 2 * plane; ++k) lo[k] = (k % 64) >= 32;
        finish(L, label, 2 * plane, out, 1, 1, 2 * plane, wp, nullptr, nullptr, nullptr, true, &lo);
        L.Kreal = in;
        L.plane = plane;
        L.x3n = true;
    }

    float *vec(const std::string &key, int n) {
        const HostTensor *t = get(key, {n});
        return t ? upload(t->data) : nullptr;
    }
};

int backbone_level_channels(const hmv_config &c, int level /*0 = layer1*/) {
    return (64 << level) * (c.backbone == HMV_RESNET50_PAPER ? 4 : 1);
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

const char *hmv_version(void) {
#ifdef HMV_DEV_KNOBS
    return "handmv-mi355x 0.4-dev (gfx950; arithmetic modes: f32 = native fp32 MFMA, f16 = fp16 storage + fp16 MFMA with fp32 accumulation, "
           "f32x3 = (hi, lo) fp16 pairs on the fp16 MFMA, fp32-equivalent; development knobs compiled in)";
#else
    return "handmv-mi355x 0.4 (gfx950; arithmetic modes: f32 = native fp32 MFMA, f16 = fp16 storage + fp16 MFMA with fp32 accumulation, "
           "f32x3 = (hi, lo) fp16 pairs on the fp16 MFMA, fp32-equivalent)";
#endif
}

// The tile conv_igemm's launcher rule gives a conv / GEMM of M output pixels, Cout channels and reduction length K ("256x256", "128x32",
// "256x128,k16,w8" ...): host logic only, no GPU call -- the CPU tests pin the rules that were measured on the hardware.
const char *hmv_tile_rule(int32_t M, int32_t Cout, int32_t K, int32_t f16, int32_t has_residual) {
    const ConvTile t = conv_pick_tile(M, Cout, K, f16 != 0, has_residual != 0);
    const char *n = f16 ? conv_tile_name_f16(t, 0) : conv_tile_name(t, 0);   // "conv_igemm_f32<256x256,taps>"
    static thread_local char buf[64];
    const char *lt = strchr(n, '<'), *comma = lt ? strrchr(n, ',') : nullptr;
    if (!lt || !comma || comma < lt) return n;
    const size_t len = (size_t)(comma - lt - 1) < sizeof(buf) - 1 ? (size_t)(comma - lt - 1) : sizeof(buf) - 1;
    memcpy(buf, lt + 1, len);
    buf[len] = 0;
    return buf;
}

const char *hmv_last_error(hmv_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int hmv_create(const hmv_config *cfg, hmv_handle *out) {
    auto bad = [&](const char *m) { g_create_err = m; return HMV_ERR_ARG; };
    if (!cfg || !out) return bad("null argument");
    if (cfg->struct_size != (int32_t)sizeof(hmv_config)) return bad("hmv_config.struct_size mismatch (ABI)");
    if (cfg->backbone < HMV_RESNET18 || cfg->backbone > HMV_HRNET_W64) return bad("Supports only 18, 34, 50_paper (resnet) and w40, w64 (hrnet)");
    if (cfg->num_views < 1 || cfg->num_views > 48) return bad("num_views must be in [1, 48]");
    if (cfg->fusion != HMV_FUSION_CROSS_ATTN && cfg->fusion != HMV_FUSION_LEARNABLE_QUERY) return bad("Invalid fusion type");
    if (cfg->fusion == HMV_FUSION_CROSS_ATTN && (cfg->fusion_layers < 1 || cfg->fusion_layers % 2 != 1))
        return bad("num_layers must be an odd number");
    if (cfg->dtype != HMV_F32 && cfg->dtype != HMV_F16 && cfg->dtype != HMV_F32X3) { g_create_err = "dtype must be HMV_F32, HMV_F16 or HMV_F32X3"; return HMV_ERR_UNSUPPORTED; }
    // ResNet backbones take any frame size the reference's convs take (resnet.py:216-254); HRNet's fuse layers add 2^k-upsampled
    // maps (hrnet.py:194-212), which only line up -- in the reference as here -- when every branch size is exact
    if (cfg->height < 32 || cfg->width < 32) return bad("frame height/width must be at least 32");
    if (cfg->backbone >= HMV_HRNET_W40 && (cfg->height % 32 || cfg->width % 32))
        return bad("HRNet frame height/width must be multiples of 32 (its fuse layers need exact 2x branch sizes)");
    if (cfg->image_size <= 0 || cfg->heatmap_size <= 0) return bad("image_size / heatmap_size must be positive");
    const bool paper = cfg->backbone == HMV_RESNET50_PAPER;
    const bool hrnet = cfg->backbone >= HMV_HRNET_W40;
    if (cfg->n_levels < 1 || cfg->n_levels > (hrnet ? 4 : (paper ? 1 : 3))) return bad("backbone_channels has too many levels for this backbone");
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->channels[i] != (hrnet ? kHrChannels[cfg->backbone - HMV_HRNET_W40][i] : backbone_level_channels(*cfg, 2 - i)))
            return bad("backbone_channels do not match the backbone");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_err = "no HIP device: libhandmv has no CPU fallback";
        return HMV_ERR_HIP;
    }
    if (cfg->device < 0 || cfg->device >= ndev) return bad("device ordinal out of range");
    hmv_engine *h = new hmv_engine();
    if (const char *g = getenv("HMV_GRAPHS")) h->graphs = atoi(g) != 0;
    h->ff_fuse = HMV_DEV_ENV("HMV_NO_FFFUSE") == nullptr;
    h->cheb_fuse = HMV_DEV_ENV("HMV_NO_CHEBFUSE") == nullptr;
    h->cfg = *cfg;
    h->paper = paper;
    h->hrnet = hrnet;
    h->lq = cfg->fusion == HMV_FUSION_LEARNABLE_QUERY;
    h->fdim = 0;
    for (int i = 0; i < cfg->n_levels; ++i) h->fdim += cfg->channels[i] / 2;
    h->d = h->fdim + ((cfg->pos_enc & HMV_POS2D) ? 2 : 0) + ((cfg->pos_enc & HMV_POS_CROP) ? 10 : 0);
    h->ldt = round_up(h->d, 32);
    *out = h;
    return HMV_OK;
}

int hmv_set_tensor(hmv_handle h, const char *key, const float *host, const int64_t *shape, int32_t ndim) {
    if (!h || !key || !host || ndim < 0 || ndim > 4) return h ? h->fail(HMV_ERR_ARG, "bad hmv_set_tensor argument") : HMV_ERR_ARG;
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(host, host + n);
    h->host[key] = std::move(t);
    h->finalized = false;
    return HMV_OK;
}

int hmv_finalize_weights(hmv_handle h) {
    if (!h) return HMV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    // re-finalisation: forwards in flight and captured graphs still point at the old weight buffers
    HIPCHK(h, hipDeviceSynchronize());
    h->drop_graphs();
    h->finalized = false;
    for (void *p : h->dev_allocs) (void)hipFree(p);
    h->dev_allocs.clear();
    for (auto &v : h->blocks) v.clear();
    h->attn.clear();
    Loader L{h};
    const hmv_config &c = h->cfg;
    const int exp = h->paper ? 4 : 1;
    const bool h16 = c.dtype != HMV_F32;   // conv stack on the fp16 kernels; heat-map logits, tokens, fusion, decoder stay fp32
    L.split = c.dtype == HMV_F32X3;        // ... with (hi, lo) pairs: fp32-equivalent

    if (h->hrnet) {
        // ---- HighResolutionNet: hrnet.py:231-311.  Every tensor keeps its real channel stride (dense-K conv mode for 40 / 80 channels).
        HrNet &hr = h->hr;
        hr = HrNet();
        for (int i = 0; i < 4; ++i) hr.ch[i] = kHrChannels[c.backbone - HMV_HRNET_W40][i];
        // the highest-resolution branch (H/4 x W/4) of w40 has 40 channels: its stride-1 3x3 convs run row-decomposed
        // when a 128-row tile covers whole image rows (conv_igemm.hip, RD)
        // (only where it removes padding: channel counts that are not multiples of 32, i.e. w40's 40- and 80-wide branches)
        bool rdb[4] = {false, false, false, false};
        for (int b = 0; b < 2; ++b)
            rdb[b] = 3 * hr.ch[b] <= 256 && hr.ch[b] % 32 != 0 && hr.ch[b] % 4 == 0 && 128 % (c.width / (4 << b)) == 0 && !L.split &&
                     !HMV_DEV_ENV("HMV_NO_ROWSUM");
        // fp16 path, 40-channel branch on frames whose H/4 x W/4 map tiles into 16 x 16 blocks: plain (r, s, c) packing instead, so
        // that the weight-stationary halo-streaming kernel (conv_hs.hip: 1.9 -> ~4 TB/s on these layers) takes the large batches
        // and conv_igemm's dense mode -- same K order, same bits -- the small ones.  Decided by the configuration, never by the batch.
        if (h16 && !L.split && hr.ch[0] == 40 && c.height % 64 == 0 && c.width % 64 == 0 && !HMV_DEV_ENV("HMV_NO_HS")) rdb[0] = false;
        // ... and the 80-channel branch (H/8 x W/8 maps in 8 x 16 blocks; conv_hs.hip's three-wave variant)
        if (h16 && !L.split && hr.ch[1] == 80 && c.height % 64 == 0 && c.width % 128 == 0 && !HMV_DEV_ENV("HMV_NO_HS") && !HMV_DEV_ENV("HMV_NO_HS80")) rdb[1] = false;
        const bool rd0 = rdb[0];
        L.conv(hr.conv1, "stem.conv1", "backbone.conv1.weight", "", "backbone.bn1", 64, 3, 3, 3, /*cin_pad=*/h16 ? 8 : 4, h16);
        L.conv(hr.conv2, "stem.conv2", "backbone.conv2.weight", "", "backbone.bn2", 64, 64, 3, 3, 0, h16);
        int inpl_ = 64;
        for (int bi = 0; bi < 4; ++bi) {
            Block b;
            const std::string p = "backbone.layer1." + std::to_string(bi), lab = "layer1." + std::to_string(bi);
            L.conv(b.c1, lab + ".conv1", p + ".conv1.weight", "", p + ".bn1", 64, inpl_, 1, 1, 0, h16);
            L.conv(b.c2, lab + ".conv2", p + ".conv2.weight", "", p + ".bn2", 64, 64, 3, 3, 0, h16);
            L.conv(b.c3, lab + ".conv3", p + ".conv3.weight", "", p + ".bn3", 256, 64, 1, 1, 0, h16);
            b.has_ds = bi == 0;
            if (b.has_ds) L.conv(b.ds, lab + ".downsample", p + ".downsample.0.weight", "", p + ".downsample.1", 256, 64, 1, 1, 0, h16);
            inpl_ = 256;
            hr.layer1.push_back(b);
        }
        static const int NMOD[3] = {1, 4, 3};
        int npre = 1, prec[4] = {256, 0, 0, 0};
        for (int st = 0; st < 3; ++st) {
            const int nbr = st + 2;
            const std::string tp = "backbone.transition" + std::to_string(st + 1);
            for (int i = 0; i < nbr; ++i) {   // _make_transition_layer: hrnet.py:287-311
                hr.trans[st][i].clear();
                if (i < npre) {
                    if (hr.ch[i] != prec[i]) {
                        Layer l;
                        const std::string q = tp + "." + std::to_string(i);
                        L.conv(l, "transition" + std::to_string(st + 1) + "." + std::to_string(i), q + ".0.weight", "", q + ".1",
                               hr.ch[i], prec[i], 3, 3, cpad(prec[i]), h16, i == 0 && rd0);
                        hr.trans[st][i].push_back(l);
                    }
                } else {
                    int cin = prec[npre - 1];
                    for (int j = 0; j < i + 1 - npre; ++j) {
                        const int outc = (j == i - npre) ? hr.ch[i] : cin;
                        Layer l;
                        const std::string q = tp + "." + std::to_string(i) + "." + std::to_string(j);
                        L.conv(l, "transition" + std::to_string(st + 1) + "." + std::to_string(i) + "." + std::to_string(j),
                               q + ".0.weight", "", q + ".1", outc, cin, 3, 3, cpad(cin), h16);
                        hr.trans[st][i].push_back(l);
                        cin = outc;
                    }
                }
            }
            hr.stage[st].clear();
            for (int m = 0; m < NMOD[st]; ++m) {
                hr.stage[st].emplace_back();
                HrModule &M = hr.stage[st].back();
                const std::string mp = "backbone.stage" + std::to_string(st + 2) + "." + std::to_string(m);
                const std::string ml = "stage" + std::to_string(st + 2) + "." + std::to_string(m);
                for (int b = 0; b < nbr; ++b)
                    for (int blk = 0; blk < 4; ++blk) {
                        const std::string bp = mp + ".branches." + std::to_string(b) + "." + std::to_string(blk);
                        const std::string bl = ml + ".b" + std::to_string(b) + "." + std::to_string(blk);
                        L.conv(M.br[b][blk][0], bl + ".conv1", bp + ".conv1.weight", "", bp + ".bn1", hr.ch[b], hr.ch[b], 3, 3, cpad(hr.ch[b]), h16, rdb[b]);
                        L.conv(M.br[b][blk][1], bl + ".conv2", bp + ".conv2.weight", "", bp + ".bn2", hr.ch[b], hr.ch[b], 3, 3, cpad(hr.ch[b]), h16, rdb[b]);
                    }
                for (int i = 0; i < nbr; ++i)
                    for (int j = 0; j < nbr; ++j) {
                        const std::string fp = mp + ".fuse_layers." + std::to_string(i) + "." + std::to_string(j);
                        const std::string fl = ml + ".fuse" + std::to_string(i) + std::to_string(j);
                        if (j > i) {
                            Layer l;
                            L.conv(l, fl, fp + ".0.weight", "", fp + ".1", hr.ch[i], hr.ch[j], 1, 1, cpad(hr.ch[j]), h16);
                            M.fuse[i][j].push_back(l);
                            if (h16 && !L.split && nbr - 1 - i >= 2)   // (only branches with two or more up-sampling terms run fused)
                                L.conv(M.fup[i][j], fl, fp + ".0.weight", "", fp + ".1", hr.ch[i], hr.ch[j], 1, 1, cpad(hr.ch[j]), false);
                        } else if (j < i) {
                            for (int q = 0; q < i - j; ++q) {
                                const int outc = (q == i - j - 1) ? hr.ch[i] : hr.ch[j];
                                Layer l;
                                const std::string fq = fp + "." + std::to_string(q);
                                L.conv(l, fl + "." + std::to_string(q), fq + ".0.weight", "", fq + ".1", outc, hr.ch[j], 3, 3, cpad(hr.ch[j]), h16);
                                M.fuse[i][j].push_back(l);
                            }
                        }
                    }
            }
            npre = nbr;
            for (int i = 0; i < nbr; ++i) prec[i] = hr.ch[i];
        }
        // pose_net = nn.Conv2d(C0, 21, 3, stride 2, padding 1): handmvnet.py:51-57
        L.conv(h->pose0, "pose_net", "pose_net.weight", "pose_net.bias", "", NJ, c.channels[0], 3, 3, cpad(c.channels[0]), h16);
    } else {
    // ---- backbone: resnet.py:162-177, 189-203
    {   // conv1 7x7 s2 p3 as the 4x4 s1 p2 conv over the 2x2 space-to-depth image (misc_kernels.hip, nchw_to_s2d_kernel):
        // kernel row r reads input row 2y - 3 + r = row pair y - 2 + (r + 1) / 2, sub-row (r + 1) % 2; s2d channel (dy, dx, c)
        const HostTensor *w7 = L.get("backbone.conv1.weight", {64, 3, 7, 7});
        std::vector<float> ws((size_t)64 * 12 * 16, 0.f);
        if (w7)
            for (int o = 0; o < 64; ++o)
                for (int c = 0; c < 3; ++c)
                    for (int r = 0; r < 7; ++r)
                        for (int q = 0; q < 7; ++q) {
                            const int ty = (r + 1) >> 1, dy = (r + 1) & 1, tx = (q + 1) >> 1, dx = (q + 1) & 1;
                            ws[(((size_t)o * 12 + (dy * 2 + dx) * 3 + c) * 4 + ty) * 4 + tx] = w7->data[(((size_t)o * 3 + c) * 7 + r) * 7 + q];
                        }
        L.conv(h->stem, "stem", "", "", "backbone.bn1", 64, 12, 4, 4, /*cin_pad=*/h16 ? 16 : 12, h16, false, ws.data());
        h->stem.Kreal = 7 * 7 * 3;   // FLOP / byte accounting sees the real convolution
        h->stem.in_real = 12;        // real input elements per (s2d) input pixel
    }
    int inpl = 64;
    for (int li = 0; li < 3; ++li) {
        const int planes = 64 << li;
        int stride = li == 0 ? 1 : 2;
        if (h->paper && li == 2) stride = 1;
        for (int bi = 0; bi < kBlocks[c.backbone][li]; ++bi) {
            Block b;
            b.stride = bi == 0 ? stride : 1;
            const std::string p = "backbone.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            const std::string lab = "layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            const int outc = planes * exp;
            if (h->paper) {
                L.conv(b.c1, lab + ".conv1", p + ".conv1.weight", "", p + ".bn1", planes, inpl, 1, 1, 0, h16);
                // conv2 at stride 1 on a map that tiles into 16 x 32 blocks: the tall-tile kernel (conv_ht.hip), chosen by the
                // configuration alone -- the map is H/8 x W/8 from layer2.1 on (layer3 keeps it: the paper variant's stride 1)
                auto hup = [](int n) { return (n - 1) / 2 + 1; };
                const int mh = hup(hup(hup(c.height))), mw = hup(hup(hup(c.width)));
                const bool tall = h16 && li >= 1 && b.stride == 1 && conv_ht_shape_ok(3, 3, 1, 1, planes, planes, mh, mw);
                L.conv(b.c2, lab + ".conv2", p + ".conv2.weight", "", p + ".bn2", planes, planes, 3, 3, 0, h16, false, nullptr, tall);
                L.conv(b.c3, lab + ".conv3", p + ".conv3.weight", "", p + ".bn3", outc, planes, 1, 1, 0, h16);
            } else {
                L.conv(b.c1, lab + ".conv1", p + ".conv1.weight", "", p + ".bn1", planes, inpl, 3, 3, 0, h16);
                L.conv(b.c2, lab + ".conv2", p + ".conv2.weight", "", p + ".bn2", planes, planes, 3, 3, 0, h16);
            }
            b.has_ds = (b.stride != 1 || inpl != outc);
            if (b.has_ds) L.conv(b.ds, lab + ".downsample", p + ".downsample.0.weight", "", p + ".downsample.1", outc, inpl, 1, 1, 0, h16);
            static const bool no_fuse = HMV_DEV_ENV("HMV_NO_DSFUSE") != nullptr;   // development knob (A/B runs)
            if (b.has_ds && h->paper && !L.split && !no_fuse) {
                const int CHK = h16 ? 64 : 32;
                const HostTensor *w3 = L.get(p + ".conv3.weight", {outc, planes, 1, 1}), *wd = L.get(p + ".downsample.0.weight", {outc, inpl, 1, 1});
                std::vector<double> s3, b3, sd_, bd_;
                if (w3 && wd && planes % CHK == 0 && inpl % CHK == 0 && L.bn_fold(p + ".bn3", outc, s3, b3) &&
                    L.bn_fold(p + ".downsample.1", outc, sd_, bd_)) {
                    const float *a3 = w3->data.data(), *ad = wd->data.data();
                    std::vector<double> shift(outc);
                    for (int o = 0; o < outc; ++o) shift[o] = b3[o] + bd_[o];
                    auto wt = [=, &s3, &sd_](int o, int k) -> double {
                        return k < planes ? (double)a3[(size_t)o * planes + k] * s3[o] : (double)ad[(size_t)o * inpl + (k - planes)] * sd_[o];
                    };
                    L.finish(b.c3ds, lab + ".conv3+downsample", planes + inpl, outc, 1, 1, planes + inpl, wt, nullptr, &shift, nullptr, h16);
                    b.c3ds.Kreal = planes + inpl;
                    b.fused_ds = true;
                }
            }
            inpl = outc;
            h->blocks[li].push_back(b);
        }
    }
    // ---- pose_net: handmvnet.py:70-86
    const int c0 = c.channels[0];
    if (h->paper) {
        L.conv(h->pose0, "pose_net.0", "pose_net.0.weight", "pose_net.0.bias", "pose_net.1", 512, c0, 1, 1, 0, h16);
        L.conv(h->pose1, "pose_net.3", "pose_net.3.weight", "pose_net.3.bias", "", NJ, 512, 1, 1, 0, h16);
    } else {
        // ConvTranspose2d(k4,s2,p1) weight [Cin][Cout][4][4] as 4 sub-pixel 2x2 convs:
        // out[2q+a] takes ky = {3,1} (a=0, window starts at q-1) or {2,0} (a=1, window starts at q)
        const HostTensor *w = L.get("pose_net.0.weight", {c0, 128, 4, 4});
        const HostTensor *cb = L.get("pose_net.0.bias", {128});
        std::vector<double> sc, sh;
        const bool ok = L.bn_fold("pose_net.1", 128, sc, sh);
        if (w && cb && ok) {
            static const int kmap[2][2] = {{3, 1}, {2, 0}};
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    const float *wd = w->data.data();
                    const int CH = h16 ? 64 : 32;
                    const bool sp = L.split && h16;          // (hi, lo) pairs: the K order walks 3 * c0 virtual channels
                    const int cv = sp ? 3 * c0 : c0;
                    auto wt = [=](int o, int k) -> float {   // K order (chunk, r, s, c % CH)
                        const int chunk = k / (CH * 4), rem = k % (CH * 4), tap = rem / CH;
                        const int ci = (chunk * CH + rem % CH) % c0, r = tap / 2, s = tap % 2;
                        return wd[(((size_t)ci * 128 + o) * 4 + kmap[a][r]) * 4 + kmap[b][s]];
                    };
                    std::vector<unsigned char> lo_plane;
                    if (sp) {
                        lo_plane.assign(4 * cv, 0);
                        for (int k = 0; k < 4 * cv; ++k) lo_plane[k] = ((k / (CH * 4)) * CH + (k % (CH * 4)) % CH) / c0 == 2;
                    }
                    Layer &dl = h->deconv[a * 2 + b];
                    L.finish(dl, "pose_net.0.phase" + std::to_string(a * 2 + b), cv, 128, 2, 2, 4 * cv, wt,
                             &sc, &sh, cb->data.data(), h16, sp ? &lo_plane : nullptr);
                    dl.Kreal = 4 * c0;
                    if (sp) dl.plane = c0;
                }
            // the four phases' weight blocks back to back: one launch runs all of them (conv_igemm.hip, p.phases).  Not in the
            // split mode, whose per-layer power-of-two weight scale differs between the phases.
            h->deconv_all = Layer();
            if (!(L.split && h16) && !HMV_DEV_ENV("HMV_NO_PHASEMERGE") && L.rc == HMV_OK) {
                const Layer &d0 = h->deconv[0];
                const size_t elems = (size_t)d0.Cout_pad * d0.Kpad, bytes = elems * (h16 ? 2 : 4);
                void *all = nullptr;
                hipError_t e = hipMalloc(&all, 4 * bytes);
                for (int i = 0; i < 4 && e == hipSuccess; ++i)
                    e = hipMemcpy(static_cast<char *>(all) + i * bytes, h->deconv[i].w, bytes, hipMemcpyDeviceToDevice);
                if (e != hipSuccess) L.rc = h->fail(HMV_ERR_HIP, "weight upload failed: %s", hipGetErrorString(e));
                if (all) h->dev_allocs.push_back(all);
                if (e == hipSuccess) {
                    h->deconv_all = d0;
                    h->deconv_all.label = "pose_net.0";
                    h->deconv_all.w = static_cast<float *>(all);
                    h->deconv_stride = elems;
                }
            }
        }
        L.conv(h->pose1, "pose_net.3", "pose_net.3.weight", "pose_net.3.bias", "pose_net.4", 64, 128, 3, 3, 0, h16);
        L.conv(h->pose2, "pose_net.6", "pose_net.6.weight", "pose_net.6.bias", "", NJ, 64, 3, 3, 0, h16);
    }
    }   // resnet backbones
    // ---- sample nets: nets.py:24-31
    for (int i = 0; i < c.n_levels; ++i) {
        const std::string p = "sample_nets." + std::to_string(i) + ".conv";
        L.conv(h->sample[i], "sample_nets." + std::to_string(i), p + ".0.weight", p + ".0.bias", p + ".1", c.channels[i] / 2,
               c.channels[i], 1, 1, h->hrnet ? cpad(c.channels[i]) : 0, h16);
    }
    // ---- fusion: layers.py:177-200
    const int d = h->d;
    // sinusoidal PE table (plain attribute in the reference, not in the state_dict): layers.py:134-158.  The learnable-query
    // blocks carry their own pos_embed and always add it.
    std::vector<float> pe_host;
    if ((c.pos_enc & HMV_POS_SIN) || h->lq) {
        const int T = c.num_views * NJ;
        pe_host.resize((size_t)T * d);
        for (int p = 0; p < T; ++p)
            for (int cc = 0; cc < d; ++cc) {
                const int k2 = cc & ~1;
                const float div = expf((float)k2 * (float)(-std::log(10000.0) / (double)d));
                const float ang = (float)p * div;
                pe_host[(size_t)p * d + cc] = (cc & 1) ? cosf(ang) : sinf(ang);
            }
        h->pe = L.upload(pe_host);
    } else {
        h->pe = nullptr;
    }
    // fp16 / f32x3 modes: the q/k/v projections (the fusion stage's largest GEMMs) run on the fused split fp16 kernels --
    // fp32-equivalent results at a third of the fp32 MFMA time.  Chosen by dtype alone, never by the batch.
    const bool x3lin = h16 && !HMV_DEV_ENV("HMV_NO_X3LIN");
    if (h->lq) {
        // CrossAttentionFusionLearnableQuery: fusion.py:33-49; MultiHeadAttentionLearnableQuery: layers.py:240-301
        for (int l = 0; l < 5; ++l) {
            AttnLayer a;
            const std::string p = "joints_late_fusion.attn_fusion." + std::to_string(l);
            const std::string lab = "fusion_lq." + std::to_string(l);
            const bool cross = l == 2;
            const HostTensor *wq = L.get(p + ".to_q.weight", {INNER_LQ, d}), *wk = L.get(p + ".to_k.weight", {INNER_LQ, d}),
                             *wv = L.get(p + ".to_v.weight", {INNER_LQ, d});
            if (wq && wk && wv) {
                const float *q = wq->data.data(), *k = wk->data.data(), *v = wv->data.data();
                if (cross) {
                    auto wt = [=](int o, int kk) -> float { return (o < INNER_LQ ? k : v)[(size_t)(o % INNER_LQ) * d + kk]; };
                    if (x3lin) L.linear_x3(a.kv, lab + ".kv", h->ldt, 2 * INNER_LQ, d, wt);
                    else L.finish(a.kv, lab + ".kv", h->ldt, 2 * INNER_LQ, 1, 1, d, wt, nullptr, nullptr, nullptr);
                    const HostTensor *pr = L.get(p + ".probe", {1, NJ, d});
                    if (pr) {   // q = to_q(probe + pe[:21]) in double, once
                        std::vector<float> qp((size_t)NJ * INNER_LQ);
                        for (int t = 0; t < NJ; ++t)
                            for (int o = 0; o < INNER_LQ; ++o) {
                                double acc = 0.0;
                                for (int kk = 0; kk < d; ++kk)
                                    acc += (double)(pr->data[(size_t)t * d + kk] + pe_host[(size_t)t * d + kk]) * (double)q[(size_t)o * d + kk];
                                qp[(size_t)t * INNER_LQ + o] = (float)acc;
                            }
                        a.qprobe = L.upload(qp);
                    }
                } else {
                    auto wt = [=](int o, int kk) -> float {
                        const float *src = o < INNER_LQ ? q : (o < 2 * INNER_LQ ? k : v);
                        return src[(size_t)(o % INNER_LQ) * d + kk];
                    };
                    if (x3lin) L.linear_x3(a.qkv, lab + ".qkv", h->ldt, 3 * INNER_LQ, d, wt);
                    else L.finish(a.qkv, lab + ".qkv", h->ldt, 3 * INNER_LQ, 1, 1, d, wt, nullptr, nullptr, nullptr);
                }
            }
            L.linear(a.out, lab + ".to_out", p + ".to_out.0.weight", p + ".to_out.0.bias", d, INNER_LQ);
            if (x3lin) {   // fp16-kernel modes: to_out once more as a split-pair GEMM (gemm_x3.hip), as in CrossAttentionFusion below
                const HostTensor *wo = L.get(p + ".to_out.0.weight", {d, INNER_LQ});
                if (wo) {
                    const float *o_ = wo->data.data();
                    L.linear_x3(a.out_x3, lab + ".to_out", INNER_LQ, d, INNER_LQ, [=](int o, int kk) -> float { return o_[(size_t)o * INNER_LQ + kk]; });
                }
            }
            L.linear(a.ff1, lab + ".ff1", p + ".ff.net.1.weight", p + ".ff.net.1.bias", DHEAD_LQ, d);
            L.linear(a.ff2, lab + ".ff2", p + ".ff.net.4.weight", p + ".ff.net.4.bias", d, DHEAD_LQ);
            a.fg = L.vec(p + ".ff.net.0.weight", d); a.fb = L.vec(p + ".ff.net.0.bias", d);
            h->attn.push_back(a);
        }
    }
    for (int l = 0; l < (h->lq ? 0 : c.fusion_layers); ++l) {
        AttnLayer a;
        const std::string p = "joints_late_fusion.attn_fusion." + std::to_string(l);
        const std::string lab = "fusion." + std::to_string(l);
        const HostTensor *wq = L.get(p + ".to_q.weight", {INNER, d}), *wk = L.get(p + ".to_k.weight", {INNER, d}),
                         *wv = L.get(p + ".to_v.weight", {INNER, d});
        if (wq && wk && wv) {
            const float *q = wq->data.data(), *k = wk->data.data(), *v = wv->data.data();
            auto wt = [=](int o, int kk) -> float {
                const float *src = o < INNER ? q : (o < 2 * INNER ? k : v);
                return src[(size_t)(o % INNER) * d + kk];
            };
            if (x3lin) L.linear_x3(a.qkv, lab + ".qkv", h->ldt, 3 * INNER, d, wt);
            else L.finish(a.qkv, lab + ".qkv", h->ldt, 3 * INNER, 1, 1, d, wt, nullptr, nullptr, nullptr);
        }
        L.linear(a.out, lab + ".to_out", p + ".to_out.weight", p + ".to_out.bias", d, INNER);
        if (x3lin) {
            const HostTensor *wo = L.get(p + ".to_out.weight", {d, INNER});
            if (wo) {
                const float *o_ = wo->data.data();
                L.linear_x3(a.out_x3, lab + ".to_out", INNER, d, INNER, [=](int o, int kk) -> float { return o_[(size_t)o * INNER + kk]; });
            }
        }
        L.linear(a.ff1, lab + ".ff1", p + ".ff.net.1.weight", p + ".ff.net.1.bias", DHEAD, d);
        L.linear(a.ff2, lab + ".ff2", p + ".ff.net.4.weight", p + ".ff.net.4.bias", d, DHEAD);
        a.n1g = L.vec(p + ".norm1.weight", d); a.n1b = L.vec(p + ".norm1.bias", d);
        a.n2g = L.vec(p + ".norm2.weight", d); a.n2b = L.vec(p + ".norm2.bias", d);
        a.fg = L.vec(p + ".ff.net.0.weight", d); a.fb = L.vec(p + ".ff.net.0.bias", d);
        h->attn.push_back(a);
    }
    // ---- decoder: nets.py:119-154
    if (c.decoder == HMV_DECODER_GCN) {
        const int dims[4] = {d, 256, 64, 3};
        for (int i = 0; i < 3; ++i) {
            const std::string p = "joints_decoder.joints_gcn" + std::to_string(i + 1);
            const int ci = dims[i], co = dims[i + 1];
            const HostTensor *w = L.get(p + ".weight", {3, 1, ci, co});
            const HostTensor *b = L.get(p + ".bias", {1, 1, co});
            if (w && b) {
                const float *wd = w->data.data();
                auto wt = [=](int o, int k) -> float { return wd[((size_t)(o / co) * ci + k) * co + (o % co)]; };
                L.finish(h->gcn[i], "decoder.gcn" + std::to_string(i + 1), round_up(ci, 32), 3 * co, 1, 1, ci, wt, nullptr,
                         nullptr, nullptr);
                h->gcn_bias[i] = L.upload(b->data);
            }
        }
        // Chebyshev polynomials of the fixed hand graph: utils.py:108-120, layers.py:405-445, constants.py:37-41
        static const int E[20][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {0, 5}, {5, 6}, {6, 7}, {7, 8}, {0, 9}, {9, 10},
                                     {10, 11}, {11, 12}, {0, 13}, {13, 14}, {14, 15}, {15, 16}, {0, 17}, {17, 18},
                                     {18, 19}, {19, 20}};
        float adj[NJ][NJ] = {}, Lp[NJ][NJ];
        for (auto &e : E) { adj[e[0]][e[1]] = 1.f; adj[e[1]][e[0]] = 1.f; }
        for (int i = 0; i < NJ; ++i) adj[i][i] += 1.f;
        for (int i = 0; i < NJ; ++i) {
            float rs = 0.f;
            for (int j = 0; j < NJ; ++j) rs += adj[i][j];
            const float inv = rs != 0.f ? 1.f / rs : 0.f;
            for (int j = 0; j < NJ; ++j) adj[i][j] *= inv;
        }
        float dsq[NJ];
        for (int i = 0; i < NJ; ++i) {
            float rs = 0.f;
            for (int j = 0; j < NJ; ++j) rs += adj[i][j];
            dsq[i] = 1.f / std::sqrt(rs);
        }
        std::vector<float> tk(3 * NJ * NJ);
        for (int i = 0; i < NJ; ++i)
            for (int j = 0; j < NJ; ++j) {
                Lp[i][j] = (i == j ? 1.f : 0.f) - dsq[i] * adj[i][j] * dsq[j];
                tk[(0 * NJ + i) * NJ + j] = i == j ? 1.f : 0.f;
                tk[(1 * NJ + i) * NJ + j] = Lp[i][j];
            }
        for (int i = 0; i < NJ; ++i)
            for (int j = 0; j < NJ; ++j) {
                float s = 0.f;
                for (int m = 0; m < NJ; ++m) s += Lp[i][m] * Lp[m][j];
                tk[(2 * NJ + i) * NJ + j] = 2.f * s - (i == j ? 1.f : 0.f);
            }
        h->cheb_t = L.upload(tk);
    } else {
        L.linear(h->fc1, "decoder.fc1", "joints_decoder.joints_fc1.weight", "joints_decoder.joints_fc1.bias", 64, d);
        L.linear(h->fc2, "decoder.fc2", "joints_decoder.joints_fc2.weight", "joints_decoder.joints_fc2.bias", 3, 64);
    }
    h->zero_bias = L.upload(std::vector<float>(4096, 0.f));
    if (L.rc != HMV_OK) return L.rc;
    HIPCHK(h, hipDeviceSynchronize());
    h->finalized = true;
    h->host.clear();  // host copies are no longer needed
    return HMV_OK;
}


// Diagnostic: time `iters` launches of one conv shape with a chosen tile (tile < 0: engine's choice).
// HMV_BENCH_DTYPE=f16 in the environment runs the shape on the fp16 kernels (fp16 activations / weights / residual / output).
int hmv_bench_conv(int32_t device, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t R, int32_t S,
                   int32_t stride, int32_t pad, int32_t with_residual, int32_t tile, int32_t iters, float *avg_ms) {
    if (hipSetDevice(device) != hipSuccess) return HMV_ERR_HIP;
    const char *edt = HMV_DEV_ENV("HMV_BENCH_DTYPE");
    const bool f16 = edt && std::string(edt) == "f16";
    const size_t eb = f16 ? 2 : 4;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    const int K = R * S * Cin, Kpad = round_up(K, f16 ? 64 : 32), Cp = round_up(Cout, 256);
    const char *ea = HMV_DEV_ENV("HMV_BENCH_APAD"), *ew = HMV_DEV_ENV("HMV_BENCH_WPAD");
    const int lda = Cin + (ea ? atoi(ea) : 0), ldw = Kpad + (ew ? atoi(ew) : 0);
    const size_t nin = (size_t)N * H * W * lda, nout = (size_t)N * Ho * Wo * Cout, nw = (size_t)Cp * ldw;
    float *din = nullptr, *dout = nullptr, *dw = nullptr, *db = nullptr, *dres = nullptr;
    const char *esk = HMV_DEV_ENV("HMV_BENCH_SKEW");   // bytes by which the output buffer is displaced inside its allocation
    const size_t skew = esk ? (size_t)atol(esk) : 0;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&din), nin * eb);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dout), nout * eb + skew);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dw), nw * eb);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&db), (size_t)Cp * 4);
    if (e == hipSuccess && with_residual) e = hipMalloc(reinterpret_cast<void **>(&dres), nout * eb);
    if (e == hipSuccess) {
        // pseudo-random (not zero: zero operands raise the clock and flatter the number)
        const size_t nmax = std::max(std::max(nin, nw), with_residual ? nout : (size_t)1);
        std::vector<float> hbuf(f16 ? 0 : nmax);
        std::vector<_Float16> hh(f16 ? nmax : 0);
        uint32_t st = 12345u;
        auto fill = [&](float *d, size_t n) {
            for (size_t i = 0; i < n; ++i) {
                st = st * 1664525u + 1013904223u;
                const float v = ((st >> 8) * (1.0f / 8388608.0f)) - 1.0f;
                if (f16) hh[i] = (_Float16)v; else hbuf[i] = v;
            }
            return hipMemcpy(d, f16 ? (const void *)hh.data() : (const void *)hbuf.data(), n * eb, hipMemcpyHostToDevice);
        };
        e = fill(din, nin);
        if (e == hipSuccess) e = fill(dw, nw);
        if (e == hipSuccess && with_residual) e = fill(dres, nout);
        if (e == hipSuccess) e = hipMemset(db, 0, (size_t)Cp * 4);
    }
    if (e == hipSuccess) {
        ConvParams p{};
        p.in = din; p.wgt = dw; p.bias = db; p.res = dres; p.out = reinterpret_cast<char *>(dout) + skew;
        p.in_f16 = f16; p.out_f16 = f16; p.res_f16 = f16 && dres;
        p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
        p.R = R; p.S = S; p.stride = stride; p.pad_h = pad; p.pad_w = pad; p.K = K; p.Kpad = Kpad;
        p.M = N * Ho * Wo; p.ldc = Cout; p.ldr = Cout; p.act = ACT_RELU; p.osy = p.osx = 1;
        p.lda = lda; p.ldw = ldw;
        p.tall = HMV_DEV_ENV("HMV_BENCH_TALL") != nullptr;   // the tall-tile 3x3 kernel (conv_ht.hip); random weights have no order
        unsigned long long *ddbg = nullptr;
        const int nblk_dbg = ((p.M + 63) / 64) * ((Cout + 31) / 32);
        if (HMV_DEV_ENV("HMV_BENCH_CLOCK")) {
            (void)hipMalloc(reinterpret_cast<void **>(&ddbg), (size_t)nblk_dbg * 64);
            (void)hipMemset(ddbg, 0, (size_t)nblk_dbg * 64);
            p.dbg = ddbg;
        }
        const ConvTile t = tile < 0 ? conv_pick_tile(p.M, Cout, K, f16, dres != nullptr) : (ConvTile)tile;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 2 && e == hipSuccess; ++i) e = launch_conv(p, t, nullptr);
        if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters && e == hipSuccess; ++i) e = launch_conv(p, t, nullptr);
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (avg_ms) *avg_ms = ms / (float)iters;
        if (ddbg) {
            std::vector<unsigned long long> hd((size_t)nblk_dbg * 8);
            (void)hipMemcpy(hd.data(), ddbg, hd.size() * 8, hipMemcpyDeviceToHost);
            if (const char *dump = HMV_DEV_ENV("HMV_BENCH_DUMP")) {
                if (FILE *f = fopen(dump, "wb")) { fwrite(hd.data(), 8, hd.size(), f); fclose(f); }
            }
            if (HMV_DEV_ENV("HMV_BENCH_PHASES"))   // conv_hs.hip: {wait, main loop, epilogue} cycles of workgroup 0's first wave, blocks, total
                fprintf(stderr, "[phases] wait %llu main %llu epilogue %llu blocks %llu total %llu\n", hd[0], hd[1], hd[2], hd[3], hd[4]);
            std::vector<double> clk, cyc;
            for (int i = 0; i < nblk_dbg; ++i)
                if (hd[8 * i + 1] > 0) { clk.push_back((double)hd[8 * i] / (double)hd[8 * i + 1] * 0.1); cyc.push_back((double)hd[8 * i]); }
            if (!clk.empty()) {
                std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
                fprintf(stderr, "[clock] blocks=%zu median %.3f GHz (p10 %.3f p90 %.3f), median main-loop cycles %.0f\n", clk.size(),
                        clk[clk.size() / 2], clk[clk.size() / 10], clk[clk.size() * 9 / 10], cyc[cyc.size() / 2]);
            }
            (void)hipFree(ddbg);
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    for (float *ptr : {din, dout, dw, db, dres}) if (ptr) (void)hipFree(ptr);
    if (e != hipSuccess) { g_create_err = std::string("hmv_bench_conv: ") + hipGetErrorString(e); return HMV_ERR_HIP; }
    return HMV_OK;
}

int hmv_op_conv2d_ex(int32_t device, int32_t dtype, const float *in, int32_t N, int32_t H, int32_t W, int32_t Cin,
                     const float *w_oihw, const float *bias_host, int32_t Cout, int32_t R, int32_t S, int32_t stride, int32_t pad,
                     const float *residual, int32_t relu, float *out, void *stream);
}  // extern "C"

// ====================================================================== forward
namespace {

struct Runner {
    hmv_engine *h;
    hipStream_t s;
    bool dry;
    int rc = HMV_OK;
    Arena &A;

    float *alloc(size_t n) {
        float *ptr = A.alloc(n);
        // a real run must stay inside the reservation its planning run sized: stop launching before anything touches memory past it
        if (!dry && h->arena_bytes && A.high > h->arena_bytes && rc == HMV_OK) rc = h->fail(HMV_ERR_STATE, "workspace plan exceeded its reservation");
        return ptr;
    }
    // Two streams (HRNet: a module's last branch beside the one above it): nothing freed inside the region may be handed out again before
    // the streams have joined -- its last user may still be running on the other stream -- so releases wait in `deferred` until join().
    // Same alloc / release sequence in the planning run.
    bool defer = false;
    std::vector<float *> deferred;
    void release(float *p) {
        if (defer && p) deferred.push_back(p);
        else A.release(p);
    }
    void fork() {   // the second stream starts behind everything enqueued so far
        defer = true;
        if (dry || rc != HMV_OK) return;
        check(hipEventRecord(h->ev_fork, s), "hipEventRecord");
        check(hipStreamWaitEvent(h->aux, h->ev_fork, 0), "hipStreamWaitEvent");
    }
    void join() {   // ... and the first one continues behind everything the second has enqueued
        if (!dry && rc == HMV_OK) {
            check(hipEventRecord(h->ev_join, h->aux), "hipEventRecord");
            check(hipStreamWaitEvent(s, h->ev_join, 0), "hipStreamWaitEvent");
        }
        defer = false;
        for (float *p : deferred) A.release(p);
        deferred.clear();
    }

    void check(hipError_t e, const char *what) {
        if (e != hipSuccess && rc == HMV_OK) rc = h->fail(HMV_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    }

    // One conv / GEMM launch.  in: NHWC [N][H][W][L.Cin] ; returns output dims through Ho/Wo.
    int ksplit = 1;   // split-K slices of the next conv() call (gemm() below)
    int phases = 0;   // 4: the next conv() call runs the four transposed-conv phases in one launch (weights phase_stride apart)
    size_t phase_stride = 0;
    // second A source of the next conv() call (conv3 + downsample as one GEMM); consumed by that call
    struct Dual { const float *in2 = nullptr; int ksplit = 0, H2 = 0, W2 = 0, lda2 = 0, stride2 = 1; } dual;
    const char **kernel_name = nullptr;   // op-level entries: receives the kernel family of the last launch
    // chained 1x1 conv of the next conv() call (conv_stream.hip: Bottleneck i's conv3 -> Bottleneck i + 1's conv1 from the output tile
    // while it is in LDS); consumed by that call.  `probe`: the next conv() call only fills *probe with the launch it WOULD make (also
    // in the planning run) -- what chain_ok() asks conv_stream_chain_ok about
    struct Chain { const Layer *L = nullptr; float *out = nullptr; } chain;
    ConvParams *probe = nullptr;
    // MaxPool2d(3, 2, 1) fused into the next conv() call (conv_hs.hip, the fp16 stem): `out` is then the pooled map; consumed by that call
    struct Pool { int h = 0, w = 0; } pool;

    // Bottleneck conv3 + BN3 + downsample conv + BN + ReLU in one launch: out = relu([t2 | x(strided)] . Wcat + b)
    void conv_dual(const Layer &L, const float *t2, int planes, const float *x, int inpl, int N, int Hx, int Wx, int stride, float *out,
                   int ldc, int Ho, int Wo, bool out_f16) {
        dual.in2 = x; dual.ksplit = planes; dual.H2 = Hx; dual.W2 = Wx; dual.lda2 = inpl; dual.stride2 = stride;
        conv(L, t2, N, Ho, Wo, 1, 0, 0, out, ldc, nullptr, 0, ACT_RELU, Ho, Wo, 0, 0, 0, 0, 0, out_f16);
        dual = Dual();
    }

    void conv(const Layer &L, const float *in, int N, int H, int W, int stride, int pad_h, int pad_w, float *out, int ldc,
              const float *res, int ldr, int act, int Ho, int Wo, int rg_out = 0, int rg_in = 0, int scatter = 0, int ooy = 0,
              int oox = 0, bool out_f16 = false, int up = 0, bool fill = false) {
        if ((dry && !probe) || rc != HMV_OK) { probe = nullptr; return; }
        ConvParams p{};
        p.in = in; p.wgt = L.w; p.bias = L.bias; p.res = res; p.out = out;
        if (ksplit > 1) { p.ksl = ksplit; p.kslice = L.Kpad / ksplit; p.out_slice = (size_t)N * Ho * Wo * ldc; }
        if (phases > 1) { p.phases = phases; p.phase_stride = phase_stride; }
        if (dual.in2) {
            p.in2 = dual.in2; p.ksplit = dual.ksplit; p.H2 = dual.H2; p.W2 = dual.W2; p.lda2 = dual.lda2; p.stride2 = dual.stride2;
            p.lda = dual.ksplit;   // the first source's pixel stride is its own channel count, not the concatenated one
        }
        p.in_f16 = L.f16; p.out_f16 = out_f16; p.res_f16 = L.f16 && res != nullptr;   // a residual always has the layer's dtype
        p.N = N; p.H = H; p.W = W; p.Cin = L.Cin;
        p.Ho = Ho; p.Wo = Wo; p.Cout = L.Cout;
        p.R = L.R; p.S = L.S; p.stride = stride; p.pad_h = pad_h; p.pad_w = pad_w;
        p.K = L.K; p.Kpad = L.Kpad;
        p.M = N * Ho * Wo;
        p.ldc = ldc; p.ldr = ldr; p.act = act;
        p.rg_out = rg_out; p.rg_in = rg_in;
        p.scatter = scatter; p.osy = scatter ? 2 : 1; p.osx = scatter ? 2 : 1; p.ooy = ooy; p.oox = oox;
        p.up = up; p.fill = fill ? 1 : 0;
        p.tall = L.tall;
        if (L.rd_cout) { p.rd_cout = L.rd_cout; p.pad_w = 0; }   // L.R x L.S is the 3x1 GEMM, the epilogue sums the s groups
        if (L.plane) {   // HMV_F32X3: [hi | lo] rows in, pairs out (unless this layer writes fp32), pairs as residual
            p.lda = 2 * L.plane;
            if (L.x3n) p.x3_plane = L.plane;
            else p.cwrap = 2 * L.plane;
            p.acc_shift = L.acc_shift;
            if (out_f16) { p.out_split = 1; p.ldc = 2 * ldc; }
            if (res) { p.res_split = 1; p.ldr = 2 * ldr; }
        }
        const bool pooled = pool.h > 0;
        if (pooled) { p.pool = 1; p.pool_h = pool.h; p.pool_w = pool.w; pool = Pool(); }
        if (probe) { *probe = p; probe = nullptr; return; }
        const Layer *Lx = chain.L;
        if (Lx) {
            p.nx_wgt = Lx->w; p.nx_bias = Lx->bias; p.nx_out = chain.out; p.nx_cout = Lx->Cout; p.nx_ldw = Lx->Kpad; p.nx_ldc = Lx->Cout;
            p.nx_act = ACT_RELU;
            chain = Chain();
        }
        // split layers walk 3x the reduction on fp16 MFMAs: the tile rules see the real reduction length
        const ConvTile tile = conv_pick_tile(p.M, p.Cout, L.plane ? p.K / (L.x3n ? 2 : 3) : p.K, L.f16, res != nullptr);
        ProfRec *pr = nullptr;
        if (h->profiling) {
            if (h->prof_used == h->prof.size()) {
                ProfRec r{};
                check(hipEventCreate(&r.e0), "hipEventCreate");
                check(hipEventCreate(&r.e1), "hipEventCreate");
                h->prof.push_back(r);
            }
            pr = &h->prof[h->prof_used++];
            pr->label = Lx ? L.label + "+" + Lx->label : (pooled ? L.label + "+maxpool" : L.label);
            // algorithmic FLOPs: the real (un-padded) reduction length; the stem's 4th channel is padding
            const double kreal = L.Kreal ? (double)L.Kreal : (double)L.K;   // real channels only (no padding FLOPs)
            const double cout_real = L.rd_cout ? (double)L.rd_cout : (double)L.Cout;
            pr->flops = 2.0 * (double)p.M * (double)L.Cout * kreal;
            // algorithmic bytes: each input pixel the window touches once, the weights once, residual and output rows once.
            // A (hi, lo) pair is 4 bytes like fp32.  A strided 1x1 conv reads only the pixels it keeps.
            const double eb_in = L.f16 ? (L.plane ? 4.0 : 2.0) : 4.0;
            const double eb_out = (L.f16 && out_f16) ? (L.plane ? 4.0 : 2.0) : 4.0;
            const double cin_real = L.in_real ? (double)L.in_real : kreal / (double)(L.R * L.S);   // (the row-decomposed form has R x S = 3 x 1 and Kreal = 3 * Cin)
            const bool pointwise = L.R == 1 && L.S == 1;
            const double in_px = (pointwise && stride > 1) ? (double)p.M : (double)N * H * W;
            pr->bytes = in_px * cin_real * eb_in + (double)L.Cout * kreal * eb_in + (double)p.M * cout_real * eb_out +
                        (res ? (double)p.M * cout_real * eb_in : 0.0);
            if (phases > 1) {   // every phase: its own weights and output pixels, the same input
                pr->flops *= phases;
                pr->bytes = in_px * cin_real * eb_in + phases * ((double)L.Cout * kreal * eb_in + (double)p.M * cout_real * eb_out);
            }
            if (dual.in2)   // [t2 | x]: t2 at every output pixel, x at the pixels the stride keeps
                pr->bytes = ((double)p.M * dual.ksplit + (double)p.M * (kreal - dual.ksplit)) * eb_in + (double)L.Cout * kreal * eb_in +
                            (double)p.M * cout_real * eb_out;
            if (pooled)   // the conv map never reaches HBM: input, weights, the pooled map
                pr->bytes = in_px * cin_real * eb_in + (double)L.Cout * kreal * eb_in + (double)N * p.pool_h * p.pool_w * cout_real * eb_out;
            if (Lx) {   // the chained conv: its weights and output rows; its input rows never leave the CU
                pr->flops += 2.0 * (double)p.M * (double)Lx->Cout * (double)Lx->K;
                pr->bytes += (double)Lx->Cout * (double)Lx->K * eb_in + (double)p.M * (double)Lx->Cout * eb_out;
            }
            check(hipEventRecord(pr->e0, s), "hipEventRecord");
        }
        const char *kname = nullptr;
        check(launch_conv(p, tile, s, &kname), L.label.c_str());
        ++h->launches;
        if (kernel_name) *kernel_name = kname;
        if (pr) {
            pr->name = kname;
            check(hipEventRecord(pr->e1, s), "hipEventRecord");
        }
    }

    // the up-sampling terms of an HRNet fuse layer in one launch (hr_fuse.hip)
    void hr_fuse_up(const HrFuseParams &p, const std::string &label) {
        if (dry || rc != HMV_OK) return;
        ProfRec *pr = nullptr;
        if (h->profiling) {
            if (h->prof_used == h->prof.size()) {
                ProfRec r{};
                check(hipEventCreate(&r.e0), "hipEventCreate");
                check(hipEventCreate(&r.e1), "hipEventCreate");
                h->prof.push_back(r);
            }
            pr = &h->prof[h->prof_used++];
            pr->label = label;
            const double eb = p.f16 ? 2.0 : 4.0;
            pr->flops = 0.0;
            pr->bytes = 2.0 * (double)p.N * p.H * p.W * p.C * eb;   // the map once in, once out
            for (int q = 0; q < p.nsrc; ++q) {   // the products at the sources' own resolution
                const HrFuseSrc &S = p.src[q];
                pr->flops += 2.0 * (double)p.N * S.H * S.W * S.C * p.C;
                pr->bytes += (double)p.N * S.H * S.W * S.C * eb + (double)S.C * p.C * 4.0;
            }
            check(hipEventRecord(pr->e0, s), "hipEventRecord");
        }
        check(launch_hr_fuse_up(p, s), label.c_str());
        ++h->launches;
        if (pr) {
            pr->name = p.f16 ? "hr_fuse_up_f16" : "hr_fuse_up_f32";
            check(hipEventRecord(pr->e1, s), "hipEventRecord");
        }
    }

    static int splitk_slices(const Layer &L, int rows) {
        static const bool no_splitk = HMV_DEV_ENV("HMV_NO_SPLITK") != nullptr;   // development knob (A/B runs)
        return (!no_splitk && !L.f16 && L.R == 1 && L.S == 1 && !L.plane && L.Kpad >= 1024 && L.Kpad % 128 == 0 &&
                L.Cout <= 4096 /* zero_bias */ && rows > 0) ? 4 : 1;
    }

    void gemm(const Layer &L, const float *a, int rows, float *out, int ldc, const float *res, int ldr, int act, int rg_out = 0,
              int rg_in = 0) {
        // split-K (layers.py:224 to_out: K = 1024, and 2048 in the learnable-query blocks): a long reduction over few token rows
        // tiles into a few dozen workgroups at a small batch.  K is cut into 4 slices that run as 4x the workgroups of ONE
        // launch; a reduction kernel adds the partial products in slice order and applies bias / residual / activation.
        // The cut depends on K alone -- never on the batch -- so a sample's result does not depend on what it is batched with
        // (tests/test_gpu_parity.py::test_full_size_properties).  Same alloc / release sequence in the dry (planning) run.
        const int S = splitk_slices(L, rows);
        if (S == 1) {
            conv(L, a, rows, 1, 1, 1, 0, 0, out, ldc, res, ldr, act, 1, 1, rg_out, rg_in);
            return;
        }
        const int lds_ = (L.Cout + 3) / 4 * 4;
        float *slab = alloc((size_t)S * rows * lds_);
        Layer Ls = L;
        Ls.bias = h->zero_bias;
        ksplit = S;
        conv(Ls, a, rows, 1, 1, 1, 0, 0, slab, lds_, nullptr, 0, ACT_NONE, 1, 1);
        ksplit = 1;
        if (!dry && rc == HMV_OK) {
            ++h->launches;
            if (ln.y)   // the consumer is a LayerNorm (gemm_ln below): slice sum, bias and residual become its load
                check(launch_splitk_layernorm(slab, S, rows, lds_, L.Cout, L.bias, res, ldr, rg_out, rg_in, ln.g1, ln.b1, ln.y, ln.ldy,
                                              ln.g2, ln.b2, ln.y2, s), "splitk_layernorm");
            else
                check(launch_splitk_reduce(slab, S, rows, lds_, L.Cout, L.bias, res, ldr, rg_out, rg_in, act, out, ldc, s), "splitk_reduce");
        }
        release(slab);
    }

    // Everything of a fusion block behind the attention in two launches: the split-K to_out GEMM, then ONE kernel for
    // (slice sum + bias + residual) -> [LayerNorm1] -> FeedForward (LayerNorm, Linear, GELU, Linear, + residual) -> [LayerNorm2]
    // (fusion_kernels.hip).  Same alloc / release sequence in the dry (planning) run.
    bool ff_fusable(const Layer &out, const Layer &ff1, const Layer &ff2, int rows, int ldt, bool have_x3 = false) const {
        return h->ff_fuse && (have_x3 || splitk_slices(out, rows) > 1) && !ff1.f16 && !ff2.f16 && !ff1.plane && !ff2.plane &&
               ff1.Kpad == ldt && ldt % 16 == 0 && (ff1.Cout == 128 || ff1.Cout == 256) && ff2.Kpad == ff1.Cout && ff2.Cout_pad >= ldt &&
               ldt <= 576;
    }
    void ff_block(const Layer &out, const Layer &ff1, const Layer &ff2, const float *att, int rows, const float *res, int ldr, int rg_out,
                  int rg_in, const float *n1g, const float *n1b, const float *fg, const float *fb, const float *n2g, const float *n2b,
                  float *y, int ldt, int d, float *y_pairs = nullptr, const Layer *out_x3 = nullptr) {
        // out_x3: `att` holds (hi, lo) pairs and to_out runs as ONE split-pair GEMM (gemm_x3.hip) instead of four fp32 split-K slices
        const int S = out_x3 ? 1 : splitk_slices(out, rows), lds_ = (out.Cout + 3) / 4 * 4;
        float *slab = alloc((size_t)S * rows * lds_);
        if (out_x3) {
            conv(*out_x3, att, rows, 1, 1, 1, 0, 0, slab, lds_, nullptr, 0, ACT_NONE, 1, 1);   // (its own bias is zero)
        } else {
            Layer Ls = out;
            Ls.bias = h->zero_bias;
            ksplit = S;
            conv(Ls, att, rows, 1, 1, 1, 0, 0, slab, lds_, nullptr, 0, ACT_NONE, 1, 1);
            ksplit = 1;
        }
        if (!dry && rc == HMV_OK) {
            FfBlockParams p{};
            p.slab = slab; p.S = S; p.slice = (size_t)rows * lds_; p.lds = lds_; p.bias0 = out.bias;
            p.res = res; p.ldr = ldr; p.rg_out = rg_out; p.rg_in = rg_in;
            p.rows = rows; p.d = d; p.ld = ldt;
            p.n1g = n1g; p.n1b = n1b; p.fg = fg; p.fb = fb; p.n2g = n2g; p.n2b = n2b;
            p.w1 = ff1.w; p.b1 = ff1.bias; p.ldw1 = ff1.Kpad; p.w2 = ff2.w; p.b2 = ff2.bias; p.ldw2 = ff2.Kpad;
            p.out = y; p.ldo = ldt; p.hid = ff1.Cout; p.out_pairs = y_pairs;
#ifdef HMV_DEV_KNOBS
            static unsigned long long *ffdbg = nullptr;   // HMV_FF_DBG=1: phase stamps of the last launch, printed by the next one (never under graph capture)
            if (HMV_DEV_ENV("HMV_FF_DBG")) {
                if (!ffdbg) (void)hipMalloc(reinterpret_cast<void **>(&ffdbg), 4096 * 64);
                else {
                    unsigned long long hst[8 * 4];
                    (void)hipStreamSynchronize(s);
                    (void)hipMemcpy(hst, ffdbg, sizeof(hst), hipMemcpyDeviceToHost);
                    for (int w = 0; w < 2; ++w)
                        fprintf(stderr, "[ff_block wg %d] phase0 %.2f us  gemm1 %.2f  gemm2 %.2f  ln2+store %.2f  (entry->exit %.2f us)\n", w * 3,
                                (hst[w * 24 + 1] - hst[w * 24]) * 0.01, (hst[w * 24 + 2] - hst[w * 24 + 1]) * 0.01, (hst[w * 24 + 3] - hst[w * 24 + 2]) * 0.01,
                                (hst[w * 24 + 4] - hst[w * 24 + 3]) * 0.01, (hst[w * 24 + 4] - hst[w * 24]) * 0.01);
                }
                p.dbg = ffdbg;
            }
#endif
            check(launch_ff_block(p, s), "ff_block");
            ++h->launches;
        }
        release(slab);
    }

    // y = LayerNorm(a . W + bias + residual) (+ a chained second LayerNorm y2): layers.py:224-229 (to_out, + _q, norm1, then the
    // FeedForward's own LayerNorm).  With split-K the GEMM's reduction kernel IS the LayerNorm kernel (one launch less per block).
    struct LnTail { const float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr; float *y = nullptr, *y2 = nullptr; int ldy = 0; } ln;
    void gemm_ln(const Layer &L, const float *a, int rows, const float *res, int ldr, int rg_out, int rg_in, const float *g1,
                 const float *b1, float *y, int ldy, const float *g2, const float *b2, float *y2) {
        if (splitk_slices(L, rows) > 1) {
            ln.g1 = g1; ln.b1 = b1; ln.g2 = g2; ln.b2 = b2; ln.y = y; ln.y2 = y2; ln.ldy = ldy;
            gemm(L, a, rows, nullptr, 0, res, ldr, ACT_NONE, rg_out, rg_in);
            ln = LnTail();
            return;
        }
        float *o = alloc((size_t)rows * ldy);
        gemm(L, a, rows, o, ldy, res, ldr, ACT_NONE, rg_out, rg_in);
        if (!dry && rc == HMV_OK) { check(launch_layernorm(o, ldy, rows, L.Cout, g1, b1, y, ldy, g2, b2, y2, s), "layernorm"); ++h->launches; }
        release(o);
    }
};

int run_forward(hmv_engine *h, int B, const float *x, const float *bbox, const float *intr, float *crop_img, float *joints_cam,
                float *heatmap, hipStream_t s, bool dry, Arena &A) {
    Runner R{h, s, dry, HMV_OK, A};
    const hmv_config &c = h->cfg;
    if (!dry && h->hrnet && h->hr_overlap && c.dtype != HMV_F32 && !h->aux) {   // (the first forward of a handle is never a captured one)
        if (hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess)
            return h->fail(HMV_ERR_HIP, "second stream for the HRNet branches");
    }
    const int V = c.num_views, N = B * V, H = c.height, W = c.width;
    const int d = h->d, ldt = h->ldt;
    // fp16 path: the conv stack (stem .. pose_net / sample convs) stores activations as fp16; heat-map logits,
    // coordinates, tokens, fusion and decoder stay fp32.  ACT(n) = arena floats for n activation elements.
    const bool h16 = c.dtype != HMV_F32;      // the conv stack runs on the fp16 kernels ...
    const bool split = c.dtype == HMV_F32X3;  // ... on (hi, lo) pairs: 4 bytes per element like fp32
#define ACT(n) ((h16 && !split) ? ((size_t)(n) + 1) / 2 : (size_t)(n))
#define LAUNCH(expr) do { if (!dry && R.rc == HMV_OK) { R.check((expr), #expr); ++h->launches; } } while (0)

    // sampled feature levels in the reference's feats[] order (handmvnet.py:165-177), channels-last with row stride ld
    float *lvl[4] = {nullptr, nullptr, nullptr, nullptr};
    int lvc[4] = {0, 0, 0, 0}, lvld[4] = {0, 0, 0, 0}, lvh[4] = {0, 0, 0, 0}, lvw[4] = {0, 0, 0, 0}, nkeep = 0;
    int hmh = 0, hmw = 0;
    float *hm = nullptr;
    if (h->hrnet) {
        // ================= HighResolutionNet.forward (hrnet.py:357-393) =================
        const HrNet &hr = h->hr;
        float *in4 = R.alloc((size_t)N * H * W * (split ? 8 : 4));
        if (h->fsrc.frames) LAUNCH(launch_frames_to_input(h->fsrc.frames, h->fsrc.boxes, N, h->fsrc.fh, h->fsrc.fw, H, W, h->fsrc.mean, h->fsrc.std, split ? 2 : (h16 ? 1 : 0), in4, s));
        else if (split) LAUNCH(launch_nchw_to_nhwc_split(x, in4, N, H, W, s));
        else if (h16) LAUNCH(launch_nchw_to_nhwc8_f16(x, in4, N, H, W, s));
        else LAUNCH(launch_nchw_to_nhwc4(x, in4, N, H, W, s));
        const int H1 = (H + 2 - 3) / 2 + 1, W1 = (W + 2 - 3) / 2 + 1, H2 = (H1 + 2 - 3) / 2 + 1, W2 = (W1 + 2 - 3) / 2 + 1;
        float *c1 = R.alloc(ACT((size_t)N * H1 * W1 * 64));
        R.conv(hr.conv1, in4, N, H, W, 2, 1, 1, c1, 64, nullptr, 0, ACT_RELU, H1, W1, 0, 0, 0, 0, 0, h16);
        R.release(in4);
        float *cur = R.alloc(ACT((size_t)N * H2 * W2 * 64));
        R.conv(hr.conv2, c1, N, H1, W1, 2, 1, 1, cur, 64, nullptr, 0, ACT_RELU, H2, W2, 0, 0, 0, 0, 0, h16);
        R.release(c1);
        float *t1_chained = nullptr;
        for (size_t bi = 0; bi < hr.layer1.size(); ++bi) {   // 4 Bottlenecks, planes 64 -> 256 channels
            const Block &b = hr.layer1[bi];
            float *t1 = t1_chained;   // (conv_stream.hip chain, as in the ResNet loop below: the previous block's conv3 launch computed it)
            t1_chained = nullptr;
            if (!t1) {
                t1 = R.alloc(ACT((size_t)N * H2 * W2 * 64));
                R.conv(b.c1, cur, N, H2, W2, 1, 0, 0, t1, 64, nullptr, 0, ACT_RELU, H2, W2, 0, 0, 0, 0, 0, h16);
            }
            float *t2 = R.alloc(ACT((size_t)N * H2 * W2 * 64));
            R.conv(b.c2, t1, N, H2, W2, 1, 1, 1, t2, 64, nullptr, 0, ACT_RELU, H2, W2, 0, 0, 0, 0, 0, h16);
            R.release(t1);
            const float *res = cur;
            float *dsb = nullptr;
            if (b.has_ds) {
                dsb = R.alloc(ACT((size_t)N * H2 * W2 * 256));
                R.conv(b.ds, cur, N, H2, W2, 1, 0, 0, dsb, 256, nullptr, 0, ACT_NONE, H2, W2, 0, 0, 0, 0, 0, h16);
                res = dsb;
            }
            float *y = R.alloc(ACT((size_t)N * H2 * W2 * 256));
            const Block *nb = bi + 1 < hr.layer1.size() ? &hr.layer1[bi + 1] : nullptr;
            bool ch = false;
            if (h->chain_fuse && nb && h16 && !split && nb->c1.f16 && !nb->c1.plane && nb->c1.R == 1 && nb->c1.S == 1 && nb->c1.Cin == 256 &&
                nb->c1.K == 256 && nb->c1.Kpad == 256 && !nb->c1.tall && !nb->c1.rd_cout) {
                ConvParams q{};
                R.probe = &q;
                R.conv(b.c3, t2, N, H2, W2, 1, 0, 0, y, 256, res, 256, ACT_RELU, H2, W2, 0, 0, 0, 0, 0, h16);
                ch = conv_stream_chain_ok(q, nb->c1.Cout);
            }
            if (ch) { t1_chained = R.alloc(ACT((size_t)N * H2 * W2 * 64)); R.chain.L = &nb->c1; R.chain.out = t1_chained; }
            R.conv(b.c3, t2, N, H2, W2, 1, 0, 0, y, 256, res, 256, ACT_RELU, H2, W2, 0, 0, 0, 0, 0, h16);
            R.chain = Runner::Chain();
            R.release(t2);
            R.release(dsb);
            R.release(cur);
            cur = y;
        }
        float *xs[4] = {nullptr, nullptr, nullptr, nullptr}, *pre[4] = {cur, nullptr, nullptr, nullptr};
        int hs[4] = {0, 0, 0, 0}, ws[4] = {0, 0, 0, 0}, prec[4] = {256, 0, 0, 0}, preh[4] = {H2, 0, 0, 0}, prew[4] = {W2, 0, 0, 0};
        int npre = 1;
        for (int st = 0; st < 3; ++st) {
            const int nbr = st + 2;
            bool moved[4] = {false, false, false, false};
            for (int i = 0; i < nbr; ++i) {   // transition layers (hrnet.py:287-311, 368-390)
                const int cp = cpad(hr.ch[i]);
                if (i < npre) {
                    hs[i] = preh[i]; ws[i] = prew[i];
                    if (!hr.trans[st][i].empty()) {
                        xs[i] = R.alloc(ACT((size_t)N * hs[i] * ws[i] * cp));
                        R.conv(hr.trans[st][i][0], pre[i], N, preh[i], prew[i], 1, 1, 1, xs[i], cp, nullptr, 0, ACT_RELU, hs[i], ws[i],
                               0, 0, 0, 0, 0, h16, 0, /*fill=*/true);
                    } else {
                        xs[i] = pre[i];   // x_list.append(y_list[i]): the same tensor
                        moved[i] = true;
                    }
                } else {   // a new, lower-resolution branch from the LAST previous branch
                    const float *src = pre[npre - 1];
                    float *tmp = nullptr;
                    int hh_ = preh[npre - 1], ww_ = prew[npre - 1];
                    for (const Layer &l : hr.trans[st][i]) {
                        const int ho = (hh_ + 2 - 3) / 2 + 1, wo = (ww_ + 2 - 3) / 2 + 1, cpo = cpad(l.Cout);
                        float *o = R.alloc(ACT((size_t)N * ho * wo * cpo));
                        R.conv(l, src, N, hh_, ww_, 2, 1, 1, o, cpo, nullptr, 0, ACT_RELU, ho, wo, 0, 0, 0, 0, 0, h16, 0, true);
                        R.release(tmp);
                        tmp = o; src = o; hh_ = ho; ww_ = wo;
                    }
                    xs[i] = tmp; hs[i] = hh_; ws[i] = ww_;
                }
            }
            for (int i = 0; i < npre; ++i)
                if (!moved[i]) R.release(pre[i]);
            for (const HrModule &M : hr.stage[st]) {   // HighResolutionModule.forward (hrnet.py:194-212)
                // Four branches: the lowest-resolution one (320 / 512 channels on 1/32-size maps: a few hundred small tiles per conv, bound
                // by request latency) runs on the second stream BESIDE the branch above it (one round of 256 x 192 tiles, latency-bound too):
                // their workgroups share the CUs.  The two highest-resolution branches own the chip with persistent kernels and stay alone.
                const bool overlap = h->hr_overlap && h16 && nbr == 4 && (dry || h->aux != nullptr);
                for (int b = 0; b < nbr; ++b) {
                    const int cp = cpad(hr.ch[b]);
                    if (overlap && b == 2) R.fork();
                    const hipStream_t main_s = R.s;
                    if (overlap && b == 3) R.s = h->aux;
                    for (int blk = 0; blk < 4; ++blk) {
                        float *t = R.alloc(ACT((size_t)N * hs[b] * ws[b] * cp));
                        R.conv(M.br[b][blk][0], xs[b], N, hs[b], ws[b], 1, 1, 1, t, cp, nullptr, 0, ACT_RELU, hs[b], ws[b], 0, 0, 0, 0, 0,
                               h16, 0, true);
                        float *y = R.alloc(ACT((size_t)N * hs[b] * ws[b] * cp));
                        R.conv(M.br[b][blk][1], t, N, hs[b], ws[b], 1, 1, 1, y, cp, xs[b], cp, ACT_RELU, hs[b], ws[b], 0, 0, 0, 0, 0, h16,
                               0, true);
                        R.release(t);
                        R.release(xs[b]);
                        xs[b] = y;
                    }
                    R.s = main_s;
                    if (overlap && b == 3) R.join();
                }
                // fuse: y_i = relu(sum_j f_ij(x_j)).  Every non-identity term is a conv at branch i's resolution whose
                // epilogue adds the running sum (the identity term x_i rides along as the first residual); the
                // "1x1 conv + BN + nearest upsample" terms run the 1x1 conv on the up-sampled index map instead.
                float *outs[4] = {nullptr, nullptr, nullptr, nullptr};
                for (int i = 0; i < nbr; ++i) {
                    const int cpi = cpad(hr.ch[i]);
                    float *running = nullptr;
                    int nterm = 0;
                    // two or more up-sampling terms (always the last of the sum): one launch for all of them (hr_fuse.hip)
                    HrFuseParams fp{};
                    bool fused = false;
                    if (h->hr_fuse && !split && nbr - 1 - i >= 2) {
                        fp.N = N; fp.H = hs[i]; fp.W = ws[i]; fp.C = hr.ch[i]; fp.ldc = cpi; fp.relu = 1; fp.f16 = h16 ? 1 : 0;
                        for (int j = i + 1; j < nbr; ++j) {
                            const Layer &l = h16 ? M.fup[i][j] : M.fuse[i][j][0];
                            HrFuseSrc &S = fp.src[fp.nsrc++];
                            S.w = l.w; S.bias = l.bias; S.C = hr.ch[j]; S.ld = cpad(hr.ch[j]); S.ldw = l.Kpad; S.shift = j - i; S.H = hs[j]; S.W = ws[j];
                        }
                        fused = fp.src[0].w != nullptr && hr_fuse_up_plan(fp);
                    }
                    for (int j = 0; j < nbr; ++j) {
                        if (j == i) continue;
                        if (fused && j > i) continue;
                        const bool first = nterm == 0, last = nterm == nbr - 2;
                        const float *res = first ? xs[i] : running;
                        const int act = last ? ACT_RELU : ACT_NONE;
                        float *o = R.alloc(ACT((size_t)N * hs[i] * ws[i] * cpi));
                        if (j > i) {
                            R.conv(M.fuse[i][j][0], xs[j], N, hs[j], ws[j], 1, 0, 0, o, cpi, res, cpi, act, hs[i], ws[i], 0, 0, 0, 0, 0,
                                   h16, /*up=*/j - i, true);
                        } else {
                            const float *src = xs[j];
                            float *tmp = nullptr;
                            int hh_ = hs[j], ww_ = ws[j];
                            const int nq = i - j;
                            for (int q = 0; q < nq; ++q) {
                                const Layer &l = M.fuse[i][j][q];
                                const int ho = (hh_ + 2 - 3) / 2 + 1, wo = (ww_ + 2 - 3) / 2 + 1;
                                if (q == nq - 1) {
                                    R.conv(l, src, N, hh_, ww_, 2, 1, 1, o, cpi, res, cpi, act, ho, wo, 0, 0, 0, 0, 0, h16, 0, true);
                                } else {
                                    const int cpo = cpad(l.Cout);
                                    float *t = R.alloc(ACT((size_t)N * ho * wo * cpo));
                                    R.conv(l, src, N, hh_, ww_, 2, 1, 1, t, cpo, nullptr, 0, ACT_RELU, ho, wo, 0, 0, 0, 0, 0, h16, 0, true);
                                    R.release(tmp);
                                    tmp = t; src = t;
                                }
                                hh_ = ho; ww_ = wo;
                            }
                            R.release(tmp);
                        }
                        if (!first) R.release(running);
                        running = o;
                        ++nterm;
                    }
                    if (fused) {
                        float *o = R.alloc(ACT((size_t)N * hs[i] * ws[i] * cpi));
                        fp.base = running ? running : xs[i];
                        fp.out = o;
                        for (int q = 0; q < fp.nsrc; ++q) fp.src[q].x = xs[i + 1 + q];
                        R.hr_fuse_up(fp, M.fuse[i][i + 1][0].label + "+up");
                        if (running) R.release(running);
                        running = o;
                    }
                    outs[i] = running;
                }
                for (int b = 0; b < nbr; ++b) { R.release(xs[b]); xs[b] = outs[b]; }
            }
            npre = nbr;
            for (int i = 0; i < nbr; ++i) { pre[i] = xs[i]; prec[i] = hr.ch[i]; preh[i] = hs[i]; prew[i] = ws[i]; xs[i] = nullptr; }
        }
        (void)prec;
        nkeep = 4;
        for (int i = 0; i < 4; ++i) { lvl[i] = pre[i]; lvc[i] = hr.ch[i]; lvld[i] = cpad(hr.ch[i]); lvh[i] = preh[i]; lvw[i] = prew[i]; }
        if (h->capture && !dry && h->cap_feat0) {
            if (split) LAUNCH(launch_nhwc_split_to_nchw(lvl[0], h->cap_feat0, N, lvh[0], lvw[0], lvc[0], s));
            else if (h16) LAUNCH(launch_nhwc_f16_to_nchw(lvl[0], h->cap_feat0, N, lvh[0], lvw[0], lvc[0], s));
            else LAUNCH(launch_nhwc_to_nchw(lvl[0], h->cap_feat0, N, lvh[0], lvw[0], lvc[0], s, lvld[0]));
        }
        // pose_net = Conv2d(C0, 21, 3, stride 2, padding 1) on the highest-resolution branch (handmvnet.py:51-57, 180)
        hmh = (lvh[0] + 2 - 3) / 2 + 1; hmw = (lvw[0] + 2 - 3) / 2 + 1;
        hm = R.alloc((size_t)N * hmh * hmw * 32);
        R.conv(h->pose0, lvl[0], N, lvh[0], lvw[0], 2, 1, 1, hm, 32, nullptr, 0, ACT_NONE, hmh, hmw);
    } else {
    // ---- stem: conv1 7x7 s2 + BN + ReLU, maxpool 3x3 s2 (resnet.py:218-221)
    // as a 4x4 stride-1 conv over the 2x2 space-to-depth frames: 12 fp32 / 16 fp16 (12 + 4 zeros) / [hi16 | lo16] per s2d pixel
    const int Hs = (H + 1) / 2, Ws = (W + 1) / 2, smode = split ? 2 : (h16 ? 1 : 0);
    float *in4 = R.alloc((size_t)N * Hs * Ws * (split ? 16 : (h16 ? 8 : 12)));
    if (h->fsrc.frames) LAUNCH(launch_frames_to_input(h->fsrc.frames, h->fsrc.boxes, N, h->fsrc.fh, h->fsrc.fw, H, W, h->fsrc.mean, h->fsrc.std, smode, in4, s, /*s2d=*/true));
    else LAUNCH(launch_nchw_to_s2d(x, in4, N, H, W, smode, s));
    const int H1 = (H + 6 - 7) / 2 + 1, W1 = (W + 6 - 7) / 2 + 1;   // == Hs, Ws
    int hh = (H1 + 2 - 3) / 2 + 1, ww = (W1 + 2 - 3) / 2 + 1, C = 64;
    // fp16, many frames: conv1 + BN + ReLU + maxpool as ONE launch (conv_hs.hip's pooled epilogue: the 64-channel conv map, 8x the
    // pooled map's bytes, never reaches HBM).  Bit-identical to the two launches, so the choice may depend on the launch size
    bool stem_pool = false;
    if (h16 && !split && h->chain_fuse) {
        ConvParams q{};
        R.probe = &q;
        R.pool.h = hh; R.pool.w = ww;
        R.conv(h->stem, in4, N, Hs, Ws, 1, 2, 2, nullptr, 64, nullptr, 0, ACT_RELU, H1, W1, 0, 0, 0, 0, 0, h16);
        R.pool = Runner::Pool();
        stem_pool = q.pool && conv_hs_supported(q);
    }
    float *cur;
    if (stem_pool) {
        cur = R.alloc(ACT((size_t)N * hh * ww * 64));
        R.pool.h = hh; R.pool.w = ww;
        R.conv(h->stem, in4, N, Hs, Ws, 1, 2, 2, cur, 64, nullptr, 0, ACT_RELU, H1, W1, 0, 0, 0, 0, 0, h16);
        R.pool = Runner::Pool();
        R.release(in4);
    } else {
        float *c1 = R.alloc(ACT((size_t)N * H1 * W1 * 64));
        R.conv(h->stem, in4, N, Hs, Ws, 1, 2, 2, c1, 64, nullptr, 0, ACT_RELU, H1, W1, 0, 0, 0, 0, 0, h16);
        R.release(in4);
        cur = R.alloc(ACT((size_t)N * hh * ww * 64));
        if (split) LAUNCH(launch_maxpool3s2_split(c1, cur, N, H1, W1, 64, hh, ww, s));
        else if (h16) LAUNCH(launch_maxpool3s2_f16(c1, cur, N, H1, W1, 64, hh, ww, s));
        else LAUNCH(launch_maxpool3s2(c1, cur, N, H1, W1, 64, hh, ww, s));
        R.release(c1);
    }

    // ---- residual layers (resnet.py:223-239; Bottleneck 124-144; BasicBlock 90-106)
    float *level[3] = {nullptr, nullptr, nullptr};
    int lc[3], lh[3], lw[3];
    float *t1_chained = nullptr;
    for (int li = 0; li < 3; ++li) {
        for (size_t bi = 0; bi < h->blocks[li].size(); ++bi) {
            const Block &b = h->blocks[li][bi];
            const int ho = (hh + 2 - 3) / b.stride + 1, wo = (ww + 2 - 3) / b.stride + 1;
            float *y;
            int outc;
            if (h->paper) {
                const int planes = b.c1.Cout;
                outc = b.c3.Cout;
                float *t1 = t1_chained;   // conv1 + BN + ReLU of this block came out of the previous block's chained conv3 launch
                t1_chained = nullptr;
                if (!t1) {
                    t1 = R.alloc(ACT((size_t)N * hh * ww * planes));
                    R.conv(b.c1, cur, N, hh, ww, 1, 0, 0, t1, planes, nullptr, 0, ACT_RELU, hh, ww, 0, 0, 0, 0, 0, h16);
                }
                float *t2 = R.alloc(ACT((size_t)N * ho * wo * planes));
                R.conv(b.c2, t1, N, hh, ww, b.stride, 1, 1, t2, planes, nullptr, 0, ACT_RELU, ho, wo, 0, 0, 0, 0, 0, h16);
                R.release(t1);
                // the block behind this one (of this layer or the first of the next): its conv1 reads this block's output and nothing
                // else, so a conv3 launch whose workgroup holds all channels of its pixels computes it from the tile in LDS
                // (conv_stream.hip "chain"; fp16 layer1 at large batches: the 256-channel tensor is read once less per block).
                // Bit-identical to the two launches, so the choice may depend on the launch size (conv_stream_chain_ok)
                const Block *nb = bi + 1 < h->blocks[li].size() ? &h->blocks[li][bi + 1] : (li + 1 < 3 && !h->blocks[li + 1].empty() ? &h->blocks[li + 1][0] : nullptr);
                const bool nb_ok = h->chain_fuse && nb && h16 && !split && nb->c1.f16 && !nb->c1.plane && nb->c1.R == 1 && nb->c1.S == 1 &&
                                   nb->c1.Cin == outc && nb->c1.K == outc && nb->c1.Kpad == outc && !nb->c1.tall && !nb->c1.rd_cout;
                const float *res = cur;
                float *dsb = nullptr;
                if (b.fused_ds) {   // conv3 and the downsample branch as one GEMM over [t2 | x]
                    y = R.alloc(ACT((size_t)N * ho * wo * outc));
                    bool ch = false;
                    if (nb_ok) {
                        ConvParams q{};
                        R.probe = &q;
                        R.conv_dual(b.c3ds, t2, planes, cur, C, N, hh, ww, b.stride, y, outc, ho, wo, h16);
                        ch = conv_stream_chain_ok(q, nb->c1.Cout);
                    }
                    if (ch) { t1_chained = R.alloc(ACT((size_t)N * ho * wo * nb->c1.Cout)); R.chain.L = &nb->c1; R.chain.out = t1_chained; }
                    R.conv_dual(b.c3ds, t2, planes, cur, C, N, hh, ww, b.stride, y, outc, ho, wo, h16);
                    R.chain = Runner::Chain();
                } else {
                    if (b.has_ds) {
                        dsb = R.alloc(ACT((size_t)N * ho * wo * outc));
                        R.conv(b.ds, cur, N, hh, ww, b.stride, 0, 0, dsb, outc, nullptr, 0, ACT_NONE, ho, wo, 0, 0, 0, 0, 0, h16);
                        res = dsb;
                    }
                    y = R.alloc(ACT((size_t)N * ho * wo * outc));
                    bool ch = false;
                    if (nb_ok) {
                        ConvParams q{};
                        R.probe = &q;
                        R.conv(b.c3, t2, N, ho, wo, 1, 0, 0, y, outc, res, outc, ACT_RELU, ho, wo, 0, 0, 0, 0, 0, h16);
                        ch = conv_stream_chain_ok(q, nb->c1.Cout);
                    }
                    if (ch) { t1_chained = R.alloc(ACT((size_t)N * ho * wo * nb->c1.Cout)); R.chain.L = &nb->c1; R.chain.out = t1_chained; }
                    R.conv(b.c3, t2, N, ho, wo, 1, 0, 0, y, outc, res, outc, ACT_RELU, ho, wo, 0, 0, 0, 0, 0, h16);
                    R.chain = Runner::Chain();
                }
                R.release(t2);
                R.release(dsb);
            } else {
                const int planes = b.c1.Cout;
                outc = planes;
                float *t1 = R.alloc(ACT((size_t)N * ho * wo * planes));
                R.conv(b.c1, cur, N, hh, ww, b.stride, 1, 1, t1, planes, nullptr, 0, ACT_RELU, ho, wo, 0, 0, 0, 0, 0, h16);
                const float *res = cur;
                float *dsb = nullptr;
                if (b.has_ds) {
                    dsb = R.alloc(ACT((size_t)N * ho * wo * outc));
                    R.conv(b.ds, cur, N, hh, ww, b.stride, 0, 0, dsb, outc, nullptr, 0, ACT_NONE, ho, wo, 0, 0, 0, 0, 0, h16);
                    res = dsb;
                }
                y = R.alloc(ACT((size_t)N * ho * wo * outc));
                R.conv(b.c2, t1, N, ho, wo, 1, 1, 1, y, outc, res, outc, ACT_RELU, ho, wo, 0, 0, 0, 0, 0, h16);
                R.release(t1);
                R.release(dsb);
            }
            bool keep = false;
            for (int q = 0; q < 3; ++q) keep |= (level[q] == cur);
            if (!keep) R.release(cur);
            cur = y; C = outc; hh = ho; ww = wo;
        }
        // handmvnet.py:165-177: feats = [layer3, layer2, layer1][:n_levels]; keep only what is sampled
        const bool needed = (li == 2) || (2 - li) < c.n_levels;
        lc[li] = C; lh[li] = hh; lw[li] = ww;
        if (needed) level[li] = cur;
    }
    float *feat0 = level[2];
    const int fh = lh[2], fw = lw[2];
    if (h->capture && !dry && h->cap_feat0) {
        if (split) LAUNCH(launch_nhwc_split_to_nchw(feat0, h->cap_feat0, N, fh, fw, lc[2], s));
        else if (h16) LAUNCH(launch_nhwc_f16_to_nchw(feat0, h->cap_feat0, N, fh, fw, lc[2], s));
        else LAUNCH(launch_nhwc_to_nchw(feat0, h->cap_feat0, N, fh, fw, lc[2], s));
    }

    // ---- pose_net (handmvnet.py:70-86, 180) -> channels-last heat map with row stride 32
    if (h->paper) {
        hmh = fh; hmw = fw;
        float *ph = R.alloc(ACT((size_t)N * fh * fw * 512));
        R.conv(h->pose0, feat0, N, fh, fw, 1, 0, 0, ph, 512, nullptr, 0, ACT_RELU, fh, fw, 0, 0, 0, 0, 0, h16);
        hm = R.alloc((size_t)N * hmh * hmw * 32);   // heat-map logits are ALWAYS fp32 (x1000 temperature)
        R.conv(h->pose1, ph, N, fh, fw, 1, 0, 0, hm, 32, nullptr, 0, ACT_NONE, fh, fw);
        R.release(ph);
    } else {
        hmh = 2 * fh; hmw = 2 * fw;
        float *p0 = R.alloc(ACT((size_t)N * hmh * hmw * 128));
        if (h->deconv_all.w) {
            R.phases = 4; R.phase_stride = h->deconv_stride;
            R.conv(h->deconv_all, feat0, N, fh, fw, 1, 1, 1, p0, 128, nullptr, 0, ACT_RELU, fh, fw, 0, 0, /*scatter=*/1, 0, 0, h16);
            R.phases = 0;
        } else {
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    R.conv(h->deconv[a * 2 + b], feat0, N, fh, fw, 1, 1 - a, 1 - b, p0, 128, nullptr, 0, ACT_RELU, fh, fw, 0, 0,
                           /*scatter=*/1, a, b, h16);
        }
        float *p1 = R.alloc(ACT((size_t)N * hmh * hmw * 64));
        R.conv(h->pose1, p0, N, hmh, hmw, 1, 1, 1, p1, 64, nullptr, 0, ACT_RELU, hmh, hmw, 0, 0, 0, 0, 0, h16);
        R.release(p0);
        hm = R.alloc((size_t)N * hmh * hmw * 32);
        R.conv(h->pose2, p1, N, hmh, hmw, 1, 1, 1, hm, 32, nullptr, 0, ACT_NONE, hmh, hmw);
        R.release(p1);
    }
    for (int i = 0; i < 3; ++i) {   // feats = [layer3, layer2, layer1]; only the kept levels are non-null
        const int li = 2 - i;
        if (!level[li]) continue;
        lvl[nkeep] = level[li]; lvc[nkeep] = lc[li]; lvld[nkeep] = lc[li]; lvh[nkeep] = lh[li]; lvw[nkeep] = lw[li];
        ++nkeep;
    }
    }   // resnet backbones
    // ---- soft-argmax (handmvnet.py:182, 252)
    float *coords = R.alloc((size_t)N * NJ * 2);
    LAUNCH(launch_soft_argmax(hm, 32, N, hmh, hmw, coords, crop_img, (float)c.image_size, (float)c.heatmap_size, heatmap, s));
    R.release(hm);
    if (h->capture && !dry && h->cap_coords)
        LAUNCH(hipMemcpyAsync(h->cap_coords, coords, (size_t)N * NJ * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));

    // ---- sample nets as gather -> conv1x1+BN+ReLU -> bilinear blend (nets.py:46-63; handmvnet.py:185-187)
    float *tokens = R.alloc((size_t)N * NJ * ldt);
    int col0 = 0;
    for (int i = 0; i < c.n_levels; ++i) {
        const int Ci = lvld[i], co = lvc[i] / 2;   // gather the (padded) channel rows; the conv's pad weights are zero
        float *g = R.alloc(ACT((size_t)N * NJ * 4 * Ci));
        // a split row is Ci (hi, lo) pairs = Ci 4-byte elements, like fp32
        LAUNCH(launch_sample_gather(lvl[i], N, lvh[i], lvw[i], Ci, coords, g, s, (h16 && !split) ? 2 : 4));
        float *s4 = R.alloc((size_t)N * NJ * 4 * co);
        R.gemm(h->sample[i], g, N * NJ * 4, s4, co, nullptr, 0, ACT_RELU);
        R.release(g);
        LAUNCH(launch_sample_blend(s4, co, co, N, lvh[i], lvw[i], coords, tokens, ldt, col0, s));
        R.release(s4);
        col0 += co;
    }
    for (int i = 0; i < nkeep; ++i) R.release(lvl[i]);
    // pos2d / FoV / zero pad / PE (handmvnet.py:189-225; fusion.py:27-28).  In the fp16-kernel modes the q/k/v projections of
    // CrossAttentionFusion read their token rows as (hi, lo) fp16 pairs: the kernel that produces a block's input rows -- this one for
    // block 0, ff_block_kernel for the others -- writes that copy itself (rows_f32_to_half's arithmetic, one launch less per block)
    float *Xpairs = nullptr;   // [hi ldt | lo ldt] halfs per row of X, when its producer wrote them
    if (!h->lq && c.fusion_layers > 0 && h->attn[0].qkv.plane) Xpairs = R.alloc((size_t)N * NJ * ldt);
    LAUNCH(launch_tokens_finalize(tokens, ldt, d, h->fdim, N, V, coords, bbox, intr, c.pos_enc,
                                  (h->lq || !(c.pos_enc & HMV_POS_SIN)) ? nullptr : h->pe,   // the learnable-query blocks add their own PE
                                  (h->capture && h->cap_tokens) ? h->cap_tokens : nullptr, s, Xpairs));
    R.release(coords);

    float *X = tokens;
    int Tcur = V * NJ;
    // q/k/v projection of fp32 token rows; in the fp16 / f32x3 modes on the fused split kernels (Loader::linear_x3)
    // pairs_out: the result rows as (hi, lo) fp16 pairs [hi ldc | lo ldc] (the GEMM's pair epilogue; same bytes as fp32 rows) for a consumer
    // that multiplies on the fp16 matrix cores (attention_x3_kernel)
    auto project = [&](const Layer &L, const float *a, int rows, float *out, int ldc, float *ready_pairs = nullptr, bool pairs_out = false) {
        if (!L.plane) { R.gemm(L, a, rows, out, ldc, nullptr, 0, ACT_NONE); return; }
        float *pairs = ready_pairs;
        if (!pairs) {
            pairs = R.alloc((size_t)rows * ldt);                           // [hi ldt | lo ldt] halfs per row
            LAUNCH(launch_rows_f32_to_half(a, pairs, (size_t)rows, ldt, 2, s));
        }
        R.conv(L, pairs, rows, 1, 1, 1, 0, 0, out, ldc, nullptr, 0, ACT_NONE, 1, 1, 0, 0, 0, 0, 0, pairs_out);
        R.release(pairs);
    };
    if (h->lq) {
        // ---- CrossAttentionFusionLearnableQuery (fusion.py:33-49; MultiHeadAttentionLearnableQuery layers.py:273-301)
        for (int l = 0; l < 5; ++l) {
            const AttnLayer &a = h->attn[l];
            const bool cross = l == 2;
            const int rows = B * Tcur, Tq = cross ? NJ : Tcur, qrows = B * Tq;
            float *xp = R.alloc((size_t)rows * ldt);                       // x = self.pos_embed(x)
            LAUNCH(launch_add_pe(X, ldt, rows, Tcur, d, h->pe, xp, ldt, s));
            R.release(X);
            float *att = R.alloc((size_t)qrows * INNER_LQ);
            const bool tx3 = a.out_x3.plane != 0 && R.ff_fusable(a.out, a.ff1, a.ff2, qrows, ldt, true);   // attention rows as (hi, lo) pairs
            if (cross) {
                float *kv = R.alloc((size_t)rows * 2 * INNER_LQ);
                project(a.kv, xp, rows, kv, 2 * INNER_LQ);
                LAUNCH(launch_attention_d256(a.qprobe, INNER_LQ, 0, kv, kv + INNER_LQ, 2 * INNER_LQ, B, Tcur, Tq, att, s, tx3 ? 1 : 0));
                R.release(kv);
            } else {
                float *qkv = R.alloc((size_t)rows * 3 * INNER_LQ);
                project(a.qkv, xp, rows, qkv, 3 * INNER_LQ);
                LAUNCH(launch_attention_d256(qkv, 3 * INNER_LQ, Tcur, qkv + INNER_LQ, qkv + 2 * INNER_LQ, 3 * INNER_LQ, B, Tcur, Tq, att, s, tx3 ? 1 : 0));
                R.release(qkv);
            }
            if (R.ff_fusable(a.out, a.ff1, a.ff2, qrows, ldt, tx3)) {
                // out = to_out(att) (+ x, not in the probe block); out = ff(out) + out: no LayerNorm around the attention, the
                // FeedForward keeps its own; pad columns come out as zeros
                float *Xf = R.alloc((size_t)qrows * ldt);
                R.ff_block(a.out, a.ff1, a.ff2, att, qrows, cross ? nullptr : xp, ldt, 0, 0, nullptr, nullptr, a.fg, a.fb, nullptr, nullptr,
                           Xf, ldt, d, nullptr, tx3 ? &a.out_x3 : nullptr);
                R.release(att);
                R.release(xp);
                X = Xf;
                Tcur = Tq;
                continue;
            }
            float *o = R.alloc((size_t)qrows * ldt);
            if (cross) R.gemm(a.out, att, qrows, o, ldt, nullptr, 0, ACT_NONE);   // out = to_out(att); no residual from the tokens
            else R.gemm(a.out, att, qrows, o, ldt, xp, ldt, ACT_NONE);            // out = to_out(att) + x
            R.release(att);
            R.release(xp);
            float *f0 = R.alloc((size_t)qrows * ldt);
            LAUNCH(launch_layernorm(o, ldt, qrows, d, a.fg, a.fb, f0, ldt, nullptr, nullptr, nullptr, s));
            float *f1 = R.alloc((size_t)qrows * DHEAD_LQ);
            R.gemm(a.ff1, f0, qrows, f1, DHEAD_LQ, nullptr, 0, ACT_GELU);
            R.release(f0);
            float *Xn = R.alloc((size_t)qrows * ldt);
            R.gemm(a.ff2, f1, qrows, Xn, ldt, o, ldt, ACT_NONE);          // out = ff(out) + out
            R.release(f1);
            R.release(o);
            // The decoder's GEMM reads all ldt columns of the last block's output (its weights are zero there, but 0 * NaN is NaN)
            // and no epilogue writes them: a GEMM epilogue stops at round4(d), the split-K reduction at d.  Inner blocks go
            // through add_pe, which writes the pad itself.
            if (l == 4 && ldt > d) LAUNCH(hipMemset2DAsync(Xn + d, (size_t)ldt * sizeof(float), 0, (size_t)(ldt - d) * sizeof(float), (size_t)qrows, s));
            X = Xn;
            Tcur = Tq;
        }
    }
    // ---- CrossAttentionFusion (fusion.py:7-30; layers.py:202-237)
    const int half = (c.fusion_layers - 1) / 2;
    for (int l = 0; l < (h->lq ? 0 : c.fusion_layers); ++l) {
        const AttnLayer &a = h->attn[l];
        const bool cross = (l == half);
        const int Tq = cross ? NJ : Tcur, koff = cross ? NJ : 0, Tk = cross ? Tcur - NJ : Tcur;
        const int rows = B * Tcur, qrows = B * Tq;
#ifdef HMV_NO_ATT_X3   // A/B builds only (python -m handmvnet_amd.build --variant noax HMV_NO_ATT_X3): the exact-fp32 attention in every mode
        const int att_x3 = 0;
#else
        // fp16-kernel modes: the projection writes q, k, v as (hi, lo) pairs and the attention multiplies on the fp16 matrix cores (by the
        // arithmetic mode alone: a sample's result never depends on the batch)
        const int att_x3 = (h16 && a.qkv.plane) ? 1 : 0;
#endif
        float *qkv = R.alloc((size_t)rows * 3 * INNER);
        project(a.qkv, X, rows, qkv, 3 * INNER, Xpairs, att_x3 != 0);
        Xpairs = nullptr;
        float *att = R.alloc((size_t)qrows * INNER);
        // fp16-kernel modes, fused tail: the attention rows leave the kernel as (hi, lo) pairs and to_out is a split-pair GEMM
        const bool tx3 = a.out_x3.plane != 0 && R.ff_fusable(a.out, a.ff1, a.ff2, qrows, ldt, true);
        if (Tk > 0) LAUNCH(launch_attention(qkv, B, Tcur, Tq, koff, Tk, att, s, tx3 ? 1 : 0, att_x3));
        else LAUNCH(hipMemsetAsync(att, 0, (size_t)qrows * INNER * sizeof(float), s));   // (zero rows are zero pairs)
        R.release(qkv);
        if (R.ff_fusable(a.out, a.ff1, a.ff2, qrows, ldt, tx3)) {   // norm1(to_out + _q) -> FeedForward -> norm2 in one launch behind the GEMM
            float *Xf = R.alloc((size_t)qrows * ldt);
            if (l + 1 < c.fusion_layers && h->attn[l + 1].qkv.plane) Xpairs = R.alloc((size_t)qrows * ldt);   // the next block's projection input
            R.ff_block(a.out, a.ff1, a.ff2, att, qrows, X, ldt, cross ? Tq : 0, cross ? Tcur : 0, a.n1g, a.n1b, a.fg, a.fb, a.n2g, a.n2b, Xf,
                       ldt, d, Xpairs, tx3 ? &a.out_x3 : nullptr);
            R.release(att);
            R.release(X);
            X = Xf;
            Tcur = Tq;
            continue;
        }
        float *n1 = R.alloc((size_t)qrows * ldt), *f0 = R.alloc((size_t)qrows * ldt);
        R.gemm_ln(a.out, att, qrows, X, ldt, cross ? Tq : 0, cross ? Tcur : 0, a.n1g, a.n1b, n1, ldt, a.fg, a.fb, f0);  // norm1(to_out + _q), ff LN
        R.release(att);
        float *f1 = R.alloc((size_t)qrows * DHEAD);
        R.gemm(a.ff1, f0, qrows, f1, DHEAD, nullptr, 0, ACT_GELU);
        R.release(f0);
        float *f2 = R.alloc((size_t)qrows * ldt);
        R.gemm(a.ff2, f1, qrows, f2, ldt, n1, ldt, ACT_NONE);
        R.release(f1);
        float *Xn = R.alloc((size_t)qrows * ldt);
        LAUNCH(launch_layernorm(f2, ldt, qrows, d, a.n2g, a.n2b, Xn, ldt, nullptr, nullptr, nullptr, s));
        R.release(f2);
        R.release(n1);
        R.release(X);
        X = Xn;
        Tcur = Tq;
    }
    if (h->capture && !dry && h->cap_fused) LAUNCH(launch_copy_rows(X, ldt, h->cap_fused, d, B * Tcur, d, s));

    // ---- decoder (nets.py:133-139 / 150-154)
    const int jr = B * NJ;
    if (c.decoder == HMV_DECODER_GCN && h->cheb_fuse && h->gcn[0].Kpad == ldt &&
        cheb_fusable(ldt, ldt, h->gcn[0].Kpad, 256, h->gcn[1].Kpad, 64, h->gcn[2].Kpad, 3)) {
        // the three ChebConv layers in two launches (fusion_kernels.hip): layer 1 per (sample, 16 channels), layers 2 + 3 per sample
        float *scr = R.alloc((size_t)jr * 256);
        if (!dry && R.rc == HMV_OK) {
            ChebFusedParams p{};
            p.x = X; p.ldx = ldt; p.B = B; p.K = ldt;
            p.w1 = h->gcn[0].w; p.ldw1 = h->gcn[0].Kpad; p.c1 = 256; p.bias1 = h->gcn_bias[0];
            p.w2 = h->gcn[1].w; p.ldw2 = h->gcn[1].Kpad; p.c2 = 64; p.bias2 = h->gcn_bias[1];
            p.w3 = h->gcn[2].w; p.ldw3 = h->gcn[2].Kpad; p.c3 = 3; p.bias3 = h->gcn_bias[2];
            p.tk = h->cheb_t; p.scratch = scr; p.out = joints_cam; p.ldo = 3;
            R.check(launch_cheb_fused(p, s), "cheb_fused");
            h->launches += 2;
        }
        R.release(X);
        R.release(scr);
    } else if (c.decoder == HMV_DECODER_GCN) {
        const int dims[4] = {d, 256, 64, 3};
        float *xin = X;
        for (int i = 0; i < 3; ++i) {
            const int co = dims[i + 1];
            float *y = R.alloc((size_t)jr * 3 * co);
            R.gemm(h->gcn[i], xin, jr, y, 3 * co, nullptr, 0, ACT_NONE);
            R.release(xin);
            float *out = i == 2 ? joints_cam : R.alloc((size_t)jr * co);
            LAUNCH(launch_cheb_mix(y, 3 * co, B, co, h->cheb_t, h->gcn_bias[i], i < 2, out, i == 2 ? 3 : co, s));
            R.release(y);
            xin = i == 2 ? nullptr : out;
        }
    } else {
        float *g1 = R.alloc((size_t)jr * 64);
        R.gemm(h->fc1, X, jr, g1, 64, nullptr, 0, ACT_LEAKY);
        R.release(X);
        R.gemm(h->fc2, g1, jr, joints_cam, 3, nullptr, 0, ACT_NONE);
        R.release(g1);
    }
#undef LAUNCH
#undef ACT
    return R.rc;
}

int ensure_capture(hmv_engine *h, int B) {
    if (!h->capture || h->cap_batch >= B) return HMV_OK;
    const hmv_config &c = h->cfg;
    const int N = B * c.num_views;
    // size of feats[0] by the backbones' conv arithmetic (any frame size): 3x3 s2 p1 (and 7x7 s2 p3, 1x1 s2) all give (n - 1) / 2 + 1
    auto half_up = [](int n) { return (n - 1) / 2 + 1; };
    int fh = half_up(half_up(c.height)), fw = half_up(half_up(c.width));           // stem: H/4
    if (!h->hrnet) { fh = half_up(fh); fw = half_up(fw); }                         // layer2
    if (!h->hrnet && !h->paper) { fh = half_up(fh); fw = half_up(fw); }            // layer3 of ResNet-18/34
    for (float **p : {&h->cap_feat0, &h->cap_coords, &h->cap_tokens, &h->cap_fused}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    h->cap_feat0_n = (size_t)N * c.channels[0] * fh * fw;
    h->cap_coords_n = (size_t)N * NJ * 2;
    h->cap_tokens_n = (size_t)N * NJ * h->d;
    h->cap_fused_n = (size_t)B * NJ * h->d;
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->cap_feat0), h->cap_feat0_n * sizeof(float)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->cap_coords), h->cap_coords_n * sizeof(float)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->cap_tokens), h->cap_tokens_n * sizeof(float)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->cap_fused), h->cap_fused_n * sizeof(float)));
    h->cap_batch = B;
    return HMV_OK;
}

}  // namespace

extern "C" {

size_t hmv_workspace_bytes(hmv_handle h, int32_t batch) {
    if (!h || batch <= 0) return 0;
    Arena dry;
    dry.reset(reinterpret_cast<char *>(uintptr_t(1) << 40));  // fake non-null base: offsets only
    const bool saved = h->profiling;
    h->profiling = false;
    run_forward(h, batch, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, /*dry=*/true, dry);
    h->profiling = saved;
    return dry.high;
}

int hmv_reserve(hmv_handle h, int32_t batch) {
    if (!h || batch <= 0) return h ? h->fail(HMV_ERR_ARG, "batch must be positive") : HMV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (batch > h->reserved_batch) {
        const size_t need = hmv_workspace_bytes(h, batch);
        if (h->arena) {
            HIPCHK(h, hipDeviceSynchronize());
            h->drop_graphs();   // captured launches point into the old workspace
            HIPCHK(h, hipFree(h->arena));
            h->arena = nullptr;
        }
        HIPCHK(h, hipMalloc(reinterpret_cast<void **>(&h->arena), need));
        h->arena_bytes = need;
        h->reserved_batch = batch;
    }
    return ensure_capture(h, batch);
}

static int forward_eager(hmv_handle h, int32_t batch, const float *x, const float *bbox, const float *intrinsic,
                         float *joints_crop_img, float *joints_cam, float *heatmap, hipStream_t stream) {
    h->plan.reset(h->arena);
    h->launches = 0;
    const int rc = run_forward(h, batch, x, bbox, intrinsic, joints_crop_img, joints_cam, heatmap, stream, false, h->plan);
    if (rc == HMV_OK && h->plan.high > h->arena_bytes) return h->fail(HMV_ERR_STATE, "workspace plan exceeded its reservation");
    return rc;
}

static int forward_common(hmv_handle h, int32_t batch, const float *x, const float *bbox, const float *intrinsic,
                          float *joints_crop_img, float *joints_cam, float *heatmap, void *stream) {
    if ((h->cfg.pos_enc & HMV_POS_CROP) && (!bbox || !intrinsic))
        return h->fail(HMV_ERR_ARG, "pos_enc contains 'crop': bbox and cam_params['intrinsic'] are required");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (batch > h->reserved_batch || (h->capture && h->cap_batch < batch)) {
        const int rc = hmv_reserve(h, batch);
        if (rc != HMV_OK) return rc;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // stage capture copies into side buffers and profiling brackets launches with events: both stay eager
    if (!h->graphs || h->capture || h->profiling)
        return forward_eager(h, batch, x, bbox, intrinsic, joints_crop_img, joints_cam, heatmap, s);

    hmv_engine::GraphKey key{};
    key[0] = (uintptr_t)batch; key[1] = (uintptr_t)x; key[2] = (uintptr_t)bbox; key[3] = (uintptr_t)intrinsic;
    key[4] = (uintptr_t)joints_crop_img; key[5] = (uintptr_t)joints_cam; key[6] = (uintptr_t)heatmap;
    key[7] = (uintptr_t)h->fsrc.frames; key[8] = (uintptr_t)h->fsrc.boxes;
    key[9] = ((uintptr_t)(unsigned)h->fsrc.fh << 32) | (uintptr_t)(unsigned)h->fsrc.fw;
    if (h->fsrc.frames) {
        uint32_t bits[6];
        memcpy(bits, h->fsrc.mean, 12);
        memcpy(bits + 3, h->fsrc.std, 12);
        key[10] = ((uintptr_t)bits[0] << 32) ^ ((uintptr_t)bits[1] << 16) ^ (uintptr_t)bits[2];
        key[11] = ((uintptr_t)bits[3] << 32) ^ ((uintptr_t)bits[4] << 16) ^ (uintptr_t)bits[5];
    }
    for (auto &e : h->gcache)
        if (e.key == key) {
            e.stamp = ++h->gclock;
            ++h->greplays;
            HIPCHK(h, hipGraphLaunch(e.exec, s));
            return HMV_OK;
        }
    auto seen = std::find(h->gseen.begin(), h->gseen.end(), key);
    if (seen == h->gseen.end()) {   // first sighting: run eagerly (also configures kernels, zero page, ...)
        if (h->gseen.size() >= 16) h->gseen.erase(h->gseen.begin());
        h->gseen.push_back(key);
        return forward_eager(h, batch, x, bbox, intrinsic, joints_crop_img, joints_cam, heatmap, s);
    }
    // second use of the same buffers: capture on the engine's own stream, then launch the graph on the caller's
    h->gseen.erase(seen);
    if (!h->gstream) HIPCHK(h, hipStreamCreateWithFlags(&h->gstream, hipStreamNonBlocking));
    HIPCHK(h, hipStreamBeginCapture(h->gstream, hipStreamCaptureModeThreadLocal));
    const int rc = forward_eager(h, batch, x, bbox, intrinsic, joints_crop_img, joints_cam, heatmap, h->gstream);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(h->gstream, &graph);
    hipGraphExec_t exec = nullptr;
    if (rc != HMV_OK || ce != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        h->graphs = false;   // capture is not available here: stay eager for the life of the handle
        if (rc != HMV_OK) return rc;
        return forward_eager(h, batch, x, bbox, intrinsic, joints_crop_img, joints_cam, heatmap, s);
    }
    (void)hipGraphDestroy(graph);
    if (h->gcache.size() >= 8) {   // least recently replayed entry makes room
        size_t lru = 0;
        for (size_t i = 1; i < h->gcache.size(); ++i)
            if (h->gcache[i].stamp < h->gcache[lru].stamp) lru = i;
        (void)hipGraphExecDestroy(h->gcache[lru].exec);
        h->gcache.erase(h->gcache.begin() + (long)lru);
    }
    h->gcache.push_back({key, exec, ++h->gclock});
    HIPCHK(h, hipGraphLaunch(exec, s));
    return HMV_OK;
}

int hmv_set_graphs(hmv_handle h, int32_t enable) {
    if (!h) return HMV_ERR_ARG;
    (void)hipSetDevice(h->cfg.device);
    h->graphs = enable != 0;
    if (!h->graphs) { (void)hipDeviceSynchronize(); h->drop_graphs(); }
    return HMV_OK;
}

int hmv_graph_stats(hmv_handle h, int32_t *cached, int64_t *replays) {
    if (!h) return HMV_ERR_ARG;
    if (cached) *cached = (int32_t)h->gcache.size();
    if (replays) *replays = (int64_t)h->greplays;
    return HMV_OK;
}

int hmv_forward(hmv_handle h, int32_t batch, const float *x, const float *bbox, const float *intrinsic, float *joints_crop_img,
                float *joints_cam, float *heatmap, void *stream) {
    if (!h) return HMV_ERR_ARG;
    if (!h->finalized) return h->fail(HMV_ERR_STATE, "hmv_finalize_weights has not succeeded on this handle");
    if (batch <= 0 || !x || !joints_crop_img || !joints_cam) return h->fail(HMV_ERR_ARG, "null or empty input/output");
    return forward_common(h, batch, x, bbox, intrinsic, joints_crop_img, joints_cam, heatmap, stream);
}

int hmv_forward_frames(hmv_handle h, int32_t batch, const uint8_t *frames, int32_t frame_h, int32_t frame_w, const int32_t *crop_boxes,
                       const float *mean, const float *std, const float *bbox, const float *intrinsic, float *joints_crop_img,
                       float *joints_cam, float *heatmap, void *stream) {
    if (!h) return HMV_ERR_ARG;
    if (!h->finalized) return h->fail(HMV_ERR_STATE, "hmv_finalize_weights has not succeeded on this handle");
    if (batch <= 0 || !frames || !crop_boxes || !mean || !std || !joints_crop_img || !joints_cam)
        return h->fail(HMV_ERR_ARG, "null or empty input/output");
    if (frame_h <= 0 || frame_w <= 0) return h->fail(HMV_ERR_ARG, "frame size must be positive");
    for (int c = 0; c < 3; ++c)
        if (!(std[c] > 0.f)) return h->fail(HMV_ERR_ARG, "std must be positive");
    h->fsrc.frames = frames;
    h->fsrc.boxes = crop_boxes;
    h->fsrc.fh = frame_h;
    h->fsrc.fw = frame_w;
    for (int c = 0; c < 3; ++c) { h->fsrc.mean[c] = mean[c]; h->fsrc.std[c] = std[c]; }
    const int rc = forward_common(h, batch, nullptr, bbox, intrinsic, joints_crop_img, joints_cam, heatmap, stream);
    h->fsrc.frames = nullptr;
    h->fsrc.boxes = nullptr;
    return rc;
}

int hmv_op_prepare_frames(int32_t device, const uint8_t *frames, int32_t n_frames, int32_t frame_h, int32_t frame_w,
                          const int32_t *crop_boxes, const float *mean, const float *std, int32_t out_h, int32_t out_w, float *out_nhwc4,
                          void *stream) {
    if (!frames || !crop_boxes || !mean || !std || !out_nhwc4 || n_frames <= 0 || frame_h <= 0 || frame_w <= 0 || out_h <= 0 || out_w <= 0)
        return HMV_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return HMV_ERR_HIP;
    return launch_frames_to_input(frames, crop_boxes, n_frames, frame_h, frame_w, out_h, out_w, mean, std, 0, out_nhwc4,
                                  static_cast<hipStream_t>(stream)) == hipSuccess ? HMV_OK : HMV_ERR_HIP;
}

void hmv_destroy(hmv_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    h->drop_graphs();
    if (h->gstream) (void)hipStreamDestroy(h->gstream);
    if (h->aux) (void)hipStreamDestroy(h->aux);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (void *p : h->dev_allocs) (void)hipFree(p);
    if (h->arena) (void)hipFree(h->arena);
    for (float *p : {h->cap_feat0, h->cap_coords, h->cap_tokens, h->cap_fused})
        if (p) (void)hipFree(p);
    for (auto &r : h->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    delete h;
}

/* Test hook: fills the workspace arena with the byte `value` (0xFF = NaNs).  Every stage must write what a later stage reads:
 * a forward after poisoning must give the bits of a forward before it (tests/test_gpu_parity.py::test_poisoned_workspace). */
int hmv_poison_workspace(hmv_handle h, int32_t value, void *stream) {
    if (!h) return HMV_ERR_ARG;
    if (!h->arena || !h->arena_bytes) return HMV_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemsetAsync(h->arena, value & 0xFF, h->arena_bytes, static_cast<hipStream_t>(stream)));
    return HMV_OK;
}

/* Fused tail kernels on (default) / off (the launch-per-op path they replace).  The choice is part of the workspace plan, so the
 * workspace is re-planned on the next forward. */
int hmv_set_tail_fusion(hmv_handle h, int32_t enable) {
    if (!h) return HMV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipDeviceSynchronize());
    h->drop_graphs();
    h->ff_fuse = h->cheb_fuse = enable != 0;
    h->reserved_batch = 0;   // forces hmv_reserve to size the workspace again
    return HMV_OK;
}

/* Chained launches (Bottleneck conv3 -> the next block's conv1 from the output tile in LDS) on (default) / off (one launch per conv:
 * the same bits).  Part of the workspace plan, so the workspace is re-planned on the next forward. */
int hmv_set_x3k16_mode(int32_t mode) {
    if (mode < -1 || mode > 1) { g_create_err = "hmv_set_x3k16_mode: -1, 0 or 1"; return HMV_ERR_ARG; }
    gemm_x3k16_set_mode(mode);
    return HMV_OK;
}

int hmv_set_chain_fusion(hmv_handle h, int32_t enable) {
    if (!h) return HMV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipDeviceSynchronize());
    h->drop_graphs();
    h->chain_fuse = enable != 0;
    h->reserved_batch = 0;
    return HMV_OK;
}

int hmv_set_hr_fusion(hmv_handle h, int32_t enable) {
    if (!h) return HMV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipDeviceSynchronize());
    h->drop_graphs();
    h->hr_fuse = (enable & 1) != 0;
    h->hr_overlap = (enable & 2) != 0;
    h->reserved_batch = 0;
    return HMV_OK;
}

int hmv_set_capture(hmv_handle h, int32_t enable) {
    if (!h) return HMV_ERR_ARG;
    h->capture = enable != 0;
    return HMV_OK;
}

int hmv_read_stage(hmv_handle h, const char *stage, float *dst, size_t capacity, void *stream) {
    if (!h || !stage || !dst) return HMV_ERR_ARG;
    const float *src = nullptr;
    size_t n = 0;
    const std::string st(stage);
    if (st == "feat0") { src = h->cap_feat0; n = h->cap_feat0_n; }
    else if (st == "coords_hm") { src = h->cap_coords; n = h->cap_coords_n; }
    else if (st == "tokens") { src = h->cap_tokens; n = h->cap_tokens_n; }
    else if (st == "fused") { src = h->cap_fused; n = h->cap_fused_n; }
    else return h->fail(HMV_ERR_ARG, "unknown stage %s", stage);
    if (!src) return h->fail(HMV_ERR_STATE, "stage capture was not enabled before the forward");
    if (capacity < n) n = capacity;
    HIPCHK(h, hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return HMV_OK;
}

int hmv_set_profiling(hmv_handle h, int32_t enable) {
    if (!h) return HMV_ERR_ARG;
    h->profiling = enable != 0;
    if (enable == 1) h->prof_used = 0;  // 1 starts a fresh record list, 0 pauses (records kept), 2 resumes; records accumulate
    return HMV_OK;
}

int hmv_profile_count(hmv_handle h) { return h ? (int)h->prof_used : 0; }

/* Device operations (kernel launches, memsets, device copies) the last eagerly run forward enqueued. */
int hmv_launch_count(hmv_handle h) { return h ? h->launches : 0; }

int hmv_profile_get(hmv_handle h, int32_t index, const char **name, const char **label, float *ms, double *flops) {
    if (!h || index < 0 || (size_t)index >= h->prof_used) return HMV_ERR_ARG;
    ProfRec &r = h->prof[index];
    float t = 0.f;
    HIPCHK(h, hipEventElapsedTime(&t, r.e0, r.e1));
    if (name) *name = r.name;
    if (label) *label = r.label.c_str();
    if (ms) *ms = t;
    if (flops) *flops = r.flops;
    return HMV_OK;
}

int hmv_profile_get_bytes(hmv_handle h, int32_t index, double *bytes) {
    if (!h || index < 0 || (size_t)index >= h->prof_used || !bytes) return HMV_ERR_ARG;
    *bytes = h->prof[index].bytes;
    return HMV_OK;
}

int hmv_op_attention(int32_t device, const float *qkv, int32_t B, int32_t T, int32_t Tq, int32_t koff, int32_t Tk, float *out,
                     void *stream) {
    if (!qkv || !out || B <= 0 || T <= 0 || Tq <= 0 || Tq > T || Tk <= 0 || koff < 0 || koff + Tk > T) {
        g_create_err = "hmv_op_attention: bad argument";
        return HMV_ERR_ARG;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return HMV_ERR_HIP; }
    const hipError_t e = launch_attention(qkv, B, T, Tq, koff, Tk, out, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) { g_create_err = std::string("attention launch failed: ") + hipGetErrorString(e); return HMV_ERR_HIP; }
    return HMV_OK;
}

int hmv_op_attention_x3(int32_t device, const float *qkv, int32_t B, int32_t T, int32_t Tq, int32_t koff, int32_t Tk, float *out,
                        void *stream) {
    if (!qkv || !out || B <= 0 || T <= 0 || Tq <= 0 || Tq > T || Tk <= 0 || koff < 0 || koff + Tk > T) {
        g_create_err = "hmv_op_attention_x3: bad argument";
        return HMV_ERR_ARG;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return HMV_ERR_HIP; }
    // the kernel takes rows of (hi, lo) fp16 pairs (what the projection GEMMs write in those modes): split the fp32 rows first
    hipStream_t s = static_cast<hipStream_t>(stream);
    void *pairs = nullptr;
    hipError_t e = hipMalloc(&pairs, (size_t)B * T * 3072 * 4);
    if (e == hipSuccess) e = launch_rows_f32_to_half(qkv, pairs, (size_t)B * T, 3072, 2, s);
    if (e == hipSuccess) e = launch_attention(static_cast<const float *>(pairs), B, T, Tq, koff, Tk, out, s, 0, 1);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (pairs) (void)hipFree(pairs);
    if (e != hipSuccess) { g_create_err = std::string("attention launch failed: ") + hipGetErrorString(e); return HMV_ERR_HIP; }
    return HMV_OK;
}

int hmv_op_attention_lq(int32_t device, const float *q, int32_t q_ld, int32_t q_bstride, const float *k, const float *v, int32_t kv_ld,
                        int32_t B, int32_t T, int32_t Tq, float *out, void *stream) {
    if (!q || !k || !v || !out || B <= 0 || T <= 0 || Tq <= 0 || q_ld < 2048 || kv_ld < 2048 || q_bstride < 0 || (q_ld & 3) || (kv_ld & 3)) {
        g_create_err = "hmv_op_attention_lq: bad argument";
        return HMV_ERR_ARG;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return HMV_ERR_HIP; }
    const hipError_t e = launch_attention_d256(q, q_ld, q_bstride, k, v, kv_ld, B, T, Tq, out, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) { g_create_err = std::string("attention launch failed: ") + hipGetErrorString(e); return HMV_ERR_HIP; }
    return HMV_OK;
}

static const char **g_op_kernel_name = nullptr;   // hmv_op_conv2d_sel: where the kernel family of the next hmv_op_conv2d goes

int hmv_op_conv2d(int32_t device, const float *in, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *w_oihw,
                  const float *bias_host, int32_t Cout, int32_t R, int32_t S, int32_t stride, int32_t pad, const float *residual,
                  int32_t relu, float *out, void *stream) {
    if (!in || !w_oihw || !out || Cin % 4 != 0) {
        g_create_err = "hmv_op_conv2d: Cin must be a multiple of 4";
        return HMV_ERR_ARG;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return HMV_ERR_HIP; }
    const int K = R * S * Cin, Kpad = round_up(K, 32), Cp = round_up(Cout, 256);
    std::vector<float> w((size_t)Cp * Kpad, 0.f), b((size_t)Cp, 0.f);
    for (int o = 0; o < Cout; ++o) {
        for (int k = 0; k < K; ++k) {
            int c, tap;
            if (Cin % 32 == 0) { const int chunk = k / (32 * R * S), rem = k % (32 * R * S); tap = rem / 32; c = chunk * 32 + rem % 32; }
            else { c = k % Cin; tap = k / Cin; }
            w[(size_t)o * Kpad + k] = w_oihw[(((size_t)o * Cin + c) * R + tap / S) * S + tap % S];
        }
        if (bias_host) b[o] = bias_host[o];
    }
    float *dw = nullptr, *db = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dw), w.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&db), b.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(dw, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        ConvParams p{};
        p.in = in; p.wgt = dw; p.bias = db; p.res = residual; p.out = out;
        p.N = N; p.H = H; p.W = W; p.Cin = Cin;
        p.Ho = (H + 2 * pad - R) / stride + 1; p.Wo = (W + 2 * pad - S) / stride + 1; p.Cout = Cout;
        p.R = R; p.S = S; p.stride = stride; p.pad_h = pad; p.pad_w = pad;
        p.K = K; p.Kpad = Kpad; p.M = N * p.Ho * p.Wo; p.ldc = Cout; p.ldr = Cout;
        p.act = relu ? ACT_RELU : ACT_NONE; p.osy = p.osx = 1;
        e = launch_conv(p, conv_pick_tile(p.M, Cout, K, false, residual != nullptr), static_cast<hipStream_t>(stream), g_op_kernel_name);
        if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    }
    if (dw) (void)hipFree(dw);
    if (db) (void)hipFree(db);
    if (e != hipSuccess) { g_create_err = std::string("hmv_op_conv2d: ") + hipGetErrorString(e); return HMV_ERR_HIP; }
    return HMV_OK;
}

}  // extern "C"

// One conv in any arithmetic mode through the engine's own packing (Loader::conv) and launch path (Runner::conv).
// out16: the layer writes fp16 rows (plain fp16 mode only), as every backbone layer of the fp16 path does
static int op_conv2d_any(const char *who, int32_t device, int32_t dtype, const float *in, int32_t N, int32_t H, int32_t W, int32_t Cin,
                         const float *w_oihw, const float *bias_host, int32_t Cout, int32_t R, int32_t S, int32_t stride,
                         int32_t pad, const float *residual, int32_t relu, void *out, bool out16, const char **kernel_name, void *stream,
                         bool tall = false) {
    if (tall && (residual || !out16 || !conv_ht_shape_ok(R, S, stride, pad, Cin, Cout, H, W))) {
        g_create_err = std::string(who) + ": the tall-tile kernel takes 3x3 stride-1 pad-1 convs without residual, Cin % 32 == 0 (>= 64), Cout % 128 == 0, H % 16 == 0, W % 32 == 0";
        return HMV_ERR_ARG;
    }
    if ((dtype != HMV_F16 && dtype != HMV_F32X3) || !in || !w_oihw || !out || Cin % 8 != 0 || Cout % 4 != 0 || (out16 && dtype != HMV_F16)) {
        g_create_err = std::string(who) + ": dtype must be HMV_F32 / HMV_F16 / HMV_F32X3; the fp16-based modes need Cin % 8 == 0, Cout % 4 == 0";
        return HMV_ERR_ARG;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return HMV_ERR_HIP; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    hmv_engine eng;
    eng.cfg.device = device;
    HostTensor wt;
    wt.shape = {Cout, Cin, R, S};
    wt.data.assign(w_oihw, w_oihw + (size_t)Cout * Cin * R * S);
    eng.host["w"] = wt;
    if (bias_host) {
        HostTensor bt;
        bt.shape = {Cout};
        bt.data.assign(bias_host, bias_host + Cout);
        eng.host["b"] = bt;
    }
    Loader L{&eng};
    L.split = dtype == HMV_F32X3;
    Layer layer;
    L.conv(layer, "op", "w", bias_host ? "b" : "", "", Cout, Cin, R, S, 0, true, false, nullptr, tall);
    int rc = L.rc;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    const size_t rows_in = (size_t)N * H * W, rows_out = (size_t)N * Ho * Wo;
    const int mode = dtype == HMV_F32X3 ? 2 : 1;
    void *din = nullptr, *dres = nullptr;
    hipError_t e = hipSuccess;
    if (rc == HMV_OK) {
        e = hipMalloc(&din, rows_in * Cin * (mode == 2 ? 4 : 2));
        if (e == hipSuccess && residual) e = hipMalloc(&dres, rows_out * Cout * (mode == 2 ? 4 : 2));
        if (e == hipSuccess) e = launch_rows_f32_to_half(in, din, rows_in, Cin, mode, s);
        if (e == hipSuccess && residual) e = launch_rows_f32_to_half(residual, dres, rows_out, Cout, mode, s);
        if (e == hipSuccess) {
            Arena dummy;
            Runner Rn{&eng, s, false, HMV_OK, dummy};
            Rn.kernel_name = kernel_name;
            Rn.conv(layer, static_cast<const float *>(din), N, H, W, stride, pad, pad, static_cast<float *>(out), Cout,
                    static_cast<const float *>(dres), Cout, relu ? ACT_RELU : ACT_NONE, Ho, Wo, 0, 0, 0, 0, 0, out16);
            rc = Rn.rc;
            if (rc == HMV_OK) e = hipStreamSynchronize(s);
        }
    }
    if (din) (void)hipFree(din);
    if (dres) (void)hipFree(dres);
    for (void *ptr : eng.dev_allocs) (void)hipFree(ptr);
    if (rc != HMV_OK) { g_create_err = std::string(who) + ": " + eng.err; return rc; }
    if (e != hipSuccess) { g_create_err = std::string(who) + ": " + hipGetErrorString(e); return HMV_ERR_HIP; }
    return HMV_OK;
}

extern "C" int hmv_op_conv2d_ex(int32_t device, int32_t dtype, const float *in, int32_t N, int32_t H, int32_t W, int32_t Cin,
                                const float *w_oihw, const float *bias_host, int32_t Cout, int32_t R, int32_t S, int32_t stride,
                                int32_t pad, const float *residual, int32_t relu, float *out, void *stream) {
    if (dtype == HMV_F32)
        return hmv_op_conv2d(device, in, N, H, W, Cin, w_oihw, bias_host, Cout, R, S, stride, pad, residual, relu, out, stream);
    return op_conv2d_any("hmv_op_conv2d_ex", device, dtype, in, N, H, W, Cin, w_oihw, bias_host, Cout, R, S, stride, pad, residual, relu,
                         out, false, nullptr, stream);
}

// The up-sampling terms of an HRNet fuse layer through hr_fuse.hip alone (op-level parity tests): out = act(base + sum_s up_{2^shift_s}(W_s x_s + b_s)),
// terms added in the order given.  base / x_s device fp32 NHWC; f16 != 0 runs the fp16 instantiation on fp16 copies of them and `out` receives
// fp16 rows (else fp32).  Weights [C][C_s] and biases [C] on the host.
extern "C" int hmv_op_hr_fuse_up(int32_t device, int32_t f16, const float *base, int32_t N, int32_t H, int32_t W, int32_t C, int32_t nsrc,
                                 const float *const *src, const int32_t *src_c, const int32_t *shift, const float *const *w_host,
                                 const float *const *bias_host, int32_t relu, void *out, void *stream) {
    if (!base || !out || !src || !src_c || !shift || !w_host || !bias_host || nsrc < 1 || nsrc > 3 || N <= 0 || H <= 0 || W <= 0 || C <= 0) {
        g_create_err = "hmv_op_hr_fuse_up: bad arguments";
        return HMV_ERR_ARG;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return HMV_ERR_HIP; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    HrFuseParams p{};
    p.N = N; p.H = H; p.W = W; p.C = C; p.ldc = C; p.nsrc = nsrc; p.relu = relu ? 1 : 0; p.f16 = f16 ? 1 : 0;
    std::vector<void *> owned;
    auto fail = [&](int rc, const std::string &msg) {
        for (void *q : owned) (void)hipFree(q);
        g_create_err = "hmv_op_hr_fuse_up: " + msg;
        return rc;
    };
    hipError_t e = hipSuccess;
    auto dev_copy = [&](const std::vector<float> &v) -> float * {
        void *d = nullptr;
        if (e == hipSuccess) e = hipMalloc(&d, v.size() * sizeof(float));
        if (e == hipSuccess) { owned.push_back(d); e = hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice); }
        return static_cast<float *>(d);
    };
    auto as_half = [&](const float *x, size_t rows, int ch) -> const void * {   // fp16 copy of fp32 rows
        void *d = nullptr;
        if (e == hipSuccess) e = hipMalloc(&d, rows * ch * 2);
        if (e == hipSuccess) { owned.push_back(d); e = launch_rows_f32_to_half(x, d, rows, ch, 1, s); }
        return d;
    };
    for (int q = 0; q < nsrc; ++q) {
        HrFuseSrc &S = p.src[q];
        if (!src[q] || !w_host[q] || !bias_host[q] || src_c[q] <= 0 || shift[q] < 1 || shift[q] > 3) return fail(HMV_ERR_ARG, "bad source");
        S.C = src_c[q]; S.ld = src_c[q]; S.shift = shift[q]; S.H = H >> shift[q]; S.W = W >> shift[q];
        S.ldw = round_up(src_c[q], 32);
        std::vector<float> wp((size_t)round_up(C, 16) * S.ldw, 0.f), bp((size_t)round_up(C, 16), 0.f);   // rows / columns past C and C_s are zeros
        for (int o = 0; o < C; ++o) {
            for (int k = 0; k < S.C; ++k) wp[(size_t)o * S.ldw + k] = w_host[q][(size_t)o * S.C + k];
            bp[o] = bias_host[q][o];
        }
        S.w = dev_copy(wp);
        S.bias = dev_copy(bp);
        S.x = f16 ? as_half(src[q], (size_t)N * S.H * S.W, S.C) : static_cast<const void *>(src[q]);
    }
    p.base = f16 ? as_half(base, (size_t)N * H * W, C) : static_cast<const void *>(base);
    p.out = out;
    if (e != hipSuccess) return fail(HMV_ERR_HIP, hipGetErrorString(e));
    if (!hr_fuse_up_plan(p)) return fail(HMV_ERR_ARG, "the shape has no fused form (C % 4 (fp16: 8), C_s % 16, exact 2^shift map sizes, LDS)");
    e = launch_hr_fuse_up(p, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(HMV_ERR_HIP, hipGetErrorString(e));
    for (void *q : owned) (void)hipFree(q);
    return HMV_OK;
}

// One fp32 3x3 stride-1 pad-1 conv C -> C in the ROW-DECOMPOSED packing (Loader::conv, rd) through Runner::conv: what HRNet-w40's 40- /
// 80-channel branches run.  kernel_sel: 0 the launcher's choice, 1 conv_igemm's row-decomposed tiles, 2 conv_rds.hip whatever the size
extern "C" int hmv_op_conv2d_rd(int32_t device, const float *in, int32_t N, int32_t H, int32_t W, int32_t C, const float *w_oihw,
                                const float *bias_host, const float *residual, int32_t relu, float *out, int32_t kernel_sel,
                                const char **kernel_name, void *stream) {
    if (!in || !w_oihw || !out || C % 4 != 0 || 3 * C > 256 || C % 32 == 0 || W <= 0 || 128 % W != 0 || kernel_sel < 0 || kernel_sel > 2) {
        g_create_err = "hmv_op_conv2d_rd: C % 4 == 0, C % 32 != 0, 3 C <= 256, 128 % W == 0, kernel_sel 0 .. 2";
        return HMV_ERR_ARG;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return HMV_ERR_HIP; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    hmv_engine eng;
    eng.cfg.device = device;
    HostTensor wt;
    wt.shape = {C, C, 3, 3};
    wt.data.assign(w_oihw, w_oihw + (size_t)C * C * 9);
    eng.host["w"] = wt;
    // (the row-decomposed packing takes its shift from a BatchNorm: an identity BN whose shift is the bias)
    HostTensor g1, b1, m0, v1;
    g1.shape = b1.shape = m0.shape = v1.shape = {C};
    g1.data.assign(C, 1.f); m0.data.assign(C, 0.f);
    v1.data.assign(C, 1.f - 1e-5f);   // scale = 1 / sqrt(var + eps) == 1 to fp32 rounding
    b1.data.assign(C, 0.f);
    if (bias_host) b1.data.assign(bias_host, bias_host + C);
    eng.host["bn.weight"] = g1; eng.host["bn.bias"] = b1; eng.host["bn.running_mean"] = m0; eng.host["bn.running_var"] = v1;
    Loader L{&eng};
    Layer layer;
    L.conv(layer, "op", "w", "", "bn", C, C, 3, 3, 0, false, true);
    int rc = L.rc;
    hipError_t e = hipSuccess;
    if (rc == HMV_OK && !layer.rd_cout) { eng.err = "the layer was not packed row-decomposed"; rc = HMV_ERR_ARG; }
    if (rc == HMV_OK) {
        conv_rds_set_mode(kernel_sel == 0 ? -1 : kernel_sel - 1);
        Arena dummy;
        Runner Rn{&eng, s, false, HMV_OK, dummy};
        Rn.kernel_name = kernel_name;
        Rn.conv(layer, in, N, H, W, 1, 1, 1, out, C, residual, C, relu ? ACT_RELU : ACT_NONE, H, W, 0, 0, 0, 0, 0, false);
        rc = Rn.rc;
        conv_rds_set_mode(-1);
        if (rc == HMV_OK) e = hipStreamSynchronize(s);
    }
    for (void *ptr : eng.dev_allocs) (void)hipFree(ptr);
    if (rc != HMV_OK) { g_create_err = std::string("hmv_op_conv2d_rd: ") + eng.err; return rc; }
    if (e != hipSuccess) { g_create_err = std::string("hmv_op_conv2d_rd: ") + hipGetErrorString(e); return HMV_ERR_HIP; }
    return HMV_OK;
}

extern "C" int hmv_op_conv2d_sel(int32_t device, const float *in, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *w_oihw,
                                 const float *bias_host, int32_t Cout, int32_t R, int32_t S, int32_t stride, int32_t pad,
                                 const float *residual, int32_t relu, float *out, int32_t kernel_sel, const char **kernel_name, void *stream) {
    if (kernel_sel < 0 || kernel_sel > 2) { g_create_err = "hmv_op_conv2d_sel: kernel_sel must be 0, 1 or 2"; return HMV_ERR_ARG; }
    conv_stream_set_mode(kernel_sel == 0 ? -1 : kernel_sel - 1);
    g_op_kernel_name = kernel_name;
    const int rc = hmv_op_conv2d(device, in, N, H, W, Cin, w_oihw, bias_host, Cout, R, S, stride, pad, residual, relu, out, stream);
    g_op_kernel_name = nullptr;
    conv_stream_set_mode(-1);
    return rc;
}

extern "C" int hmv_op_conv2d_f16(int32_t device, const float *in, int32_t N, int32_t H, int32_t W, int32_t Cin, const float *w_oihw,
                                 const float *bias_host, int32_t Cout, int32_t R, int32_t S, int32_t stride, int32_t pad,
                                 const float *residual, int32_t relu, void *out_f16, int32_t kernel_sel, const char **kernel_name,
                                 void *stream) {
    if (kernel_sel < 0 || kernel_sel > 8) { g_create_err = "hmv_op_conv2d_f16: kernel_sel must be 0 .. 8"; return HMV_ERR_ARG; }
    const int force = kernel_sel == 0 ? -1 : ((kernel_sel == 2 || kernel_sel == 8) ? 1 : 0);   // 3 .. 7 (tall-tile packing) keep the other special kernels out
    conv_gemm8_set_persistent(kernel_sel == 8 ? 2 : (kernel_sel == 2 ? 0 : 1));
    conv_ht_set_mode((kernel_sel == 3 || kernel_sel == 5 || kernel_sel == 7) ? 1 : ((kernel_sel == 4 || kernel_sel == 6) ? 0 : -1));
    conv_ht_set_shape((kernel_sel == 5 || kernel_sel == 6) ? 0 : 1);
    conv_ht_set_persistent(kernel_sel == 7 ? 2 : (kernel_sel == 3 ? 0 : 1));
    conv_stream_set_mode(force);
    conv_gemm8_set_mode(force);
    conv_hs_set_mode(force);
    const int rc = op_conv2d_any("hmv_op_conv2d_f16", device, HMV_F16, in, N, H, W, Cin, w_oihw, bias_host, Cout, R, S, stride, pad,
                                 residual, relu, out_f16, true, kernel_name, stream, kernel_sel >= 3 && kernel_sel <= 7);   // (3 .. 7: the tall-tile packing)
    conv_ht_set_mode(-1);
    conv_ht_set_shape(1);
    conv_ht_set_persistent(1);
    conv_gemm8_set_persistent(1);
    conv_stream_set_mode(-1);
    conv_gemm8_set_mode(-1);
    conv_hs_set_mode(-1);
    return rc;
}
