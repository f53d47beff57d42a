/*
 * handmv.h -- C ABI of libhandmv.so, the MI355X-native (gfx950) HandMvNet inference engine.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference has no FFI
 * layer of its own; the boundary it exposes is the Python object protocol of
 *   HandMvNet(train_params, model_params, data_params)      /root/reference/src/models/handmvnet.py:28
 *   HandMvNet.forward(x, bbox=None, cam_params=None)->dict  /root/reference/src/models/handmvnet.py:158-266
 *   load_state_dict(state_dict, strict=True)                /root/reference/src/eval.py:46,50
 * handmvnet_amd/model.py mirrors that protocol and binds the entry points below through
 * ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions: plain C types only; integer status codes (0 = HMV_OK); no exceptions cross
 * the ABI; the caller owns every input/output buffer, the engine owns weights + workspace;
 * hmv_forward is asynchronous on the stream it is given (caller synchronises); a handle
 * is bound to one device and is not re-entrant; different handles are independent.
 *
 * Threading / stream contract.  One handle owns ONE workspace arena whose offsets every forward reuses (cached
 * hipGraphs bake them in), plus one set of stage-capture and profiling buffers.  A handle is therefore
 *   - not thread-safe: calls on one handle must be serialised by the caller;
 *   - single-stream at a time: a forward may be enqueued behind another forward of the SAME handle only on the same
 *     stream; to move a handle to another stream, synchronise (or event-order) the first stream before the next
 *     hmv_forward.  Two forwards of one handle in flight on two streams race on the workspace.
 * Concurrency comes from several handles (one per stream / rank), which share nothing.
 * hmv_set_tensor + hmv_finalize_weights may be repeated on a live handle: finalisation synchronises the device and drops
 * every cached graph before it frees the previous weight buffers.
 */
#ifndef HANDMV_H
#define HANDMV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hmv_engine *hmv_handle;

enum { HMV_OK = 0, HMV_ERR_ARG = 1, HMV_ERR_STATE = 2, HMV_ERR_MISSING_TENSOR = 3, HMV_ERR_SHAPE = 4, HMV_ERR_HIP = 5,
       HMV_ERR_UNSUPPORTED = 6 };

/* model_params["backbone"] = "resnet": backbone_type "18" | "34" | "50_paper"   (handmvnet.py:59-68)
 * model_params["backbone"] = "hrnet":  backbone_type "w40" | "w64"            (handmvnet.py:41-57, hrnet.py:430-447) */
enum { HMV_RESNET18 = 0, HMV_RESNET34 = 1, HMV_RESNET50_PAPER = 2, HMV_HRNET_W40 = 3, HMV_HRNET_W64 = 4 };
/* model_params["pos_enc"] subset of {pos2d, crop, sin}       (handmvnet.py:89-95) */
enum { HMV_POS2D = 1, HMV_POS_CROP = 2, HMV_POS_SIN = 4 };
/* model_params["use_gcn"]: JointsDecoderNN | JointsDecoderGCN (handmvnet.py:152-155) */
enum { HMV_DECODER_NN = 0, HMV_DECODER_GCN = 1 };
/* HMV_F16 = BASELINE configs[4]: conv stack in fp16 storage + fp16 MFMA with fp32 accumulation; heat-map
 * logits, soft-argmax, tokens, fusion transformer and decoder stay fp32.  I/O buffers are fp32 either way. */
/* HMV_F32X3: fp32-equivalent arithmetic on the fp16 matrix cores (every backbone).  Every fp32 value of the
 * conv stack travels as a (hi, lo) fp16 pair (hi = fp16(v), lo = fp16(v - hi); hi + lo == v to 2^-22) and every product is
 * evaluated as hi*hi + lo*hi + hi*lo with fp32 accumulation -- three 2.5 PFLOP/s fp16 MFMAs instead of one 157 TFLOP/s fp32
 * MFMA.  Same bytes in HBM as fp32.  Heat-map logits, soft-argmax, tokens, fusion and decoder are plain fp32 as always. */
enum { HMV_F32 = 0, HMV_F16 = 1, HMV_F32X3 = 2 };
/* model_params["fusion"] (handmvnet.py:137-149): "cross_attn" = CrossAttentionFusion (fusion.py:7-30, every release config);
 * "cross_attn_learnable_query" = CrossAttentionFusionLearnableQuery (fusion.py:33-49; layers.py:240-301): five blocks of
 * heads 8 x 256, a learnable 21-token probe as the query of the middle block, a positional embedding inside every block,
 * no LayerNorm around the attention.  fusion_layers is ignored for it (always 5), and so is HMV_POS_SIN. */
enum { HMV_FUSION_CROSS_ATTN = 0, HMV_FUSION_LEARNABLE_QUERY = 1 };

typedef struct hmv_config {
    int32_t struct_size;   /* sizeof(hmv_config), ABI guard */
    int32_t backbone;      /* HMV_RESNET* | HMV_HRNET_* */
    int32_t n_levels;      /* len(model_params["backbone_channels"]) */
    int32_t channels[4];   /* model_params["backbone_channels"] (ResNet: last level first; HRNet: highest resolution first) */
    int32_t num_views;     /* model_params["num_views"] */
    int32_t height, width; /* frame size the plan is built for (x.shape[-2:]); ResNet backbones: any size >= 32 (the heat map is
                            * ceil-chained like the reference's convs: resnet.py:216-254); HRNet: multiples of 32 (its fuse layers add
                            * 2^k-upsampled maps, hrnet.py:194-212, which only line up at those sizes -- the reference raises otherwise) */
    int32_t image_size;    /* data_params["image_size"]   -- config constant, handmvnet.py:252 */
    int32_t heatmap_size;  /* data_params["heatmap_size"] -- config constant, handmvnet.py:252 */
    int32_t pos_enc;       /* bitmask of HMV_POS* */
    int32_t fusion_layers; /* model_params["fusion_layers"], odd */
    int32_t decoder;       /* HMV_DECODER_* */
    int32_t dtype;         /* HMV_F32 | HMV_F16 | HMV_F32X3 */
    int32_t device;        /* HIP device ordinal */
    int32_t fusion;        /* HMV_FUSION_* */
} hmv_config;

/* Replaces HandMvNet.__init__ (handmvnet.py:28-125): validates the configuration and
 * builds the layer plan; no weights yet. */
int hmv_create(const hmv_config *cfg, hmv_handle *out);

/* Replaces one entry of load_state_dict (eval.py:46,50): `key` is the reference's
 * state_dict key, `host` fp32 data in the reference's layout (OIHW convs, [out,in]
 * linears, ...), copied.  Unknown keys (layer4.*, fc.*) are accepted and ignored. */
int hmv_set_tensor(hmv_handle h, const char *key, const float *host, const int64_t *shape, int32_t ndim);

/* After the last hmv_set_tensor: checks every key the forward reads is present with the
 * right shape (HMV_ERR_MISSING_TENSOR / HMV_ERR_SHAPE name the key in hmv_last_error),
 * folds BatchNorm into conv scale/bias, repacks to the MFMA-friendly K-major layout and
 * uploads.  Replaces .to(device).eval().freeze() (eval_fps.py:63-65). */
int hmv_finalize_weights(hmv_handle h);

/* Bytes of device workspace a forward of `batch` multi-view samples needs. */
size_t hmv_workspace_bytes(hmv_handle h, int32_t batch);

/* (Re)allocates the workspace for up to `batch` samples.  hmv_forward calls it on demand;
 * call it up front to keep allocation out of a timed or graph-captured region. */
int hmv_reserve(hmv_handle h, int32_t batch);

/* Fused tail kernels (fusion_kernels.hip: FeedForward + LayerNorms behind the to_out GEMM as one launch, the ChebConv decoder
 * as two) on (default) or off (the launch-per-op path; also HMV_NO_FFFUSE=1 / HMV_NO_CHEBFUSE=1 at hmv_create time).  A/B runs
 * and the equivalence test; the workspace is re-planned on the next forward. */
int hmv_set_tail_fusion(hmv_handle h, int32_t enable);

/* Cross-layer launches of the fp16 backbone at large batches: conv_stream.hip's chain (a Bottleneck's conv3 + residual and the NEXT
 * Bottleneck's conv1 + BN + ReLU, /root/reference/src/models/backbones/resnet.py:124-144, as ONE launch that computes the second conv
 * from the first one's output tile while it is still on the CU; layer1) and conv_hs.hip's pooled stem (conv1 + BN + ReLU + MaxPool2d,
 * resnet.py:218-221, as one launch), on (default) or off (one launch per op; also HMV_NO_CHAIN=1 / HMV_NO_STEMPOOL=1 in the environment).
 * Both give the same bits; A/B runs and the identity test; the workspace is re-planned on the next forward. */
int hmv_set_chain_fusion(hmv_handle h, int32_t enable);

/* HRNet fuse layers (/root/reference/src/models/backbones/hrnet.py:194-212, y_i = relu(sum_j f_ij(x_j))): the up-sampling terms of an
 * output branch (1x1 conv + BN + nearest up-sampling, j > i: always the last terms of the sum) as ONE launch where there are two or
 * more of them (hr_fuse.hip: the branch's map is read once and written once), on (default) or off (one conv launch per term, each
 * adding the running sum).  fp32 mode: equal up to the summation order inside a dot product; fp16 mode: the fused launch rounds the sum
 * to fp16 once instead of after every term.  The split-precision mode always runs one launch per term.
 * Bit 1 of `enable` (round 4; fp16-kernel modes only -- measured hr40 fp16 -1.2 %, f32x3 -0.9 %, fp32 +0.3 %): a four-branch module's lowest-resolution branch (its eight 3x3 convs) is enqueued on a second stream of the
 * handle beside the branch above it -- both are a few hundred latency-bound tiles per conv and share the CUs; forked from and joined
 * into the caller's stream by events (also under hipGraph capture), same kernels and bits.  enable: 0 neither, 1 fused fuse layers
 * only, 2 branch overlap only, 3 both (default).  A/B runs and the equivalence test; the workspace is re-planned on the next forward. */
int hmv_set_hr_fusion(hmv_handle h, int32_t enable);

/* Test hook: fills the reserved workspace with the byte `value` (0xFF: NaNs) on `stream`.  No stage may read workspace bytes that
 * an earlier stage of the SAME forward has not written, so a forward after poisoning returns the bits of one before it. */
int hmv_poison_workspace(hmv_handle h, int32_t value, void *stream);

/* Replaces HandMvNet.forward (handmvnet.py:158-266).  All pointers are DEVICE pointers.
 *   x               [B][V][3][H][W] fp32 (the reference's NCHW frames)
 *   bbox            [B][V][4]  (x1,y1,x2,y2), may be NULL unless HMV_POS_CROP
 *   intrinsic       [B][V][4]  (fx,fy,cx,cy), may be NULL unless HMV_POS_CROP
 *   joints_crop_img [B][V][21][2]  out
 *   joints_cam      [B][21][3]     out
 *   heatmap         [B][V][21][H/8][W/8] out, may be NULL (skips the NCHW copy)
 *   stream          hipStream_t (NULL = default stream) */
int hmv_forward(hmv_handle h, int32_t batch, const float *x, const float *bbox, const float *intrinsic,
                float *joints_crop_img, float *joints_cam, float *heatmap, void *stream);

/* Human-readable description of the last failure on this handle (or on creation when h is NULL). */
const char *hmv_last_error(hmv_handle h);

void hmv_destroy(hmv_handle h);

/* ---- introspection used by tests and bench.py (not part of the reference's surface) ---- */

/* Copies an intermediate of the LAST forward to a device buffer, converted to the
 * reference's layout.  stage: "feat0" [N][C][h][w], "coords_hm" [N][21][2],
 * "tokens" [B][V*21][d] (before the sinusoidal PE is added), "fused" [B][21][d].
 * Stage capture must have been enabled before that forward. */
int hmv_set_capture(hmv_handle h, int32_t enable);
int hmv_read_stage(hmv_handle h, const char *stage, float *dst_device, size_t capacity_floats, void *stream);

/* Per-launch timing with hipEvents on the forward's stream (0 = off, 1 = on).  When on,
 * hmv_forward records an event pair around every kernel launch of the conv/GEMM kernel
 * family; records accumulate over successive forwards (calling hmv_set_profiling again
 * clears them; enable = 0 pauses and keeps the records, enable = 2 resumes without clearing);
 * hmv_profile_* read them back after the caller has synchronised the stream. */
int hmv_set_profiling(hmv_handle h, int32_t enable);
int hmv_profile_count(hmv_handle h);
/* name: kernel family = one device symbol ("conv_igemm_f32<256x256,1x1>" ...); label: layer ("layer3.2.conv2");
 * ms: duration; flops: algorithmic 2*M*N*K of that launch. */
int hmv_profile_get(hmv_handle h, int32_t index, const char **name, const char **label, float *ms, double *flops);
/* algorithmic HBM bytes of that launch: input pixels, weights, residual and output rows each moved once in the storage type of
 * the arithmetic mode (what bench.py prices the launch's HBM roofline with). */
int hmv_profile_get_bytes(hmv_handle h, int32_t index, double *bytes);
/* Device operations (kernel launches, memsets, device-to-device copies) that the last eagerly run forward of this handle
 * enqueued: the "launches per forward" figure of bench.py (a hipGraph replay enqueues ONE graph of as many nodes). */
int hmv_launch_count(hmv_handle h);

/* One NHWC convolution through the engine's conv kernel (op-level parity tests).
 * in [N][H][W][Cin] device; weight OIHW host (Cin must be a multiple of 4);
 * bias host[Cout] or NULL; residual device [N][Ho][Wo][Cout] or NULL; out device NHWC. */
int hmv_op_conv2d(int32_t device, const float *in, int32_t N, int32_t H, int32_t W, int32_t Cin,
                  const float *weight_oihw_host, const float *bias_host, int32_t Cout, int32_t R, int32_t S,
                  int32_t stride, int32_t pad, const float *residual, int32_t relu, float *out, void *stream);

/* TEST HOOK (process-global, single-threaded like the kernel_sel entries below): which kernel the split-pair token GEMMs with a short
 * reduction (the q / k / v projections of the fp16-kernel modes: layers.py:213-215) take -- -1 the engine's size rule (gemm_x3.hip's
 * 256 x 256 tiles from ~200 tiles up, conv_igemm's fused split loop below), 0 never the 256 x 256 tiles, 1 whenever the shape allows.
 * The two give the same bits; the identity test drives both through hmv_op_conv2d_ex(dtype = HMV_F32X3). */
int hmv_set_x3k16_mode(int32_t mode);

/* The same op in any arithmetic mode (dtype = HMV_F32 | HMV_F16 | HMV_F32X3): the fp32 input / residual are converted to the
 * mode's storage format on the device, the layer is packed exactly as hmv_finalize_weights packs it (no BatchNorm), the
 * output is fp32.  Cin must be a multiple of 8 for the fp16-based modes. */
int hmv_op_conv2d_ex(int32_t device, int32_t dtype, const float *in, int32_t N, int32_t H, int32_t W, int32_t Cin,
                     const float *weight_oihw_host, const float *bias_host, int32_t Cout, int32_t R, int32_t S, int32_t stride,
                     int32_t pad, const float *residual, int32_t relu, float *out, void *stream);

/* One fp32 3x3 stride-1 pad-1 conv C -> C (+ bias, optional residual / ReLU) in the engine's ROW-DECOMPOSED packing -- what HRNet-w40's
 * 40- and 80-channel branch convs run (hrnet.py:96-221).  C % 4 == 0, C % 32 != 0, 3 C <= 256, 128 % W == 0.  kernel_sel: 0 = the
 * launcher's choice, 1 = conv_igemm's row-decomposed tiles, 2 = the persistent weight-stationary kernel (conv_rds.hip; C = 40 / 80,
 * 64 % W == 0) whatever the size.  *kernel_name (optional) receives the family that ran. */
int hmv_op_conv2d_rd(int32_t device, const float *in, int32_t N, int32_t H, int32_t W, int32_t C, const float *weight_oihw_host,
                     const float *bias_host, const float *residual, int32_t relu, float *out, int32_t kernel_sel,
                     const char **kernel_name, void *stream);

/* hmv_op_conv2d with a kernel selector (op-level parity tests): 0 = the launcher's choice, 1 = conv_igemm only, 2 = the persistent
 * weight-stationary kernel (conv_stream.hip, fp32 variant: residual-bearing 1x1 convs with K = 64 / 128 / 256, Cout % 256 == 0)
 * wherever the shape has one, whatever its size.  *kernel_name (optional) receives the family that ran. */
int hmv_op_conv2d_sel(int32_t device, const float *in, int32_t N, int32_t H, int32_t W, int32_t Cin,
                      const float *weight_oihw_host, const float *bias_host, int32_t Cout, int32_t R, int32_t S, int32_t stride,
                      int32_t pad, const float *residual, int32_t relu, float *out, int32_t kernel_sel, const char **kernel_name,
                      void *stream);

/* The fp16-storage op with fp16 OUTPUT rows (what a backbone layer of the fp16 path writes): out_f16 device [N][Ho][Wo][Cout]
 * halfs.  kernel_sel: 0 = the launcher's choice, 1 = conv_igemm only, 2 = the round-3 kernels (conv_stream.hip: persistent
 * weight-stationary residual 1x1; conv_gemm8.hip: phase-interleaved 256 x 256 1x1; conv_hs.hip: halo-streaming few-channel 3x3)
 * wherever the shape has one, whatever its size; 3 = the tall-tile 3x3 kernel (conv_ht.hip: 3x3 stride 1 pad 1 without residual,
 * Cin % 32 == 0 >= 64, Cout % 128 == 0, H % 16 == 0, W % 32 == 0, else HMV_ERR_ARG) with the weights packed in ITS reduction order,
 * as hmv_finalize_weights packs the layers the engine gives to it, on the 16x16x32 fp16 MFMA (the engine's shape since round 4);
 * 4 = the same packing on the small-launch tiles (conv_m16.hip: what such a layer runs on when the batch is too small for
 * 512-pixel tiles: same bits as 3); 5 / 6 = as 3 / 4 on the 32x32x16 fp16 MFMA (conv_ht's other instantiation / conv_igemm's
 * 32-channel-chunk tiles: the partner of the MFMA-shape A/B of round 4, not used by the engine; 5 and 6 agree bit for bit with
 * each other, and with 3 / 4 to the last fp16 bit of a few outputs); 7 = as 3 on conv_ht's PERSISTENT form (one workgroup per CU walks
 * its tiles with the next tile's operands in flight; what a launch of two or more tiles per CU runs on; 3 is one workgroup per tile
 * whatever the size; same bits; channel-tile counts other than 1, 2, 4 have no persistent form and run as 3); 8 = as 2 with conv_gemm8's
 * PERSISTENT form wherever it exists (2 is one workgroup per tile whatever the size; same bits).  *kernel_name (optional) receives the family that ran.
 * TEST HOOKS: the hmv_op_* entries with a kernel_sel argument switch PROCESS-GLOBAL kernel-selection state for the duration of the
 * call.  They are single-threaded test / probe entries: never call one concurrently with any other hmv_* call of the process
 * (an hmv_forward running on another thread would see the forced selection). */
int hmv_op_conv2d_f16(int32_t device, const float *in, int32_t N, int32_t H, int32_t W, int32_t Cin,
                      const float *weight_oihw_host, const float *bias_host, int32_t Cout, int32_t R, int32_t S, int32_t stride,
                      int32_t pad, const float *residual, int32_t relu, void *out_f16, int32_t kernel_sel,
                      const char **kernel_name, void *stream);

/* The up-sampling terms of an HRNet fuse layer (hrnet.py:194-212) through hr_fuse.hip alone (op-level parity tests):
 *   out = act(base + sum_s Upsample_{2^shift_s, nearest}(W_s x_s + b_s)),   terms added in the order given (1 <= nsrc <= 3, 1 <= shift <= 3).
 * base [N][H][W][C] and src[s] [N][H >> shift_s][W >> shift_s][src_c[s]]: device fp32 NHWC; w_host[s] [C][src_c[s]] and bias_host[s] [C] on
 * the host.  f16 != 0: the fp16 instantiation on fp16 copies of base / src, `out` receives fp16 rows; else fp32 rows.  HMV_ERR_ARG for
 * shapes without a fused form (C % 4 -- fp16: 8 --, src_c % 16, map sizes that are not exact multiples of 2^shift). */
int hmv_op_hr_fuse_up(int32_t device, int32_t f16, const float *base, int32_t N, int32_t H, int32_t W, int32_t C, int32_t nsrc,
                      const float *const *src, const int32_t *src_c, const int32_t *shift, const float *const *weight_host,
                      const float *const *bias_host, int32_t relu, void *out, void *stream);

/* One multi-head attention of the fusion transformer (layers.py:216-221; 8 heads x 128) through the engine's kernel
 * (op-level parity tests).  qkv device [B][T][3 * 1024] = [q | k | v] per token; queries are tokens [0, Tq), keys / values
 * tokens [koff, koff + Tk); out device [B][Tq][1024]. */
int hmv_op_attention(int32_t device, const float *qkv, int32_t B, int32_t T, int32_t Tq, int32_t koff, int32_t Tk, float *out,
                     void *stream);
/* The same attention as the fp16-kernel modes (HMV_F16, HMV_F32X3) run it: q, k, v, P as fp16 (hi, lo) pairs on the fp16 matrix cores,
 * three products per step, fp32 accumulation and softmax -- fp32-equivalent (2^-22 per operand).  Same arguments; the fp32 rows are split into
 * pairs first (in those modes the projection GEMMs write pairs themselves); synchronises the stream. */
int hmv_op_attention_x3(int32_t device, const float *qkv, int32_t B, int32_t T, int32_t Tq, int32_t koff, int32_t Tk, float *out,
                        void *stream);

/* The attention of the learnable-query fusion (MultiHeadAttentionLearnableQuery, layers.py:284-291; 8 heads x 256) through the
 * engine's kernel: q rows at q + (b * q_bstride + i) * q_ld (q_bstride = 0: the same probe queries for every sample), k / v rows
 * at k + (b * T + j) * kv_ld, j < T; softmax(q k^T / 16) v per (sample, head); out device [B][Tq][2048].  HMV_LQ_SCALAR_ATT=1 in
 * the environment selects the non-MFMA kernel (A/B). */
int hmv_op_attention_lq(int32_t device, const float *q, int32_t q_ld, int32_t q_bstride, const float *k, const float *v, int32_t kv_ld,
                        int32_t B, int32_t T, int32_t Tq, float *out, void *stream);

/* Diagnostic micro-benchmark: average milliseconds of `iters` launches of one NHWC conv shape on
 * pseudo-random data.  tile: -1 = the engine's own choice, else 0..7 = 128x32, 128x64, 128x128, 256x128,
 * 128x256, 256x256, 128x128 (k-step 16), 128x256 (k-step 16) (BM x BN).  HMV_BENCH_CLOCK=1 adds in-kernel clock stamps (stderr). */
int hmv_bench_conv(int32_t device, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t R, int32_t S,
                   int32_t stride, int32_t pad, int32_t with_residual, int32_t tile, int32_t iters, float *avg_ms);

/* hipGraph replay (opt-in: hmv_set_graphs(h, 1) or HMV_GRAPHS=1 in the environment; it saves host time per forward,
 * not GPU time -- measured throughput on MI355X is the same as eager launches).
 * A forward whose batch and caller buffers (x / frames, bbox, intrinsic and the three outputs) equal those of an
 * earlier call is captured into a hipGraph on its second occurrence and replayed as ONE launch afterwards
 * (up to 8 buffer sets per handle, least recently used evicted).  Results are bit-identical to the eager path;
 * stage capture and profiling force the eager path.  hmv_graph_stats reports cached graphs / replays so far. */
int hmv_set_graphs(hmv_handle h, int32_t enable);
int hmv_graph_stats(hmv_handle h, int32_t *cached, int64_t *replays);

/* hmv_forward from raw camera frames: the reference prepares every view on DataLoader workers
 * (datasets/ho3d.py:35-40, 136-149: crop_and_pad_image (datasets/utils.py:40-77) -> ToTensor -> Resize((S,S), antialias=True)
 * -> Normalize(mean, std)); here that is one kernel in front of the stem conv and the fp32 NCHW batch never exists.
 * frames: device uint8 [batch*V][frame_h][frame_w][3] (HWC, the channel order the weights were trained on);
 * crop_boxes: device int32 [batch*V][4] = x1,y1,x2,y2 of the (square or not) crop window in frame pixels -- may leave the
 * frame (zeros are read there); an EMPTY window (x2<=x1 or y2<=y1, or wider than 65536 px) yields the reference's black "no visible joint" view;
 * mean/std: host float[3] (ho3d.py:38-39 uses the ImageNet constants).  The window is resized to cfg.height x cfg.width.
 * bbox / intrinsic / outputs / stream exactly as hmv_forward (bbox is normally crop_boxes as fp32, ho3d.py:198). */
int hmv_forward_frames(hmv_handle h, int32_t batch, const uint8_t *frames, int32_t frame_h, int32_t frame_w, const int32_t *crop_boxes,
                       const float *mean, const float *std, const float *bbox, const float *intrinsic, float *joints_crop_img,
                       float *joints_cam, float *heatmap, void *stream);

/* The frame preparation alone (op-level parity tests): out_nhwc4 device fp32 [n_frames][out_h][out_w][4] (4th channel 0). */
int hmv_op_prepare_frames(int32_t device, const uint8_t *frames, int32_t n_frames, int32_t frame_h, int32_t frame_w,
                          const int32_t *crop_boxes, const float *mean, const float *std, int32_t out_h, int32_t out_w, float *out_nhwc4,
                          void *stream);

/* Evaluation metrics of HandMvNet._get_metrics (handmvnet.py:352-368) on the device, replacing
 * PoseMetrics.mpjpe / pa_mpjpe / pck / pck_auc / compute_similarity_transform (models/metrics.py:6-24, 64-176).
 * pred, target: device fp32 [n_sets][n_pts][dim] (dim 2 or 3; units as given -- the caller applies the x1000 the
 * reference applies).  Thresholds = torch.linspace(thr_min, thr_max, steps), 1 <= steps <= 256.
 * procrustes != 0 (dim == 3 only): also the similarity-aligned error; aligned (device [n_sets][n_pts][3] or NULL)
 * receives the aligned predictions.  result: device fp32 [4 + 2*steps] =
 *   { mpjpe, pa_mpjpe (NaN if not requested), auc, norm_auc, pck[steps], thresholds[steps] }.
 * Asynchronous on `stream`; one single-workgroup launch with fixed-order reductions (bit-reproducible). */
int hmv_pose_metrics(int32_t device, const float *pred, const float *target, int32_t n_sets, int32_t n_pts, int32_t dim,
                     float thr_min, float thr_max, int32_t steps, int32_t procrustes, float *aligned, float *result,
                     void *stream);

const char *hmv_version(void);

/* The tile shape the general conv / GEMM kernel's launcher rule picks for M output pixels, Cout channels, reduction length K
 * (e.g. "256x256", "128x32", "256x128,k16,w8"); host logic only, valid until the calling thread's next call. */
const char *hmv_tile_rule(int32_t M, int32_t Cout, int32_t K, int32_t f16, int32_t has_residual);

#ifdef __cplusplus
}
#endif
#endif /* HANDMV_H */
