// Internal launcher interface between engine.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// Development knobs (A/B switches, probes: DESIGN.md section 9) exist only in a -DHMV_DEV_KNOBS build
// (`python -m handmvnet_amd.build --variant dev HMV_DEV_KNOBS` -> build/libhandmv_dev.so, loaded with HMV_LIB=...).  The product
// library reads ONE environment variable, HMV_GRAPHS; tests/test_abi_cpu.py holds its strings to that.
#ifdef HMV_DEV_KNOBS
#define HMV_DEV_ENV(name) getenv(name)
#else
#define HMV_DEV_ENV(name) (static_cast<const char *>(nullptr))
#endif

namespace hmv {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_LEAKY = 3 };

// Implicit-GEMM convolution / plain GEMM on fp32 MFMA.
//   out[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] + res[m'][n] )
// A[m][k] is gathered on the fly from an NHWC tensor: m = (img, ho, wo), k = (chunk, r, s, c % 32).
// A plain row-major GEMM is the special case H = W = R = S = 1, N = M, Cin = K.
struct ConvParams {
    const void *in;     // NHWC [N][H][W][Cin], fp32 or fp16 (in_f16)
    const void *wgt;    // [Cout_pad][Kpad], zero padded, same element type as `in`; K ordered
                        // (c / CH, r, s, c % CH) with CH = 32 (fp32) / 64 (fp16), and (r, s, c) for the stem
    const float *bias;  // [Cout_pad] fp32 (folded BN shift + conv bias), never null
    const void *res;    // residual, row-major [M'][ldr] or null; fp16 if res_f16
    void *out;          // [M''][ldc]; fp16 if out_f16
    int in_f16, res_f16, out_f16;
    int up;             // 1x1 and dense modes (R = S = 1, no padding) only: output pixel (ho, wo) reads input pixel (ho >> up, wo >> up) -- a 1x1 conv
                        // commutes with nearest-neighbour upsampling (HRNet fuse layers); H, W are the INPUT dims
    int fill;           // write columns [Cout, ldc) too (exact zeros): keeps padded channel strides clean
    int N, H, W, Cin;
    int Ho, Wo, Cout;
    int R, S, stride, pad_h, pad_w;
    int K, Kpad;        // K = R*S*Cin; Kpad = K rounded up to 32
    int M;              // N*Ho*Wo
    int ldc, ldr;
    int act;
    int rg_out, rg_in;  // residual row remap: row' = (m / rg_out) * rg_in + m % rg_out (rg_out = 0: identity)
    int scatter;        // output pixel (ho*osy + ooy, wo*osx + oox) in an (Ho*osy) x (Wo*osx) image
    int osy, osx, ooy, oox;
    int mtiles, ntiles;
    const float *zero;  // 256 bytes of zeros (filled in by launch_conv)
    int cpt;            // dense mode: 16-byte vectors per tap (Cin / 4, or Cin / 8 in fp16); filled in by launch_conv
    unsigned cpt_magic, s_magic;   // ceil(2^32 / cpt), ceil(2^32 / S): exact division by multiply-high for k indices < 2^16
    // fp32-equivalent "split" operands on the fp16 kernels (HMV_F32X3): a tensor row is [hi plane | lo plane] of fp16
    // and the reduction walks the virtual channel sequence hi, lo, hi against weights packed [W_hi | W_hi | W_lo]
    int cwrap;          // physical channels per pixel (2 * C): a virtual channel offset >= cwrap wraps back to the hi plane (0 = off)
    int res_split;      // residual rows are [hi | lo] pairs (plane stride ldr / 2): value = hi + lo
    int x3_plane;       // fused split reduction (fp16 kernels, chunked modes, plane % 32 == 0): physical channels per plane.  One
                        // k-step then carries [A_hi | A_lo] and [W_hi | W_lo] of 32 channels and issues hi*hi, lo*hi, hi*lo from ONE
                        // tile load (the cwrap scheme loads three); weights are packed per step as 32 hi values, 32 lo values
    int acc_shift;      // split layers: the packed weights are W * 2^acc_shift (keeps W_lo out of the fp16 subnormals);
    float acc_scale;    //   the epilogue multiplies the accumulator by 2^-acc_shift (filled in by launch_conv)
    int out_split;      // write (hi, lo) pairs (plane stride ldc / 2) instead of one fp16 value
    int ngroup;         // gemm_x3.hip: > 0 = N-tiles per XCD (each XCD owns a column range of the weights, which then stay in ITS L2)
    int m16;            // fp16 1x1 that multiplies on v_mfma_f32_16x16x32_f16 at every batch size (filled in by launch_conv from conv_m16_rule)
    int tall;           // fp16 3x3 whose weights are packed in conv_ht.hip's K order: that kernel or an error, at every batch size
    int rd_cout;        // row-decomposed 3x3 (narrow Cout): the real channel count; Cout is then 3 * rd_cout, R = 3, S = 1
    // second A source of a plain 1x1 conv (Bottleneck conv3 and its block's downsample conv as ONE GEMM over the concatenated
    // reduction [t2 | x] . [W3 ; Wds], resnet.py:124-144): reduction indices >= ksplit read `in2`, an NHWC tensor
    // [N][H2][W2][lda2] sampled at pixel (ho * stride2, wo * stride2); indices < ksplit read `in` at pixel (ho, wo)
    const void *in2;
    int ksplit, H2, W2, lda2, stride2;
    int lda;            // input pixel stride in floats (0 = Cin)
    int ldw;            // weight row stride in floats (0 = Kpad)
    // split-K (plain fp32 1x1 / GEMM launches): `ksl` slices of `kslice` reduction elements run as ksl x the workgroups; slice s
    // reads in + s * kslice, wgt + s * kslice and writes its partial product to out + s * out_slice (floats); the caller adds them
    int ksl, kslice;
    size_t out_slice;
    int phases;            // 4: the sub-pixel phases of a k4 s2 p1 transposed conv as ONE launch (generic epilogue, scatter)
    size_t phase_stride;   // elements between the phases' packed weight blocks
    int burst;          // residual tile by one LDS-DMA burst per wave after the main loop (filled in by launch_conv)
    int prefetch;       // software L2 prefetch of the residual tile / later activation k-steps (filled in by launch_conv)
    int stagger;        // start-up stagger (filled in by launch_conv): the workgroups of the first dispatch round are split into 4
    int stagger_mode;   //   (mode 1: groups by blockIdx / 256, i.e. by dispatch round, for kernels with several workgroups per CU)
    int stagger_blocks; //   phase groups that begin 0 / 1 / 2 / 3 x `stagger` 10-ns ticks late, so that the load / compute / drain
                        //   phases of the 256 CUs do not march in lockstep (speed only: any placement gives the same result)
    unsigned long long *dbg;  // diagnostic builds only: per-block {shader cycles, 100 MHz ticks} of the main loop
    // conv_stream.hip chain: a FOLLOWING 1x1 conv + BN + ReLU over this launch's output pixels (Bottleneck i's conv3 -> Bottleneck
    // i + 1's conv1, resnet.py:124-130) computed from the output tile while it is still in LDS -- the wide tensor is written once and
    // not read again by that conv.  nx_out[m][nx_ldc] = act(sum_k out[m][k] * nx_wgt[n][k] + nx_bias[n]), k < Cout; fp16 rows.
    // Only launches for which conv_stream_chain_ok(p, nx_cout) holds may set these.
    // conv_hs.hip: the launch also applies MaxPool2d(3, stride 2, padding 1) to its (ReLU) output (the stem, resnet.py:218-221): `out` is the
    // pooled map [N][pool_h][pool_w][ldc]; Ho x Wo stay the conv map's dims.  Only launches conv_hs_supported(p) accepts may set it.
    int pool, pool_h, pool_w;
    const void *nx_wgt;
    const float *nx_bias;
    void *nx_out;
    int nx_cout, nx_ldw, nx_ldc, nx_act;
};

enum ConvTile { TILE_128x32 = 0, TILE_128x64 = 1, TILE_128x128 = 2, TILE_256x128 = 3, TILE_128x256 = 4, TILE_256x256 = 5,
                TILE_128x128_K16 = 6, TILE_128x256_K16 = 7, TILE_256x128_K16 = 8, TILE_64x64 = 9,
                TILE_256x128_K16W8 = 10,   // fp32: 8 waves, 64 accumulators per lane, 48 KB of tile buffers -> two workgroups per CU
                TILE_256x256_RING = 11,    // fp16: 256x256, 32-element k-step, four 32 KB stages with three tiles in flight
                TILE_COUNT = 12 };
int conv_tile_bn(ConvTile t);                       // N-tile width of a tile config
const char *conv_tile_name(ConvTile t, int mode);   // mode: 0 taps, 1 1x1/GEMM, 2 dense K (stem, Cin % 32 != 0)
const char *conv_tile_name_f16(ConvTile t, int mode);
ConvTile conv_pick_tile(int M, int Cout, int K, bool f16 = false, bool has_res = false);
ConvTile conv_dense_tile(ConvTile t, bool f16);     // the tile a dense-K (Cin % 32 != 0) launch really uses
bool conv_partial_n(ConvTile t, int Cout);          // launch_conv uses the block-skipping instantiation
// fills mtiles/ntiles and launches
// `name` (optional) receives the kernel family actually launched ("conv_igemm_f32<256x128,dense,skipN>" ...)
hipError_t launch_conv(ConvParams p, ConvTile tile, hipStream_t s, const char **name = nullptr);
// conv_stream.hip: persistent weight-stationary 1x1 convolution (fp16 rows, residual-bearing, short reductions, many pixels);
// launch_conv routes to it when conv_stream_supported(p).  Bit-identical to conv_igemm's result.
bool conv_stream_supported(const ConvParams &p);
// true when launch_conv(p) would run on a conv_stream instantiation that can also compute a following 1x1 conv with nx_cout output
// channels (p as for the launch itself, nx_* fields unset or set)
bool conv_stream_chain_ok(const ConvParams &p, int nx_cout);
void conv_stream_set_mode(int mode);   // -1 launcher's rule, 0 never, 1 whenever the shape has an instantiation (op-level tests)
hipError_t launch_conv_stream(const ConvParams &p, hipStream_t s, const char **name);

// ---- small kernels
hipError_t launch_nchw_to_nhwc4(const float *x, float *out, int N, int H, int W, hipStream_t s);
hipError_t launch_maxpool3s2(const float *in, float *out, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s);
// hm: [N*h*w][ld] (channels-last heat map, 21 valid columns) -> coords [N][21][2] (heat-map px),
// joints_crop_img = coords * image_size / heatmap_size, optional NCHW heat map copy.
hipError_t launch_soft_argmax(const float *hm, int ld, int N, int h, int w, float *coords, float *crop_img,
                              float image_size, float heatmap_size, float *hm_nchw, hipStream_t s);
// gathers the 4 bilinear neighbours of every joint: out [(n*21+j)*4 + t][C] (zeros when out of range)
hipError_t launch_sample_gather(const float *feat, int N, int H, int W, int C, const float *coords, float *out,
                                hipStream_t s, int elem_bytes = 4);
// fp16 path (BASELINE configs[4])
hipError_t launch_nchw_to_nhwc8_f16(const float *x, void *out, int N, int H, int W, hipStream_t s);
// uint8 HWC camera frames + integer crop windows -> normalised NHWC4 fp32 / NHWC8 fp16 stem input (ho3d.py:35-40, 136-149)
// out_mode: 0 = NHWC4 fp32, 1 = NHWC8 fp16, 2 = split [hi8 | lo8] fp16 pairs (HMV_F32X3)
hipError_t launch_frames_to_input(const uint8_t *frames, const int *boxes, int N, int Hf, int Wf, int S_h, int S_w, const float *mean,
                                  const float *std, int out_mode, void *out, hipStream_t s, bool s2d = false);
// HMV_F32X3 helpers: tensors whose rows are [hi plane | lo plane] fp16 pairs
// space-to-depth stem input (misc_kernels.hip): mode 0 fp32 [12], 1 fp16 [16], 2 split [hi16 | lo16] per 2x2 pixel block
hipError_t launch_nchw_to_s2d(const float *x, void *out, int N, int H, int W, int mode, hipStream_t s);
hipError_t launch_nchw_to_nhwc_split(const float *x, void *out, int N, int H, int W, hipStream_t s);
hipError_t launch_maxpool3s2_split(const void *in, void *out, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s);
hipError_t launch_nhwc_split_to_nchw(const void *in, float *out, int N, int H, int W, int C, hipStream_t s);
// fp32 rows -> fp16 rows (mode 1) / split [hi | lo] rows (mode 2)
hipError_t launch_rows_f32_to_half(const float *in, void *out, size_t rows, int C, int mode, hipStream_t s);
hipError_t launch_maxpool3s2_f16(const void *in, void *out, int N, int H, int W, int C, int Ho, int Wo, hipStream_t s);
hipError_t launch_nhwc_f16_to_nchw(const void *in, float *out, int N, int H, int W, int C, hipStream_t s);
// tokens[(n*21+j)][col0 + c] = sum_t w_t * s[(n*21+j)*4+t][c]
hipError_t launch_sample_blend(const float *s4, int lds4, int C, int N, int H, int W, const float *coords,
                               float *tokens, int ldt, int col0, hipStream_t s);
// pos2d / FoV columns, zero padding columns; optional raw copy (before PE) and PE add
hipError_t launch_tokens_finalize(float *tokens, int ldt, int d, int fdim, int N, int V, const float *coords,
                                  const float *bbox, const float *intr, int pos_mask, const float *pe,
                                  float *raw_copy, hipStream_t s, void *pairs = nullptr);   // pairs: rows again as [hi ldt | lo ldt] halfs
// y = LN(x) ; optional second LN applied to y -> y2.  Pads [d, ld) are written as zeros.
hipError_t launch_layernorm(const float *x, int ldx, int rows, int d, const float *g1, const float *b1, float *y,
                            int ldy, const float *g2, const float *b2, float *y2, hipStream_t s);
// softmax(q k^T / sqrt(128)) v per (sample, head); qkv rows are [q | k | v] of width 3*1024.
hipError_t launch_attention(const float *qkv, int B, int T, int Tq, int koff, int Tk, float *out, hipStream_t s, int pairs = 0, int x3 = 0);
// (x3: the fp16-kernel modes' form -- every operand as fp16 (hi, lo) pairs on the fp16 matrix cores, fp32-equivalent: attention_x3_kernel)   // pairs: the rows as (hi, lo) fp16 pairs [hi 1024 | lo 1024] instead of fp32
// MultiHeadAttentionLearnableQuery (layers.py:240-301): softmax(q k^T / sqrt(256)) v per (sample, head), 8 heads x 256.
// q rows: q + (b * q_bstride + i) * q_ld (q_bstride = 0: the same 21 probe queries for every sample); k / v rows:
// k + (b * T + j) * kv_ld, j < T.  out [B*Tq][2048].
hipError_t launch_attention_d256(const float *q, int q_ld, int q_bstride, const float *k, const float *v, int kv_ld, int B, int T,
                                 int Tq, float *out, hipStream_t s, int pairs = 0);
// y[r][c] = x[r][c] + pe[r % T][c] for c < d, 0 for d <= c < ldy (PositionalEncoding inside every learnable-query block)
hipError_t launch_add_pe(const float *x, int ldx, int rows, int T, int d, const float *pe, float *y, int ldy, hipStream_t s);
// split-K GEMM tail: out[r][c] = act(sum_s slab[s][r][c] + bias[c] + res[r'][c]) for c < N (slices summed in index order)
hipError_t launch_splitk_layernorm(const float *slab, int S, int rows, int lds, int d, const float *bias, const float *res, int ldr,
                                   int rg_out, int rg_in, const float *g1, const float *b1, float *y, int ldy, const float *g2,
                                   const float *b2, float *y2, hipStream_t s);
hipError_t launch_splitk_reduce(const float *slab, int S, int rows, int lds, int N, const float *bias, const float *res, int ldr,
                                int rg_out, int rg_in, int act, float *out, int ldc, hipStream_t s);
// Chebyshev mix: out[b][i][o] = act(sum_k sum_j Tk[k][i][j] * y[b*21+j][k*co + o] + bias[o])
hipError_t launch_cheb_mix(const float *y, int ldy, int B, int co, const float *tk, const float *bias, int leaky,
                           float *out, int ldo, hipStream_t s);
// conv_gemm8.hip: the fp16 256 x 256 1x1 tile (no residual; optional second source) on the counted-vmcnt, phase-interleaved
// main loop; launch_conv routes to it when conv_gemm8_supported(p).  Bit-identical to conv_igemm's result.
bool conv_gemm8_supported(const ConvParams &p);
void conv_gemm8_set_mode(int mode);   // -1 launcher's rule, 0 never, 1 whenever supported (op-level tests)
hipError_t launch_conv_gemm8(ConvParams p, hipStream_t s, const char **name);
void conv_gemm8_set_persistent(int on);   // 1 (default): the persistent form from two tiles per CU up; 0: one workgroup per tile; 2: wherever it exists (tests)

// conv_hs.hip: persistent weight-stationary R x S convolution for few-channel fp16 layers (3x3 64 -> 64; the 4x4 space-to-depth
// stem): weights in registers, one halo image per 16 x 16 output block.  Bit-identical to conv_igemm's result.
bool conv_hs_supported(const ConvParams &p);
void conv_hs_set_mode(int mode);
hipError_t launch_conv_hs(const ConvParams &p, hipStream_t s, const char **name);
// conv_rds.hip: fp32 row-decomposed 3x3 convs (HRNet-w40's 40- / 80-channel branches) on the persistent weight-stationary
// structure; bit-identical to conv_igemm's row-decomposed tiles
bool conv_rds_supported(const ConvParams &p);
void conv_rds_set_mode(int mode);   // -1 launch_conv's rule, 0 never, 1 whenever supported (op-level tests)
hipError_t launch_conv_rds(const ConvParams &p, hipStream_t s, const char **name);
// conv_ht.hip: fp16 3x3 stride-1 convs on tall 512-pixel x 128-channel tiles (K order (32-channel chunk, r, s, c % 32)); the
// shape rule is asked at weight-packing time, so a layer it takes runs there at every batch size
bool conv_ht_shape_ok(int R, int S, int stride, int pad, int Cin, int Cout, int H, int W);
hipError_t launch_conv_ht(ConvParams p, hipStream_t s, const char **name);
void conv_ht_set_mode(int mode);   // -1 launch_conv's rule, 0 never (the c32 tiles of conv_igemm.hip), 1 always: op-level tests
int conv_ht_mode();
void conv_ht_set_shape(int m16);   // MFMA shape of the tall-tile layers: 1 = 16x16x32 (the engine's, round 4), 0 = 32x32x16 (the A/B partner)
int conv_ht_shape();
void conv_ht_set_persistent(int on);   // 1 (default): the persistent form from two tiles per CU up; 0: one workgroup per tile; 2: wherever it exists (tests)
// conv_m16.hip: the small-launch companion of the 16x16x32-MFMA kernels (64 x 64 / 128 x 128 tiles; 3x3 in conv_ht's K order, plain
// 1x1): same bits as conv_ht<..., m16>, so a layer's result does not depend on which of the two its batch size selects
bool conv_m16_supported(const ConvParams &p);
// the plain 1x1 layers that take the 16x16x32 MFMA: a rule on the layer's SHAPE and epilogue only (never on M), so that the large
// launches (conv_gemm8<..., m16>) and the small ones (conv_m16's tiles) of one layer agree bit for bit
bool conv_m16_rule(const ConvParams &p);
void conv_m16_set_rule(int on);   // 1 (default) / 0: op-level A/B against the 32x32x16 kernels
hipError_t launch_conv_m16(ConvParams p, hipStream_t s, const char **name);

// gemm_x3.hip: token GEMMs over (hi, lo) fp16 pairs with fp32 output rows (Loader::linear_x3's packing), at every size
bool gemm_x3_rule(const ConvParams &p);
hipError_t launch_gemm_x3(ConvParams p, hipStream_t s, const char **name);
// ... and the short-reduction ones (q / k / v projections) at large row counts: 256 x 256 tiles on a deep ring, bit-identical to
// conv_igemm's fused split loop (so chosen by size)
bool gemm_x3k16_ok(const ConvParams &p);
void gemm_x3k16_set_mode(int mode);   // -1 the size rule, 0 never, 1 whenever the shape allows (op-level tests)
hipError_t launch_gemm_x3k16(ConvParams p, hipStream_t s, const char **name);

// ---- fusion_kernels.hip: the launch-bound tail as fused kernels
// Everything of a fusion block behind its to_out GEMM (layers.py:224-233 / 161-174; learnable-query blocks: layers.py:293-299):
//   t = (slab: sum of S split-K slices + bias0 + residual row | x row);  n1 = n1g ? LN(t) : t;  f0 = LN_ff(n1);
//   h = GELU(f0 W1^T + b1);  f2 = h W2^T + b2 + n1;  out = n2g ? LN(f2) : f2   (pad columns [d, ldo) written as zeros)
struct FfBlockParams {
    const float *slab; int S; size_t slice; int lds; const float *bias0; const float *res; int ldr, rg_out, rg_in;
    const float *x; int ldx;
    int rows, d, ld;                       // token rows, feature width, padded width (= K of W1, multiple of 16)
    const float *n1g, *n1b, *fg, *fb;
    const float *w1, *b1; int ldw1;        // [hid (+pad)][ldw1 >= ld]
    const float *w2, *b2; int ldw2;        // [>= ld rows][ldw2 >= hid]
    const float *n2g, *n2b;
    float *out; int ldo;
    void *out_pairs;                       // optional: the output rows once more as (hi, lo) fp16 pairs [row][hi ldo | lo ldo] (the next
                                           // block's q/k/v projection reads them in the fp16-kernel modes: rows_f32_to_half's arithmetic)
    int hid;                               // 128 | 256
    unsigned long long *dbg;               // diagnostic builds only: 100 MHz stamps at the phase boundaries of workgroup 0
};
hipError_t launch_ff_block(const FfBlockParams &p, hipStream_t s);
// hr_fuse.hip: the up-sampling terms of an HRNet fuse layer (hrnet.py:194-212) as one launch:
//   out = act(((base + G_0) + G_1) + G_2),  G_s = W_s x_s + b_s at x_s's resolution, read at (y >> shift_s, x >> shift_s)
// base / out / x_s: NHWC rows of fp32 or fp16 (f16); w_s fp32 [>= C rows][ldw], bias_s fp32; sources in the reference's order (ascending j)
struct HrFuseSrc { const void *x; const float *w; const float *bias; int C, ld, ldw, shift, H, W, a_off, g_off; unsigned rcpu; };
struct HrFuseParams {
    const void *base; void *out;
    int N, H, W, C, ldc, nsrc, relu, f16;
    HrFuseSrc src[3];
    int th, lds, ldg; unsigned rcpu;        // filled by hr_fuse_up_plan: tile height, LDS bytes, G row stride, ceil(2^32 / 16-byte units per pixel)
    int nitems[4]; unsigned char items[4][12];   // per wave: its items (source << 4 | 16-channel block), longest first
};
bool hr_fuse_up_plan(HrFuseParams &p);      // false: no fused form for this shape (run the per-term launches)
hipError_t launch_hr_fuse_up(const HrFuseParams &p, hipStream_t s);
// JointsDecoderGCN (nets.py:133-139): three ChebConv layers K -> c1 -> c2 -> c3 in two launches (layer 1 per (sample, 16
// channels); layers 2 + 3 per sample).  w_i: the packed [3 * c_i (+pad)][ld] matrices of the unfused GEMMs (row k * c_i + o).
struct ChebFusedParams {
    const float *x; int ldx; int B, K;
    const float *w1; int ldw1; int c1; const float *bias1;
    const float *w2; int ldw2; int c2; const float *bias2;
    const float *w3; int ldw3; int c3; const float *bias3;
    const float *tk;                       // [3][21][21]
    float *scratch;                        // [B*21][c1]
    float *out; int ldo;
};
hipError_t launch_cheb_fused(const ChebFusedParams &p, hipStream_t s);
bool cheb_fusable(int K, int ldx, int ldw1, int c1, int ldw2, int c2, int ldw3, int c3);   // the shapes launch_cheb_fused takes

// NHWC -> NCHW copy (stage capture)
hipError_t launch_nhwc_to_nchw(const float *in, float *out, int N, int H, int W, int C, hipStream_t s, int ld = 0);
hipError_t launch_copy_rows(const float *in, int ldi, float *out, int ldo, int rows, int cols, hipStream_t s);

}  // namespace hmv
