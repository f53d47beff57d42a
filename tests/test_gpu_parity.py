"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, called through the
C ABI (handmvnet_amd.HandMvNet -> ctypes -> libhandmv.so), against
  (1) the CPU oracle on the same seeded inputs,
  (2) the committed golden fixtures (outputs of the REAL reference),
  (3) size-independent properties at BASELINE.json's full size (B=32, V=8, 256x256).

Tolerance (north_star): joints_cam within 1e-3 rel-L2 of the reference in fp32.  The
reference itself sits ~8e-5 from an fp64 evaluation, so dense stage tensors
are held to 2e-4 and heat-map coordinates to 0.05 heat-map px.
"""
import ctypes

import json
import os

import numpy as np
import pytest
import torch

from cases import CASES
from helpers import check_against_fixture, cond_bounds, load_case, rel_l2

pytestmark = pytest.mark.gpu

TOL_CAM = 1e-3
TOL_STAGE = 2e-4


def _model(name):
    from handmvnet_amd import HandMvNet
    cfg, (tp, mp, dp), sd, inputs, fx = load_case(name)
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd, strict=True)
    m.to("cuda").eval()
    m.freeze()
    return m, cfg, sd, inputs, fx


def _run(m, x, bbox, intr, stages=True):
    dev = torch.device("cuda:0")
    m.capture_stages(stages)
    out = m(torch.from_numpy(x).to(dev), torch.from_numpy(bbox).to(dev), {"intrinsic": torch.from_numpy(intr).to(dev)})
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items()}
    if stages:
        for nm in ("feat0", "coords_hm", "tokens", "fused"):
            res[nm] = m.read_stage(nm).cpu().numpy()
        torch.cuda.synchronize()
    return res


def test_extension_is_loaded_and_has_no_fallback():
    from handmvnet_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.hmv_version()
    m, *_ = _model("tiny_r50")
    with pytest.raises(_lib.HandMvError, match="no CPU fallback"):
        m(torch.zeros(1, 2, 3, 64, 64))


@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_reference_fixture(name):
    m, cfg, sd, (x, bbox, intr), fx = _model(name)
    got = _run(m, x, bbox, intr)
    rep = check_against_fixture(got, fx, tol_cam=TOL_CAM, tol_coord_px=0.05 * cfg.image_size / cfg.heatmap_size,
                                tol_stage=TOL_STAGE)
    assert np.abs(got["coords_hm"] - fx["coords_hm"]).max() < 0.05, rep
    print(name, rep)


@pytest.mark.parametrize("name", ["tiny_r50", "tiny_r18", "cfg1_r50_v4_128", "cfg2s_r18_v4_256", "cfg3s_r50_v8_256",
                                  "r50_wocam_nn", "r34_onelevel"])
def test_forward_matches_oracle(name):
    from oracle.oracle import Oracle
    m, cfg, sd, (x, bbox, intr), fx = _model(name)
    got = _run(m, x, bbox, intr)
    ref = Oracle(cfg, sd, "f64").forward(x, bbox, intr, stages=True)
    rep = {k: rel_l2(got[k], ref[k]) for k in ("joints_cam", "heatmap", "feat0", "tokens", "fused")}
    rep["coords"] = float(np.abs(got["coords_hm"] - ref["coords_hm"]).max())
    print(name, rep)
    assert rep["joints_cam"] <= TOL_CAM, rep
    assert rep["heatmap"] <= TOL_STAGE and rep["feat0"] <= TOL_STAGE, rep
    assert rep["tokens"] <= TOL_STAGE and rep["fused"] <= TOL_STAGE, rep
    assert rep["coords"] < 0.05, rep
    assert got["joints_crop_img"].shape == ref["joints_crop_img"].shape
    assert np.abs(got["joints_crop_img"] - ref["joints_crop_img"]).max() < 0.05 * cfg.image_size / cfg.heatmap_size


@pytest.mark.parametrize("name,mode", [("tiny_r18", "f32"), ("tiny_r50", "f32"), ("hr40_tiny", "f32"), ("tiny_r50", "f32x3"),
                                       ("hr40_tiny", "f32x3")])
def test_non_square_frames_match_oracle(name, mode):
    """H != W (the reference takes any frame size; every fixture is square): 64 x 96 and 96 x 64 frames against the oracle."""
    from oracle.oracle import Oracle
    from handmvnet_amd.synth import normalish
    m, cfg, sd, (x, bbox, intr), fx = _model(name)
    if mode == "f32x3":
        m.float32x3()
    orc = Oracle(cfg, sd, "f64")
    for hh, ww in ((64, 96), (96, 64)):
        b, v = x.shape[:2]
        xs = normalish("input.nonsquare", 23 + hh, b * v * 3 * hh * ww).astype(np.float32).reshape(b, v, 3, hh, ww)
        got = _run(m, xs, bbox, intr)
        ref = orc.forward(xs, bbox, intr, stages=True)
        assert got["heatmap"].shape == ref["heatmap"].shape == (b, v, 21, hh // 8, ww // 8)
        rep = {k: rel_l2(got[k], ref[k]) for k in ("joints_cam", "heatmap", "feat0", "tokens", "fused")}
        rep["coords"] = float(np.abs(got["coords_hm"] - ref["coords_hm"]).max())
        print(name, mode, (hh, ww), rep)
        assert rep["joints_cam"] <= TOL_CAM and rep["coords"] < 0.05, rep
        assert max(rep["heatmap"], rep["feat0"], rep["tokens"], rep["fused"]) <= TOL_STAGE, rep


POISON_CASES = ["r50_lq", "r18_lq_wocam", "tiny_r50", "r18_frozen_nosin", "hr40_tiny", "r50_odd_96"]


@pytest.mark.parametrize("name", [n for n in POISON_CASES if n in CASES])
@pytest.mark.parametrize("mode", ["f32", "f16", "f32x3"])
def test_poisoned_workspace(name, mode):
    """No stage may read workspace bytes that an earlier stage of the same forward did not write (zero weights do not help:
    0 * NaN = NaN).  The arena is filled with 0xFF bytes (NaN patterns in fp32 and fp16) between two forwards: the second one
    must return the bits of the first.  Covers the pad columns [d, ldt) of the token matrices of both fusion modules."""
    m, cfg, sd, (x, bbox, intr), _ = _model(name)
    if mode == "f16":
        m.half()
    elif mode == "f32x3":
        m.float32x3()
    a = _run(m, x, bbox, intr, stages=False)
    m.poison_workspace(0xFF)
    b = _run(m, x, bbox, intr, stages=False)
    for k in ("joints_cam", "joints_crop_img", "heatmap"):
        assert np.isfinite(b[k]).all(), (name, mode, k)
        assert np.array_equal(a[k], b[k]), (name, mode, k, float(np.abs(a[k] - b[k]).max()))


FUSED_TAIL_CASES = ["tiny_r50", "cfg2s_r18_v4_256", "r50_lq", "r18_lq_wocam", "r18_13views", "hr40_tiny", "r50_wocam_nn"]


@pytest.mark.parametrize("name", [n for n in FUSED_TAIL_CASES if n in CASES])
def test_fused_tail_kernels_match_the_unfused_launches(name):
    """fusion_kernels.hip (FeedForward + LayerNorms behind the split-K to_out GEMM as ONE launch; the three ChebConv layers as
    two) against the launch-per-op path it replaces (hmv_set_tail_fusion(h, 0)): same arithmetic up to
    the GEMMs' summation order, so the two agree far inside the parity tolerance -- and the fused path is the one the
    reference fixtures are checked against (test_forward_matches_reference_fixture).  Fewer launches, too."""
    m, cfg, sd, (x, bbox, intr), _ = _model(name)
    fused = _run(m, x, bbox, intr)
    n_fused = m.launch_count()
    m.set_tail_fusion(False)
    plain = _run(m, x, bbox, intr)
    n_plain = m.launch_count()
    rep = {k: rel_l2(fused[k], plain[k]) for k in ("fused", "joints_cam")}
    print(name, rep, n_fused, n_plain)
    assert rep["fused"] <= 2e-5 and rep["joints_cam"] <= 2e-5, rep
    assert np.array_equal(fused["tokens"], plain["tokens"])
    assert n_fused < n_plain, (n_fused, n_plain)


TAIL_CASES = ["hr40_lq", "r50_lq", "r18_lq_wocam", "cfg3s_r50_v8_256", "tiny_r18", "hr40_tiny", "r50_wocam_nn", "r18_13views",
              "r18_single_view", "r18_frozen_nosin"]
TOL_TAIL_FUSED, TOL_TAIL_CAM = 5e-5, 2e-4


def _set_mode(m, mode):
    if mode == "f16":
        m.half()
    elif mode == "f32x3":
        m.float32x3()


@pytest.mark.parametrize("mode", ["f32", "f16", "f32x3"])
@pytest.mark.parametrize("name", TAIL_CASES)
def test_fusion_tail_on_engine_tokens(name, mode):
    """Implementation error apart from conditioning: the engine's OWN captured token matrix goes through the f64
    oracle's fusion + decoder (oracle.fuse_tokens), and the engine's `fused` / joints_cam must match THAT -- at a quarter of the
    stage bar / a fifth of the north-star bar (the tail alone, no backbone noise amplified through it), or, for fixtures of
    ill-conditioned configurations, at 4 x the distance of the reference's own fp32 run from its float64 evaluation (hr40_lq:
    3.0e-4 / 7.1e-4 where worst-case amplification x token error would allow 1e-2 .. 1).  In every arithmetic mode the tail is
    fp32 (the fp16 path's q/k/v projections run as (hi, lo) pairs), so the fp16 path's learnable-query tail is pinned here even where
    its end-to-end joints_cam floor is vacuous (fp16_noise.json: hr40_lq 1.1).  For the (hi, lo) modes of an ill-conditioned fixture the bar
    follows the conditioning AT THE ENGINE'S OWN TOKENS (below)."""
    from oracle.oracle import Oracle
    m, cfg, sd, (x, bbox, intr), fx = _model(name)
    _set_mode(m, mode)
    got = _run(m, x, bbox, intr)
    o64 = Oracle(cfg, sd, "f64")
    ref = o64.fuse_tokens(got["tokens"])
    tol_cam, tol_fused = cond_bounds(fx, TOL_TAIL_CAM, TOL_TAIL_FUSED)
    if "amp_fused" in fx and mode != "f32":
        # The fixture's conditioning figures were measured at the REFERENCE's tokens; the fp16 path's tokens are ~1e-3 away from them,
        # and the un-normalised learnable-query blocks respond to their input very unevenly (logits of 1e4: measured in float64 at two
        # token sets 2.4e-3 apart, a 1e-6 relative perturbation moves `fused` by 3.4e-5 at one and by 2.0e-2 at the other).  The (hi, lo)
        # tail carries its operands to 2^-22, so its bar is taken at the operating point: twice what a 2^-19 relative perturbation of
        # THESE tokens does to the float64 tail (tight where the point is benign -- 7e-5 --, honest where it is not).
        rng = np.random.default_rng(1234)
        tk = got["tokens"].astype(np.float64)
        moved = o64.fuse_tokens((tk * (1.0 + 2.0 ** -19 * rng.standard_normal(tk.shape))).astype(np.float32))
        tol_fused = max(tol_fused, 2.0 * rel_l2(moved["fused"], ref["fused"]))
        tol_cam = max(tol_cam, 2.0 * rel_l2(moved["joints_cam"], ref["joints_cam"]))
    rep = {"fused": rel_l2(got["fused"], ref["fused"]), "joints_cam": rel_l2(got["joints_cam"], ref["joints_cam"]),
           "bounds": (tol_fused, tol_cam)}
    print(name, mode, rep)
    assert rep["fused"] <= tol_fused and rep["joints_cam"] <= tol_cam, rep


@pytest.mark.parametrize("mode", ["f32", "f32x3", "f16"])
def test_hrnet_release_shape(mode):
    """The *_HR release configs' shape (configs/release/*_HR*.yaml; hrnet.py:372-407): HRNet-w40, levels [40, 80, 160, 320],
    V = 8, 256 x 256, B = 9 -> 72 frames, above every size gate of the engine (conv_rds_f32, conv_hs<40|80>, the 256 x 192 tile,
    the >= 65 536-pixel tile rules, the layer1 chain), which the 128 x 128 HRNet fixtures never reach.  Sample 0 against the
    fixture of the REAL reference (hr40_v8_256: the same frames, synth_inputs is a counter hash) and against the f64 oracle at
    the mode's bar; samples 0, 4, 8 bit-equal to their single-sample runs; everything finite."""
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.synth import synth_inputs
    from oracle.oracle import Oracle
    cfg, (tp, mp, dp), sd, (x1, bbox1, intr1), fx = load_case("hr40_v8_256")
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd, strict=True)
    m.to("cuda").eval()
    _set_mode(m, mode)
    spec = CASES["hr40_v8_256"]
    x, bbox, intr = synth_inputs(cfg, 9, spec["iseed"], 256)
    assert np.array_equal(x[:1], x1) and np.array_equal(bbox[:1], bbox1) and np.array_equal(intr[:1], intr1)
    got = _run(m, x, bbox, intr)
    for k in ("joints_cam", "joints_crop_img", "heatmap", "feat0", "tokens", "fused"):
        assert np.isfinite(got[k]).all(), k
    assert got["heatmap"].shape == (9, 8, 21, 32, 32) and got["feat0"].shape[0] == 72
    for i in (0, 4, 8):
        one = _run(m, x[i:i + 1], bbox[i:i + 1], intr[i:i + 1])
        for k in ("joints_cam", "joints_crop_img", "heatmap"):
            assert np.array_equal(one[k][0], got[k][i]), (mode, i, k)
        if i == 0:
            first = one
    # sample 0 vs the real reference's fixture (stages of the B = 1 run have the fixture's shapes) ...
    ref = Oracle(cfg, sd, "f64").forward(x[:1], bbox[:1], intr[:1], stages=True)
    if mode == "f16":
        bound = fp16_bounds("hr40_v8_256")
        hm = rel_l2(first["heatmap"].reshape(-1)[fx["heatmap_idx"]], fx["heatmap_val"])
        feat = rel_l2(first["feat0"].reshape(-1)[fx["feat0_idx"]], fx["feat0_val"])
        dc = np.abs(first["coords_hm"] - fx["coords_hm"])
        cam = rel_l2(first["joints_cam"], fx["joints_cam"])
        rep = {"feat0": feat, "heatmap": hm, "coord_median_px": float(np.median(dc)), "coord_flip_frac": float((dc > 0.5).mean()),
               "joints_cam": cam, "bounds": bound}
        print(mode, rep)
        assert feat <= 2e-3 and hm <= bound["heatmap"], rep
        assert rep["coord_median_px"] <= 0.02 and rep["coord_flip_frac"] <= bound["flip"], rep
        assert cam <= bound["joints_cam"], rep
    else:
        rep = check_against_fixture(first, fx, tol_cam=TOL_CAM, tol_coord_px=0.05 * cfg.image_size / cfg.heatmap_size, tol_stage=TOL_STAGE)
        rep["coords"] = float(np.abs(first["coords_hm"] - fx["coords_hm"]).max())
        # ... and vs the f64 oracle (whole tensors, not samples of them)
        rep["oracle"] = {k: rel_l2(first[k], ref[k]) for k in ("joints_cam", "heatmap", "feat0", "tokens", "fused")}
        print(mode, rep)
        assert rep["coords"] < 0.05, rep
        assert rep["oracle"]["joints_cam"] <= TOL_CAM and max(rep["oracle"][k] for k in ("heatmap", "feat0", "tokens", "fused")) <= TOL_STAGE, rep


# (backbone_type, channels, V, B, size): frame sizes that do and do not tile into the size-gated kernels' blocks (conv_ht 16 x 32,
# conv_hs 16 x 16 / 8 x 16, conv_stream's pixel tiles, conv_rds' 64-pixel rows), at batches above their gates
SIZE_ROWS = [("w40", [40, 80, 160, 320], 2, 8, 320), ("50_paper", [1024], 4, 20, 192), ("50_paper", [1024], 4, 16, 320),
             ("50_paper", [1024], 2, 6, 512)]


@pytest.mark.parametrize("row", SIZE_ROWS, ids=lambda r: f"{r[0]}-V{r[2]}-B{r[3]}-{r[4]}")
def test_size_gated_kernels_at_other_frame_sizes(row):
    """tools/size_smoke.py's rows as assertions: fp32 and fp16 forwards at frame sizes other than 256 x 256 with
    batches large enough for the persistent / tall-tile kernels -- finite, a sample alone == the same sample inside the batch bit
    for bit in both modes (an engine-level stride or packing slip at these sizes breaks exactly that), and the fp16 pose within
    the fp16-storage noise level of the fp32 one (0.05 .. 0.13 measured in round 3; a layout slip is O(1))."""
    from cases import case_params
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.spec import config_from_params
    from handmvnet_amd.synth import synth_inputs, synth_state_dict
    bt, ch, V, B, size = row
    spec = dict(bt=bt, ch=ch, V=V, B=B, size=size, pos=["pos2d", "crop", "sin"], gcn=True, wseed=3, iseed=11)
    tp, mp, dp = case_params(spec)
    cfg = config_from_params(tp, mp, dp)
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(synth_state_dict(cfg, 3), strict=True)
    m.to("cuda").eval()
    x, bbox, intr = synth_inputs(cfg, B, 11, size)
    res = {}
    for mode in ("f32", "f16"):
        _set_mode(m, mode)
        out = _run(m, x, bbox, intr, stages=False)
        last = B - 1
        one = _run(m, x[last:], bbox[last:], intr[last:], stages=False)
        for k in ("joints_cam", "joints_crop_img", "heatmap"):
            assert np.isfinite(out[k]).all(), (mode, k)
            assert np.array_equal(one[k][0], out[k][last]), (mode, k)
        res[mode] = out["joints_cam"]
    rel = rel_l2(res["f16"], res["f32"])
    print(row, "fp16 vs fp32 joints_cam rel-L2", rel)
    assert rel < 0.3, rel


CONV_SHAPES = [
    # N, H, W, Cin, Cout, k, stride, pad, residual, relu
    (2, 16, 16, 64, 64, 1, 1, 0, False, True),
    (2, 16, 16, 64, 256, 1, 1, 0, True, True),
    (1, 16, 16, 256, 128, 1, 2, 0, False, False),
    (2, 12, 20, 64, 64, 3, 1, 1, False, True),
    (1, 16, 16, 128, 128, 3, 2, 1, False, True),
    (1, 8, 8, 512, 21, 1, 1, 0, False, False),
    (3, 32, 32, 4, 64, 7, 2, 3, False, True),       # the stem: NHWC4 frames
    (1, 9, 7, 32, 160, 3, 1, 1, True, False),        # ragged M and N tails
    # dense K order over the real channels (Cin % 32 != 0: HRNet's 40 / 80-channel tensors)
    (2, 16, 16, 40, 40, 3, 1, 1, True, True),
    (1, 16, 16, 80, 80, 3, 1, 1, False, True),
    (2, 16, 16, 40, 80, 3, 2, 1, False, False),
    (1, 16, 16, 80, 40, 1, 1, 0, False, False),
    (1, 9, 7, 12, 20, 3, 1, 1, True, True),           # 3 vectors per tap, ragged everything
    (32, 64, 64, 80, 80, 3, 1, 1, True, True),        # 256x128 dense, last 32-column block skipped
    # block skipping in the chunked modes (Cout = 160 / 320 under 128- and 256-wide tiles)
    (8, 64, 64, 160, 160, 3, 1, 1, False, True),      # 128x128 taps
    (32, 32, 32, 96, 160, 1, 1, 0, True, False),      # 128x128 1x1
    (16, 64, 64, 64, 320, 1, 1, 0, False, True),      # 256x256 1x1
    (32, 64, 64, 32, 160, 3, 1, 1, True, True),       # 256x256 taps
]


def _conv_case(shape):
    N, H, W, Cin, Cout, k, stride, pad, use_res, relu = shape
    g = torch.Generator().manual_seed(sum(shape[:8]))
    x = torch.randn(N, Cin, H, W, generator=g)
    if Cin == 4:
        x[:, 3] = 0
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    res = torch.randn(N, Cout, *ref.shape[2:], generator=g) if use_res else None
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)
    return x, w, b, res, ref


def _run_conv(shape, dtype):
    from handmvnet_amd import _lib
    lib = _lib.load()
    N, H, W, Cin, Cout, k, stride, pad, use_res, relu = shape
    x, w, b, res, ref = _conv_case(shape)
    Ho, Wo = ref.shape[2:]
    dev = torch.device("cuda:0")
    xin = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.full((N, Ho, Wo, Cout), float("nan"), device=dev)
    rdev = res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None
    wc, bc = w.contiguous().numpy(), b.contiguous().numpy()
    rc = lib.hmv_op_conv2d_ex(0, dtype, xin.data_ptr(), N, H, W, Cin, wc.ctypes.data_as(ctypes.c_void_p),
                              bc.ctypes.data_as(ctypes.c_void_p), Cout, k, k, stride, pad,
                              rdev.data_ptr() if use_res else None, int(relu), out.data_ptr(), None)
    assert rc == 0, lib.hmv_last_error(None)
    got = out.cpu().permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv_kernel_vs_torch(shape):
    """op-level: one NHWC implicit-GEMM conv (native fp32 MFMA) vs torch fp64 conv2d on the CPU."""
    assert _run_conv(shape, 0) < 2e-6


HALF_SHAPES = [sh for sh in CONV_SHAPES if sh[3] % 8 == 0 and sh[4] % 4 == 0]


@pytest.mark.parametrize("shape", HALF_SHAPES)
def test_conv_kernel_split_precision_vs_torch(shape):
    """The same op with (hi, lo) fp16 pairs and three fp16 MFMAs per product (HMV_F32X3): held to fp32-grade error.
    (The inputs and the residual are themselves rounded to hi + lo, 2^-22 relative; the fp32 kernel's bound is 2e-6.)"""
    assert _run_conv(shape, 2) < 3e-6


# plain GEMMs over split rows with fp32 output rows and a long reduction (gemm_x3.hip: to_out of the fusion blocks in the fp16-kernel
# modes, layers.py:224: K = 1 024, 524 columns -- no multiple of the tile; the learnable-query blocks' K = 2 048): both tile sizes,
# ragged row counts; and the shorter reductions that stay on conv_igemm's fused split loop (the q / k / v projections, K = 544)
X3_SHAPES = [(3, 21, 8, 1024, 524, 1, 1, 0, False, False), (40, 21, 8, 1024, 524, 1, 1, 0, False, False), (1, 1, 33, 1024, 128, 1, 1, 0, False, False),
             (2, 16, 16, 2048, 312, 1, 1, 0, False, False), (1, 7, 3, 1024, 64, 1, 1, 0, False, False), (3, 21, 8, 544, 3072, 1, 1, 0, False, False)]


@pytest.mark.parametrize("shape", X3_SHAPES)
def test_split_pair_gemm_vs_torch(shape):
    """gemm_x3.hip against torch fp64: fp32-grade error (hi + lo carries 22 bits, the dropped lo * lo term is 2^-22 relative)."""
    assert _run_conv(shape, 2) < 3e-6


# q / k / v projection shapes (K = d pairs, 3 x 1 024 columns): whole and ragged 256-row tiles, one and several N-tiles, an odd step count
X3K16_SHAPES = [(2, 21, 8, 544, 3072, 1, 1, 0, False, False), (1, 1, 300, 256, 768, 1, 1, 0, False, False), (3, 9, 7, 96, 256, 1, 1, 0, False, False),
                (16, 21, 16, 544, 3072, 1, 1, 0, False, False)]


@pytest.mark.parametrize("shape", X3K16_SHAPES)
def test_split_pair_gemm_large_tiles_are_bit_identical(shape):
    """gemm_x3k16_f16 (256 x 256 tiles, k-steps of 16 channels on a four-stage ring: what the projections run on at large row counts)
    against conv_igemm's fused split loop (what they run on otherwise): the same per-element sequence of 32x32x16 MFMAs, the same
    bits -- which is what lets the launcher choose by size -- and torch fp64 at fp32-grade error."""
    from handmvnet_amd import _lib
    lib = _lib.load()
    N, H, W, Cin, Cout, k, stride, pad, use_res, relu = shape
    x, w, b, res, ref = _conv_case(shape)
    dev = torch.device("cuda:0")
    xin = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wc, bc = w.contiguous().numpy(), b.contiguous().numpy()
    outs = []
    try:
        for mode in (1, 0):
            assert lib.hmv_set_x3k16_mode(mode) == 0
            out = torch.full((N, H, W, Cout), float("nan"), device=dev)
            rc = lib.hmv_op_conv2d_ex(0, 2, xin.data_ptr(), N, H, W, Cin, wc.ctypes.data_as(ctypes.c_void_p), bc.ctypes.data_as(ctypes.c_void_p), Cout, 1, 1, 1, 0,
                                      None, 0, out.data_ptr(), None)
            assert rc == 0, lib.hmv_last_error(None)
            outs.append(out.cpu())
    finally:
        lib.hmv_set_x3k16_mode(-1)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), (outs[0] - outs[1]).abs().max()
    got = outs[0].permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 3e-6


@pytest.mark.parametrize("shape", HALF_SHAPES)
def test_conv_kernel_fp16_vs_torch(shape):
    """... and with plain fp16 operands, fp32 accumulation: fp16-grade error."""
    err = _run_conv(shape, 1)
    assert 1e-6 < err < 2e-3, err


# 3x3 stride-1 convs large enough for the fp16 256x256 tile: 16 x 16 pixel blocks with the halo image in LDS (MODE_HALO);
# one, two and three 64-channel chunks (the halo buffers alternate), a non-square map, image borders on every side
HALO_SHAPES = [(32, 64, 64, 64, 256, 3, 1, 1, False, True), (32, 64, 64, 128, 256, 3, 1, 1, False, False),
               (16, 32, 128, 192, 512, 3, 1, 1, False, True), (64, 32, 32, 128, 128, 3, 1, 1, False, True)]   # the last: 256x128 tile


@pytest.mark.parametrize("shape", HALO_SHAPES)
def test_conv_halo_tiles_fp16_vs_torch(shape):
    err = _run_conv(shape, 1)
    assert 1e-6 < err < 2e-3, err


# 1x1 residual convs with the short reductions of Bottleneck conv3 (resnet.py:137-144), the shapes conv_stream.hip instantiates:
# (K, channel slice) = (256, 512), (128, 512), (64, 256); pixel counts that are no multiple of the 64- / 128-pixel tiles, streams
# with zero, one and several tiles, one to four channel slices
STREAM_SHAPES = [(4, 32, 32, 256, 1024, 1, 1, 0, True, True), (3, 17, 19, 256, 512, 1, 1, 0, True, False),
                 (64, 32, 32, 256, 1024, 1, 1, 0, True, True), (8, 32, 32, 128, 512, 1, 1, 0, True, True),
                 (1, 33, 31, 128, 1024, 1, 1, 0, True, True), (40, 32, 32, 128, 512, 1, 1, 0, True, False),
                 (4, 64, 64, 64, 256, 1, 1, 0, True, True), (2, 30, 35, 64, 512, 1, 1, 0, True, False),
                 (40, 64, 64, 64, 256, 1, 1, 0, True, True),
                 # without a residual: the squeezing conv1 shapes (K, Cout) = (256, 64), (512, 128), (256, 128)
                 (4, 64, 64, 256, 64, 1, 1, 0, False, True), (3, 17, 19, 256, 64, 1, 1, 0, False, False), (40, 64, 64, 256, 64, 1, 1, 0, False, True),
                 (8, 32, 32, 512, 128, 1, 1, 0, False, True), (1, 33, 31, 512, 128, 1, 1, 0, False, True), (12, 64, 64, 256, 128, 1, 1, 0, False, True),
                 # layer1.0's conv1 (64 -> 64)
                 (4, 64, 64, 64, 64, 1, 1, 0, False, True), (3, 17, 19, 64, 64, 1, 1, 0, False, False), (40, 64, 64, 64, 64, 1, 1, 0, False, True)]


# 1x1 convs without a residual on the phase-interleaved 256 x 256 tile (conv_gemm8.hip): K from 2 to 16 k-steps, ragged pixel and
# channel tails, with and without ReLU
GEMM8_SHAPES = [(4, 32, 32, 1024, 256, 1, 1, 0, False, True), (3, 17, 19, 512, 512, 1, 1, 0, False, False),
                (2, 32, 32, 128, 320, 1, 1, 0, False, True), (1, 40, 40, 192, 256, 1, 1, 0, False, True),
                (8, 32, 32, 1024, 512, 1, 1, 0, False, True)]


# 3x3 64 -> 64 convs on the halo-streaming weight-stationary kernel (conv_hs.hip): images of 1 x 1 to 4 x 4 blocks of 16 x 16 pixels,
# non-square, more blocks than workgroups' first round and fewer, with and without a residual / ReLU
HS_SHAPES = [(4, 32, 32, 64, 64, 3, 1, 1, False, True), (3, 48, 16, 64, 64, 3, 1, 1, True, True), (1, 16, 16, 64, 64, 3, 1, 1, False, False),
             (40, 64, 64, 64, 64, 3, 1, 1, False, True), (24, 64, 32, 64, 64, 3, 1, 1, True, False),
             # 40 -> 40 channels (HRNet-w40's highest-resolution branch): 80-byte pixels, two taps inside one MFMA step
             (4, 32, 32, 40, 40, 3, 1, 1, False, True), (3, 48, 16, 40, 40, 3, 1, 1, True, True), (40, 64, 64, 40, 40, 3, 1, 1, True, True),
             (1, 16, 16, 40, 40, 3, 1, 1, True, False),
             # 80 -> 80 channels (w40's second branch): 160-byte pixels, 8 x 16 blocks, three waves with the weights of 32 channels each
             (8, 32, 32, 80, 80, 3, 1, 1, False, True), (3, 24, 16, 80, 80, 3, 1, 1, True, True), (40, 32, 32, 80, 80, 3, 1, 1, True, True),
             (1, 8, 16, 80, 80, 3, 1, 1, True, False)]


def _run_conv_f16(shape, sel):
    from handmvnet_amd import _lib
    lib = _lib.load()
    N, H, W, Cin, Cout, k, stride, pad, use_res, relu = shape
    g = torch.Generator().manual_seed(sum(shape[:8]))
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(N, H, W, Cout, generator=g) if use_res else None
    dev = torch.device("cuda:0")
    xin = x.to(dev)
    rdev = res.to(dev) if use_res else None
    out = torch.full((N, H, W, Cout), float("nan"), device=dev, dtype=torch.float16)
    wc, bc = w.contiguous().numpy(), b.contiguous().numpy()
    kname = ctypes.c_char_p()
    rc = lib.hmv_op_conv2d_f16(0, xin.data_ptr(), N, H, W, Cin, wc.ctypes.data_as(ctypes.c_void_p), bc.ctypes.data_as(ctypes.c_void_p),
                               Cout, k, k, stride, pad, rdev.data_ptr() if use_res else None, int(relu), out.data_ptr(), sel,
                               ctypes.byref(kname), None)
    assert rc == 0, lib.hmv_last_error(None)
    return out.cpu(), kname.value.decode(), (x, w, b, res, relu)


@pytest.mark.parametrize("shape", STREAM_SHAPES + GEMM8_SHAPES + HS_SHAPES)
def test_stream_kernel_is_bit_identical(shape):
    """The persistent weight-stationary kernel against conv_igemm on the same operands: same bits (it keeps conv_igemm's operand
    roles, accumulation order and epilogue arithmetic -- which is what makes a sample's result independent of the batch whichever
    kernel the launcher picks), and both against torch fp64 at fp16 accuracy."""
    a, ka, (x, w, b, res, relu) = _run_conv_f16(shape, 2)
    c, kc, _ = _run_conv_f16(shape, 1)
    want = "conv_hs_f16" if shape[5] == 3 else ("conv_stream_f16" if shape in STREAM_SHAPES else "conv_gemm8_f16<256x256,1x1,m16>")
    # (the MFMA-heavy 1x1 layers without a residual multiply on the 16x16x32 MFMA at every size since round 4: their small-launch
    # partner is conv_m16.hip's tiles, not conv_igemm's)
    assert ka.startswith(want) and kc.startswith("conv_m16_f16" if shape in GEMM8_SHAPES else "conv_igemm_f16"), (ka, kc)
    assert torch.isfinite(a.float()).all()
    assert torch.equal(a.view(torch.int16), c.view(torch.int16)), (ka, kc, (a.float() - c.float()).abs().max())
    if shape[0] * shape[1] * shape[2] <= 8192:   # fp64 reference on the CPU for the small cases
        xh, wh = x.half().double(), w.half().double()
        ref = torch.nn.functional.conv2d(xh.permute(0, 3, 1, 2), wh, b.double(), stride=shape[6], padding=shape[7]).permute(0, 2, 3, 1)
        if res is not None:
            ref = ref + res.half().double()
        if relu:
            ref = ref.clamp_min(0)
        err = (a.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-3, err


def test_gemm8_refuses_a_padded_reduction():
    """conv_gemm8's loop walks Kpad / 64 k-steps with no column mask, so a reduction that is not a whole number of 64-channel
    steps (Cin = 160 -> Kpad = 192) must NOT reach it -- nor conv_m16's tiles (whole 32-channel steps of real channels only): with
    kernel_sel = 2 ("the special kernels wherever the shape has one") such a conv falls back to conv_igemm, and the result is right."""
    shape = (2, 16, 16, 160, 256, 1, 1, 0, False, True)
    a, ka, (x, w, b, res, relu) = _run_conv_f16(shape, 2)
    assert ka.startswith("conv_igemm_f16"), ka
    ref = torch.nn.functional.conv2d(x.half().double().permute(0, 3, 1, 2), w.half().double(), b.double()).permute(0, 2, 3, 1).clamp_min(0)
    assert torch.isfinite(a.float()).all()
    assert (a.double() - ref).abs().max().item() / ref.abs().max().item() < 1e-3


# 3x3 convs on the tall 512-pixel x 128-channel tiles (conv_ht.hip): 1 to 6 blocks per image, one and several N-tiles, 2 to 8
# sub-chunks, ReLU on and off, more tiles than CUs
HT_SHAPES = [(2, 16, 32, 64, 128, 3, 1, 1, False, True), (1, 32, 32, 128, 128, 3, 1, 1, False, False), (3, 48, 64, 256, 256, 3, 1, 1, False, True),
             (5, 32, 32, 256, 384, 3, 1, 1, False, True), (72, 32, 32, 128, 256, 3, 1, 1, False, True)]
# ... and launches of two and more tiles per CU (the persistent form's own size: 4-5 and 2-3 tiles per workgroup, ragged per XCD)
HTP_SHAPES = HT_SHAPES + [(67, 32, 64, 64, 512, 3, 1, 1, False, True), (150, 32, 32, 128, 256, 3, 1, 1, False, False)]


@pytest.mark.parametrize("shape", HT_SHAPES)
def test_tall_tile_kernel(shape):
    """conv_ht.hip walks the reduction in 32-channel chunks (its halo images are 32 channels deep) and, since round 4, multiplies on
    v_mfma_f32_16x16x32_f16 (one MFMA per tap and sub-chunk: 12-15 % less time than 32x32x16, profiles/r04_probe_mfma_shape.txt);
    conv_igemm's fp16 kernels walk 64-channel chunks on 32x32x16: same products, another summation order, so the two agree to fp32
    accumulation noise under the fp16 output rounding, not bit for bit.  The engine therefore decides at weight-packing time, from the
    layer's shape and map size alone, which order a layer uses; a layer packed for conv_ht runs on conv_ht when the batch fills the
    chip with its tiles and on conv_m16.hip's 64 x 64 / 128 x 128 tiles (same MFMA, same order) when it does not -- and THOSE two must
    agree bit for bit, or a sample's result would depend on its batch.  Checked: conv_ht == conv_m16 bitwise; against the 64-chunk
    kernel (at most the last fp16 bit, rarely); against torch fp64 at fp16 accuracy; image 0 alone == image 0 inside the batch."""
    a, ka, (x, w, b, res, relu) = _run_conv_f16(shape, 3)
    e, ke, _ = _run_conv_f16(shape, 4)
    c, kc, _ = _run_conv_f16(shape, 1)
    assert ka == "conv_ht_f16<512x128,3x3,m16>" and ke.startswith("conv_m16_f16") and ke.endswith("taps,c32>") and kc.startswith("conv_igemm_f16"), (ka, ke, kc)
    assert torch.equal(a.view(torch.int16), e.view(torch.int16)), (ka, ke, (a.float() - e.float()).abs().max())
    assert torch.isfinite(a.float()).all()
    d = (a.float() - c.float()).abs()
    scale = c.float().abs().max().item()
    assert d.max().item() <= 2e-3 * scale, (d.max().item(), scale)           # one fp16 ulp near the largest value
    assert (d > 0).float().mean().item() < 0.02                              # and only where the fp32 sums straddle a rounding boundary
    if shape[0] * shape[1] * shape[2] <= 8192:
        xh, wh = x.half().double(), w.half().double()
        ref = torch.nn.functional.conv2d(xh.permute(0, 3, 1, 2), wh, b.double(), stride=1, padding=1).permute(0, 2, 3, 1)
        if relu:
            ref = ref.clamp_min(0)
        err = (a.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-3, err
    if shape[0] > 1:
        one = (1,) + shape[1:]
        # same seed stream: _run_conv_f16 seeds by the shape, so rebuild image 0 from the batch's own tensors
        from handmvnet_amd import _lib
        lib = _lib.load()
        dev = torch.device("cuda:0")
        x0 = x[:1].contiguous().to(dev)
        out = torch.full((1,) + tuple(a.shape[1:]), float("nan"), device=dev, dtype=torch.float16)
        wc, bc = w.contiguous().numpy(), b.contiguous().numpy()
        rc = lib.hmv_op_conv2d_f16(0, x0.data_ptr(), 1, one[1], one[2], one[3], wc.ctypes.data_as(ctypes.c_void_p),
                                   bc.ctypes.data_as(ctypes.c_void_p), one[4], 3, 3, 1, 1, None, int(relu), out.data_ptr(), 4, None, None)   # on the small-launch tiles
        assert rc == 0, lib.hmv_last_error(None)
        assert torch.equal(out.cpu()[0].view(torch.int16), a[0].view(torch.int16))


# ... conv_gemm8's: the op-level shapes (idle workgroups, ragged pixel tails, one and two channel tiles) and launches of 2-5 tiles per CU
G8P_SHAPES = GEMM8_SHAPES + [(136, 32, 32, 512, 512, 1, 1, 0, False, True), (131, 17, 19, 1024, 256, 1, 1, 0, False, False),
                             (70, 32, 32, 256, 1024, 1, 1, 0, False, True)]


@pytest.mark.parametrize("shape", G8P_SHAPES)
def test_gemm8_persistent_form(shape):
    """conv_gemm8's persistent form (round 4: one workgroup per CU walks its tiles; the half-tiles a tile's last two k-steps used to
    request as dummies are the next tile's first seven) against one workgroup per tile: same bits at every size, so the launcher picks
    between them by tile count.  kernel_sel 8 forces the persistent form on small launches too (channel-tile counts other than
    1, 2, 4 have none and run as kernel_sel 2)."""
    a, ka, _ = _run_conv_f16(shape, 2)
    e, ke, _ = _run_conv_f16(shape, 8)
    assert ka == "conv_gemm8_f16<256x256,1x1,m16>", ka
    assert ke == ("conv_gemm8_f16<256x256,1x1,m16,persistent>" if (shape[4] + 255) // 256 in (1, 2, 4) else ka), ke
    assert torch.isfinite(a.float()).all()
    assert torch.equal(a.view(torch.int16), e.view(torch.int16)), (a.float() - e.float()).abs().max()


@pytest.mark.parametrize("shape", HTP_SHAPES)
def test_tall_tile_kernel_persistent_form(shape):
    """conv_ht's persistent form (round 4: one workgroup per CU walks its tiles, the next tile's weight stages and first halo image
    in flight under the last taps and the epilogue of this one) against one workgroup per tile: same MFMA sequence per output, so the
    same bits at every size -- the launcher may pick between them by tile count (>= 2 per CU) without a sample's result depending on
    its batch.  kernel_sel 7 forces the persistent form on small launches too (idle workgroups, one tile per workgroup, ragged XCDs)."""
    a, ka, _ = _run_conv_f16(shape, 3)
    e, ke, _ = _run_conv_f16(shape, 7)
    assert ka == "conv_ht_f16<512x128,3x3,m16>", ka
    assert ke == ("conv_ht_f16<512x128,3x3,m16,persistent>" if shape[4] // 128 in (1, 2, 4) else ka), ke
    assert torch.isfinite(a.float()).all()
    assert torch.equal(a.view(torch.int16), e.view(torch.int16)), (a.float() - e.float()).abs().max()


# shapes that give conv_m16 its 128 x 128 tiles (>= 256 of them; a tall layer's M is always a multiple of 512, so there is no ragged M here)
M16_SHAPES = [(40, 32, 32, 128, 128, 3, 1, 1, False, True), (33, 16, 32, 64, 256, 3, 1, 1, False, False)]


@pytest.mark.parametrize("shape", M16_SHAPES)
def test_small_launch_tiles_of_the_tall_layers(shape):
    """conv_m16.hip's two tile sizes against each other's partner: kernel_sel 4 picks 128 x 128 tiles for these shapes (>= 256 tiles),
    conv_ht (sel 3) must give the same bits, torch fp64 agrees on three images."""
    e, ke, (x, w, b, res, relu) = _run_conv_f16(shape, 4)
    a, ka, _ = _run_conv_f16(shape, 3)
    assert ke == "conv_m16_f16<128x128,taps,c32>" and ka == "conv_ht_f16<512x128,3x3,m16>", (ke, ka)
    assert torch.isfinite(e.float()).all()
    assert torch.equal(a.view(torch.int16), e.view(torch.int16))
    for i in (0, shape[0] // 2, shape[0] - 1):
        ref = torch.nn.functional.conv2d(x[i:i + 1].half().double().permute(0, 3, 1, 2), w.half().double(), b.double(), stride=1, padding=1).permute(0, 2, 3, 1)
        if relu:
            ref = ref.clamp_min(0)
        assert (e[i:i + 1].double() - ref).abs().max().item() / ref.abs().max().item() < 1e-3


@pytest.mark.parametrize("shape", HT_SHAPES)
def test_tall_tile_kernel_mfma_16x16x32(shape):
    """The two MFMA shapes of conv_ht against each other (the A/B of round 4, profiles/r04_probe_mfma_shape.txt): v_mfma_f32_16x16x32_f16
    (the engine's: one MFMA per (tap, 32-channel sub-chunk) and 16 x 16 block, its own LDS swizzle and accumulator -> channel map)
    and 32x32x16 (kernel_sel 5 / 6, kept as the partner).  Against torch fp64 at fp16 accuracy, and against each other: the same
    products summed inside other MFMA instructions, so equal up to the last fp16 bit of a few outputs -- never more."""
    a, ka, (x, w, b, res, relu) = _run_conv_f16(shape, 3)
    c, kc, _ = _run_conv_f16(shape, 5)
    e, ke, _ = _run_conv_f16(shape, 6)
    assert ka == "conv_ht_f16<512x128,3x3,m16>" and kc == "conv_ht_f16<512x128,3x3>" and ke.startswith("conv_igemm_f16") and ke.endswith("c32>"), (ka, kc, ke)
    assert torch.equal(c.view(torch.int16), e.view(torch.int16))     # the 32x32x16 pair agrees bit for bit with itself, as in round 3
    assert torch.isfinite(a.float()).all()
    d = (a.float() - c.float()).abs()
    scale = c.float().abs().max().item()
    assert d.max().item() <= 2e-3 * scale, (d.max().item(), scale)
    assert (d > 0).float().mean().item() < 0.02
    if shape[0] * shape[1] * shape[2] <= 8192:
        xh, wh = x.half().double(), w.half().double()
        ref = torch.nn.functional.conv2d(xh.permute(0, 3, 1, 2), wh, b.double(), stride=1, padding=1).permute(0, 2, 3, 1)
        if relu:
            ref = ref.clamp_min(0)
        err = (a.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-3, err


# fp32 residual 1x1 convs on the persistent weight-stationary kernel (conv_stream_f32): K = 64 / 128 / 256, one to four channel slices,
# ragged pixel tails, with and without ReLU, more tiles than workgroups
STREAM32_SHAPES = [(4, 32, 32, 256, 1024, 1, 1, 0, True, True), (3, 17, 19, 128, 512, 1, 1, 0, True, True), (2, 32, 32, 64, 256, 1, 1, 0, True, False),
                   (1, 8, 8, 256, 256, 1, 1, 0, True, True), (20, 32, 32, 256, 1024, 1, 1, 0, True, True), (40, 32, 32, 128, 512, 1, 1, 0, True, True),
                   # the squeezing conv1 256 -> 64 without a residual (128 x 64 tiles)
                   (2, 32, 32, 256, 64, 1, 1, 0, False, True), (3, 17, 19, 256, 64, 1, 1, 0, False, False), (48, 32, 32, 256, 64, 1, 1, 0, False, True),
                   (3, 32, 32, 64, 64, 1, 1, 0, False, True), (40, 32, 32, 64, 64, 1, 1, 0, False, False)]


@pytest.mark.parametrize("shape", STREAM32_SHAPES)
def test_stream32_kernel_is_bit_identical(shape):
    """conv_stream_f32 against conv_igemm's fp32 kernels on the same operands: same bits (the k order and pairing of the 32x32x2 MFMA
    steps, the zero-initialised accumulator and the (acc + bias) + residual epilogue are conv_igemm's), and against torch fp64."""
    from handmvnet_amd import _lib
    lib = _lib.load()
    N, H, W, Cin, Cout, k, stride, pad, use_res, relu = shape
    g = torch.Generator().manual_seed(sum(shape[:8]))
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(N, H, W, Cout, generator=g)
    dev = torch.device("cuda:0")
    xin, rdev = x.to(dev), res.to(dev)
    wc, bc = w.contiguous().numpy(), b.contiguous().numpy()
    outs, names = [], []
    for sel in (2, 1):
        out = torch.full((N, H, W, Cout), float("nan"), device=dev)
        kname = ctypes.c_char_p()
        rc = lib.hmv_op_conv2d_sel(0, xin.data_ptr(), N, H, W, Cin, wc.ctypes.data_as(ctypes.c_void_p), bc.ctypes.data_as(ctypes.c_void_p),
                                   Cout, k, k, stride, pad, rdev.data_ptr() if use_res else None, int(relu), out.data_ptr(), sel, ctypes.byref(kname), None)
        assert rc == 0, lib.hmv_last_error(None)
        outs.append(out.cpu())
        names.append(kname.value.decode())
    assert names[0].startswith("conv_stream_f32") and names[1].startswith("conv_igemm_f32"), names
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), (names, (outs[0] - outs[1]).abs().max())
    if N * H * W <= 8192:
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double()).permute(0, 2, 3, 1)
        if use_res:
            ref = ref + res.double()
        if relu:
            ref = ref.clamp_min(0)
        err = (outs[0].double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 4e-6, err


# row-decomposed fp32 3x3 convs on the persistent kernel (conv_rds.hip): (images, H, W, C, residual, relu); one and two image rows per
# 64-pixel tile, fewer tiles than workgroups and more, a ragged last round
RDS_SHAPES = [(2, 64, 64, 40, True, True), (1, 64, 64, 40, False, False), (3, 32, 32, 80, True, True), (5, 32, 32, 80, False, True),
              (24, 64, 64, 40, True, True), (70, 32, 32, 80, True, False), (3, 16, 16, 40, True, True)]


@pytest.mark.parametrize("shape", RDS_SHAPES)
def test_rds_kernel_is_bit_identical(shape):
    """conv_rds_f32 against conv_igemm's row-decomposed tiles on the same packed operands: same bits (MFMA pairing and order, the
    epilogue's (((bias + G_0) + G_1) + G_2) + residual), and against torch fp64."""
    from handmvnet_amd import _lib
    lib = _lib.load()
    N, H, W, C, use_res, relu = shape
    g = torch.Generator().manual_seed(sum(shape[:4]) + 7)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(C, generator=g)
    res = torch.randn(N, H, W, C, generator=g)
    dev = torch.device("cuda:0")
    xin, rdev = x.to(dev), res.to(dev)
    wc, bc = w.contiguous().numpy(), b.contiguous().numpy()
    outs, names = [], []
    for sel in (2, 1):
        out = torch.full((N, H, W, C), float("nan"), device=dev)
        kname = ctypes.c_char_p()
        rc = lib.hmv_op_conv2d_rd(0, xin.data_ptr(), N, H, W, C, wc.ctypes.data_as(ctypes.c_void_p), bc.ctypes.data_as(ctypes.c_void_p),
                                  rdev.data_ptr() if use_res else None, int(relu), out.data_ptr(), sel, ctypes.byref(kname), None)
        assert rc == 0, lib.hmv_last_error(None)
        outs.append(out.cpu())
        names.append(kname.value.decode())
    assert names[0].startswith("conv_rds_f32") and names[1].startswith("conv_igemm_f32") and "rowsum" in names[1], names
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), (names, (outs[0] - outs[1]).abs().max())
    if N * H * W <= 16384:
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
        if use_res:
            ref = ref + res.double()
        if relu:
            ref = ref.clamp_min(0)
        err = (outs[0].double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-5, err


@pytest.mark.parametrize("shape", [(256, 16, 16, 160, 160, 3, 1, 1, True, True), (256, 16, 16, 160, 160, 3, 1, 1, False, False),
                                   (300, 16, 16, 80, 136, 3, 1, 1, False, True)])
def test_wide_n_tile_192(shape):
    """HRNet-w40's 160-channel branch at bench size (hrnet.py:96-221): 129 .. 192 output channels run on ONE 256 x 192 tile per
    256 pixels instead of two 128-wide ones.  Checked against torch fp64 on three of the images (the whole batch in fp64 on the
    CPU would take minutes) and for finiteness everywhere."""
    a, ka, (x, w, b, res, relu) = _run_conv_f16(shape, 0)
    assert ka == "conv_igemm_f16<256x192,dense>", ka
    assert torch.isfinite(a.float()).all()
    for i in (0, shape[0] // 2, shape[0] - 1):
        ref = torch.nn.functional.conv2d(x[i:i + 1].half().double().permute(0, 3, 1, 2), w.half().double(), b.double(), stride=1,
                                         padding=1).permute(0, 2, 3, 1)
        if res is not None:
            ref = ref + res[i:i + 1].half().double()
        if relu:
            ref = ref.clamp_min(0)
        err = (a[i:i + 1].double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-3, (i, err)


def _random_conv_shapes(n, seed):
    rng = np.random.default_rng(seed)
    shapes = []
    while len(shapes) < n:
        cin = int(rng.choice([4, 8, 12, 16, 24, 32, 40, 64, 80, 96, 160, 256]))
        cout = int(rng.choice([4, 8, 20, 32, 40, 64, 80, 96, 128, 160, 200, 256, 320]))
        k = int(rng.choice([1, 1, 3, 3, 5]))
        stride = int(rng.choice([1, 1, 2]))
        pad = int(rng.choice([0, k // 2]))
        nimg, hh, ww = int(rng.integers(1, 4)), int(rng.integers(k, 34)), int(rng.integers(k, 34))
        shapes.append((nimg, hh, ww, cin, cout, k, stride, pad, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    return shapes


@pytest.mark.parametrize("dtype,tol", [(0, 4e-6), (2, 5e-6), (1, 3e-3)])   # reductions up to K = 6400: sqrt(K) growth
def test_conv_kernel_random_shapes(dtype, tol):
    """48 seeded random shapes per arithmetic mode (ragged M / N / K tails, dense and chunked K orders, every tile rule the
    small sizes reach) against torch fp64."""
    worst = 0.0
    for shape in _random_conv_shapes(48, seed=1234):
        if dtype != 0 and shape[3] % 8 != 0:
            continue
        err = _run_conv(shape, dtype)
        assert err < tol, (shape, dtype, err)
        worst = max(worst, err)
    print("worst", dtype, worst)


@pytest.mark.parametrize("mode", ["f32", "f32x3", "f16"])
def test_full_size_properties(mode):
    """BASELINE.json configs[2] / configs[4] size (B=32, V=8, 256x256, r50-paper) in every arithmetic mode (their tile rules
    differ): determinism and batch-independence (sample i alone == sample i inside the batch, bit for bit); and sample 0
    against the f64 CPU oracle at the mode's own bar."""
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.synth import synth_inputs
    from oracle.oracle import Oracle
    cfg, (tp, mp, dp), sd, _, _ = load_case("cfg3s_r50_v8_256")
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd)
    if mode == "f16":
        m.half()
    elif mode == "f32x3":
        m.float32x3()
    x, bbox, intr = synth_inputs(cfg, 32, 99, 256)
    dev = torch.device("cuda:0")
    xt, bt, it = torch.from_numpy(x).to(dev), torch.from_numpy(bbox).to(dev), torch.from_numpy(intr).to(dev)
    a = m(xt, bt, {"intrinsic": it})
    b = m(xt, bt, {"intrinsic": it})
    torch.cuda.synchronize()
    assert torch.equal(a["joints_cam"], b["joints_cam"]) and torch.equal(a["joints_crop_img"], b["joints_crop_img"])
    assert torch.isfinite(a["joints_cam"]).all() and torch.isfinite(a["heatmap"]).all()
    assert a["joints_cam"].shape == (32, 21, 3) and a["heatmap"].shape == (32, 8, 21, 32, 32)
    for i in (0, 17, 31):
        one = m(xt[i:i + 1], bt[i:i + 1], {"intrinsic": it[i:i + 1]})
        torch.cuda.synchronize()
        assert torch.equal(one["joints_cam"][0], a["joints_cam"][i])
        assert torch.equal(one["joints_crop_img"][0], a["joints_crop_img"][i])
    # soft-argmax coordinates live inside the heat map
    hs = 32
    assert (a["joints_crop_img"] >= -1e-3).all() and (a["joints_crop_img"] <= (hs - 1) * 8 + 1e-3).all()
    # one sample of the full-size batch against the f64 oracle (~5 s of CPU): the fp32-grade modes at the north-star bar,
    # the fp16 path at its noise-floor bound (same model as the cfg3s fixture)
    ref = Oracle(cfg, sd, "f64").forward(x[:1], bbox[:1], intr[:1])
    cam = rel_l2(a["joints_cam"][0].cpu().numpy(), ref["joints_cam"][0])
    hm = rel_l2(a["heatmap"][0].cpu().numpy(), ref["heatmap"][0])
    dc = np.abs(a["joints_crop_img"][0].cpu().numpy() - ref["joints_crop_img"][0]) / 8.0       # heat-map px
    rep = {"joints_cam": cam, "heatmap": hm, "coord_median_px": float(np.median(dc)), "coord_flip_frac": float((dc > 0.5).mean())}
    print(mode, rep)
    if mode == "f16":      # the fp16 path at the fp16-storage floor of this model (fixture cfg3s_r50_v8_256), heat map and coordinates too
        bound = fp16_bounds("cfg3s_r50_v8_256")
        assert cam <= bound["joints_cam"] and hm <= bound["heatmap"], (rep, bound)
        assert rep["coord_median_px"] <= 0.02 and rep["coord_flip_frac"] <= bound["flip"], (rep, bound)
    else:
        assert cam <= TOL_CAM and hm <= TOL_STAGE and dc.max() < 0.05, (mode, rep)


@pytest.mark.parametrize("case,frames_per_sample,fewer,size,samples", [("cfg3s_r50_v8_256", 8, 4, 256, (4, 5)), ("hr40_v4_128", 4, 3, 256, (8, 9)),
                                                                       # 128 x 128 frames: 32 x 32 pooled maps = 5 x 5 ragged pooled blocks, partial last pixel tiles;
                                                                       # 96 x 96: only the pooled stem is large enough (24 x 24 pooled = 4 x 4 ragged blocks)
                                                                       ("cfg3s_r50_v8_256", 8, 4, 128, (16, 17)), ("cfg3s_r50_v8_256", 8, 1, 96, (15,))])
def test_chained_launches_give_the_bits_of_one_launch_per_conv(case, frames_per_sample, fewer, size, samples):
    """Cross-layer launches of the fp16 backbone at large batches.  conv_stream.hip "chain": a layer1 Bottleneck's conv3 launch also
    computes the NEXT block's conv1 from its output tile in LDS (resnet.py:124-144 / 128-130; three launches fewer, the 256-channel
    tensor read once less per block).  conv_hs.hip "+maxpool": the stem conv's epilogue applies the 3x3 / 2 max pool to its block in
    LDS (resnet.py:218-221; the 64-channel conv map never reaches HBM).  Same operand roles, k order and epilogue arithmetic as the
    launches they absorb: every output and the layer3 feature map must equal the one-launch-per-op forward's BIT FOR BIT (32 frames
    of 256 x 256 = the smallest batch the chain takes), a ragged batch (40 / 36 frames) too, and the poisoned-workspace rule must still
    hold.  HRNet-w40's layer1 (four Bottlenecks, hrnet.py:96-140) takes the chain too (no pooled stem there: three launches fewer)."""
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.synth import synth_inputs
    cfg, (tp, mp, dp), sd, _, _ = load_case(case)
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd)
    m.half()
    dev = torch.device("cuda:0")
    for nb in samples:
        x, bbox, intr = synth_inputs(cfg, nb, 7 + nb, size)
        m.set_chain_fusion(True)
        chained = _run(m, x, bbox, intr)
        n_chained = m.launch_count()
        m.poison_workspace(0xFF)
        again = _run(m, x, bbox, intr)
        m.set_chain_fusion(False)
        plain = _run(m, x, bbox, intr)
        n_plain = m.launch_count()
        assert n_plain - n_chained == fewer, (nb, n_plain, n_chained)
        for k in ("feat0", "heatmap", "tokens", "joints_cam", "joints_crop_img"):
            assert np.isfinite(chained[k]).all(), k
            assert np.array_equal(chained[k], plain[k]), (nb, k, float(np.abs(chained[k] - plain[k]).max()))
            assert np.array_equal(chained[k], again[k]), (nb, k)
    m.set_chain_fusion(True)


# hr_fuse.hip alone: (N, H, W, C, [(C_s, shift_s), ...], relu) -- whole and ragged 16 x 32 tiles, one to three sources, HRNet-w40's and
# w64's channel counts (40 = 2.5 sixteen-channel blocks; 128 / 512-channel sources take the 8-row tiles in fp32), maps smaller than a tile
HRF_SHAPES = [(2, 16, 32, 40, [(80, 1), (160, 2), (320, 3)], True), (3, 24, 40, 40, [(80, 1), (160, 2), (320, 3)], True),
              (1, 8, 8, 80, [(160, 1), (320, 2)], True), (2, 16, 16, 64, [(128, 1), (256, 2), (512, 3)], True),
              (1, 8, 16, 128, [(256, 1), (512, 2)], False), (2, 48, 64, 48, [(96, 1)], True), (5, 32, 32, 40, [(80, 1), (160, 2)], False)]


@pytest.mark.parametrize("shape", HRF_SHAPES)
@pytest.mark.parametrize("f16", [0, 1])
def test_hr_fuse_up_vs_torch(shape, f16):
    """The fused up-sampling terms of an HRNet fuse layer (hrnet.py:194-212) through hmv_op_hr_fuse_up against torch in float64:
    out = act(base + sum_s nearest_up_{2^shift}(conv1x1(x_s) + b_s)), terms added in order.  fp32: exact-fp32 MFMA products, 1e-5 of the
    output scale; fp16: the same arithmetic on fp16-rounded inputs with ONE rounding of the result (1e-3)."""
    from handmvnet_amd import _lib
    lib = _lib.load()
    N, H, W, C, srcs, relu = shape
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W + C)
    dev = torch.device("cuda:0")
    base = torch.randn(N, H, W, C, generator=g)
    xs = [torch.randn(N, H >> sh, W >> sh, cs, generator=g) for cs, sh in srcs]
    ws = [torch.randn(C, cs, generator=g) / cs ** 0.5 for cs, _ in srcs]
    bs = [torch.randn(C, generator=g) for _ in srcs]
    rnd = (lambda t: t.half().double()) if f16 else (lambda t: t.double())
    ref = rnd(base)
    for x, w, b, (_, sh) in zip(xs, ws, bs, srcs):
        gq = rnd(x) @ w.double().t() + b.double()
        ref = ref + gq.repeat_interleave(1 << sh, dim=1).repeat_interleave(1 << sh, dim=2)
    if relu:
        ref = ref.clamp_min(0)
    dbase = base.to(dev)
    dxs = [x.contiguous().to(dev) for x in xs]
    out = torch.full((N, H, W, C), float("nan"), device=dev, dtype=torch.float16 if f16 else torch.float32)
    n = len(srcs)
    vp = ctypes.c_void_p
    src_p = (vp * n)(*[vp(t.data_ptr()) for t in dxs])
    wn = [w.contiguous().numpy() for w in ws]
    bn = [b.contiguous().numpy() for b in bs]
    w_p = (vp * n)(*[a.ctypes.data_as(vp) for a in wn])
    b_p = (vp * n)(*[a.ctypes.data_as(vp) for a in bn])
    cs_a = (ctypes.c_int32 * n)(*[cs for cs, _ in srcs])
    sh_a = (ctypes.c_int32 * n)(*[sh for _, sh in srcs])
    rc = lib.hmv_op_hr_fuse_up(0, f16, vp(dbase.data_ptr()), N, H, W, C, n, src_p, cs_a, sh_a, w_p, b_p, int(relu), vp(out.data_ptr()), None)
    assert rc == 0, lib.hmv_last_error(None)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < (1e-3 if f16 else 1e-5), err


def test_hr_fuse_up_refuses_shapes_without_a_fused_form():
    from handmvnet_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    vp = ctypes.c_void_p
    base, x = torch.zeros(1, 10, 16, 40, device=dev), torch.zeros(1, 2, 4, 80, device=dev)     # 10 rows are not a multiple of 2^2
    w, b = np.zeros((40, 80), np.float32), np.zeros(40, np.float32)
    out = torch.zeros(1, 10, 16, 40, device=dev)
    rc = lib.hmv_op_hr_fuse_up(0, 0, vp(base.data_ptr()), 1, 10, 16, 40, 1, (vp * 1)(vp(x.data_ptr())), (ctypes.c_int32 * 1)(80), (ctypes.c_int32 * 1)(2),
                               (vp * 1)(w.ctypes.data_as(vp)), (vp * 1)(b.ctypes.data_as(vp)), 1, vp(out.data_ptr()), None)
    assert rc != 0 and b"fused form" in lib.hmv_last_error(None)


@pytest.mark.parametrize("case,size,nb", [("hr40_v4_128", 128, 2), ("hr40_v4_128", 256, 3), ("hr40_v4_128", 96, 2), ("hr64_tiny", 64, 2), ("hr64_tiny", 160, 1)])
@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_hrnet_fuse_layers_in_one_launch(case, size, nb, mode):
    """hr_fuse.hip: the up-sampling terms of an HRNet fuse layer (hrnet.py:194-212; 1x1 conv + BN + nearest up-sampling of every
    coarser branch, the last terms of y_i = relu(sum_j f_ij(x_j))) as ONE launch per output branch that has two or more of them, against one
    conv launch per term (each adding the running sum).  fp32: the same sums up to the order inside a dot product (exact-fp32 MFMAs
    either way); fp16: the fused launch rounds the sum once, the per-term launches after every term.  Frame sizes that leave ragged
    16 x 32 tiles (96: 24 x 24 maps; 160: 40 x 40) included; fewer launches; the poisoned-workspace rule holds.  The reference
    fixtures themselves run with the fused launch (test_reference_fixture / test_fp16_path_within_the_noise_floor)."""
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.synth import synth_inputs
    cfg, (tp, mp, dp), sd, _, _ = load_case(case)
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd)
    if mode == "f16":
        m.half()
    x, bbox, intr = synth_inputs(cfg, nb, 11 + nb, size)
    m.set_hr_fusion(True)
    fused = _run(m, x, bbox, intr)
    n_fused = m.launch_count()
    m.poison_workspace(0xFF)
    again = _run(m, x, bbox, intr)
    m.set_hr_fusion(False)
    plain = _run(m, x, bbox, intr)
    n_plain = m.launch_count()
    m.set_hr_fusion(True)
    # stage 3: branch 0 has two up-sampling terms (one launch instead of two) x 4 modules; stage 4: branches 0 and 1 (3 -> 1, 2 -> 1) x 3 modules
    assert n_plain - n_fused == 4 * 1 + 3 * (2 + 1), (n_plain, n_fused)
    tol = 2e-5 if mode == "f32" else 6e-3
    for k in ("feat0", "heatmap"):
        assert np.isfinite(fused[k]).all(), k
        assert np.array_equal(fused[k], again[k]), k
        assert rel_l2(fused[k], plain[k]) <= tol, (k, rel_l2(fused[k], plain[k]))
    one = _run(m, x[:1], bbox[:1], intr[:1])
    assert np.array_equal(one["joints_cam"][0], fused["joints_cam"][0])        # batch independence


@pytest.mark.parametrize("mode", ["f16", "f32x3"])
def test_hrnet_last_branch_on_a_second_stream(mode):
    """fp16-kernel modes, four-branch HRNet modules: the lowest-resolution branch's eight convs are enqueued on a second stream of the handle
    beside the branch above it (fork / join by events; nothing freed inside the region is handed out again before the join).  Same
    kernels, same bits as one stream (hmv_set_hr_fusion mode 1); the poisoned-workspace rule holds (hipGraph capture with the second stream:
    tests/test_gpu_graphs.py)."""
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.synth import synth_inputs
    cfg, (tp, mp, dp), sd, _, _ = load_case("hr40_v4_128")
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd)
    _set_mode(m, mode)
    x, bbox, intr = synth_inputs(cfg, 3, 21, 256)
    m.set_hr_fusion(3)
    two = _run(m, x, bbox, intr)
    m.poison_workspace(0xFF)
    again = _run(m, x, bbox, intr)
    m.set_hr_fusion(1)
    one = _run(m, x, bbox, intr)
    m.set_hr_fusion(3)
    for k in ("feat0", "heatmap", "tokens", "joints_cam", "joints_crop_img"):
        assert np.isfinite(two[k]).all(), k
        assert np.array_equal(two[k], one[k]), (k, float(np.abs(two[k] - one[k]).max()))
        assert np.array_equal(two[k], again[k]), k


_CFG2_REF = {}


@pytest.mark.parametrize("mode", ["f32", "f32x3"])
def test_cfg2_full_batch_against_the_oracle(mode):
    """BASELINE.json configs[1] at its REAL batch (B=8, V=4, 256x256, ResNet-18; the fixture of that shape is B=2): every
    sample against the f64 CPU oracle, plus batch independence; the split-precision mode to the same fp32 bars (its ResNet-18
    transposed conv runs as four phase launches, the fp32 one as a single merged launch)."""
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.synth import synth_inputs
    from oracle.oracle import Oracle
    cfg, (tp, mp, dp), sd, _, _ = load_case("cfg2s_r18_v4_256")
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd)
    if mode == "f32x3":
        m.float32x3()
    x, bbox, intr = synth_inputs(cfg, 8, 123, 256)
    dev = torch.device("cuda:0")
    xt, bt, it = torch.from_numpy(x).to(dev), torch.from_numpy(bbox).to(dev), torch.from_numpy(intr).to(dev)
    got = m(xt, bt, {"intrinsic": it})
    torch.cuda.synchronize()
    if "ref" not in _CFG2_REF:   # ~6 s of CPU: once for both modes
        _CFG2_REF["ref"] = Oracle(cfg, sd, "f64").forward(x, bbox, intr)
    ref = _CFG2_REF["ref"]
    assert rel_l2(got["joints_cam"].cpu().numpy(), ref["joints_cam"]) <= TOL_CAM
    assert np.abs(got["joints_crop_img"].cpu().numpy() - ref["joints_crop_img"]).max() <= 0.05 * 8
    assert rel_l2(got["heatmap"].cpu().numpy(), ref["heatmap"]) <= TOL_STAGE
    one = m(xt[5:6], bt[5:6], {"intrinsic": it[5:6]})
    assert torch.equal(one["joints_cam"][0], got["joints_cam"][5])


def test_state_dict_errors_match_reference_behaviour():
    from handmvnet_amd import HandMvNet
    cfg, (tp, mp, dp), sd, _, _ = load_case("tiny_r50")
    m = HandMvNet(tp, mp, dp)
    bad = dict(sd)
    del bad["pose_net.3.bias"]
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["extra.weight"] = np.zeros(3, np.float32)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        m.load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["pose_net.3.bias"] = np.zeros(22, np.float32)
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(bad, strict=True)
    legacy = {k.replace("pose_net.", "pose_net.conv.").replace("sample_nets.0.", "sample_net."): v for k, v in sd.items()}
    m.load_state_dict(legacy, strict=True)           # eval.py:27-52 remap
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 3, 64, 64, device="cuda"))   # 3 frames cannot be viewed as [-1, 2, ...]


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: the fp16 path (model.half()): conv stack in fp16 storage + fp16 MFMA
# with fp32 accumulation; heat-map logits, soft-argmax, tokens, fusion and decoder stay fp32.
# Its tolerance is stated SEPARATELY from the fp32 north-star bar (hard part 1) and is tied to the
# NOISE FLOOR OF fp16 STORAGE MEASURED ON THE REFERENCE ITSELF: tests/golden/make_fp16_noise.py runs the real reference
# with exactly the roundings any fp16-storage implementation must make (frames, conv weights, every conv+BN unit's and
# every residual block's output -> fp16; fp32 accumulation) and records, per case, how far THAT lands from the unrounded
# reference (tests/golden/fp16_noise.json: heat map 4e-4..1.1e-3 rel-L2, 0-3 % of the coordinates moved by > 1/2 px because
# soft_argmax_2d multiplies the logits by 1000, joints_cam 1e-3..1.2e-1 because the random-weight fusion transformer
# amplifies token noise ~50x).  The same script shows that keeping the last bottleneck + pose_net (or even all of
# layer3) in fp32-grade arithmetic does not move these numbers (flip rate 3.0 % -> 2.5 % -> 0.9 %): the noise is injected
# in layer1/2 already, so a mixed-precision tail is not a remedy and is not built.  The engine is held to:
#   * heat map:           <= 2 x the floor's rel-L2 (+ 2e-4)
#   * coordinates:        median <= 0.02 px; fraction moved by > 1/2 px <= floor + 2 %
#   * joints_cam:         <= min(3 x the floor's rel-L2, max(0.15, 1.5 x the floor)) (+ 2e-3)
# i.e. "as close to the fp32 reference as fp16 storage lets ANY implementation be, within a small factor".  The
# fp32-grade alternative on the same matrix cores is HMV_F32X3 (next test), which meets the fp32 bars.
# ---------------------------------------------------------------------------------------------
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp16_noise.json")) as _f:
    FP16_NOISE = json.load(_f)["cases"]


def fp16_bounds(name):
    fl = FP16_NOISE[name]
    # joints_cam: 3 x the floor where the floor is small; capped at max(0.15, 1.5 x floor) so that a large floor (random-weight
    # fusion amplifying soft-argmax flips) cannot turn the factor 3 into room for a 30 % pose error
    cam = fl["joints_cam_rel_l2"]
    # where make_fp16_noise.py measured several samples of the same input distribution (NOISE_SAMPLES), the floor is their maximum:
    # the pose error of an fp16-storage run is a few soft-argmax flips amplified by the random-weight fusion, a heavy-tailed draw per
    # sample (hr40_v8_256: 0.011 .. 0.132 over six samples of one model, heat maps 9.7e-4 .. 9.8e-4 on all of them)
    cam = max([cam] + list(fl.get("joints_cam_rel_l2_samples", [])))
    return {"heatmap": 2.0 * fl["heatmap_rel_l2"] + 2e-4, "flip": fl["coord_flip_frac"] + 0.02,
            "joints_cam": min(3.0 * cam + 2e-3, max(0.15, 1.5 * cam + 2e-3))}


@pytest.mark.parametrize("name", list(CASES))
def test_fp16_path_within_its_stated_tolerance(name):
    m, cfg, sd, (x, bbox, intr), fx = _model(name)
    m.half()
    got = _run(m, x, bbox, intr)
    feat = rel_l2(got["feat0"].reshape(-1)[fx["feat0_idx"]], fx["feat0_val"])
    hm = rel_l2(got["heatmap"].reshape(-1)[fx["heatmap_idx"]], fx["heatmap_val"])
    dc = np.abs(got["coords_hm"] - fx["coords_hm"])
    cam = rel_l2(got["joints_cam"], fx["joints_cam"])
    bound = fp16_bounds(name)
    rep = {"feat0": feat, "heatmap": hm, "coord_median_px": float(np.median(dc)), "coord_flip_frac": float((dc > 0.5).mean()),
           "joints_cam": cam, "bounds": bound}
    print(name, rep)
    assert feat <= 2e-3 and hm <= bound["heatmap"], rep
    # flips are discrete events on near-tied peaks: on a small case (84 coordinate values) 2 % is less than two of them, and which
    # ties flip changes with the summation order of the fp16 products (the heat-map error itself stays at the floor): allow four values
    assert rep["coord_median_px"] <= 0.02 and rep["coord_flip_frac"] <= max(bound["flip"], bound["flip"] - 0.02 + 4.0 / dc.size), rep
    if "amp_joints_cam" in fx:
        # ill-conditioned by construction (hr40_lq: the module amplifies token noise up to 6 000 x; the fp16-storage floor measured on
        # the reference is 1.1, i.e. a bound an all-zero pose would pass): the END-TO-END fp16 pose of this case is NOT pinned -- said
        # here instead of asserting a vacuous number.  What is pinned: backbone features, heat map and coordinates above, and the
        # fusion + decoder tail on the engine's own fp16-path tokens at fp32-grade bars (test_fusion_tail_on_engine_tokens[...-f16]).
        print(name, "joints_cam of the fp16 path: parity unpinned (conditioning); tail pinned by test_fusion_tail_on_engine_tokens")
    else:
        assert bound["joints_cam"] < 0.5, bound
        assert cam <= bound["joints_cam"], rep
    assert np.isfinite(got["joints_cam"]).all()
    # and the fp16 engine really is a different numerical path from the fp32 one
    m.float()
    ref32 = _run(m, x, bbox, intr)
    assert rel_l2(ref32["feat0"], got["feat0"]) > 1e-5


# ---------------------------------------------------------------------------------------------
# HMV_F32X3 (model.float32x3()): fp32-equivalent arithmetic on the fp16 matrix cores -- every conv-stack value is a
# (hi, lo) fp16 pair, every product hi*hi + lo*hi + hi*lo with fp32 accumulation (weights pre-scaled by a power of two per
# layer so that W_lo stays a normal fp16).  It is held to EXACTLY the bar of the fp32 path: joints_cam within 1e-3 rel-L2 of
# the real reference, dense stages within 2e-4, coordinates within 0.05 heat-map px (measured: backbone features and heat
# maps 1e-6, tokens <= 5.5e-5, joints_cam 2e-6 .. 1.6e-4 -- the same as the fp32 engine).
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny_r50", "cfg1_r50_v4_128", "cfg3s_r50_v8_256", "r50_wocam_nn", "r50_odd_96",
                                  "hr40_tiny", "hr40_v4_128", "hr64_tiny", "tiny_r18", "cfg2s_r18_v4_256", "r34_onelevel",
                                  "r18_frozen_nosin", "r18_single_view", "r18_13views", "r50_lq", "r18_lq_wocam", "r50_200", "r18_100", "hr40_lq", "hr40_v8_256"])
def test_split_precision_path_meets_the_fp32_bar(name):
    m, cfg, sd, (x, bbox, intr), fx = _model(name)
    m.float32x3()
    got = _run(m, x, bbox, intr)
    rep = check_against_fixture(got, fx, TOL_CAM, 0.05 * cfg.image_size / cfg.heatmap_size, TOL_STAGE)
    rep["coords"] = float(np.abs(got["coords_hm"] - fx["coords_hm"]).max())
    print(name, rep)
    assert rep["coords"] < 0.05, rep
    assert rep["feat0"] <= 2e-5, rep
    # it is a different numerical path from the fp32 engine, but only just
    m.float()
    ref32 = _run(m, x, bbox, intr)
    d = rel_l2(got["feat0"], ref32["feat0"])
    assert 0 < d < 2e-5, d


# ---------------------------------------------------------------------------------------------
# Seeded random configurations against the f64 oracle: backbone x views x frame size (incl. sizes that are not multiples of
# 32 and H != W) x positional-encoding subset x decoder x fusion kind x fusion depth x batch.  The fixtures pin named
# configurations; this sweeps the combinations between them.
# ---------------------------------------------------------------------------------------------
def _random_case(seed):
    rng = np.random.default_rng(seed)
    bt = ["18", "34", "50_paper"][int(rng.integers(0, 3))]
    ch = [1024] if bt == "50_paper" else [[256, 128, 64], [256, 128], [256]][int(rng.integers(0, 3))]
    pos = [p for p in ("pos2d", "crop", "sin") if rng.random() < 0.6]
    lq = bool(rng.random() < 0.3)
    spec = dict(bt=bt, ch=ch, V=int(rng.integers(1, 6)), B=int(rng.integers(1, 4)), size=64, pos=pos, gcn=bool(rng.random() < 0.5),
                wseed=1000 + seed, iseed=2000 + seed, fusion="cross_attn_learnable_query" if lq else "cross_attn",
                fusion_layers=int([1, 3, 5][int(rng.integers(0, 3))]), freeze_bn=bool(rng.random() < 0.3))
    hh, ww = [int(v) for v in rng.choice([64, 72, 77, 88, 96, 99, 100, 120], 2)]
    if seed >= 12:   # HRNet (fuse layers need frame sizes that are multiples of 32: hrnet.py:194-212), fewer levels kept too
        bt = ["w40", "w64"][seed % 2]
        full = [40, 80, 160, 320] if bt == "w40" else [64, 128, 256, 512]
        spec.update(bt=bt, ch=full[:int(rng.integers(2, 5))], V=int(rng.integers(1, 4)), B=int(rng.integers(1, 3)))
        hh, ww = [int(v) for v in rng.choice([64, 96, 128], 2)]
    if seed == 16:   # the configuration family of profiles/r02_probe_lq_conditioning.txt's outlier: w40, four levels, probe queries
        spec.update(bt="w40", ch=[40, 80, 160, 320], fusion="cross_attn_learnable_query")
    return spec, hh, ww


def _amplification(cfg, sd, x, bbox, intr, ref):
    """By how much the map frames -> tokens -> (fused, joints_cam) amplifies a relative perturbation of the tokens, measured on the
    f64 oracle with three 1e-6-level perturbations of the frames (the backbone is well conditioned: they reach the tokens as
    ~1e-6).  Learnable-query blocks on un-normalised HRNet features: 10^2 .. 10^3 (profiles/r03_probe_lq_hr40.txt)."""
    from handmvnet_amd.synth import normalish
    from oracle.oracle import Oracle
    amp = {"fused": 0.0, "joints_cam": 0.0}
    for i in range(3):
        xp = (x.astype(np.float64) * (1.0 + 1e-6 * normalish("input.perturb", 500 + i, x.size).reshape(x.shape))).astype(np.float32)
        o = Oracle(cfg, sd, "f64").forward(xp, bbox, intr, stages=True)
        din = max(rel_l2(o["tokens"], ref["tokens"]), 1e-12)
        for k in amp:
            amp[k] = max(amp[k], rel_l2(o[k], ref[k]) / din)
    return amp


@pytest.mark.parametrize("seed", list(range(17)))   # seed 16: HRNet-w40, all four levels, learnable-query fusion
def test_random_configurations_match_oracle(seed):
    from cases import case_params
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.spec import config_from_params, heatmap_size_of
    from handmvnet_amd.synth import normalish, synth_inputs, synth_state_dict
    from oracle.oracle import Oracle
    spec, hh, ww = _random_case(seed)
    tp, mp, dp = case_params(spec)
    cfg = config_from_params(tp, mp, dp)
    sd = synth_state_dict(cfg, spec["wseed"])
    _, bbox, intr = synth_inputs(cfg, spec["B"], spec["iseed"], 64)
    x = normalish("input.random", spec["iseed"], spec["B"] * spec["V"] * 3 * hh * ww).astype(np.float32).reshape(spec["B"], spec["V"], 3, hh, ww)
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd, strict=True)
    if seed % 3 == 2:
        m.float32x3()     # every third configuration on the split-precision kernels: the same fp32 bars
    got = _run(m, x, bbox, intr)
    ref = Oracle(cfg, sd, "f64").forward(x, bbox, intr, stages=True)
    assert got["heatmap"].shape == ref["heatmap"].shape == (spec["B"], spec["V"], 21) + tuple(heatmap_size_of(cfg, hh, ww))
    rep = {k: rel_l2(got[k], ref[k]) for k in ("joints_cam", "heatmap", "feat0", "tokens", "fused")}
    rep["coords"] = float(np.abs(got["coords_hm"] - ref["coords_hm"]).max())
    # conditioning: how far an fp32 CPU evaluation of the same graph (the oracle built as float) lands from the f64 one.  The
    # learnable-query blocks have no LayerNorm; on some random weights their activations grow to ~450 and the softmax turns
    # near-one-hot, where fp32 itself is only good to 1e-2 (profiles/r02_probe_lq_conditioning.txt).  The engine has to be as good as fp32 is there.
    ref32 = Oracle(cfg, sd, "f32").forward(x, bbox, intr, stages=True)
    cond = {k: rel_l2(ref32[k], ref[k]) for k in ("joints_cam", "fused")}
    print(seed, spec, (hh, ww), rep, cond)
    assert max(rep["heatmap"], rep["feat0"], rep["tokens"]) <= TOL_STAGE and rep["coords"] < 0.05, (spec, hh, ww, rep)
    bound = {"joints_cam": max(TOL_CAM, 4 * cond["joints_cam"]), "fused": max(TOL_STAGE, 4 * cond["fused"])}
    if rep["joints_cam"] > bound["joints_cam"] or rep["fused"] > bound["fused"]:
        # one fp32 evaluation is ONE draw of a heavy-tailed error (softmax flips): before calling the engine wrong, measure the
        # amplification of this configuration and allow amplification x the engine's own token error
        amp = _amplification(cfg, sd, x, bbox, intr, ref)
        bound = {k: max(bound[k], 2.0 * amp[k] * rep["tokens"]) for k in bound}
        print(seed, "amplification", amp, "bounds", bound)
    assert rep["joints_cam"] <= bound["joints_cam"], (spec, hh, ww, rep, cond, bound)
    assert rep["fused"] <= bound["fused"], (spec, hh, ww, rep, cond, bound)


# (B, T, Tq, koff, Tk): self-attention over V * 21 tokens, the middle block's 21 queries against the rest, ragged key ranges
ATTENTION_SHAPES = [(2, 21, 21, 0, 21), (3, 84, 84, 0, 84), (1, 168, 168, 0, 168), (2, 168, 21, 21, 147), (2, 84, 21, 21, 63),
                    (1, 1008, 1008, 0, 1008), (2, 105, 21, 21, 84), (1, 33, 33, 0, 33), (1, 129, 129, 0, 129), (2, 64, 64, 0, 64)]


@pytest.mark.parametrize("shape", ATTENTION_SHAPES)
@pytest.mark.parametrize("kernel", ["f32", "x3"])
def test_attention_kernel_vs_torch(shape, kernel):
    """op-level: the fusion transformer's attention (layers.py:216-221, 8 heads x 128) vs torch fp64 on the CPU; large
    logits included (a softmax that is nearly one-hot) and the result must not depend on what else is in the batch.
    f32: exact-fp32 MFMA products (the fp32 mode).  x3 (round 4): what the fp16-kernel modes run -- q, k, v and P as fp16 (hi, lo) pairs
    on the fp16 matrix cores, three products per step, operands carried to 2^-22: the same bar."""
    from handmvnet_amd import _lib
    lib = _lib.load()
    op = lib.hmv_op_attention if kernel == "f32" else lib.hmv_op_attention_x3
    B, T, Tq, koff, Tk = shape
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = torch.randn(B, T, 3, 8, 128, generator=g)
    qkv[:, :, 0] *= 3.0                                   # logits of std ~3 * sqrt(128) / sqrt(128): sharp rows
    dev = torch.device("cuda:0")
    qd = qkv.reshape(B, T, 3072).contiguous().to(dev)
    out = torch.full((B, Tq, 1024), float("nan"), device=dev)
    rc = op(0, qd.data_ptr(), B, T, Tq, koff, Tk, out.data_ptr(), None)
    assert rc == 0, lib.hmv_last_error(None)
    torch.cuda.synchronize()
    q = qkv[:, :Tq, 0].double().permute(0, 2, 1, 3)       # [B, 8, Tq, 128]
    k = qkv[:, koff:koff + Tk, 1].double().permute(0, 2, 1, 3)
    v = qkv[:, koff:koff + Tk, 2].double().permute(0, 2, 1, 3)
    att = torch.softmax(q @ k.transpose(-1, -2) * 128 ** -0.5, dim=-1) @ v
    ref = att.permute(0, 2, 1, 3).reshape(B, Tq, 1024)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    # fp32 logits of magnitude ~12 carry ~1e-6 absolute rounding, which the exponential turns into ~1e-6 relative
    err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1.0)
    print(kernel, shape, err)
    assert err < 4e-6, err       # (measured: f32 0.8 .. 2.1e-6, x3 0.6 .. 1.5e-6)
    # sample 0 alone gives the same bits
    out1 = torch.full((1, Tq, 1024), float("nan"), device=dev)
    rc = op(0, qd[:1].contiguous().data_ptr(), 1, T, Tq, koff, Tk, out1.data_ptr(), None)
    assert rc == 0
    assert torch.equal(out1[0], out[0])


# (B, T, Tq, shared probe queries): self-attention blocks (q = k = v rows of one qkv matrix) and the probe block (21 learnable
# queries shared by every sample, keys / values from a [k | v] matrix)
LQ_ATTENTION_SHAPES = [(2, 42, 42, False), (1, 168, 168, False), (3, 63, 21, True), (2, 21, 21, False), (1, 273, 21, True), (2, 33, 33, False)]


@pytest.mark.parametrize("shape", LQ_ATTENTION_SHAPES)
def test_lq_attention_kernel_vs_torch(shape):
    """op-level: the learnable-query fusion's attention (layers.py:284-291, 8 heads x 256) on the fp32 matrix cores vs torch fp64,
    sharp rows included; batch independence.  (The round-2 scalar kernel it was first checked against is reachable only in a
    -DHMV_DEV_KNOBS build now.)"""
    from handmvnet_amd import _lib
    lib = _lib.load()
    B, T, Tq, probe = shape
    g = torch.Generator().manual_seed(B * 1000 + T)
    dev = torch.device("cuda:0")
    if probe:
        q = torch.randn(Tq, 8, 256, generator=g) * 3.0
        kv = torch.randn(B, T, 2, 8, 256, generator=g)
        qd, kvd = q.reshape(Tq, 2048).contiguous().to(dev), kv.reshape(B, T, 4096).contiguous().to(dev)
        args = (qd.data_ptr(), 2048, 0, kvd.data_ptr(), kvd.data_ptr() + 2048 * 4, 4096)
        q64 = q.double().permute(1, 0, 2)[None].expand(B, -1, -1, -1)
        k64, v64 = kv[:, :, 0].double().permute(0, 2, 1, 3), kv[:, :, 1].double().permute(0, 2, 1, 3)
    else:
        qkv = torch.randn(B, T, 3, 8, 256, generator=g)
        qkv[:, :, 0] *= 3.0
        qd = qkv.reshape(B, T, 6144).contiguous().to(dev)
        args = (qd.data_ptr(), 6144, T, qd.data_ptr() + 2048 * 4, qd.data_ptr() + 4096 * 4, 6144)
        q64 = qkv[:, :Tq, 0].double().permute(0, 2, 1, 3)
        k64, v64 = qkv[:, :, 1].double().permute(0, 2, 1, 3), qkv[:, :, 2].double().permute(0, 2, 1, 3)
    ref = (torch.softmax(q64 @ k64.transpose(-1, -2) * 256 ** -0.5, dim=-1) @ v64).permute(0, 2, 1, 3).reshape(B, Tq, 2048)

    def run(nb):
        out = torch.full((nb, Tq, 2048), float("nan"), device=dev)
        rc = lib.hmv_op_attention_lq(0, *args, nb, T, Tq, out.data_ptr(), None)
        assert rc == 0, lib.hmv_last_error(None)
        torch.cuda.synchronize()
        return out
    got = run(B)
    assert torch.isfinite(got).all()
    tol = 4e-6 * max(ref.abs().max().item(), 1.0)
    assert (got.cpu().double() - ref).abs().max().item() < tol
    assert torch.equal(run(1)[0], got[0])                 # sample 0 alone gives the same bits


@pytest.mark.parametrize("mode", ["f32", "f16", "f32x3"])
def test_odd_frame_size_through_the_space_to_depth_stem(mode):
    """The stem runs as a 4x4 conv over 2x2 space-to-depth frames; an odd H / W leaves a half-empty last row / column pair.
    75 x 91 frames in every arithmetic mode against the f64 oracle (fp16: backbone features to the fp16-storage level --
    a layout slip would be an O(1) error)."""
    from cases import case_params
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.spec import config_from_params
    from handmvnet_amd.synth import normalish, synth_inputs, synth_state_dict
    from oracle.oracle import Oracle
    spec = dict(bt="18", ch=[256, 128, 64], V=2, B=2, size=64, pos=["pos2d", "crop", "sin"], gcn=True, wseed=77, iseed=78)
    tp, mp, dp = case_params(spec)
    cfg = config_from_params(tp, mp, dp)
    sd = synth_state_dict(cfg, spec["wseed"])
    hh, ww = 75, 91
    _, bbox, intr = synth_inputs(cfg, spec["B"], spec["iseed"], 64)
    x = normalish("input.odd", 5, spec["B"] * spec["V"] * 3 * hh * ww).astype(np.float32).reshape(spec["B"], spec["V"], 3, hh, ww)
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd, strict=True)
    if mode == "f16":
        m.half()
    elif mode == "f32x3":
        m.float32x3()
    got = _run(m, x, bbox, intr)
    ref = Oracle(cfg, sd, "f64").forward(x, bbox, intr, stages=True)
    feat = rel_l2(got["feat0"], ref["feat0"])
    print(mode, feat, rel_l2(got["joints_cam"], ref["joints_cam"]))
    if mode == "f16":
        assert feat <= 1e-2
    else:
        assert feat <= TOL_STAGE and rel_l2(got["joints_cam"], ref["joints_cam"]) <= TOL_CAM
