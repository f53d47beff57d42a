"""GPU parity of the frame preparation (uint8 camera frames -> model input):
  (1) hmv_op_prepare_frames against the reference fixture (crop_and_pad_image of the real reference + torch's
      antialiased resize) and against the numpy oracle on fresh inputs;
  (2) hmv_forward_frames == hmv_forward on the oracle-prepared batch, end to end;
  (3) properties at full size: determinism, per-frame independence, value range.
Tolerance: 5e-6 in normalised units for the op (fp32 filter arithmetic in a different summation order);
joints_cam of the fused path within 1e-4 rel-L2 of the two-step path.
"""
import os

import numpy as np
import pytest
import torch

from test_gpu_parity import fp16_bounds
from helpers import load_case, rel_l2
from oracle import frames_oracle as fo

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = np.load(os.path.join(ROOT, "tests", "golden", "frames_cases.npz"))
NAMES = sorted({k.split(".")[0] for k in FIX.files})


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("name", NAMES)
def test_prepare_frames_matches_reference_fixture(name):
    from handmvnet_amd.frames import prepare_frames
    got = prepare_frames(_dev(FIX[f"{name}.frames"]), _dev(FIX[f"{name}.boxes"]), int(FIX[f"{name}.size"]))
    assert tuple(got.shape) == FIX[f"{name}.out"].shape
    assert np.abs(got.cpu().numpy() - FIX[f"{name}.out"]).max() < 5e-6


def test_prepare_frames_matches_oracle_on_camera_sized_frames():
    """480x640 frames (ho3d.py:26), windows like batch_center_scale_to_box produces, 256x256 output."""
    from handmvnet_amd.frames import prepare_frames
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (2, 3, 480, 640, 3), dtype=np.uint8)
    side = rng.integers(90, 420, (2, 3))
    cx, cy = rng.integers(0, 640, (2, 3)), rng.integers(0, 480, (2, 3))
    boxes = np.stack([cx - side // 2, cy - side // 2, cx - side // 2 + side, cy - side // 2 + side], axis=-1).astype(np.int32)
    boxes[1, 2] = [700, 500, 900, 700]                      # entirely outside the frame: all zeros before Normalize
    got = prepare_frames(_dev(frames), _dev(boxes), 256).cpu().numpy()
    want = fo.prepare_batch(frames, boxes, 256)
    assert got.shape == (2, 3, 3, 256, 256)
    assert np.abs(got - want).max() < 5e-6
    assert np.allclose(got[1, 2], ((0 - fo.MEAN) / fo.STD)[:, None, None])


@pytest.mark.parametrize("name,mode", [("tiny_r18", "f32"), ("cfg1_r50_v4_128", "f32"), ("hr40_tiny", "f32"), ("tiny_r50", "f16"),
                                       ("cfg1_r50_v4_128", "f32x3"), ("hr40_tiny", "f32x3"),
                                       ("r50_200", "f32"), ("r18_100", "f32"), ("r50_lq", "f32")])   # odd frame sizes, learnable-query fusion
def test_forward_frames_equals_forward_on_prepared_batch(name, mode):
    from handmvnet_amd import HandMvNet
    cfg, (tp, mp, dp), sd, (x, bbox, intr), fx = load_case(name)
    b, v, size = x.shape[0], x.shape[1], x.shape[-1]
    rng = np.random.default_rng(17)
    frames = rng.integers(0, 256, (b, v, 120, 160, 3), dtype=np.uint8)
    # smooth the noise a little so that the heat maps are not pathologically flat
    frames = ((frames.astype(np.float32) + np.roll(frames, 1, 2) + np.roll(frames, 1, 3)) / 3).astype(np.uint8)
    side = rng.integers(50, 140, (b, v))
    x1, y1 = rng.integers(-20, 100, (b, v)), rng.integers(-20, 60, (b, v))
    boxes = np.stack([x1, y1, x1 + side, y1 + side], axis=-1).astype(np.int32)
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd, strict=True)
    m.to("cuda").eval()
    half = mode == "f16"
    if half:
        m.half()
    elif mode == "f32x3":
        m.float32x3()
    cam = {"intrinsic": _dev(intr)}
    two_step = m(_dev(fo.prepare_batch(frames, boxes, size)), _dev(boxes.astype(np.float32)), cam)
    fused = m.forward_frames(_dev(frames), _dev(boxes), cam, image_size=size)
    torch.cuda.synchronize()
    assert set(fused) == {"joints_crop_img", "joints_cam", "heatmap"}
    # fp16 storage: the two paths' inputs differ by ~5e-6, which is enough to flip fp16 roundings, so they land a noise
    # floor apart (tests/golden/fp16_noise.json, measured on the reference itself; see test_gpu_parity.py)
    tol = fp16_bounds(name)["joints_cam"] if half else 1e-4
    assert rel_l2(fused["joints_cam"].cpu().numpy(), two_step["joints_cam"].cpu().numpy()) < tol
    assert np.abs(fused["joints_crop_img"].cpu().numpy() - two_step["joints_crop_img"].cpu().numpy()).max() < (1.0 if half else 0.02)   # image px; 1.0 = 1/8 heat-map px, the fp16 noise floor on soft coordinates
    assert rel_l2(fused["heatmap"].cpu().numpy(), two_step["heatmap"].cpu().numpy()) < (5e-3 if half else 1e-4)


@pytest.mark.parametrize("mode", ["f32", "f16", "f32x3"])
def test_forward_frames_odd_output_size(mode):
    """75 x 75 crops: the frame-preparation kernel writes the space-to-depth stem layout directly, and an odd size leaves a
    half-empty last row / column pair that must read as zeros in every storage mode."""
    from handmvnet_amd import HandMvNet
    cfg, (tp, mp, dp), sd, (x, bbox, intr), fx = load_case("tiny_r18")
    b, v, size = x.shape[0], x.shape[1], 75
    rng = np.random.default_rng(23)
    frames = rng.integers(0, 256, (b, v, 120, 160, 3), dtype=np.uint8)
    frames = ((frames.astype(np.float32) + np.roll(frames, 1, 2) + np.roll(frames, 1, 3)) / 3).astype(np.uint8)
    side = rng.integers(50, 140, (b, v))
    x1, y1 = rng.integers(-20, 100, (b, v)), rng.integers(-20, 60, (b, v))
    boxes = np.stack([x1, y1, x1 + side, y1 + side], axis=-1).astype(np.int32)
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd, strict=True)
    m.to("cuda").eval()
    if mode == "f16":
        m.half()
    elif mode == "f32x3":
        m.float32x3()
    cam = {"intrinsic": _dev(intr)}
    two_step = m(_dev(fo.prepare_batch(frames, boxes, size)), _dev(boxes.astype(np.float32)), cam)
    fused = m.forward_frames(_dev(frames), _dev(boxes), cam, image_size=size)
    torch.cuda.synchronize()
    half = mode == "f16"
    assert rel_l2(fused["heatmap"].cpu().numpy(), two_step["heatmap"].cpu().numpy()) < (5e-3 if half else 1e-4)
    assert rel_l2(fused["joints_cam"].cpu().numpy(), two_step["joints_cam"].cpu().numpy()) < (fp16_bounds("tiny_r18")["joints_cam"] if half else 1e-4)


def test_prepare_frames_full_size_properties():
    """BASELINE-sized batch (256 frames of 480x640 -> 256x256): deterministic, every frame independent of its
    neighbours, output bounded by the normalised range of [0, 255]."""
    from handmvnet_amd.frames import prepare_frames
    g = torch.Generator(device="cuda:0").manual_seed(3)
    frames = torch.randint(0, 256, (32, 8, 480, 640, 3), dtype=torch.uint8, device="cuda:0", generator=g)
    side = torch.randint(100, 400, (32, 8, 1), device="cuda:0", generator=g)
    org = torch.randint(-50, 400, (32, 8, 2), device="cuda:0", generator=g)
    boxes = torch.cat([org, org + side], dim=-1).int()
    a = prepare_frames(frames, boxes, 256)
    b = prepare_frames(frames, boxes, 256)
    assert torch.equal(a, b)
    sub = prepare_frames(frames[5:6, 2:5], boxes[5:6, 2:5], 256)
    assert torch.equal(sub, a[5:6, 2:5])
    lo = torch.tensor([(0 - m) / s for m, s in zip(fo.MEAN, fo.STD)], device="cuda:0").view(1, 1, 3, 1, 1)
    hi = torch.tensor([(1 - m) / s for m, s in zip(fo.MEAN, fo.STD)], device="cuda:0").view(1, 1, 3, 1, 1)
    assert bool(((a >= lo - 1e-5) & (a <= hi + 1e-5)).all())


def test_prepare_frames_random_windows_match_oracle():
    """40 seeded random (frame, window, output size) combinations: 1-pixel windows, heavy up- and down-sampling, windows
    straddling every edge or missing the frame entirely, non-square windows."""
    from handmvnet_amd.frames import prepare_frames
    rng = np.random.default_rng(99)
    worst = 0.0
    for _ in range(40):
        hf, wf = int(rng.integers(8, 70)), int(rng.integers(8, 90))
        size = int(rng.choice([8, 16, 24, 32, 48]))
        frames = rng.integers(0, 256, (3, hf, wf, 3), dtype=np.uint8)
        x1, y1 = rng.integers(-30, wf + 10, 3), rng.integers(-30, hf + 10, 3)
        bw, bh = rng.integers(1, 120, 3), rng.integers(1, 120, 3)
        boxes = np.stack([x1, y1, x1 + bw, y1 + bh], axis=-1).astype(np.int32)
        got = prepare_frames(_dev(frames), _dev(boxes), size).cpu().numpy()
        want = fo.prepare_batch(frames, boxes, size)
        err = float(np.abs(got - want).max())
        assert err < 1e-5, (hf, wf, size, boxes.tolist(), err)
        worst = max(worst, err)
    print("worst", worst)


def test_garbage_boxes_are_bounded_and_black():
    """int32 extremes and absurd windows: no overflow, no unbounded loop; they come out as the black view."""
    from handmvnet_amd.frames import prepare_frames
    frames = torch.full((4, 48, 64, 3), 200, dtype=torch.uint8, device="cuda:0")
    boxes = torch.tensor([[-2 ** 31, -2 ** 31, 2 ** 31 - 1, 2 ** 31 - 1], [0, 0, 2 ** 30, 2 ** 30], [10, 10, 5, 40],
                          [-1000, -1000, 1000, 1000]], dtype=torch.int32, device="cuda:0")
    out = prepare_frames(frames, boxes, 32).cpu().numpy()
    black = ((0 - fo.MEAN) / fo.STD)[:, None, None]
    assert np.allclose(out[:3], black)
    want = fo.prepare_view(np.full((48, 64, 3), 200, np.uint8), [-1000, -1000, 1000, 1000], 32)   # a legal window 30x the frame
    assert np.abs(out[3] - want).max() < 5e-6


def test_forward_frames_argument_errors():
    from handmvnet_amd import HandMvNet, _lib
    cfg, (tp, mp, dp), sd, (x, bbox, intr), fx = load_case("tiny_r18")
    m = HandMvNet(tp, mp, dp).to("cuda").eval()
    frames = torch.zeros(1, 2, 32, 32, 3, dtype=torch.uint8, device="cuda:0")
    boxes = torch.tensor([[[0, 0, 32, 32], [0, 0, 32, 32]]], device="cuda:0")
    with pytest.raises(ValueError):
        m.forward_frames(frames.float(), boxes, {"intrinsic": _dev(intr)})
    with pytest.raises(_lib.HandMvError):
        m.forward_frames(frames.cpu(), boxes, {"intrinsic": _dev(intr)})
    with pytest.raises(TypeError):
        m.forward_frames(frames, boxes, None)                      # 'crop' in pos_enc needs intrinsics
    with pytest.raises(RuntimeError):
        m.forward_frames(frames, boxes[:, :1], {"intrinsic": _dev(intr)})
    with pytest.raises(_lib.HandMvError):
        m.forward_frames(frames, boxes, {"intrinsic": _dev(intr)}, std=(0.2, 0.0, 0.2))
