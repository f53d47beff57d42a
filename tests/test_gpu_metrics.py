"""GPU parity of the evaluation metrics (hmv_pose_metrics through handmvnet_amd.metrics.PoseMetrics) against
  (1) outputs of the real reference's models/metrics.py (tests/golden/metrics_cases.npz),
  (2) the numpy oracle on fresh seeded inputs, including degenerate point sets,
  (3) the evaluation step end to end: checkpoint file -> model -> test_step -> metrics.

Tolerances: PCK values and thresholds are bit-exact (fp32 comparisons of fp32 distances, integer counts);
mpjpe / pa_mpjpe within 2e-5 relative (the reference sums in fp32, we sum in fp64); aligned points 2e-6 m.
"""
import os
import sys

import numpy as np
import pytest
import torch

from cases import CASES
from helpers import load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import metrics_oracle as mo  # noqa: E402

pytestmark = pytest.mark.gpu

FIX = np.load(os.path.join(ROOT, "tests", "golden", "metrics_cases.npz"))
NAMES = sorted({k.split(".")[0] for k in FIX.files if k.endswith(".pa_mpjpe")})


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("name", NAMES)
def test_metrics_match_reference_fixture(name):
    from handmvnet_amd.metrics import PoseMetrics
    p, g = _dev(FIX[f"{name}.pred"]), _dev(FIX[f"{name}.gt"])
    lo, hi, steps = FIX[f"{name}.range"]
    assert PoseMetrics.mpjpe(p, g).item() == pytest.approx(float(FIX[f"{name}.mpjpe"]), rel=2e-5)
    assert PoseMetrics.pa_mpjpe(p, g).item() == pytest.approx(float(FIX[f"{name}.pa_mpjpe"]), rel=2e-5)
    al = PoseMetrics.compute_similarity_transform(p, g)
    assert al.shape == p.shape and al.is_cuda
    assert np.abs(al.cpu().numpy()[:32] - FIX[f"{name}.aligned"]).max() < 2e-6
    auc, norm_auc, vals, thr = PoseMetrics.pck_auc(p, g, min_threshold=lo, max_threshold=hi, steps=int(steps))
    assert isinstance(auc, float) and isinstance(vals, list) and len(vals) == int(steps)
    assert np.array_equal(np.array(thr, np.float32), FIX[f"{name}.thr"].astype(np.float32))
    assert np.array_equal(np.array(vals, np.float32), FIX[f"{name}.pck"].astype(np.float32))
    assert auc == pytest.approx(FIX[f"{name}.auc"][0], rel=1e-6)
    assert norm_auc == pytest.approx(FIX[f"{name}.auc"][1], rel=1e-6)
    assert PoseMetrics.pck(p, g, 0.01).item() == pytest.approx(float(FIX[f"{name}.pck_at_10mm"]), abs=1e-7)
    # the fused call the model uses returns the same numbers
    m, pa, auc2, norm2, vals2, thr2 = PoseMetrics.all_metrics(p, g, lo, hi, int(steps))
    assert (m.item(), pa.item(), auc2, norm2, vals2, thr2) == \
        (PoseMetrics.mpjpe(p, g).item(), PoseMetrics.pa_mpjpe(p, g).item(), auc, norm_auc, vals, thr)


def test_mpjpe_2d_matches_reference_fixture():
    from handmvnet_amd.metrics import PoseMetrics
    got = PoseMetrics.mpjpe(_dev(FIX["crop2d.pred"]), _dev(FIX["crop2d.gt"])).item()
    assert got == pytest.approx(float(FIX["crop2d.mpjpe"]), rel=2e-5)


def test_metrics_match_oracle_on_degenerate_sets():
    """Planar and collinear hands (rank-deficient 3x3 cross-covariance), identical sets, huge offsets."""
    from handmvnet_amd.metrics import PoseMetrics
    rng = np.random.default_rng(5)
    gt = (rng.standard_normal((6, 21, 3)) * 0.04).astype(np.float32)
    pred = gt + rng.standard_normal(gt.shape).astype(np.float32) * 0.004
    gt[0, :, 2] = 0.0                                  # planar target
    pred[1, :, 2] = 0.1                                # planar prediction
    pred[2] = pred[2] + 5.0                            # 5 m offset: translation must be removed exactly
    pred[3] = gt[3]                                    # identical: zero error, d <= 0 counts at threshold 0
    pred[4] = gt[4] * 3.0                              # pure scale
    p, g = _dev(pred), _dev(gt)
    assert PoseMetrics.mpjpe(p, g).item() == pytest.approx(mo.mpjpe(pred, gt), rel=2e-6)
    assert PoseMetrics.pa_mpjpe(p, g).item() == pytest.approx(mo.pa_mpjpe(pred, gt), rel=1e-5, abs=1e-9)
    al = PoseMetrics.compute_similarity_transform(p, g).cpu().numpy()
    assert np.abs(al - mo.compute_similarity_transform(pred, gt)).max() < 2e-6
    assert np.abs(al[3] - gt[3]).max() < 1e-7 and np.abs(al[4] - gt[4]).max() < 1e-7
    auc, norm_auc, vals, thr = PoseMetrics.pck_auc(p, g, 0.0, 0.02, 20)
    o_auc, o_norm, o_vals, o_thr = mo.pck_auc(pred, gt, 0.0, 0.02, 20)
    assert vals == o_vals and thr == o_thr
    assert vals[0] == pytest.approx(21 / (6 * 21))     # only the identical pose sits at distance 0
    assert auc == pytest.approx(o_auc, rel=1e-6) and norm_auc == pytest.approx(o_norm, rel=1e-6)


def test_metrics_are_deterministic_and_reject_bad_arguments():
    from handmvnet_amd import _lib
    from handmvnet_amd.metrics import PoseMetrics
    rng = np.random.default_rng(9)
    p, g = _dev(rng.standard_normal((300, 21, 3)).astype(np.float32)), _dev(rng.standard_normal((300, 21, 3)).astype(np.float32))
    a = [PoseMetrics.all_metrics(p, g, 0.0, 3.0, 20) for _ in range(3)]
    assert all(x[0].item() == a[0][0].item() and x[1].item() == a[0][1].item() and x[4] == a[0][4] for x in a)
    with pytest.raises(_lib.HandMvError):
        PoseMetrics.pa_mpjpe(p[..., :2], g[..., :2])             # Procrustes is 3-D only
    with pytest.raises(_lib.HandMvError):
        PoseMetrics.pck_auc(p, g, 0.0, 1.0, steps=1000)          # more thresholds than the kernel holds
    with pytest.raises(_lib.HandMvError):
        PoseMetrics.mpjpe(p.cpu(), g.cpu())                      # no CPU path
    with pytest.raises(NotImplementedError):
        PoseMetrics.pck(p, g, 0.01, reference_len=torch.ones(300))


def test_evaluation_step_end_to_end(tmp_path):
    """checkpoint file -> load_checkpoint_with_legacy_fix -> test_step: the metrics of the engine's own forward,
    checked against the oracle metrics evaluated on the REAL reference's forward output for the same case."""
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.checkpoint import load_checkpoint_with_legacy_fix
    name = "cfg1_r50_v4_128"
    cfg, (tp, mp, dp), sd, (x, bbox, intr), fx = load_case(name)
    path = str(tmp_path / "m.ckpt")
    torch.save({"state_dict": {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, "epoch": 1}, path)
    model = load_checkpoint_with_legacy_fix(path, HandMvNet(tp, mp, dp)).to("cuda").eval()
    rng = np.random.default_rng(3)
    ref_cam, ref_crop = fx["joints_cam"], fx["joints_crop_img"]
    gt_cam_mm = ((ref_cam + rng.standard_normal(ref_cam.shape) * 0.006) * 1000).astype(np.float32)
    gt_crop = (ref_crop + rng.standard_normal(ref_crop.shape) * 2).astype(np.float32)
    mask = rng.random(ref_crop.shape[:3]) < 0.2
    batch = {"data": {"rgb": _dev(x), "bboxes": _dev(bbox), "joints_cam": _dev(gt_cam_mm), "root_joint": _dev(np.zeros((1, 3), np.float32)),
                      "joints_crop_img": _dev(gt_crop), "joints_img_mask": _dev(mask)},
             "cam_params": {"intrinsic": _dev(intr)}}
    own = model(batch["data"]["rgb"], batch["data"]["bboxes"], batch["cam_params"])     # the forward is deterministic
    own_cam, own_crop = own["joints_cam"].cpu().numpy(), own["joints_crop_img"].cpu().numpy()
    res = model.test_step(batch, 0)
    assert res["loss"] is None
    mt = res["metrics"]
    gt_m = gt_cam_mm / np.float32(1000)
    assert np.allclose(batch["data"]["joints_cam"].cpu().numpy(), gt_m)            # converted in place, like the reference
    keep = (~mask)[..., None]
    # (a) exactly the oracle's metrics of the engine's own forward output
    assert mt["test_mpjpe"].item() == pytest.approx(mo.mpjpe(own_cam, gt_m) * 1000, rel=2e-5)
    assert mt["test_pa_mpjpe"].item() == pytest.approx(mo.pa_mpjpe(own_cam, gt_m) * 1000, rel=2e-5)
    assert mt["test_mpjpe2d"].item() == pytest.approx(mo.mpjpe(own_crop * keep, gt_crop * keep), rel=2e-5)
    o_auc, o_norm, o_vals, _ = mo.pck_auc(own_cam, gt_m, 0.0, 0.05, 20)                # ho3d: thresholds 0..50 mm
    assert mt["test_pck_j"] == o_vals
    assert mt["test_auc_j"] == pytest.approx(o_auc, rel=1e-6) and mt["test_norm_auc_j"] == pytest.approx(o_norm, rel=1e-6)
    # (b) and, by the triangle inequality, within the forward's own deviation of the metrics of the REAL reference's output
    dev3d = np.linalg.norm(own_cam - ref_cam, axis=-1).mean() * 1000
    assert abs(mt["test_mpjpe"].item() - mo.mpjpe(ref_cam, gt_m) * 1000) <= dev3d + 1e-4
    dev2d = np.linalg.norm((own_crop - ref_crop) * keep, axis=-1).mean()
    assert abs(mt["test_mpjpe2d"].item() - mo.mpjpe(ref_crop * keep, gt_crop * keep)) <= dev2d + 1e-4
    # (c) validation_step (handmvnet.py:468-491) is the same body with "val_" keys; ground truth left on the HOST is moved
    #     to the device like the reference's Lightning batch transfer would have done
    batch_v = {"data": {"rgb": _dev(x), "bboxes": _dev(bbox), "joints_cam": torch.from_numpy(gt_cam_mm.copy()),
                        "root_joint": torch.zeros(1, 3), "joints_crop_img": torch.from_numpy(gt_crop.copy()),
                        "joints_img_mask": torch.from_numpy(mask)},
               "cam_params": {"intrinsic": _dev(intr)}}
    rv = model.validation_step(batch_v, 0)
    assert rv["loss"] is None and set(rv["metrics"]) == {k.replace("test_", "val_") for k in mt}
    # (the host-side /1000 of the ground truth may differ from the device's by an ulp)
    assert rv["metrics"]["val_mpjpe"].item() == pytest.approx(mt["test_mpjpe"].item(), rel=1e-5)
    assert rv["metrics"]["val_pa_mpjpe"].item() == pytest.approx(mt["test_pa_mpjpe"].item(), rel=1e-5)
    assert rv["metrics"]["val_pck_j"] == mt["test_pck_j"]
