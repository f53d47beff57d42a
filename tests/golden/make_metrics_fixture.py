"""Generates tests/golden/metrics_cases.npz by running the REAL reference metrics module
(/root/reference/src/models/metrics.py, imports cleanly with torch alone) on seeded inputs.

    python tests/golden/make_metrics_fixture.py

Stored per case: the inputs (small) and every output HandMvNet._get_metrics / _calculate_mpjpe reads
(handmvnet.py:352-383).  Runs only in the build container; the fixture is data.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
from models.metrics import PoseMetrics  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def hand_like(rng, b, n=21):
    """Root-relative hand-sized point sets in metres."""
    return (rng.standard_normal((b, n, 3)) * 0.04).astype(np.float32)


def make_cases():
    rng = np.random.default_rng(20250310)
    cases = {}
    gt = hand_like(rng, 16)
    cases["noise_5mm"] = (gt + rng.standard_normal(gt.shape).astype(np.float32) * 0.005, gt, (0.0, 0.02), 20)
    gt = hand_like(rng, 32)
    cases["noise_20mm_auc50"] = (gt + rng.standard_normal(gt.shape).astype(np.float32) * 0.02, gt, (0.0, 0.05), 20)
    # a rotated, scaled and shifted copy plus a little noise: alignment must remove almost all of the error
    gt = hand_like(rng, 8)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    q *= np.sign(np.linalg.det(q))
    cases["similarity"] = ((1.3 * gt @ q.T + 0.05 + rng.standard_normal(gt.shape) * 1e-3).astype(np.float32), gt, (0.0, 0.02), 20)
    # mirrored prediction: det(U V^T) < 0, the Z fix-up must keep R a proper rotation (metrics.py:158-160)
    gt = hand_like(rng, 8)
    mir = gt.copy()
    mir[..., 0] *= -1
    cases["mirrored"] = ((mir + rng.standard_normal(gt.shape) * 2e-3).astype(np.float32), gt, (0.0, 0.02), 20)
    # a single pose, odd step count
    gt = hand_like(rng, 1)
    cases["single_pose_7steps"] = (gt + rng.standard_normal(gt.shape).astype(np.float32) * 0.01, gt, (0.0, 0.02), 7)
    # many poses (an epoch's worth in one call)
    gt = hand_like(rng, 1100)   # more poses than lanes in the workgroup
    cases["many_poses"] = (gt + rng.standard_normal(gt.shape).astype(np.float32) * 0.008, gt, (0.0, 0.02), 20)
    return cases


def main():
    out = {}
    for name, (pred, gt, (lo, hi), steps) in make_cases().items():
        p, g = torch.from_numpy(pred), torch.from_numpy(gt)
        auc, norm_auc, vals, thr = PoseMetrics.pck_auc(p, g, min_threshold=lo, max_threshold=hi, steps=steps)
        out[f"{name}.pred"], out[f"{name}.gt"] = pred, gt
        out[f"{name}.range"] = np.array([lo, hi, steps], np.float64)
        out[f"{name}.mpjpe"] = np.float64(PoseMetrics.mpjpe(p, g).item())
        out[f"{name}.pa_mpjpe"] = np.float64(PoseMetrics.pa_mpjpe(p, g).item())
        out[f"{name}.aligned"] = PoseMetrics.compute_similarity_transform(p, g).numpy()[:32]
        out[f"{name}.auc"] = np.array([auc, norm_auc], np.float64)
        out[f"{name}.pck"] = np.array(vals, np.float64)
        out[f"{name}.thr"] = np.array(thr, np.float64)
        out[f"{name}.pck_at_10mm"] = np.float64(PoseMetrics.pck(p, g, 0.01).item())
    # 2-D error of joints_crop_img [b, v, 21, 2] (handmvnet.py:381)
    rng = np.random.default_rng(7)
    g2 = (rng.random((4, 8, 21, 2)) * 256).astype(np.float32)
    p2 = g2 + rng.standard_normal(g2.shape).astype(np.float32) * 3
    out["crop2d.pred"], out["crop2d.gt"] = p2, g2
    out["crop2d.mpjpe"] = np.float64(PoseMetrics.mpjpe(torch.from_numpy(p2), torch.from_numpy(g2)).item())
    path = os.path.join(HERE, "metrics_cases.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
