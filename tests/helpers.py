"""Shared helpers for the parity tests."""
from __future__ import annotations

import json
import os

import numpy as np

from cases import CASES, case_params
from handmvnet_amd.spec import config_from_params
from handmvnet_amd.synth import synth_inputs, synth_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def load_case(name: str):
    """-> (cfg, params triple, state_dict, (x, bbox, intr), fixture dict)"""
    fx = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    spec = json.loads(str(fx["spec"]))
    assert spec == json.loads(json.dumps(CASES[name])), "fixture is stale: regenerate with make_fixtures.py"
    tp, mp, dp = case_params(spec)
    cfg = config_from_params(tp, mp, dp)
    sd = synth_state_dict(cfg, spec["wseed"])
    inputs = synth_inputs(cfg, spec["B"], spec["iseed"], spec["size"])
    return cfg, (tp, mp, dp), sd, inputs, fx


def check_against_fixture(out: dict, fx: dict, tol_cam: float, tol_coord_px: float, tol_stage: float = None):
    """out: dict with joints_cam / joints_crop_img / heatmap (+ optional stages), numpy arrays."""
    # ill-conditioned cases carry the amplification measured on the reference in float64 (make_fixtures.py, `cond`): the
    # tolerance behind the token matrix is then amplification x the implementation's own token error (x2), never below the
    # fixed one.  Un-normalised learnable-query blocks on HRNet features amplify token rounding noise by 10^2 .. 10^3.
    if "amp_fused" in fx and "tokens" in out and out["tokens"] is not None:
        tok = max(rel_l2(np.asarray(out["tokens"]).reshape(-1)[fx["tokens_idx"]], fx["tokens_val"]), 1e-6)
        tol_cam = max(tol_cam, 2.0 * float(fx["amp_joints_cam"]) * tok)
        tol_fused = max(tol_stage or 0.0, 2.0 * float(fx["amp_fused"]) * tok)
    else:
        tol_fused = tol_stage
    report = {"joints_cam": rel_l2(out["joints_cam"], fx["joints_cam"]),
              "joints_crop_img_maxabs": float(np.abs(out["joints_crop_img"] - fx["joints_crop_img"]).max())}
    assert out["joints_cam"].shape == fx["joints_cam"].shape
    assert out["joints_crop_img"].shape == fx["joints_crop_img"].shape
    assert report["joints_cam"] <= tol_cam, report
    assert report["joints_crop_img_maxabs"] <= tol_coord_px, report
    for nm in ("heatmap", "feat0", "tokens", "fused"):
        if nm in out and out[nm] is not None:
            assert tuple(out[nm].shape) == tuple(fx[nm + "_shape"]), nm
            got = np.asarray(out[nm]).reshape(-1)[fx[nm + "_idx"]]
            report[nm] = rel_l2(got, fx[nm + "_val"])
            if tol_stage is not None:
                assert report[nm] <= (tol_fused if nm == "fused" else tol_stage), report
    return report
