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


def cond_bounds(fx: dict, tol_cam: float, tol_fused, tok_err: float = None, cond_cap: float = 4.0):
    """Tolerances behind the token matrix for fixtures of ill-conditioned configurations (make_fixtures.py, `cond`).

    Such a fixture carries (a) the amplification of a relative token perturbation into `fused` / joints_cam measured on the
    reference's own fusion module in float64 -- worst and median of eight draws (`amp_*`, `amp_*_med`: up to 6 000 / typically
    1 200 for un-normalised learnable-query blocks on HRNet features) -- and (b) how far the REFERENCE's own fp32 run sits from its
    float64 fusion -> decoder on the same tokens (`cond_*32_vs_64`).

    * tok_err None (the tail alone, run on given tokens): cond_cap x cond_*32_vs_64, never below the fixed bar.
    * tok_err given (end to end: the implementation's own token error travels through the tail as well):
      cond_cap x cond_*32_vs_64 + median amplification x token error, never above 2 x worst amplification x token error and
      never below the fixed bar.  (Round 3 allowed the worst-case term alone: 1e-2 .. 1 where fp32 really does 3e-4)"""
    if "amp_fused" not in fx:
        return tol_cam, tol_fused
    out = []
    for fixed, nm in ((tol_cam, "joints_cam"), (tol_fused or 0.0, "fused")):
        allow = cond_cap * float(fx[f"cond_{nm}32_vs_64"])
        if tok_err is not None:
            t = max(tok_err, 1e-6)
            allow = min(2.0 * float(fx[f"amp_{nm}"]) * t, allow + float(fx[f"amp_{nm}_med"]) * t)
        out.append(max(fixed, allow))
    return out[0], out[1]


def check_against_fixture(out: dict, fx: dict, tol_cam: float, tol_coord_px: float, tol_stage: float = None, cond_cap: float = 4.0):
    """out: dict with joints_cam / joints_crop_img / heatmap (+ optional stages), numpy arrays."""
    if "amp_fused" in fx and "tokens" in out and out["tokens"] is not None:
        tok = rel_l2(np.asarray(out["tokens"]).reshape(-1)[fx["tokens_idx"]], fx["tokens_val"])
        tol_cam, tol_fused = cond_bounds(fx, tol_cam, tol_stage, tok, cond_cap)
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
