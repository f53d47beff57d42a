"""The numpy restatement of the evaluation metrics (oracle/metrics_oracle.py) against outputs of the real
reference module (tests/golden/metrics_cases.npz, made by tests/golden/make_metrics_fixture.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import metrics_oracle as mo  # noqa: E402

FIX = np.load(os.path.join(ROOT, "tests", "golden", "metrics_cases.npz"))
CASES = sorted({k.split(".")[0] for k in FIX.files if k.endswith(".pa_mpjpe")})


@pytest.mark.parametrize("name", CASES)
def test_metrics_oracle_matches_reference(name):
    pred, gt = FIX[f"{name}.pred"], FIX[f"{name}.gt"]
    lo, hi, steps = FIX[f"{name}.range"]
    assert mo.mpjpe(pred, gt) == pytest.approx(float(FIX[f"{name}.mpjpe"]), rel=2e-6)
    assert mo.pa_mpjpe(pred, gt) == pytest.approx(float(FIX[f"{name}.pa_mpjpe"]), rel=2e-5)
    al = mo.compute_similarity_transform(pred, gt)[:32]
    assert np.abs(al - FIX[f"{name}.aligned"]).max() < 2e-6     # metres; fp32 SVD in the reference
    auc, norm_auc, vals, thr = mo.pck_auc(pred, gt, lo, hi, int(steps))
    assert np.array_equal(np.array(thr, np.float32), FIX[f"{name}.thr"].astype(np.float32))
    assert np.array_equal(np.array(vals, np.float32), FIX[f"{name}.pck"].astype(np.float32))
    assert auc == pytest.approx(FIX[f"{name}.auc"][0], rel=1e-6)
    assert norm_auc == pytest.approx(FIX[f"{name}.auc"][1], rel=1e-6)
    assert mo.pck(pred, gt, 0.01) == pytest.approx(float(FIX[f"{name}.pck_at_10mm"]), abs=1e-7)


def test_mpjpe_2d_matches_reference():
    assert mo.mpjpe(FIX["crop2d.pred"], FIX["crop2d.gt"]) == pytest.approx(float(FIX["crop2d.mpjpe"]), rel=2e-6)
