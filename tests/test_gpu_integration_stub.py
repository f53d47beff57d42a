"""INTEGRATION.md promises that its ~50-line ctypes stub is all a reference maintainer has to add.  This test executes
exactly that code block (only the library path is substituted) and checks it against a reference fixture."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import load_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_documented_stub_runs_and_matches_the_reference():
    from handmvnet_amd import _lib
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# src/models/handmvnet_mi355x\.py.*?)```", md, re.S).group(1)
    code = code.replace('ctypes.CDLL("libhandmv.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    cfg, (tp, mp, dp), sd, (x, bbox, intr), fx = load_case("cfg1_r50_v4_128")
    model = ns["HandMvNetMI355X"](tp, mp, dp)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    dev = torch.device("cuda:0")
    out = model(torch.from_numpy(x).to(dev), torch.from_numpy(bbox).to(dev), {"intrinsic": torch.from_numpy(intr).to(dev)})
    torch.cuda.synchronize()
    assert set(out) == {"joints_crop_img", "joints_cam", "heatmap"}
    err = np.linalg.norm(out["joints_cam"].cpu().numpy() - fx["joints_cam"]) / np.linalg.norm(fx["joints_cam"])
    assert err < 1e-3, err
    assert np.abs(out["joints_crop_img"].cpu().numpy() - fx["joints_crop_img"]).max() < 0.4
