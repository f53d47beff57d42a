"""The numpy restatement of the frame preparation (oracle/frames_oracle.py) against tests/golden/frames_cases.npz
(the reference's crop_and_pad_image + torch's antialiased resize, made by tests/golden/make_frames_fixture.py)."""
import os

import numpy as np
import pytest

from oracle import frames_oracle as fo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = np.load(os.path.join(ROOT, "tests", "golden", "frames_cases.npz"))
NAMES = sorted({k.split(".")[0] for k in FIX.files})


@pytest.mark.parametrize("name", NAMES)
def test_frames_oracle_matches_reference(name):
    got = fo.prepare_batch(FIX[f"{name}.frames"], FIX[f"{name}.boxes"], int(FIX[f"{name}.size"]))
    assert got.shape == FIX[f"{name}.out"].shape and got.dtype == np.float32
    assert np.abs(got - FIX[f"{name}.out"]).max() < 3e-6      # normalised units (values span about +-2.6)


def test_empty_box_is_the_black_view():
    out = fo.prepare_view(np.full((20, 20, 3), 200, np.uint8), [5, 5, 5, 9], 8)
    assert np.allclose(out, ((0 - fo.MEAN) / fo.STD)[:, None, None])
