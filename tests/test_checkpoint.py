"""Checkpoint ingestion (handmvnet_amd/checkpoint.py, mirror of eval.py:15-52) -- host logic, no GPU."""
import os
import pickle
from collections import OrderedDict

import numpy as np
import pytest
import torch

from handmvnet_amd.checkpoint import is_legacy_version, load_checkpoint_with_legacy_fix  # noqa: E402
from handmvnet_amd.model import HandMvNet  # noqa: E402
from handmvnet_amd.spec import config_from_params  # noqa: E402
from handmvnet_amd.synth import synth_state_dict  # noqa: E402
from cases import CASES, case_params  # noqa: E402


def _lightning_ckpt(sd, path, legacy=False):
    state = OrderedDict()
    for k, v in sd.items():
        if legacy:   # the layout eval.py:18 detects: pose_net wrapped in .conv, one un-indexed sample_net
            k = k.replace("pose_net.", "pose_net.conv.").replace("sample_nets.0.", "sample_net.")
        state[k] = torch.from_numpy(np.array(v))
    torch.save({"epoch": 3, "global_step": 1234, "pytorch-lightning_version": "2.2.0", "state_dict": state,
                "hyper_parameters": {"model_params": {"num_views": 2}}, "optimizer_states": [], "lr_schedulers": []}, path)


@pytest.mark.parametrize("legacy", [False, True])
def test_load_checkpoint_roundtrip(tmp_path, legacy, capsys):
    tp, mp, dp = case_params(CASES["tiny_r50"])
    sd = synth_state_dict(config_from_params(tp, mp, dp), seed=77)
    path = str(tmp_path / "model.ckpt")
    _lightning_ckpt(sd, path, legacy)
    assert is_legacy_version(torch.load(path, weights_only=True)["state_dict"]) == legacy
    model = HandMvNet(tp, mp, dp)            # constructed with different (seed 0) weights
    assert load_checkpoint_with_legacy_fix(path, model) is model
    for k, v in sd.items():
        assert np.array_equal(model._weights[k], v), k
    printed = capsys.readouterr().out
    assert ("Legacy version detected" in printed) == legacy


def test_checkpoint_for_another_architecture_is_rejected(tmp_path):
    tp, mp, dp = case_params(CASES["tiny_r18"])
    sd = synth_state_dict(config_from_params(tp, mp, dp), seed=1)
    path = str(tmp_path / "r18.ckpt")
    _lightning_ckpt(sd, path)
    model = HandMvNet(*case_params(CASES["tiny_r50"]))
    with pytest.raises(RuntimeError, match="Error\\(s\\) in loading state_dict for HandMvNet"):
        load_checkpoint_with_legacy_fix(path, model)


class _Payload:
    def __reduce__(self):
        return (os.getenv, ("HOME",))


def test_checkpoint_with_code_is_refused(tmp_path):
    """weights_only=True: a checkpoint that would run code on load is refused instead of executed."""
    path = str(tmp_path / "evil.ckpt")
    torch.save({"state_dict": {}, "callbacks": _Payload()}, path)
    model = HandMvNet(*case_params(CASES["tiny_r50"]))
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint_with_legacy_fix(path, model)
