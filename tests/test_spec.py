"""Host logic: config distillation and the state_dict layout (handmvnet.py:28-125)."""
import numpy as np
import pytest

from cases import CASES, case_params
from handmvnet_amd import spec as S
from handmvnet_amd.synth import synth_inputs, synth_state_dict


def _cfg(name):
    return S.config_from_params(*case_params(CASES[name]))


def test_r50_paper_layout_counts():
    cfg = _cfg("cfg3s_r50_v8_256")
    lay = S.state_dict_layout(cfg)
    assert cfg.feat_dim == 524                      # 512 + 2 + 10
    assert len(lay) == 355                          # 355 tensors
    nbytes = sum(int(np.prod(s)) * 4 for k, s in lay.items() if not k.endswith("num_batches_tracked"))
    assert 85e6 < nbytes < 88e6                     # "86 MB for r50-paper-V8"
    assert lay["joints_late_fusion.attn_fusion.2.to_q.weight"] == (1024, 524)
    assert lay["joints_decoder.joints_gcn1.weight"] == (3, 1, 524, 256)
    assert "backbone.layer4.0.conv1.weight" not in lay


def test_r18_layout_has_unused_tail_and_convtranspose():
    cfg = _cfg("cfg2s_r18_v4_256")
    lay = S.state_dict_layout(cfg)
    assert cfg.feat_dim == 236
    assert lay["pose_net.0.weight"] == (256, 128, 4, 4)      # ConvTranspose2d [in, out, kh, kw]
    assert "backbone.layer4.1.conv2.weight" in lay and "backbone.fc.weight" in lay
    ex = S.executed_keys(cfg)
    assert not any(k.startswith("backbone.layer4") or k.startswith("backbone.fc") for k in ex)


def test_frozen_bn_has_no_num_batches_tracked():
    lay = S.state_dict_layout(_cfg("r18_frozen_nosin"))
    assert not any(k.startswith("backbone") and k.endswith("num_batches_tracked") for k in lay)
    assert "pose_net.1.num_batches_tracked" in lay          # pose_net keeps nn.BatchNorm2d


def test_config_errors_match_reference():
    tp, mp, dp = case_params(CASES["tiny_r50"])
    with pytest.raises(AssertionError):
        S.config_from_params(tp, dict(mp, backbone="vgg"), dp)
    with pytest.raises(AssertionError):
        S.config_from_params(tp, dict(mp, backbone_type="101"), dp)
    with pytest.raises(NotImplementedError):
        S.config_from_params(tp, dict(mp, fusion="mean"), dp)
    with pytest.raises(NotImplementedError):
        S.config_from_params(tp, mp, dict(dp, name="freihand"))
    with pytest.raises(AssertionError):
        S.config_from_params(tp, dict(mp, fusion_layers=4), dp)
    with pytest.raises(KeyError):
        S.config_from_params({"root_relative": True}, mp, dp)
    mp2 = dict(mp)
    del mp2["num_views"]
    mp2["selected_views"] = [0, 2, 5]
    assert S.config_from_params(tp, mp2, dp).num_views == 3      # config.py:46-49


def test_legacy_remap():
    sd = {"pose_net.conv.0.weight": 1, "pose_net.conv.1.bias": 2, "sample_net.conv.0.weight": 3, "backbone.conv1.weight": 4}
    out = S.remap_legacy_keys(sd)
    assert set(out) == {"pose_net.0.weight", "pose_net.1.bias", "sample_nets.0.conv.0.weight", "backbone.conv1.weight"}
    same = {"pose_net.0.weight": 1}
    assert S.remap_legacy_keys(same) == same


def test_synth_is_deterministic_and_nontrivial():
    cfg = _cfg("tiny_r18")
    a, b = synth_state_dict(cfg, 3), synth_state_dict(cfg, 3)
    c = synth_state_dict(cfg, 4)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert not np.array_equal(a["backbone.conv1.weight"], c["backbone.conv1.weight"])
    rv = a["backbone.layer1.0.bn1.running_var"]
    assert rv.min() >= 0.5 and rv.max() <= 1.5 and rv.std() > 0.1
    x, bbox, intr = synth_inputs(cfg, 2, 5, 64)
    assert x.shape == (2, 2, 3, 64, 64) and abs(float(x.std()) - 1.0) < 0.05
    assert (bbox[..., 2] > bbox[..., 0]).all() and (intr[..., 0] >= 400).all()
    # bit-stability across machines: a few known values
    assert float(a["backbone.conv1.weight"].reshape(-1)[0]) == pytest.approx(float(b["backbone.conv1.weight"].reshape(-1)[0]), abs=0)


def test_flop_model_matches_survey():
    cfg = _cfg("cfg3s_r50_v8_256")
    f = S.conv_flops_per_image(cfg, 256)
    backbone = f["stem"] + f["layer1"] + f["layer2"] + f["layer3"]
    assert backbone == pytest.approx(19.23e9, rel=0.01)
    assert f["pose_net"] == pytest.approx(1.10e9, rel=0.01)
    assert f["sample_net"] == pytest.approx(1.07e9, rel=0.01)


def test_state_dict_follows_the_nn_module_protocol():
    """prefix / destination (keyword or positional), and a parent module's state_dict() / load_state_dict() see the keys."""
    import torch
    from collections import OrderedDict
    from handmvnet_amd import HandMvNet
    m = HandMvNet(*case_params(CASES["tiny_r18"]))
    plain = m.state_dict()
    assert list(plain) == list(S.state_dict_layout(m.cfg))
    dest = OrderedDict(other=torch.zeros(1))
    out = m.state_dict(destination=dest, prefix="net.")
    assert out is dest and "other" in dest and all(("net." + k) in dest for k in plain)
    assert list(m.state_dict(OrderedDict(), "p.")) == ["p." + k for k in plain]     # positional form

    class Parent(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = m
            self.extra = torch.nn.Linear(2, 2)

    p = Parent()
    sd = p.state_dict()
    assert "extra.weight" in sd and "model.pose_net.0.weight" in sd
    new = {k: (v + 1 if k == "model.pose_net.0.bias" else v) for k, v in sd.items()}
    p.load_state_dict(new, strict=True)
    assert np.allclose(m._weights["pose_net.0.bias"], plain["pose_net.0.bias"].numpy() + 1)
    bad = dict(new)
    del bad["model.pose_net.0.bias"]
    with pytest.raises(RuntimeError, match="model.pose_net.0.bias"):
        p.load_state_dict(bad, strict=True)


def test_learnable_query_fusion_warns_about_keys_it_ignores():
    """cross_attn_learnable_query is an extension beyond the reference's runnable surface (handmvnet.py:227 raises TypeError): the
    module always has 5 blocks and its own per-block PE, so 'sin' in pos_enc / fusion_layers != 5 have no effect -- and say so."""
    tp, mp, dp = case_params(CASES["r50_lq"])
    mp = dict(mp, pos_enc=["pos2d", "crop", "sin"], fusion_layers=3)
    with pytest.warns(UserWarning, match="ignores .*'sin' in pos_enc.* and fusion_layers=3"):
        cfg = S.config_from_params(tp, mp, dp)
    assert cfg.learnable_query
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        S.config_from_params(tp, dict(mp, pos_enc=["pos2d", "crop"], fusion_layers=5), dp)
