"""Status codes and messages of the raw C ABI (include/handmv.h) on a real device: every misuse comes back as an
integer status + hmv_last_error text, nothing throws and nothing falls back."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import load_case

pytestmark = pytest.mark.gpu

OK, ARG, STATE, MISSING, SHAPE, HIP, UNSUPPORTED = range(7)


def _cfg(lib_mod, cfg, hh, ww, **over):
    from handmvnet_amd.spec import BACKBONE_IDS
    c = lib_mod.HmvConfig()
    c.struct_size = ctypes.sizeof(lib_mod.HmvConfig)
    c.backbone = BACKBONE_IDS[cfg.backbone_type]
    c.n_levels = len(cfg.backbone_channels)
    for i, ch in enumerate(cfg.backbone_channels):
        c.channels[i] = ch
    c.num_views, c.height, c.width = cfg.num_views, hh, ww
    c.image_size, c.heatmap_size = cfg.image_size, cfg.heatmap_size
    c.pos_enc, c.fusion_layers, c.decoder, c.dtype, c.device = cfg.pos_mask, cfg.fusion_layers, int(cfg.use_gcn), 0, 0
    c.fusion = int(cfg.learnable_query)
    for k, v in over.items():
        setattr(c, k, v)
    return c


def _set(lib, h, key, a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
    return lib.hmv_set_tensor(h, key.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim)


def test_create_rejects_bad_configurations():
    from handmvnet_amd import _lib
    lib = _lib.load()
    cfg, _, sd, _, _ = load_case("tiny_r18")
    h = ctypes.c_void_p()
    for over, code, text in [({"struct_size": 8}, ARG, b"struct_size"), ({"backbone": 9}, ARG, b"Supports only"),
                             ({"fusion_layers": 4}, ARG, b"odd number"), ({"height": 16}, ARG, b"at least 32"),
                             ({"fusion": 7}, ARG, b"Invalid fusion type"), ({"backbone": 3, "height": 100}, ARG, b"multiples of 32"),
                             ({"num_views": 0}, ARG, b"num_views"), ({"dtype": 7}, UNSUPPORTED, b"dtype"),
                             ({"device": 99}, ARG, b"device ordinal")]:
        rc = lib.hmv_create(ctypes.byref(_cfg(_lib, cfg, 64, 64, **over)), ctypes.byref(h))
        assert rc == code, (over, rc, lib.hmv_last_error(None))
        assert text in lib.hmv_last_error(None), (over, lib.hmv_last_error(None))
    assert lib.hmv_create(None, ctypes.byref(h)) == ARG


def test_weight_loading_and_forward_state_machine():
    from handmvnet_amd import _lib
    from handmvnet_amd.spec import executed_keys
    lib = _lib.load()
    cfg, _, sd, (x, bbox, intr), fx = load_case("tiny_r18")
    h = ctypes.c_void_p()
    assert lib.hmv_create(ctypes.byref(_cfg(_lib, cfg, 64, 64)), ctypes.byref(h)) == OK
    try:
        dev = torch.device("cuda:0")
        xt, bt, it = torch.from_numpy(x).to(dev), torch.from_numpy(bbox).to(dev), torch.from_numpy(intr).to(dev)
        b, v = x.shape[:2]
        crop, cam = torch.empty(b, v, 21, 2, device=dev), torch.empty(b, 21, 3, device=dev)
        fwd = lambda xp=xt.data_ptr(), bp=bt.data_ptr(), ip=it.data_ptr(), batch=b: lib.hmv_forward(
            h, batch, xp, bp, ip, crop.data_ptr(), cam.data_ptr(), None, None)
        assert fwd() == STATE and b"finalize" in lib.hmv_last_error(h)          # no weights yet
        keys = executed_keys(cfg)
        for k in keys[:-1]:
            assert _set(lib, h, k, sd[k]) == OK
        assert lib.hmv_finalize_weights(h) == MISSING and keys[-1].encode() in lib.hmv_last_error(h)
        assert _set(lib, h, keys[-1], np.zeros(5)) == OK                        # stored; the shape is checked at finalize
        assert lib.hmv_finalize_weights(h) == SHAPE and b"size mismatch" in lib.hmv_last_error(h)
        assert _set(lib, h, keys[-1], sd[keys[-1]]) == OK
        assert _set(lib, h, "backbone.layer4.0.conv1.weight", np.zeros((4, 4))) == OK   # unread keys are accepted and ignored
        assert lib.hmv_finalize_weights(h) == OK
        assert fwd(xp=None) == ARG and fwd(batch=0) == ARG
        assert fwd(bp=None) == ARG and b"crop" in lib.hmv_last_error(h)         # pos_enc has 'crop': bbox / intrinsic required
        assert lib.hmv_reserve(h, 0) == ARG
        assert lib.hmv_workspace_bytes(h, 2) > lib.hmv_workspace_bytes(h, 1) > 0
        assert fwd() == OK                                                       # heatmap output is optional (NULL)
        torch.cuda.synchronize()
        assert np.abs(cam.cpu().numpy() - fx["joints_cam"]).max() <= 1e-3 * np.abs(fx["joints_cam"]).max()
        assert lib.hmv_read_stage(h, b"no_such_stage", crop.data_ptr(), 4, None) != OK
        assert lib.hmv_profile_get(h, 10 ** 6, None, None, None, None) == ARG
    finally:
        lib.hmv_destroy(h)
    assert lib.hmv_forward(None, 1, None, None, None, None, None, None, None) == ARG
    assert lib.hmv_version().startswith(b"handmv")


def test_op_attention_argument_errors():
    """hmv_op_attention (op-level test entry): ranges that would read outside [B][T][3072] are refused before any launch."""
    import torch
    from handmvnet_amd import _lib
    lib = _lib.load()
    qkv = torch.zeros(2, 42, 3072, device="cuda:0")
    out = torch.zeros(2, 42, 1024, device="cuda:0")
    ok = lib.hmv_op_attention(0, qkv.data_ptr(), 2, 42, 21, 21, 21, out.data_ptr(), None)
    assert ok == 0
    for args in ((2, 42, 43, 0, 42), (2, 42, 21, 30, 21), (2, 42, 21, 0, 0), (0, 42, 21, 0, 21), (2, 42, 0, 0, 21), (2, 42, 21, -1, 21)):
        assert lib.hmv_op_attention(0, qkv.data_ptr(), *args, out.data_ptr(), None) != 0, args
    assert lib.hmv_op_attention(0, None, 2, 42, 21, 0, 42, out.data_ptr(), None) != 0
