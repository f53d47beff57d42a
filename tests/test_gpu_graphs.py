"""hipGraph replay of the forward (hmv_set_graphs, opt-in): bit-identical to the eager path, keyed by the caller's buffers."""
import numpy as np
import pytest
import torch

from helpers import load_case

pytestmark = pytest.mark.gpu


def _model(name):
    from handmvnet_amd import HandMvNet
    cfg, (tp, mp, dp), sd, (x, bbox, intr), fx = load_case(name)
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd, strict=True)
    m.to("cuda").eval()
    dev = torch.device("cuda:0")
    return m, torch.from_numpy(x).to(dev), torch.from_numpy(bbox).to(dev), {"intrinsic": torch.from_numpy(intr).to(dev)}


def _call_into(m, x, bbox, cam, outs):
    """forward through the C ABI into caller-owned output buffers (so that the buffer set is under the test's control)."""
    import ctypes
    from handmvnet_amd import _lib
    b, v = x.shape[:2]
    h = m._engine(x.shape[-2], x.shape[-1], 0)
    bb = bbox.reshape(-1, 4).contiguous().float()
    it = cam["intrinsic"].reshape(-1, 4).contiguous().float()
    m._keep = (bb, it)
    rc = _lib.load().hmv_forward(h, b, x.data_ptr(), bb.data_ptr(), it.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
                                 outs[2].data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, h)
    m._last_key = (x.shape[-2], x.shape[-1], 0, b, m._dtype)


def _outs(x, m=None):
    b, v, _, hh, ww = x.shape
    from handmvnet_amd.spec import heatmap_size_of
    hs = heatmap_size_of(m.cfg, hh, ww) if m is not None else (hh // 8, ww // 8)
    return [torch.zeros(b, v, 21, 2, device=x.device), torch.zeros(b, 21, 3, device=x.device),
            torch.zeros((b, v, 21) + tuple(hs), device=x.device)]


@pytest.mark.parametrize("name", ["tiny_r18", "cfg1_r50_v4_128", "hr40_tiny", "r18_single_view", "r50_lq", "r18_100"])
def test_graph_replay_is_bit_identical_to_eager(name):
    m, x, bbox, cam = _model(name)
    bb = bbox.reshape(-1, 4).contiguous().float()
    cam = {"intrinsic": cam["intrinsic"].reshape(-1, 4).contiguous().float()}
    m.use_graphs(False)
    eager = _outs(x, m)
    _call_into(m, x, bb, cam, eager)
    torch.cuda.synchronize()
    m.use_graphs(True)
    outs = _outs(x, m)
    for i in range(4):                      # eager, capture + first launch, replay, replay
        for o in outs:
            o.fill_(float("nan"))
        _call_into(m, x, bb, cam, outs)
        torch.cuda.synchronize()
        for a, b in zip(outs, eager):
            assert torch.equal(a, b), (name, i)
    cached, replays = m.graph_stats()
    assert cached == 1 and replays == 2
    # new input VALUES in the same buffers: the replay must read them (nothing but addresses is baked in)
    x.mul_(0.5)
    m.use_graphs(False)
    _call_into(m, x, bb, cam, eager)
    torch.cuda.synchronize()
    m.use_graphs(True)
    for _ in range(3):
        _call_into(m, x, bb, cam, outs)
    torch.cuda.synchronize()
    for a, b in zip(outs, eager):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["hr40_tiny", "hr40_v4_128"])
def test_graph_capture_takes_the_second_stream_along(name):
    """fp16-kernel modes run a four-branch HRNet module's last branch on a second stream of the handle (forked from / joined into the
    caller's stream by events): under capture that stream joins the capture, and the replay gives the eager bits."""
    m, x, bbox, cam = _model(name)
    m.half()
    bb = bbox.reshape(-1, 4).contiguous().float()
    cam = {"intrinsic": cam["intrinsic"].reshape(-1, 4).contiguous().float()}
    m.use_graphs(False)
    eager = _outs(x, m)
    _call_into(m, x, bb, cam, eager)
    torch.cuda.synchronize()
    m.use_graphs(True)
    outs = _outs(x, m)
    for i in range(4):                      # eager, capture + first launch, replay, replay
        for o in outs:
            o.fill_(float("nan"))
        _call_into(m, x, bb, cam, outs)
        torch.cuda.synchronize()
        for a, b in zip(outs, eager):
            assert torch.equal(a, b), (name, i)
    cached, replays = m.graph_stats()
    assert cached == 1 and replays == 2
    m.use_graphs(False)


def test_graph_cache_is_keyed_by_buffers_and_bounded():
    m, x, bbox, cam = _model("tiny_r18")
    bb = bbox.reshape(-1, 4).contiguous().float()
    cam = {"intrinsic": cam["intrinsic"].reshape(-1, 4).contiguous().float()}
    m.use_graphs(False)
    ref = _outs(x)
    _call_into(m, x, bb, cam, ref)
    m.use_graphs(True)
    sets = [_outs(x) for _ in range(10)]
    for outs in sets:                        # 10 distinct buffer sets, each used 3 times: 10 captures, cache holds 8
        for _ in range(3):
            _call_into(m, x, bb, cam, outs)
    torch.cuda.synchronize()
    for outs in sets:
        for a, b in zip(outs, ref):
            assert torch.equal(a, b)
    cached, replays = m.graph_stats()
    assert cached == 8 and replays == 10
    # a bigger batch re-plans the workspace: every cached graph is dropped, results stay right
    x2 = torch.cat([x, x * 0.5], dim=0)
    bb2, it2 = torch.cat([bb, bb]), {"intrinsic": torch.cat([cam["intrinsic"], cam["intrinsic"]])}
    o2 = _outs(x2)
    for _ in range(3):
        _call_into(m, x2, bb2, it2, o2)
    torch.cuda.synchronize()
    assert torch.equal(o2[1][:x.shape[0]], ref[1])
    cached, _ = m.graph_stats()
    assert cached == 1


def test_module_forward_uses_graphs_in_a_steady_loop():
    """The Python drop-in allocates its outputs per call; the caching allocator hands the same blocks back in a steady
    loop, so graphs get captured and replayed without any help from the caller."""
    m, x, bbox, cam = _model("cfg1_r50_v4_128")
    m.use_graphs(True)
    first = m(x, bbox, cam)["joints_cam"].clone()
    for _ in range(12):
        out = m(x, bbox, cam)
    torch.cuda.synchronize()
    assert torch.equal(out["joints_cam"], first)
    cached, replays = m.graph_stats()
    assert cached >= 1 and replays >= 4


def test_refinalising_weights_drops_cached_graphs():
    """The C ABI allows hmv_set_tensor + hmv_finalize_weights again on a live handle.  Finalisation frees the old weight
    buffers, so every cached hipGraph (whose kernel arguments point at them) must go: a forward with the SAME caller
    buffers afterwards has to compute with the NEW weights, bit-identical to an eager run of a fresh engine."""
    import ctypes
    from handmvnet_amd import _lib
    from handmvnet_amd.spec import executed_keys
    from handmvnet_amd.synth import synth_state_dict
    m, x, bbox, cam = _model("tiny_r18")
    bb = bbox.reshape(-1, 4).contiguous().float()
    cam = {"intrinsic": cam["intrinsic"].reshape(-1, 4).contiguous().float()}
    m.use_graphs(True)
    outs = _outs(x)
    for _ in range(3):                      # eager, capture, replay
        _call_into(m, x, bb, cam, outs)
    torch.cuda.synchronize()
    old = [o.clone() for o in outs]
    assert m.graph_stats()[0] == 1
    # new weights into the SAME handle through the raw ABI
    lib = _lib.load()
    h = m._engine(x.shape[-2], x.shape[-1], 0)
    sd2 = synth_state_dict(m.cfg, 4242)
    for k in executed_keys(m.cfg):
        a = np.ascontiguousarray(sd2[k], dtype=np.float32)
        shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
        _lib.check(lib.hmv_set_tensor(h, k.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim), h)
    _lib.check(lib.hmv_finalize_weights(h), h)
    assert m.graph_stats()[0] == 0          # nothing stale left to replay
    for _ in range(3):
        _call_into(m, x, bb, cam, outs)
    torch.cuda.synchronize()
    # reference: a fresh eager engine with the new weights
    from handmvnet_amd import HandMvNet
    m2 = HandMvNet(m.train_params, m.model_params, m.data_params)
    m2.load_state_dict(sd2, strict=True)
    m2.use_graphs(False)
    ref = _outs(x)
    _call_into(m2, x, bb, cam, ref)
    torch.cuda.synchronize()
    for a, b, c in zip(outs, ref, old):
        assert torch.equal(a, b)
    assert not torch.equal(outs[1], old[1])
