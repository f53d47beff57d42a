"""N>1 path on CPU: world_size-2 gloo processes run the sample-sharded forward
(handmvnet_amd.dist) with the CPU oracle standing in for the per-rank engine, and the gathered
result must equal the single-process full-batch result bit for bit (pure sample sharding:
SURVEY.md section 8(e))."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from handmvnet_amd.dist import forward_sharded, gather_outputs, shard_range
from helpers import load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for total in (1, 2, 3, 7, 32, 256):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class _OracleModel:
    """Stands in for handmvnet_amd.HandMvNet on a CPU-only box (same call signature)."""

    def __init__(self, case):
        from oracle.oracle import Oracle
        self.cfg, _, sd, self.inputs, _ = load_case(case)
        self.oracle = Oracle(self.cfg, sd, "f32")

    def __call__(self, x, bbox, cam):
        out = self.oracle.forward(x.numpy(), bbox.numpy(), cam["intrinsic"].numpy())
        return {k: torch.from_numpy(v) for k, v in out.items()}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, port2, case, batch, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from handmvnet_amd.synth import synth_inputs
        model = _OracleModel(case)
        x, bbox, intr = synth_inputs(model.cfg, batch, 77, 64)
        xt, bt, it = torch.from_numpy(x), torch.from_numpy(bbox), torch.from_numpy(intr)
        # count what the communicator is really asked to do: ONE packed all-gather per step, nothing else
        calls = {"n": 0}
        real = dist.all_gather_into_tensor

        def counted(*a, **k):
            calls["n"] += 1
            return real(*a, **k)
        dist.all_gather_into_tensor = counted
        for forbidden in ("all_gather", "all_reduce", "broadcast", "all_to_all"):
            setattr(dist, forbidden, lambda *a, _n=forbidden, **k: (_ for _ in ()).throw(AssertionError(f"unexpected collective {_n}")))
        out = forward_sharded(model, xt, bt, {"intrinsic": it})
        assert calls["n"] == 1, calls
        # equal-shard fast path: gather a local result directly
        a, b = shard_range(batch, rank, world)
        if b > a:
            local = model(xt[a:b], bt[a:b], {"intrinsic": it[a:b]})
        else:   # batch < world: this rank holds no sample and contributes zero rows to the collective
            local = {"joints_cam": torch.zeros(0, 21, 3), "joints_crop_img": torch.zeros(0, xt.shape[1], 21, 2)}
        g2 = gather_outputs(local, total=batch)
        assert calls["n"] == 2, calls
        assert torch.equal(g2["joints_cam"], out["joints_cam"]) and torch.equal(g2["joints_crop_img"], out["joints_crop_img"])
        # the packed buffers are allocated once and reused: same gatherer, same storage, its own count agrees
        from handmvnet_amd.dist import gatherer_for
        n_max = max(b_ - a_ for a_, b_ in (shard_range(batch, r, world) for r in range(world)))
        g = gatherer_for(n_max, xt.shape[1], xt.device)
        assert g.collectives == 2 and g.send.shape == (n_max, 63 + 42 * xt.shape[1]) and g.recv.shape[0] == world * n_max
        ptr = g.send.data_ptr()
        g3 = gather_outputs(local, total=batch)
        assert g.send.data_ptr() == ptr and g.collectives == 3 and torch.equal(g3["joints_cam"], out["joints_cam"])
        # errors are raised (not asserted: `python -O`), BEFORE any collective is entered
        with pytest.raises(ValueError, match="gathers"):
            gather_outputs(local, total=batch, keys=("joints_cam", "heatmap"))
        wrong = {"joints_cam": torch.zeros(n_max + 1, 21, 3), "joints_crop_img": torch.zeros(n_max + 1, xt.shape[1], 21, 2)}
        with pytest.raises(ValueError, match="does not fit"):
            g.gather(wrong, total=batch)
        assert calls["n"] == 3, calls
        only_cam = gather_outputs(local, total=batch, keys=("joints_cam",))
        assert list(only_cam) == ["joints_cam"] and torch.equal(only_cam["joints_cam"], out["joints_cam"])
        # a new process group (destroy + init) must not be served the old group's gatherer
        dist.destroy_process_group()
        os.environ["MASTER_PORT"] = str(port2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        gnew = gatherer_for(n_max, xt.shape[1], xt.device)
        assert gnew is not g and gnew.collectives == 0
        g4 = gather_outputs(local, total=batch)
        assert torch.equal(g4["joints_cam"], out["joints_cam"])
        if rank == 0:
            q.put({k: v.numpy() for k, v in out.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [1, 2, 3])   # 3 = ragged shards (2 + 1); 1 = fewer samples than ranks (1 + 0)
def test_two_rank_gloo_matches_single_process(batch):
    case, world = "tiny_r50", 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, port2 = _free_port(), _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, port2, case, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from handmvnet_amd.synth import synth_inputs
    model = _OracleModel(case)
    x, bbox, intr = synth_inputs(model.cfg, batch, 77, 64)
    ref = model(torch.from_numpy(x), torch.from_numpy(bbox), {"intrinsic": torch.from_numpy(intr)})
    assert got["joints_cam"].shape == (batch, 21, 3) and got["joints_crop_img"].shape == (batch, 2, 21, 2)
    assert np.array_equal(got["joints_cam"], ref["joints_cam"].numpy())
    assert np.array_equal(got["joints_crop_img"], ref["joints_crop_img"].numpy())
