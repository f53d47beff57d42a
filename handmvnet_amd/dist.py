"""Sample sharding across GPUs (one process per GPU, torch.distributed).

Different multi-view samples are independent (the only cross-frame dependency is the
fusion over the V views of ONE sample, handmvnet.py:225-227), so the path shards by sample
with no data-path collective; the only communication is ONE all-gather of the results per step
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).  Payload per rank:
joints_cam [B_local,21,3] + joints_crop_img [B_local,V,21,2] packed into one row of
63 + 42 V floats per sample (~1.6 KB/sample) -> latency bound; heat maps stay local.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

KEYS = ("joints_cam", "joints_crop_img")


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of samples for `rank`; the first total % world ranks get one extra."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class ResultGatherer:
    """One packed all-gather per step.  Both result tensors of a rank travel as ONE [n_max, 63 + 42 V] fp32 block
    (`joints_cam` | `joints_crop_img` per sample row); the send and receive buffers are allocated once, here, so a timed
    loop allocates nothing and issues exactly one collective per step.  Ragged shards (total % world != 0, or fewer
    samples than ranks) are padded to the largest shard: the collective has the same fixed size on every rank."""

    def __init__(self, n_max: int, views: int, device, group=None, dtype=torch.float32):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_max, self.views = int(n_max), int(views)
        self.row = 63 + 42 * self.views
        self.send = torch.zeros((self.n_max, self.row), dtype=dtype, device=device)
        self.recv = torch.empty((self.world * self.n_max, self.row), dtype=dtype, device=device)
        self.collectives = 0   # all-gathers issued so far (the gloo test holds this to one per step)

    def gather(self, local: Dict[str, torch.Tensor], total: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """With equal shards the returned tensors are VIEWS of the receive buffer, valid until the next gather (a timed loop
        allocates nothing); gather_outputs() hands out copies."""
        cam, img = local["joints_cam"], local["joints_crop_img"]
        n = cam.shape[0]
        # raised, never asserted: under `python -O` a rank with a wrong shard would otherwise enter an all-gather of another size
        # and the job would hang instead of failing
        if n > self.n_max or img.shape[0] != n:
            raise ValueError(f"local shard of {n} samples does not fit the gatherer (built for {self.n_max})")
        if n:
            self.send[:n, :63].copy_(cam.reshape(n, 63))
            self.send[:n, 63:].copy_(img.reshape(n, self.row - 63))
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        self.collectives += 1
        blocks = self.recv.view(self.world, self.n_max, self.row)
        if total is None or total == self.n_max * self.world:
            if n != self.n_max:
                raise ValueError("equal shards expected (pass total= for ragged ones)")
            full = blocks.reshape(self.world * self.n_max, self.row)
        else:
            spans = [shard_range(total, r, self.world) for r in range(self.world)]
            if spans[self.rank][1] - spans[self.rank][0] != n:
                raise ValueError(f"local shard of {n} samples does not match shard_range{spans[self.rank]} of total={total}")
            full = torch.cat([blocks[r, :b_ - a_] for r, (a_, b_) in enumerate(spans)], dim=0)
        rows = full.shape[0]
        return {"joints_cam": full[:, :63].reshape(rows, 21, 3), "joints_crop_img": full[:, 63:].reshape(rows, self.views, 21, 2)}


_GATHERERS: Dict[tuple, ResultGatherer] = {}


def _default_group():
    """The live default ProcessGroup object (group=None means "whatever the default is NOW": after destroy_process_group +
    init_process_group that is a NEW object, and a gatherer cached for the old one holds a dead handle)."""
    try:
        return dist.distributed_c10d._get_default_group()
    except Exception:   # noqa: BLE001 -- private API; fall back to "never reuse"
        return None


def gatherer_for(n_max: int, views: int, device, group=None) -> ResultGatherer:
    """The cached ResultGatherer of this (group, shard size, views, device): buffers are allocated on first use only.  A cache hit
    must be for the SAME process-group object (held strongly by the entry, so its id cannot be recycled while the entry lives)."""
    pg = group if group is not None else _default_group()
    key = (id(pg), int(n_max), int(views), str(device))
    hit = _GATHERERS.get(key)
    if hit is not None and pg is not None and hit[0] is pg and hit[1].world == dist.get_world_size(group):
        return hit[1]
    g = ResultGatherer(n_max, views, device, group)
    _GATHERERS[key] = (pg, g)
    return g


def reset_gatherers() -> None:
    """Drops the cached buffers (call after dist.destroy_process_group())."""
    _GATHERERS.clear()


def gather_outputs(local: Dict[str, torch.Tensor], total: Optional[int] = None, group=None,
                   keys=KEYS) -> Dict[str, torch.Tensor]:
    """All-gathers per-sample results along dim 0 in rank order with ONE collective.  `total` (global sample
    count) is needed only for ragged shards.  `keys` selects among the two packed results (joints_cam, joints_crop_img); heat maps
    stay local by design (22 MB per rank), so any other key is refused with a clear error rather than a KeyError mid-gather."""
    keys = tuple(keys)
    bad = [k for k in keys if k not in KEYS]
    if bad:
        raise ValueError(f"gather_outputs gathers {KEYS} only (one packed collective); cannot gather {bad}")
    if not (dist.is_available() and dist.is_initialized()):
        return {k: local[k] for k in keys}
    world = dist.get_world_size(group)
    n_local = local["joints_cam"].shape[0]
    views = local["joints_crop_img"].shape[1]
    if total is None or total == n_local * world:
        n_max = n_local
    else:
        n_max = max(b_ - a_ for a_, b_ in (shard_range(total, r, world) for r in range(world)))
    out = gatherer_for(n_max, views, local["joints_cam"].device, group).gather(local, total)
    return {k: out[k].clone() for k in keys}   # the caller may keep these across later gathers


def forward_sharded(model, x: torch.Tensor, bbox=None, cam_params=None, group=None) -> Dict[str, torch.Tensor]:
    """Every rank passes the GLOBAL batch; each runs its contiguous shard and the results are
    all-gathered, so every rank returns the full joints_cam / joints_crop_img."""
    if not (dist.is_available() and dist.is_initialized()):
        return model(x, bbox, cam_params)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    total = x.shape[0]
    a, b = shard_range(total, rank, world)
    if b > a:
        cam = None if cam_params is None else {k: v[a:b] for k, v in cam_params.items()}
        local = model(x[a:b], None if bbox is None else bbox[a:b], cam)
    else:
        # more ranks than samples: this rank has no work, but it must still enter the collective with zero rows
        # (the engine rejects an empty batch, and the other ranks are already waiting in the all-gather)
        v = x.shape[1]
        local = {"joints_cam": torch.zeros(0, 21, 3, device=x.device, dtype=torch.float32),
                 "joints_crop_img": torch.zeros(0, v, 21, 2, device=x.device, dtype=torch.float32)}
    return gather_outputs(local, total=total, group=group)
