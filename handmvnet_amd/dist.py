"""Sample sharding across GPUs (one process per GPU, torch.distributed).

Different multi-view samples are independent (the only cross-frame dependency is the
fusion over the V views of ONE sample, handmvnet.py:225-227), so the path shards by sample
with no data-path collective; the only communication is one all-gather of the results
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).  Payload per rank:
joints_cam [B_local,21,3] + joints_crop_img [B_local,V,21,2] ~ 1.6 KB/sample -> latency
bound; heat maps stay local.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of samples for `rank`; the first total % world ranks get one extra."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_outputs(local: Dict[str, torch.Tensor], total: Optional[int] = None, group=None,
                   keys=("joints_cam", "joints_crop_img")) -> Dict[str, torch.Tensor]:
    """All-gathers per-sample results along dim 0 in rank order.  `total` (global sample
    count) is needed only for ragged shards; equal shards use one all_gather_into_tensor."""
    if not (dist.is_available() and dist.is_initialized()):
        return {k: local[k] for k in keys}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out = {}
    for k in keys:
        t = local[k].contiguous()
        n_local = t.shape[0]
        if total is None or total == n_local * world:
            full = torch.empty((n_local * world,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(full, t, group=group)
        else:
            # ragged shards: pad every shard to the largest one so that ONE fixed-size all-gather does
            # (works on RCCL and gloo alike), then keep each rank's valid rows
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[rank][1] - spans[rank][0] == n_local, "local shard does not match shard_range"
            n_max = max(b_ - a_ for a_, b_ in spans)
            padded = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            padded[:n_local] = t
            buf = torch.empty((n_max * world,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(buf, padded, group=group)
            full = torch.cat([buf[r * n_max:r * n_max + (b_ - a_)] for r, (a_, b_) in enumerate(spans)], dim=0)
        out[k] = full
    return out


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
