"""The collective of the sharded forward on real hardware: RCCL ("nccl" backend) with the world this one-GPU box
allows (1 rank), plus the ragged-shard padding path.  The N > 1 partitioning itself is covered by the 2-rank gloo
test on the CPU (tests/test_dist_gloo.py); the driver's 8-GPU run exercises both together."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from helpers import load_case

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_forward_sharded_over_rccl_world_of_one():
    from handmvnet_amd import HandMvNet
    from handmvnet_amd.dist import forward_sharded, gather_outputs
    cfg, (tp, mp, dp), sd, (x, bbox, intr), fx = load_case("r18_frozen_nosin")      # B = 2
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(sd, strict=True)
    m.to("cuda").eval()
    dev = torch.device("cuda:0")
    xt, bt, cam = torch.from_numpy(x).to(dev), torch.from_numpy(bbox).to(dev), {"intrinsic": torch.from_numpy(intr).to(dev)}
    plain = m(xt, bt, cam)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        out = forward_sharded(m, xt, bt, cam)
        torch.cuda.synchronize()
        assert set(out) == {"joints_cam", "joints_crop_img"} and out["joints_cam"].is_cuda
        assert torch.equal(out["joints_cam"], plain["joints_cam"])
        assert torch.equal(out["joints_crop_img"], plain["joints_crop_img"])
        # the padded fixed-size gather used for ragged shards (total != n_local * world is what selects it)
        part = {k: v[:1].contiguous() for k, v in plain.items()}
        g = gather_outputs(part, total=1)
        assert torch.equal(g["joints_cam"], plain["joints_cam"][:1])
        assert np.allclose(out["joints_cam"].cpu().numpy(), fx["joints_cam"], rtol=0, atol=1e-3 * np.abs(fx["joints_cam"]).max())
    finally:
        dist.destroy_process_group()
