#!/usr/bin/env python3
"""Robustness smoke for the size-gated kernels (conv_ht / conv_hs / conv_stream / conv_rds): frame sizes that do and do not tile into
their blocks, in fp32 and fp16 -- finite outputs, fp16 close to fp32, and a sample alone == the same sample inside a batch (bit for bit).
Development tool; `python tools/size_smoke.py`."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import params  # noqa: E402
from handmvnet_amd import HandMvNet  # noqa: E402
from handmvnet_amd.spec import config_from_params  # noqa: E402
from handmvnet_amd.synth import synth_inputs, synth_state_dict  # noqa: E402


def run(bt, ch, V, B, size, dtype):
    tp, mp, dp = params(bt, ch, V, B, size)
    cfg = config_from_params(tp, mp, dp)
    m = HandMvNet(tp, mp, dp)
    m.load_state_dict(synth_state_dict(cfg, 3), strict=True)
    dev = torch.device("cuda:0")
    m.to(dev).eval()
    m.freeze()
    if dtype == "f16":
        m.half()
    x, bbox, intr = synth_inputs(cfg, B, 11, size)
    xt, bt_, it = (torch.from_numpy(a).to(dev) for a in (x, bbox, intr))
    out = m(xt, bt_, {"intrinsic": it})
    one = m(xt[:1].contiguous(), bt_[:1].contiguous(), {"intrinsic": it[:1].contiguous()})
    torch.cuda.synchronize()
    jc = out["joints_cam"].float().cpu()
    assert torch.isfinite(jc).all(), (bt, size, dtype)
    same = torch.equal(one["joints_cam"].cpu()[0], out["joints_cam"].cpu()[0])
    return jc, same


def main():
    for bt, ch, V, B, size in (("50_paper", [1024], 4, 24, 256), ("50_paper", [1024], 2, 6, 512), ("50_paper", [1024], 4, 20, 192),
                               ("50_paper", [1024], 4, 16, 320), ("w40", [40, 80, 160, 320], 4, 24, 256), ("w40", [40, 80, 160, 320], 2, 8, 320),
                               ("18", [256, 128, 64], 4, 16, 256)):
        r32, s32 = run(bt, ch, V, B, size, "f32")
        r16, s16 = run(bt, ch, V, B, size, "f16")
        rel = ((r16 - r32).norm() / r32.norm()).item()
        print(f"{bt:9s} V{V} B{B:3d} {size}x{size}: fp32 batch-independent {s32}, fp16 batch-independent {s16}, fp16 vs fp32 rel-L2 {rel:.3e}", flush=True)
        assert s32 and s16


if __name__ == "__main__":
    main()
