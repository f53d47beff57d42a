#!/usr/bin/env python3
"""MFMA-shape A/B for the tall-tile 3x3 kernel: conv_ht_f16 on v_mfma_f32_32x32x16_f16 against the same kernel on
v_mfma_f32_16x16x32_f16, the cfg-3 fp16 shapes (layer3 / layer2 conv2 at 256 frames), RANDOM operands, interleaved rounds in ONE
process (cdna_hip_programming.md section 5.4 rules 24, 25, 28).  Run it under `rocprofv3 --kernel-trace --stats`: the two instantiations
are two kernel symbols, their average durations are the result.  `python tools/mfma_shape_probe.py [rounds] [relu_sparse] [shape 0|1]` (one shape per
process, so that --stats averages one shape per kernel symbol)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handmvnet_amd import _lib  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    sparse = len(sys.argv) > 2 and sys.argv[2] == "1"
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    shapes = (("layer3 conv2 3x3 256->256", 256, 32, 256), ("layer2 conv2 3x3 128->128", 256, 32, 128))
    if len(sys.argv) > 3:
        shapes = shapes[int(sys.argv[3]):int(sys.argv[3]) + 1]
    for name, n, hw, c in shapes:
        x = torch.randn(n, hw, hw, c, generator=g)
        if sparse:
            x = x.clamp_min(0)          # what the network feeds these layers: ReLU outputs, half of them zero
        w = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
        b = torch.randn(c, generator=g)
        xin = x.to(dev)
        out = torch.empty((n, hw, hw, c), device=dev, dtype=torch.float16)
        wc, bc = w.contiguous().numpy(), b.contiguous().numpy()
        res = {}
        for r in range(rounds):
            for sel in (5, 3):   # 5 = 32x32x16 (the partner), 3 = 16x16x32 (the engine's)
                kname = ctypes.c_char_p()
                rc = lib.hmv_op_conv2d_f16(0, xin.data_ptr(), n, hw, hw, c, wc.ctypes.data_as(ctypes.c_void_p), bc.ctypes.data_as(ctypes.c_void_p),
                                           c, 3, 3, 1, 1, None, 1, out.data_ptr(), sel, ctypes.byref(kname), None)
                assert rc == 0, lib.hmv_last_error(None)
                res[sel] = (kname.value.decode(), out.float().abs().mean().item())
        print(name, "sparse" if sparse else "dense", res, flush=True)


if __name__ == "__main__":
    main()
