#!/usr/bin/env python3
"""Sweeps HRNet-w40's branch conv shapes (256 images of 256x256: branches at 64^2 / 32^2 / 16^2 / 8^2) through
hmv_bench_conv for a list of tiles.  Development tool (run on the GPU box).
    python tools/hr_sweep.py [n_img] [tiles, comma separated; -1 = engine's choice]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handmvnet_amd import _lib  # noqa: E402

SHAPES = [  # name, H, Cin, Cout, k, stride, pad, residual
    ("b0 3x3 40->40 +res", 64, 40, 40, 3, 1, 1, 1),
    ("b1 3x3 80->80 +res", 32, 80, 80, 3, 1, 1, 1),
    ("b2 3x3 160->160 +res", 16, 160, 160, 3, 1, 1, 1),
    ("b3 3x3 320->320 +res", 8, 320, 320, 3, 1, 1, 1),
    ("fuse 1x1 80->40", 32, 80, 40, 1, 1, 0, 0),
    ("fuse 3x3/2 40->80", 64, 40, 80, 3, 2, 1, 0),
]
NAMES = {-1: "auto", 0: "128x32", 1: "128x64", 2: "128x128", 3: "256x128", 4: "128x256", 5: "256x256", 9: "64x64"}


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1]
    lib = _lib.load()
    for name, H, Cin, Cout, k, st, pad, res in SHAPES:
        Ho = (H + 2 * pad - k) // st + 1
        fl = 2.0 * n_img * Ho * Ho * Cout * k * k * Cin
        line = f"{name:24s} {fl / 1e9:7.1f} GF"
        for t in tiles:
            ms = ctypes.c_float()
            rc = lib.hmv_bench_conv(0, n_img, H, H, Cin, Cout, k, k, st, pad, res, t, 10, ctypes.byref(ms))
            line += f" | {NAMES.get(t, t)}: " + (f"{ms.value:6.3f} ms {fl / (ms.value * 1e-3) / 1e12:6.1f} TF" if rc == 0 else "err")
        print(line, flush=True)


if __name__ == "__main__":
    main()
