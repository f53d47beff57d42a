#!/usr/bin/env python3
"""Sweeps the r50-paper conv shapes (SURVEY.md Appendix A, N images) through hmv_bench_conv
and prints TFLOP/s per layer class and tile.  Development tool (run on the GPU box)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handmvnet_amd import _lib  # noqa: E402

SHAPES = [  # name, H, Cin, Cout, k, stride, pad, residual, count per forward
    ("stem 7x7/2 4->64", 256, 4, 64, 7, 2, 3, 0, 1),
    ("l1 1x1 64->64", 64, 64, 64, 1, 1, 0, 0, 1),
    ("l1 1x1 256->64", 64, 256, 64, 1, 1, 0, 0, 2),
    ("l1 3x3 64->64", 64, 64, 64, 3, 1, 1, 0, 3),
    ("l1 1x1 64->256 +res", 64, 64, 256, 1, 1, 0, 1, 4),
    ("l2 1x1 256->128", 64, 256, 128, 1, 1, 0, 0, 1),
    ("l2 3x3/2 128->128", 64, 128, 128, 3, 2, 1, 0, 1),
    ("l2 ds 1x1/2 256->512", 64, 256, 512, 1, 2, 0, 0, 1),
    ("l2 1x1 512->128", 32, 512, 128, 1, 1, 0, 0, 3),
    ("l2 3x3 128->128", 32, 128, 128, 3, 1, 1, 0, 3),
    ("l2 1x1 128->512 +res", 32, 128, 512, 1, 1, 0, 1, 4),
    ("l3 1x1 512->256", 32, 512, 256, 1, 1, 0, 0, 1),
    ("l3 ds 1x1 512->1024", 32, 512, 1024, 1, 1, 0, 0, 1),
    ("l3 1x1 1024->256", 32, 1024, 256, 1, 1, 0, 0, 5),
    ("l3 3x3 256->256", 32, 256, 256, 3, 1, 1, 0, 6),
    ("l3 1x1 256->1024 +res", 32, 256, 1024, 1, 1, 0, 1, 6),
    ("pose 1x1 1024->512", 32, 1024, 512, 1, 1, 0, 0, 1),
    ("pose 1x1 512->21", 32, 512, 21, 1, 1, 0, 0, 1),
]


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [-1]
    lib = _lib.load()
    tot_ms = {t: 0.0 for t in tiles}
    tot_fl = 0.0
    rows = []
    for name, H, Cin, Cout, k, st, pad, res, cnt in SHAPES:
        Ho = (H + 2 * pad - k) // st + 1
        kreal = k * k * (3 if Cin == 4 else Cin)
        fl = 2.0 * n_img * Ho * Ho * Cout * kreal
        eb = 2 if os.environ.get("HMV_BENCH_DTYPE") == "f16" else 4
        in_px = n_img * Ho * Ho if (k == 1 and st > 1) else n_img * H * H
        nbytes = eb * (in_px * (3 if Cin == 4 else Cin) + Cout * kreal + n_img * Ho * Ho * Cout * (2 if res else 1))
        line = f"{name:24s} {fl / 1e9:8.1f} GF {nbytes / 1e6:7.1f} MB x{cnt}"
        for t in tiles:
            ms = ctypes.c_float()
            rc = lib.hmv_bench_conv(0, n_img, H, H, Cin, Cout, k, k, st, pad, res, t, 5, ctypes.byref(ms))
            if rc:
                line += f" | t{t}: err {lib.hmv_last_error(None)}"
                continue
            tf = fl / (ms.value * 1e-3) / 1e12
            tot_ms[t] += ms.value * cnt
            line += f" | t{t}: {ms.value:7.3f} ms {tf:6.1f} TF {nbytes / (ms.value * 1e-3) / 1e12:4.2f} TB/s"
            rows.append({"layer": name, "tile": t, "ms": ms.value, "tflops": tf, "count": cnt})
        tot_fl += fl * cnt
        print(line, flush=True)
    for t in tiles:
        print(f"tile {t}: total {tot_ms[t]:.2f} ms per forward -> {tot_fl / (tot_ms[t] * 1e-3) / 1e12:.1f} TFLOP/s over {tot_fl / 1e9:.0f} GFLOP")
    if len(sys.argv) > 3:
        json.dump(rows, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
