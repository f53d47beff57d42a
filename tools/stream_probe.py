#!/usr/bin/env python3
"""A/B of the persistent weight-stationary 1x1 kernel (conv_stream.hip) against conv_igemm on the cfg-3 fp16 shapes it covers.
Run twice on one box: `HMV_BENCH_DTYPE=f16 python tools/stream_probe.py` and the same with HMV_NO_STREAM=1.  Development tool."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handmvnet_amd import _lib  # noqa: E402

SHAPES = [  # name, H, Cin, Cout, residual, launches per forward
    ("l1 conv3 64->256 +res", 64, 64, 256, 1, 2),
    ("l2 conv3 128->512 +res", 32, 128, 512, 1, 3),
    ("l3 conv3 256->1024 +res", 32, 256, 1024, 1, 5),
    ("l1 conv1 256->64", 64, 256, 64, 0, 2),
    ("l2 conv1 512->128", 32, 512, 128, 0, 3),
    ("l2.0 conv1 256->128", 64, 256, 128, 0, 1),
]


def main():
    os.environ.setdefault("HMV_BENCH_DTYPE", "f16")
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lib = _lib.load()
    total = 0.0
    for name, H, Cin, Cout, res, cnt in SHAPES:
        ms = ctypes.c_float()
        rc = lib.hmv_bench_conv(0, n_img, H, H, Cin, Cout, 1, 1, 1, 0, res, -1, iters, ctypes.byref(ms))
        if rc:
            print(name, "error", lib.hmv_last_error(None))
            continue
        nbytes = 2.0 * (n_img * H * H * (Cin + Cout * (2 if res else 1)) + Cin * Cout)
        print(f"{name:26s} {ms.value * 1e3:8.1f} us  {nbytes / (ms.value * 1e-3) / 1e12:5.2f} TB/s  x{cnt}", flush=True)
        total += ms.value * cnt
    print(f"sum over a forward: {total * 1e3:.0f} us  (HMV_NO_STREAM={os.environ.get('HMV_NO_STREAM', '')})")


if __name__ == "__main__":
    main()
