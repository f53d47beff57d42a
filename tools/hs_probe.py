#!/usr/bin/env python3
"""A/B of conv_hs.hip (weight-stationary halo-streaming 3x3 / stem) against conv_igemm on the cfg-3 fp16 shapes it covers.
`HMV_BENCH_DTYPE=f16 python tools/hs_probe.py` and the same with HMV_NO_HS=1 (one box).  Development tool."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handmvnet_amd import _lib  # noqa: E402


def main():
    os.environ.setdefault("HMV_BENCH_DTYPE", "f16")
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lib = _lib.load()
    for name, H, Cin, Cout, k, pad, res in (("l1 conv2 3x3 64->64", 64, 64, 64, 3, 1, 0), ("r18 l1 3x3 64->64 +res", 64, 64, 64, 3, 1, 1),
                                            ("w40 b0 3x3 40->40 +res", 64, 40, 40, 3, 1, 1), ("w40 b1 3x3 80->80", 32, 80, 80, 3, 1, 0),
                                            ("w40 b1 3x3 80->80 +res", 32, 80, 80, 3, 1, 1)):
        ms = ctypes.c_float()
        rc = lib.hmv_bench_conv(0, n_img, H, H, Cin, Cout, k, k, 1, pad, res, -1, 20, ctypes.byref(ms))
        if rc:
            print(name, "error", lib.hmv_last_error(None))
            continue
        nbytes = 2.0 * (n_img * H * H * (Cin + Cout * (2 if res else 1)) + Cin * Cout * k * k)
        print(f"{name:26s} {ms.value * 1e3:8.1f} us  {nbytes / (ms.value * 1e-3) / 1e12:5.2f} TB/s  (HMV_NO_HS={os.environ.get('HMV_NO_HS', '')})", flush=True)


if __name__ == "__main__":
    main()
