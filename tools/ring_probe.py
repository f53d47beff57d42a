#!/usr/bin/env python3
"""Development tool (GPU box): fp16 conv shapes on the two-stage 256x256 tile (5) vs the four-stage ring (11)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("l3 1x1 256->1024 +res", 32, 256, 1024, 1, 1, 0, 1), ("l3 1x1 1024->256", 32, 1024, 256, 1, 1, 0, 0),
          ("l3 3x3 256->256", 32, 256, 256, 3, 1, 1, 0), ("l1 1x1 64->256 +res", 64, 64, 256, 1, 1, 0, 1),
          ("l2 1x1 128->512 +res", 32, 128, 512, 1, 1, 0, 1), ("pose 1x1 1024->512", 32, 1024, 512, 1, 1, 0, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from handmvnet_amd import _lib
    lib = _lib.load()
    out = []
    for name, H, Cin, Cout, k, st, pad, res in SHAPES:
        ms = ctypes.c_float()
        rc = lib.hmv_bench_conv(0, 256, H, H, Cin, Cout, k, k, st, pad, res, int(os.environ["PROBE_TILE"]), 20, ctypes.byref(ms))
        out.append(f"{ms.value:.3f}" if rc == 0 else "err")
    print(" ".join(out), flush=True)
else:
    print("setting".ljust(28), " | ".join(n for n, *_ in SHAPES))
    for rep in range(2):
        for tile in ("5", "11"):
            env = dict(os.environ, HMV_BENCH_DTYPE="f16", PROBE_TILE=tile, HMV_F16_RING="0")
            r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
            print(f"f16 tile {tile}".ljust(28), r.stdout.strip() or r.stderr[-300:], flush=True)
