#!/usr/bin/env python3
"""Timing anatomy of conv_gemm8.hip's ring kernel on the cfg-3 fp16 1x1 shapes (in-kernel stamps, HMV_BENCH_CLOCK).  Development tool."""
import ctypes
import os
import struct
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HMV_BENCH_DTYPE", "f16")
from handmvnet_amd import _lib  # noqa: E402


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else float("nan")


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lib = _lib.load()
    for name, H, Cin, Cout in (("layer3 conv1 1x1 1024->256", 32, 1024, 256), ("pose_net.0 1x1 1024->512", 32, 1024, 512), ("layer3.0 conv1 512->256", 32, 512, 256)):
        for clock in (0, 1):
            dump = os.path.join(tempfile.gettempdir(), "g8_dump.bin")
            if clock:
                os.environ["HMV_BENCH_CLOCK"] = "1"
                os.environ["HMV_BENCH_DUMP"] = dump
            else:
                os.environ.pop("HMV_BENCH_CLOCK", None)
            ms = ctypes.c_float()
            rc = lib.hmv_bench_conv(0, n_img, H, H, Cin, Cout, 1, 1, 1, 0, 0, -1, 20, ctypes.byref(ms))
            if rc:
                print(name, "error", lib.hmv_last_error(None))
                break
            nbytes = 2.0 * n_img * H * H * (Cin + Cout)
            if not clock:
                print(f"{name:28s} {ms.value * 1e3:8.1f} us  {nbytes / (ms.value * 1e-3) / 1e12:5.2f} TB/s", flush=True)
                continue
            raw = open(dump, "rb").read()
            rows = [struct.unpack_from("<8Q", raw, 64 * i) for i in range(len(raw) // 64)]
            rows = [r for r in rows if r[1] > 0]
            if not rows:
                print("   (no stamps: not the ring kernel)")
                continue
            t00 = min(r[2] for r in rows)
            print(f"   stamps: {len(rows)} workgroups; prologue {med([r[3] - r[2] for r in rows]) / 100:.2f} us, main loop {med([r[4] - r[3] for r in rows]) / 100:.2f} us"
                  f" ({med([r[0] for r in rows]):.0f} cycles, {med([r[0] / r[1] * 0.1 for r in rows]):.2f} GHz), epilogue {med([r[5] - r[4] for r in rows]) / 100:.2f} us;"
                  f" last exit {max(r[5] for r in rows) / 100 - t00 / 100:.1f} us after the first entry", flush=True)
            starts = sorted(r[2] - t00 for r in rows)
            print("   entry times (us) of workgroups 0, 255, 256, 511, 512, 767, 768, 1023:", [round(starts[i] / 100, 1) for i in (0, 255, 256, 511, 512, 767, 768, len(starts) - 1) if i < len(starts)])


if __name__ == "__main__":
    main()
