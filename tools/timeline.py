#!/usr/bin/env python3
"""Per-CU timeline analysis of a conv launch from the diagnostic stamps (development tool)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handmvnet_amd import _lib
lib = _lib.load()
shape = [int(v) for v in sys.argv[1].split(",")]   # N,H,Cin,Cout,k,stride,pad,res
tile = int(sys.argv[2])
N, H, Cin, Cout, k, st, pad, res = shape
os.environ["HMV_BENCH_CLOCK"] = "1"
os.environ["HMV_BENCH_DUMP"] = "/tmp/hmv_dump.bin"
ms = ctypes.c_float()
rc = lib.hmv_bench_conv(0, N, H, H, Cin, Cout, k, k, st, pad, res, tile, 1, ctypes.byref(ms))
assert rc == 0, lib.hmv_last_error(None)
d = np.fromfile("/tmp/hmv_dump.bin", dtype=np.uint64).reshape(-1, 8)
d = d[d[:, 1] > 0]
t0 = d[:, 2].min()
entry, ms0, ms1, end = [(d[:, i] - t0).astype(np.float64) / 100.0 for i in (2, 3, 4, 5)]  # us (100 MHz ticks)
hw, xcc = d[:, 6].astype(np.int64), d[:, 7].astype(np.int64)
cu = (xcc & 0xf) * 4096 + ((hw >> 8) & 0xf) * 256 + ((hw >> 12) & 0x1) * 128 + ((hw >> 13) & 0x7) * 16  # cu_id, sh_id, se_id
cu = (xcc & 0xf) * 100000 + ((hw >> 13) & 0x7) * 1000 + ((hw >> 12) & 0x1) * 100 + ((hw >> 8) & 0xf)
print(f"launch {ms.value:.3f} ms, blocks {len(d)}, distinct CUs {len(np.unique(cu))}")
print(f"prologue (entry->main) median {np.median(ms0 - entry):.2f} us  main median {np.median(ms1 - ms0):.2f} us  epilogue (main end->exit) median {np.median(end - ms1):.2f} us")
print(f"kernel span {end.max():.1f} us")
# per CU: time covered by at least one block's main loop
cov, both = [], []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    ev = sorted([(ms0[i], 1) for i in idx] + [(ms1[i], -1) for i in idx])
    depth, last, c1, c2 = 0, 0.0, 0.0, 0.0
    for t, dlt in ev:
        if depth >= 1: c1 += t - last
        if depth >= 2: c2 += t - last
        depth += dlt; last = t
    cov.append(c1 / end.max()); both.append(c2 / end.max())
print(f"fraction of kernel span with >=1 main loop active per CU: mean {np.mean(cov):.3f}; with >=2: {np.mean(both):.3f}")
c = np.unique(cu)[0]
idx = np.where(cu == c)[0]
order = idx[np.argsort(entry[idx])]
print("CU", c, "timeline (us): entry main_start main_end exit")
for i in order[:10]:
    print(f"   {entry[i]:8.1f} {ms0[i]:8.1f} {ms1[i]:8.1f} {end[i]:8.1f}")
