#!/usr/bin/env python3
"""profiles/rNN_per_layer_<dtype>.json (bench.py --per-layer) -> a markdown table: every conv / GEMM launch of one step with its
algorithmic FLOPs and bytes, both floors (MFMA peak of the dtype it multiplies in, HBM at 8 TB/s) and the fraction of the higher one.
usage: python tools/per_layer_table.py profiles/r02_per_layer_f32.json f32 > profiles/r02_per_layer_f32.md"""
import json, sys
rows = json.load(open(sys.argv[1]))
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
mult = 3.0 if dtype == "f32x3" else 1.0
print(f"| layer | kernel | µs | GFLOP | MB | TFLOP/s | GB/s | MFMA floor µs | HBM floor µs (8 TB/s) | bound | fraction |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
tot = tot_floor = 0.0
for r in rows:
    peak = 2500.0 if "f16" in r["kernel"] else 157.3
    m = mult if "f16" in r["kernel"] else 1.0
    t_m = m * r["gflop"] / peak * 1e3          # us
    t_h = r["mbytes"] / 8000.0 * 1e3           # us
    us = r["avg_ms"] * 1e3
    floor = max(t_m, t_h)
    tot += us; tot_floor += floor
    print(f"| {r['layer']} | `{r['kernel']}` | {us:.0f} | {r['gflop']:.1f} | {r['mbytes']:.0f} | {r['tflops']:.0f} | {r['gbs']:.0f} | {t_m:.0f} | {t_h:.0f} | "
          f"{'hbm' if t_h > t_m else 'mfma'} | {floor / us:.2f} |")
print(f"| **sum** | | **{tot:.0f}** | | | | | | | | **{tot_floor / tot:.2f}** |")
