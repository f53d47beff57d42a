#!/usr/bin/env python3
"""Development tool: per-kernel L2 (TCC) hit / miss / request counts from rocprofv3 --pmc passes.
usage: python tools/l2_hits.py <dir with *_counter_collection.csv> [...]"""
import csv, glob, os, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
def short(n):
    m = re.search(r"conv_igemm<(_Float16|float), (\d+), (\d+), \d+, \d+, (\d), (true|false), (\d+)", n)
    if m:
        return f"conv_{'f16' if m.group(1)=='_Float16' else 'f32'}<{m.group(2)}x{m.group(3)},{['taps','1x1','dense','halo'][int(m.group(4))]}{',generic' if m.group(5)=='true' else ''},kb{m.group(6)}>"
    m = re.search(r"conv_igemmIDF16_Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d)ELb(\d)ELi(\d+)", n)
    if m:
        return f"conv_f16<{m.group(1)}x{m.group(2)},{['taps','1x1','dense','halo'][int(m.group(3))]}{',generic' if m.group(4)=='1' else ''},kb{m.group(5)}>"
    return n[:60]
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
names = sorted({c for k in agg for c in agg[k]})
print("kernel".ljust(44), " ".join(n.rjust(22) for n in names), "  hit%")
for k in sorted(agg, key=lambda k: -agg[k].get("TCC_REQ_sum", 0)):
    row = [agg[k].get(n, 0.0) / max(calls[k].get(n, 1), 1) for n in names]
    h, m = agg[k].get("TCC_HIT_sum", 0), agg[k].get("TCC_MISS_sum", 0)
    print(k.ljust(44), " ".join(f"{v:22.0f}" for v in row), f"  {100*h/max(h+m,1):5.1f}")
