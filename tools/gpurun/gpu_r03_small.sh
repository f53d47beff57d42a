O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 100 --warmup 10 --per-layer $O/per_layer_cfg2.json > $O/cfg2_pl.json 2> $O/cfg2_pl.err || exit 56
timeout -k 10 300 python bench.py --workload cfg3 --batch 1 --no-cpu-baseline --steps 100 --warmup 10 --per-layer $O/per_layer_b1.json > $O/b1_pl.json 2> $O/b1_pl.err || exit 57
python - <<'PY'
import json
for tag in ("cfg2", "b1"):
    d = json.load(open(f"gpurun_out/r03/{tag}_pl.json")); print(tag, d["ms_per_step"], d["launches_per_forward"])
    rows = json.load(open(f"gpurun_out/r03/per_layer_{tag}.json"))
    print("  sum of conv/GEMM launches", round(sum(r["avg_ms"] for r in rows), 3), len(rows))
    for r in rows: print(f"   {r['layer']:30s} {r['kernel']:38s} {r['avg_ms']*1e3:6.1f} us {r['gflop']:7.2f} GF {r['tflops']:6.1f} TF {r['mbytes']:6.1f} MB")
PY
