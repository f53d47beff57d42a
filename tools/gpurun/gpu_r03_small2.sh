O=gpurun_out/r03; mkdir -p $O
for t in 1 2 6 9; do
HMV_FORCE_TILE=$t timeout -k 10 300 python bench.py --workload cfg3 --batch 1 --no-cpu-baseline --steps 50 --warmup 10 --per-layer $O/per_layer_b1_t$t.json > $O/b1_t$t.json 2> $O/b1_t$t.err || exit 57
HMV_FORCE_TILE=$t timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 50 --warmup 10 --per-layer $O/per_layer_cfg2_t$t.json > $O/cfg2_t$t.json 2> $O/cfg2_t$t.err || exit 58
done
python - <<'PY'
import json
for tag in ("b1", "cfg2"):
    base = {r["layer"]: r for r in json.load(open(f"gpurun_out/r03/per_layer_{tag}.json"))}
    alts = {t: {r["layer"]: r for r in json.load(open(f"gpurun_out/r03/per_layer_{tag}_t{t}.json"))} for t in (1, 2, 6, 9)}
    print(tag, {t: json.load(open(f"gpurun_out/r03/{tag}_t{t}.json"))["ms_per_step"] for t in (1, 2, 6, 9)})
    for l, r in base.items():
        if r["avg_ms"] < 0.025: continue
        print(f"  {l:28s} {r['kernel'][11:]:22s} {r['avg_ms']*1e3:6.1f} | " + " ".join(f"t{t}:{alts[t][l]['avg_ms']*1e3:6.1f}" for t in (1, 2, 6, 9) if l in alts[t]))
PY
