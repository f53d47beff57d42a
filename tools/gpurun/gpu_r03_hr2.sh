O=gpurun_out/r03; mkdir -p $O
timeout -k 10 400 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 6 --warmup 2 --per-layer $O/per_layer_hr40_f16.json > $O/hr40_f16.json 2> $O/hr40_f16.err || { tail -5 $O/hr40_f16.err; exit 56; }
timeout -k 10 400 python bench.py --workload hr40 --no-cpu-baseline --steps 4 --warmup 1 --per-layer $O/per_layer_hr40_f32.json > $O/hr40_f32.json 2> $O/hr40_f32.err || { tail -5 $O/hr40_f32.err; exit 57; }
python - <<'PY'
import json, collections
for dt in ("f16", "f32"):
    d = json.load(open(f"gpurun_out/r03/hr40_{dt}.json")); print(dt, d["ms_per_step"])
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:14]: print("   ", k, v)
PY
