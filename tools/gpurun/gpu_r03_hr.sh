O=gpurun_out/r03; mkdir -p $O
timeout -k 10 400 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2 --per-layer $O/per_layer_hr40_f16.json > $O/bench_hr40_f16.json 2> $O/bench_hr40_f16.err || { tail $O/bench_hr40_f16.err; exit 50; }
timeout -k 10 400 python bench.py --workload hr40 --no-cpu-baseline --steps 6 --warmup 2 --per-layer $O/per_layer_hr40_f32.json > $O/bench_hr40_f32.json 2> $O/bench_hr40_f32.err || exit 51
python - <<'PY'
import json, collections
for dt in ("f16", "f32"):
    d = json.load(open(f"gpurun_out/r03/bench_hr40_{dt}.json")); print(dt, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"])
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:12]:
        print(f"   {k:46s} {v}")
PY
