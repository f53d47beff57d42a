O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stream_kernel" > $O/tests_hs40.log 2>&1 || { tail -30 $O/tests_hs40.log; exit 40; }
tail -1 $O/tests_hs40.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hr40 or hr64 or (random_configurations and (12 or 13 or 14 or 15 or 16))" > $O/tests_hr.log 2>&1 || { tail -30 $O/tests_hr.log; exit 41; }
tail -1 $O/tests_hr.log
timeout -k 10 400 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2 --per-layer $O/per_layer_hr40_f16_hs.json > $O/bench_hr40_f16_hs.json 2> $O/bench_hr40_f16_hs.err || { tail $O/bench_hr40_f16_hs.err; exit 50; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_hr40_f16_hs.json")); print("hr40 f16", d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"])
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]:
    print(f"   {k:46s} {v}")
PY
