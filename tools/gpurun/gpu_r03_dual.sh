O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "(full_size and f16) or stream_kernel or fp16_path" > $O/tests_dual.log 2>&1 || { tail -30 $O/tests_dual.log; exit 40; }
tail -1 $O/tests_dual.log
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_dual.json > $O/bench_f16_dual.json 2> $O/bench_f16_dual.err || exit 56
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_f16_dual.json")); print("f16", d["ms_per_step"], d["launches_per_forward"], d["roofline"]["kernel"], d["roofline"]["frac"])
for r in json.load(open("gpurun_out/r03/per_layer_f16_dual.json")):
    if "stream" in r["kernel"] or "downsample" in r["layer"]: print(f"{r['layer']:28s} {r['kernel']:40s} {r['avg_ms']*1e3:7.1f} us {r['gbs']:6.0f} GB/s")
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]: print(f"   {k:46s} {v}")
PY
