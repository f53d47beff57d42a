O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stream32" > $O/tests_s32.log 2>&1 || { tail -30 $O/tests_s32.log; exit 40; }
tail -1 $O/tests_s32.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --per-layer $O/per_layer_f32_k64.json > $O/bench_f32_k64.json 2> $O/bench_f32_k64.err || exit 56
python - <<PY
import json
d = json.load(open("gpurun_out/r03/bench_f32_k64.json")); print("f32", d["ms_per_step"])
print("   ", [(r['layer'], r['kernel'], round(r['avg_ms']*1e3)) for r in json.load(open("gpurun_out/r03/per_layer_f32_k64.json")) if r['layer'] in ('layer1.0.conv1','stem')])
PY
