# round 3: DUAL conv_stream with the second source's pixel offsets kept per tile: identity tests + per-layer lines (one box)
O=gpurun_out/r03dual2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_kernel or chained or full_size_properties[f16] or fixture" > $O/tests.log 2>&1; rc=$?
tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16.json > $O/bench_f16.json 2> $O/bench_f16.err || { tail -20 $O/bench_f16.err; exit 52; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03dual2/bench_f16.json")); print(d["ms_per_step"], d["value"], d["roofline"]["frac"])
for r in json.load(open("gpurun_out/r03dual2/per_layer_f16.json")):
    if "dual" in r["kernel"] or "downsample" in r["layer"]: print(f"{r['layer']:45s} {r['kernel']:50s} {r['avg_ms']*1000:7.1f} us {r['gbs']:6.0f} GB/s")
PY
