O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stream_kernel" > $O/tests_hs.log 2>&1 || { tail -30 $O/tests_hs.log; exit 40; }
tail -1 $O/tests_hs.log
timeout -k 10 200 python tools/hs_probe.py; HMV_NO_HS=1 timeout -k 10 200 python tools/hs_probe.py; timeout -k 10 200 python tools/hs_probe.py
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size and f16" > $O/tests_hs_full.log 2>&1 || { tail -30 $O/tests_hs_full.log; exit 41; }
tail -1 $O/tests_hs_full.log
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_hs.json > $O/bench_f16_hs.json 2> $O/bench_f16_hs.err || exit 56
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_f16_hs.json")); print("f16", d["ms_per_step"], d["launches_per_forward"], d["roofline"]["kernel"], d["roofline"]["frac"])
for r in json.load(open("gpurun_out/r03/per_layer_f16_hs.json")):
    if "hs" in r["kernel"] or r["layer"] in ("stem",): print(r["layer"], r["kernel"], round(r["avg_ms"] * 1e3, 1), "us", round(r["gbs"]), "GB/s")
PY
