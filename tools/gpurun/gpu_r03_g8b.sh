O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stream_kernel_is_bit_identical" > $O/tests_g8.log 2>&1 || { tail -40 $O/tests_g8.log; exit 40; }
tail -1 $O/tests_g8.log
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_g8.json > $O/bench_f16_g8.json 2> $O/bench_f16_g8.err || exit 56
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_f16_g8.json")); print("f16", d["ms_per_step"])
for r in json.load(open("gpurun_out/r03/per_layer_f16_g8.json")):
    if "gemm8" in r["kernel"]: print("  ", r["layer"], r["kernel"], round(r["avg_ms"] * 1e3, 1), "us", round(r["gbs"]), "GB/s")
PY
