O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tall_tile" > $O/tests_ht.log 2>&1 || { tail -40 $O/tests_ht.log; exit 40; }
tail -1 $O/tests_ht.log
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_ht.json > $O/bench_f16_ht.json 2> $O/bench_f16_ht.err || { tail -5 $O/bench_f16_ht.err; exit 56; }
HMV_NO_HT=1 timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_noht.json > $O/bench_f16_noht.json 2> $O/bench_f16_noht.err || exit 57
python - <<'PY'
import json
for tag in ("ht", "noht"):
    d = json.load(open(f"gpurun_out/r03/bench_f16_{tag}.json")); print(tag, d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
    for r in json.load(open(f"gpurun_out/r03/per_layer_f16_{tag}.json")):
        if r["layer"].endswith(".conv2") and (r["layer"].startswith("layer3") or r["layer"].startswith("layer2")): print("  ", r["layer"], r["kernel"], round(r["avg_ms"] * 1e3, 1), "us")
PY
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "f16" > $O/tests_ht_f16.log 2>&1 || { tail -40 $O/tests_ht_f16.log; exit 41; }
tail -1 $O/tests_ht_f16.log
