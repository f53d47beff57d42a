# round 3: the whole GPU suite, then the bench lines the round's records come from (one box)
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?
tail -12 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16.json > $O/bench_f16.json 2> $O/bench_f16.err || { tail -20 $O/bench_f16.err; exit 52; }
HMV_NO_STREAM=1 timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline > $O/bench_f16_nostream.json 2> $O/bench_f16_nostream.err || exit 53
timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_cfg2.json 2> $O/bench_cfg2.err || exit 54
timeout -k 10 300 python bench.py --workload cfg3 --batch 1 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_b1.json 2> $O/bench_b1.err || exit 55
timeout -k 10 300 python bench.py --workload cfg1 --steps 200 --warmup 20 --instrument-every 0 > $O/bench_cfg1.json 2> $O/bench_cfg1.err || exit 56
python - <<'PY'
import json
for n in ("bench_f16", "bench_f16_nostream", "bench_cfg2", "bench_b1", "bench_cfg1"):
    d = json.load(open(f"gpurun_out/r03/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"), d["forward"])
PY
