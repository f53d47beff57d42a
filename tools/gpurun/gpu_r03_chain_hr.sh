# round 3: chain in HRNet's layer1 -- identity test + A/B (one box)
O=gpurun_out/r03chainhr; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained or fixture or poisoned" > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_hr40_f16.json 2> $O/bench_hr40_f16.err || { tail -20 $O/bench_hr40_f16.err; exit 52; }
HMV_NO_CHAIN=1 timeout -k 10 300 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_hr40_f16_nochain.json 2> $O/bench_hr40_f16_nochain.err || exit 53
python - <<'PY'
import json
for n in ("bench_hr40_f16", "bench_hr40_f16_nochain"):
    d = json.load(open(f"gpurun_out/r03chainhr/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
PY
