# round 3: conv_hs pipelined fragment reads (64-channel column swizzle; 40-channel pixels with two-tap steps): identity tests, phases, bench lines (one box)
O=gpurun_out/r03hspipe; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_kernel or chained or fixture" > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
HMV_BENCH_CLOCK=1 HMV_BENCH_PHASES=1 HMV_BENCH_DTYPE=f16 timeout -k 10 200 python tools/hs_probe.py > $O/phases.txt 2>&1 || { tail -5 $O/phases.txt; exit 51; }
grep -v clock $O/phases.txt
timeout -k 10 300 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_hr40_f16.json 2> $O/bench_hr40_f16.err || { tail -20 $O/bench_hr40_f16.err; exit 52; }
python - <<'PY'
import json
for n in ("bench_hr40_f16",):
    d = json.load(open(f"gpurun_out/r03hspipe/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]: print("   ", k, v)
PY
