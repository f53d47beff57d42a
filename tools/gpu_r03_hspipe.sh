# round 3: conv_hs 64-channel variant on the column swizzle + pipelined fragment reads: identity tests, phases, bench lines (one box)
O=gpurun_out/r03hspipe; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_kernel or chained or full_size_properties or fixture" > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
HMV_BENCH_CLOCK=1 HMV_BENCH_PHASES=1 HMV_BENCH_DTYPE=f16 timeout -k 10 200 python tools/hs_probe.py > $O/phases.txt 2>&1 || { tail -5 $O/phases.txt; exit 51; }
grep -v clock $O/phases.txt
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16.json > $O/bench_f16.json 2> $O/bench_f16.err || { tail -20 $O/bench_f16.err; exit 52; }
timeout -k 10 300 python bench.py --workload cfg2 --dtype f16 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_cfg2_f16.json 2> $O/bench_cfg2_f16.err || exit 55
python - <<'PY'
import json
for n in ("bench_f16", "bench_cfg2_f16"):
    d = json.load(open(f"gpurun_out/r03hspipe/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
pl = json.load(open("gpurun_out/r03hspipe/per_layer_f16.json"))
for r in pl[:8]:
    print(f"{r['layer']:45s} {r['kernel']:50s} {r['avg_ms']*1000:7.1f} us")
PY
