# round 3: cross-layer launches of the fp16 backbone (conv_stream chain, pooled stem): the GPU suite, then A/B bench lines on one box
O=gpurun_out/r03chain; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?
tail -8 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16.json > $O/bench_f16.json 2> $O/bench_f16.err || { tail -20 $O/bench_f16.err; exit 52; }
HMV_NO_CHAIN=1 HMV_NO_STEMPOOL=1 timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline > $O/bench_f16_nochain.json 2> $O/bench_f16_nochain.err || exit 53
HMV_NO_STEMPOOL=1 timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline > $O/bench_f16_nopool.json 2> $O/bench_f16_nopool.err || exit 53
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --instrument-every 0 > $O/bench_f16_noev.json 2> $O/bench_f16_noev.err || exit 54
timeout -k 10 300 python bench.py --workload cfg2 --dtype f16 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_cfg2_f16.json 2> $O/bench_cfg2_f16.err || exit 55
python - <<'PY'
import json
for n in ("bench_f16", "bench_f16_nochain", "bench_f16_nopool", "bench_f16_noev", "bench_cfg2_f16"):
    d = json.load(open(f"gpurun_out/r03chain/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
pl = json.load(open("gpurun_out/r03chain/per_layer_f16.json"))
for r in pl[:10]:
    print(f"{r['layer']:45s} {r['kernel']:50s} {r['avg_ms']*1000:7.1f} us {r['mbytes']:7.0f} MB {r['gbs']:6.0f} GB/s")
print("conv sum", sum(r['avg_ms'] for r in pl))
PY
