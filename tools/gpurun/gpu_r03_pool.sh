# round 3: pooled stem A/B (one box): two four-wave workgroups per CU vs one eight-wave workgroup vs conv + pool as two launches
O=gpurun_out/r03pool; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained or full_size_properties[f16]" > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16.json > $O/bench_f16.json 2> $O/bench_f16.err || { tail -20 $O/bench_f16.err; exit 52; }
HMV_STEM_POOL8=1 timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_pool8.json > $O/bench_f16_pool8.json 2> $O/bench_f16_pool8.err || exit 53
HMV_NO_STEMPOOL=1 timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_nopool.json > $O/bench_f16_nopool.json 2> $O/bench_f16_nopool.err || exit 53
python - <<'PY'
import json
for n in ("bench_f16", "bench_f16_pool8", "bench_f16_nopool"):
    d = json.load(open(f"gpurun_out/r03pool/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
for f in ("per_layer_f16", "per_layer_f16_pool8", "per_layer_f16_nopool"):
    pl = json.load(open(f"gpurun_out/r03pool/{f}.json"))
    for r in pl[:1]:
        print(f"{f:24s} {r['layer']:20s} {r['kernel']:40s} {r['avg_ms']*1000:7.1f} us {r['mbytes']:7.0f} MB {r['gbs']:6.0f} GB/s")
PY
