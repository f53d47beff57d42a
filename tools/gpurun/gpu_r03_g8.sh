# round 3: conv_gemm8 parity + A/B; fused tail tests; launch-bound benches
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "stream_kernel or fused_tail or lq_attention or random_configurations or full_size" > $O/tests_g8.log 2>&1; rc=$?
grep -a "amplification" $O/tests_g8.log | cut -c1-600 | tail -5
tail -5 $O/tests_g8.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_g8.json > $O/bench_f16_g8.json 2> $O/bench_f16_g8.err || { tail $O/bench_f16_g8.err; exit 56; }
HMV_NO_GEMM8=1 timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_nog8.json > $O/bench_f16_nog8.json 2> $O/bench_f16_nog8.err || exit 57
for wl in cfg2 cfg3b1; do
  if [ $wl = cfg2 ]; then A="--workload cfg2"; else A="--workload cfg3 --batch 1"; fi
  timeout -k 10 300 python bench.py $A --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_${wl}_fused.json 2> $O/bench_${wl}_fused.err || exit 54
  HMV_NO_FFFUSE=1 HMV_NO_CHEBFUSE=1 timeout -k 10 300 python bench.py $A --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_${wl}_unfused.json 2> $O/bench_${wl}_unfused.err || exit 55
done
python - <<'PY'
import json
for n in ("bench_f16_g8", "bench_f16_nog8", "bench_cfg2_fused", "bench_cfg2_unfused", "bench_cfg3b1_fused", "bench_cfg3b1_unfused"):
    d = json.load(open(f"gpurun_out/r03/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d.get("launches_per_forward"), d["roofline"]["kernel"], d["roofline"]["frac"])
a = {r["layer"]: r for r in json.load(open("gpurun_out/r03/per_layer_f16_g8.json"))}
b = {r["layer"]: r for r in json.load(open("gpurun_out/r03/per_layer_f16_nog8.json"))}
for k in a:
    if a[k]["kernel"] != b[k]["kernel"]:
        print(f"{k:28s} {b[k]['kernel']:36s} {b[k]['avg_ms']*1e3:7.1f} us -> {a[k]['kernel']:36s} {a[k]['avg_ms']*1e3:7.1f} us  {a[k]['tflops']:.0f} TF")
PY
