# round 3: fused tail kernels, MFMA learnable-query attention, hr40_lq fixture; then the launch-bound benches
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "fused_tail or lq_attention or hr40_lq or random_configurations or poisoned or reference_fixture or lq" > $O/tests_fuse.log 2>&1; rc=$?
grep -a "amplification\|hr40_lq\|^tiny_r50 \|cfg2s\|r50_lq\|r18_lq" $O/tests_fuse.log | cut -c1-400 | tail -30
tail -5 $O/tests_fuse.log
[ $rc -eq 0 ] || exit $rc
for wl in cfg2 cfg3b1; do
  if [ $wl = cfg2 ]; then A="--workload cfg2"; else A="--workload cfg3 --batch 1"; fi
  timeout -k 10 300 python bench.py $A --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_${wl}_fused.json 2> $O/bench_${wl}_fused.err || exit 54
  HMV_NO_FFFUSE=1 HMV_NO_CHEBFUSE=1 timeout -k 10 300 python bench.py $A --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_${wl}_unfused.json 2> $O/bench_${wl}_unfused.err || exit 55
done
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --instrument-every 0 > $O/bench_f16_fused.json 2> $O/bench_f16_fused.err || exit 56
python - <<'PY'
import json
for n in ("bench_cfg2_fused", "bench_cfg2_unfused", "bench_cfg3b1_fused", "bench_cfg3b1_unfused", "bench_f16_fused"):
    d = json.load(open(f"gpurun_out/r03/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d.get("launches_per_forward"), d["forward"])
PY
