# round 3: conv_hs 64- / 40-channel variants as two four-wave workgroups per CU on 8 x 16 blocks: identity tests, phases, A/B (one box)
O=gpurun_out/r03hs4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_kernel or chained or fixture or full_size_properties[f16]" > $O/tests.log 2>&1; rc=$?
tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
HMV_BENCH_CLOCK=1 HMV_BENCH_PHASES=1 HMV_BENCH_DTYPE=f16 timeout -k 10 200 python tools/hs_probe.py > $O/phases.txt 2>&1 || { tail -5 $O/phases.txt; exit 51; }
grep -v clock $O/phases.txt
for v in new old; do
  if [ $v = old ]; then export HMV_HS_8WAVE=1; fi
  timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline > $O/bench_f16_$v.json 2> $O/bench_f16_$v.err || exit 52
  timeout -k 10 300 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_hr40_f16_$v.json 2> $O/bench_hr40_f16_$v.err || exit 53
done
python - <<'PY'
import json
for n in ("bench_f16_new", "bench_f16_old", "bench_hr40_f16_new", "bench_hr40_f16_old"):
    d = json.load(open(f"gpurun_out/r03hs4/{n}.json"))
    hs = {k: v["ms_per_step"] for k, v in d["kernels"].items() if "conv_hs" in k}
    print(n, d["ms_per_step"], d["value"], hs)
PY
