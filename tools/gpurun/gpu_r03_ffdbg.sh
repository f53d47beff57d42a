O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_tail or poisoned" > $O/tests_tail6.log 2>&1 || { tail -20 $O/tests_tail5.log; exit 40; }
tail -1 $O/tests_tail6.log
HMV_FF_DBG=1 python bench.py --workload cfg2 --no-cpu-baseline --steps 3 --warmup 2 --instrument-every 0 2>&1 | grep ff_block | tail -4
for wl in cfg2 cfg3b1; do
  if [ $wl = cfg2 ]; then A="--workload cfg2"; else A="--workload cfg3 --batch 1"; fi
  timeout -k 10 300 python bench.py $A --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_${wl}_fused6.json 2> $O/bench_${wl}_fused6.err || exit 54
  python -c "import json; d=json.load(open('$O/bench_${wl}_fused6.json')); print('$wl', d['ms_per_step'], d['launches_per_forward'])"
done
