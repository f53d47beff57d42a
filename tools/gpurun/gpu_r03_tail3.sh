O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_tail or poisoned or stream_kernel" > $O/tests_tail4.log 2>&1 || { tail -20 $O/tests_tail3.log; exit 40; }
tail -2 $O/tests_tail4.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2_fused4 -o cfg2 -- python3 bench.py --workload cfg2 --no-cpu-baseline --steps 50 --warmup 5 --instrument-every 0 > $O/prof_cfg2_fused4.json 2> $O/prof_cfg2_fused4.err || exit 60
grep -a "ff_block\|cheb_" $O/prof_cfg2_fused4/cfg2_kernel_stats.csv | cut -c1-130
find gpurun_out/r03 -name "*kernel_trace.csv" -size +5M -delete
for wl in cfg2 cfg3b1; do
  if [ $wl = cfg2 ]; then A="--workload cfg2"; else A="--workload cfg3 --batch 1"; fi
  timeout -k 10 300 python bench.py $A --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_${wl}_fused4.json 2> $O/bench_${wl}_fused4.err || exit 54
  python -c "import json; d=json.load(open('$O/bench_${wl}_fused4.json')); print('$wl', d['ms_per_step'], d['launches_per_forward'])"
done
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --instrument-every 0 > $O/bench_f16_4.json 2> $O/bench_f16_4.err || exit 56
python -c "import json; d=json.load(open('$O/bench_f16_4.json')); print('f16', d['ms_per_step'], d['launches_per_forward'])"
