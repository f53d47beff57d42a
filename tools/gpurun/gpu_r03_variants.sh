# round 3: conv_stream schedule variants (spread residual DMAs / non-temporal loads), fused tail kernels after the burst-prefetch rewrite
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_tail or poisoned" > $O/tests_tail2.log 2>&1 || { tail -20 $O/tests_tail2.log; exit 40; }
tail -2 $O/tests_tail2.log
for v in 1 2 3 4; do
  HMV_STREAM_VARIANT=$v timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "stream_kernel" > $O/tests_stream_v$v.log 2>&1 || { tail -20 $O/tests_stream_v$v.log; exit 41; }
  tail -1 $O/tests_stream_v$v.log
done
for v in 0 1 2 3 4 0; do
  echo "variant $v"; HMV_STREAM_VARIANT=$v HMV_BENCH_DTYPE=f16 timeout -k 10 300 python tools/stream_probe.py 256 20 2>&1 | head -3
done > $O/stream_variants.txt
cat $O/stream_variants.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2_fused2 -o cfg2 -- python3 bench.py --workload cfg2 --no-cpu-baseline --steps 50 --warmup 5 --instrument-every 0 > $O/prof_cfg2_fused2.json 2> $O/prof_cfg2_fused2.err || exit 60
grep -a "ff_block\|cheb_" $O/prof_cfg2_fused2/cfg2_kernel_stats.csv | cut -c1-160
find gpurun_out/r03 -name "*kernel_trace.csv" -size +5M -delete
for wl in cfg2 cfg3b1; do
  if [ $wl = cfg2 ]; then A="--workload cfg2"; else A="--workload cfg3 --batch 1"; fi
  timeout -k 10 300 python bench.py $A --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_${wl}_fused2.json 2> $O/bench_${wl}_fused2.err || exit 54
  python -c "import json; d=json.load(open('$O/bench_${wl}_fused2.json')); print('$wl', d['ms_per_step'], d['launches_per_forward'])"
done
