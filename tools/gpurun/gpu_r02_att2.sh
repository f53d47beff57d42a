# attention kernel variants (2 vs 1 workgroups per CU) + gather / LN fixes: op tests, then benches
O=gpurun_out/r02att2; mkdir -p $O
python -m pytest tests -m gpu -q --timeout=1200 -x -k "attention or random or cfg2 or graph" > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for v in def occ1; do
  if [ $v = occ1 ]; then export HMV_LIB=build/libhandmv_attocc1.so; else unset HMV_LIB; fi
  python bench.py --workload cfg2 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/cfg2_$v.json 2> $O/cfg2_$v.err || exit 41
  python bench.py --batch 1 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/b1_$v.json 2> $O/b1_$v.err || exit 43
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg3_$v.json 2> $O/cfg3_$v.err || exit 44
  for f in cfg2_$v b1_$v cfg3_$v; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['value'])"; done
done
unset HMV_LIB
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload cfg2 --steps 100 --warmup 10 --instrument-every 0 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err || exit 31
f=$(ls -t $O/stats/*/*_kernel_stats.csv | head -1); cp $f $O/kernel_stats_cfg2.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -- python3 bench.py --steps 10 --warmup 3 --instrument-every 0 --no-cpu-baseline > $O/bench_prof3.json 2> $O/prof3.err || exit 32
f=$(ls -t $O/stats3/*/*_kernel_stats.csv | head -1); cp $f $O/kernel_stats_cfg3.csv
HMV_LIB=build/libhandmv_attocc1.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3b -- python3 bench.py --steps 10 --warmup 3 --instrument-every 0 --no-cpu-baseline > $O/bench_prof3b.json 2> $O/prof3b.err || exit 33
f=$(ls -t $O/stats3b/*/*_kernel_stats.csv | head -1); cp $f $O/kernel_stats_cfg3_occ1.csv
find $O -name "*kernel_trace.csv" -size +20M -delete
grep -h attention $O/kernel_stats_cfg2.csv $O/kernel_stats_cfg3.csv $O/kernel_stats_cfg3_occ1.csv | cut -c1-160
