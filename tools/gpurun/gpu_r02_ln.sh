O=gpurun_out/r02ln; mkdir -p $O
python -m pytest tests -m gpu -q --timeout=1200 -x > $O/tests.log 2>&1; rc=$?
tail -6 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
python bench.py --workload cfg2 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/cfg2_$rep.json 2> $O/cfg2.err || exit 43
done
python bench.py --workload cfg2 --steps 300 --warmup 30 --no-cpu-baseline > $O/cfg2_ev.json 2> $O/cfg2.err || exit 43
python bench.py --batch 1 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/b1.json 2> $O/b1.err || exit 44
for f in $O/cfg2_*.json $O/b1.json; do python -c "
import json; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'])"; done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload cfg2 --steps 100 --warmup 10 --instrument-every 0 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err || exit 31
f=$(ls -t $O/stats/*/*_kernel_stats.csv | head -1); cp $f $O/kernel_stats_cfg2.csv
find $O -name "*kernel_trace.csv" -size +20M -delete
