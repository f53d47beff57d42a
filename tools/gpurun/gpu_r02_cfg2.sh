# cfg-2 / batch-1 work: op test of the attention kernel, full GPU suite, then the small-batch benches
O=gpurun_out/r02cfg2; mkdir -p $O
python -m pytest tests -m gpu -q --timeout=1200 -x > $O/tests.log 2>&1; rc=$?
tail -12 $O/tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py --workload cfg2 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/cfg2_noev.json 2> $O/cfg2_noev.err || exit 41
python bench.py --workload cfg2 --steps 300 --warmup 30 --no-cpu-baseline > $O/cfg2.json 2> $O/cfg2.err || exit 42
python bench.py --batch 1 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/b1_noev.json 2> $O/b1.err || exit 43
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg3.json 2> $O/cfg3.err || exit 44
python bench.py --workload cfg2 --dtype f16 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/cfg2_f16_noev.json 2> $O/cfg2_f16.err || exit 45
for f in cfg2_noev cfg2 b1_noev cfg3 cfg2_f16_noev; do python -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['value'])"; done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload cfg2 --steps 100 --warmup 10 --instrument-every 0 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err || exit 31
f=$(ls -t $O/stats/*/*_kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +20M -delete
