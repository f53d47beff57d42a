# what the driver runs at round end: the GPU suite, smoke(), the default bench line
O=gpurun_out/r03check; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?
tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 51; }
tail -2 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 52; }
python -c "
import json; d=json.load(open('$O/bench_default.json')); print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','dtype','scaling','vs_baseline')}); print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline']['kind'])"
