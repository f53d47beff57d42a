O=gpurun_out/r02sec; mkdir -p $O
python -m pytest tests -m gpu -q --timeout=1200 -x -k "random" > $O/tests.log 2>&1; rc=$?
tail -6 $O/tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 52; }
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value']); print(json.dumps(d['configs4_fp16'])[:900])"
