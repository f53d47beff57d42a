O=gpurun_out/r02ph; mkdir -p $O
python -m pytest tests -m gpu -q --timeout=1200 -x -k "r18 or r34 or 18 or 34 or random or cfg2 or frames or graph" > $O/tests.log 2>&1; rc=$?
tail -6 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
python bench.py --workload cfg2 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/cfg2_on_$rep.json 2> $O/cfg2.err || exit 43
HMV_NO_PHASEMERGE=1 python bench.py --workload cfg2 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/cfg2_off_$rep.json 2> $O/cfg2.err || exit 44
done
python bench.py --workload cfg2 --dtype f16 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/cfg2_f16_on.json 2> $O/cfg2.err || exit 45
HMV_NO_PHASEMERGE=1 python bench.py --workload cfg2 --dtype f16 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/cfg2_f16_off.json 2> $O/cfg2.err || exit 46
for f in $O/cfg2_*.json; do python -c "
import json; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'])"; done
