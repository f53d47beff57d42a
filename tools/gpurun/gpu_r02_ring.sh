O=gpurun_out/r02ring; mkdir -p $O
python tools/ring_probe.py > $O/probe.txt 2>&1 || exit 40
cat $O/probe.txt
python -m pytest tests -m gpu -q --timeout=1200 -x -k "f16 or fp16 or full_size or half" > $O/tests.log 2>&1; rc=$?
tail -6 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
python bench.py --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline > $O/b_ring_$rep.json 2> $O/b.err || exit 42
HMV_F16_RING=0 python bench.py --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline > $O/b_off_$rep.json 2> $O/b.err || exit 43
done
for f in $O/b_*.json; do python -c "
import json; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d.get('parity_rel_l2_vs_oracle'))"; done
