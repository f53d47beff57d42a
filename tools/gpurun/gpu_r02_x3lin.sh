# q/k/v projections on the split kernels in the fp16 / f32x3 modes: tests of those modes, then A/B
O=gpurun_out/r02x3; mkdir -p $O
python -m pytest tests -m gpu -q --timeout=1200 -k "f16 or fp16 or x3 or split or full_size or random or frames or graphs" > $O/tests.log 2>&1; rc=$?
tail -8 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for dt in f16 f32x3; do
  python bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --per-layer $O/pl_$dt.json > $O/b_$dt.json 2> $O/b_$dt.err || exit 42
  HMV_NO_X3LIN=1 python bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline > $O/b_${dt}_off.json 2> $O/b_${dt}_off.err || exit 43
  python -c "
import json
a=json.load(open('$O/b_$dt.json')); b=json.load(open('$O/b_${dt}_off.json'))
print('$dt', 'x3lin', a['ms_per_step'], 'off', b['ms_per_step'], a.get('parity_rel_l2_vs_oracle'), b.get('parity_rel_l2_vs_oracle'))"
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02x3/pl_f16.json'))
rows=d if isinstance(d,list) else d.get('layers',d)
for r in rows:
    if 'qkv' in r.get('label','') or 'fusion' in r.get('label',''):
        print(r)
PY
