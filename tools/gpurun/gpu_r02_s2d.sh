# space-to-depth stem: full GPU suite, then benches in the three modes + cfg2
O=gpurun_out/r02s2d; mkdir -p $O
python -m pytest tests -m gpu -q --timeout=1200 -x > $O/tests.log 2>&1; rc=$?
tail -12 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for dt in f32 f16 f32x3; do
  python bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --per-layer $O/pl_$dt.json > $O/b_$dt.json 2> $O/b_$dt.err || exit 42
done
python bench.py --workload cfg2 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/cfg2_noev.json 2> $O/cfg2.err || exit 43
python bench.py --batch 1 --steps 300 --warmup 30 --instrument-every 0 --no-cpu-baseline > $O/b1_noev.json 2> $O/b1.err || exit 44
python - <<'PY'
import json
O='gpurun_out/r02s2d'
for dt in ['f32','f16','f32x3']:
    b=json.load(open(f'{O}/b_{dt}.json')); pl=json.load(open(f'{O}/pl_{dt}.json'))
    st=[r for r in pl if r['layer']=='stem'][0]
    print(dt, b['ms_per_step'], b.get('parity_rel_l2_vs_oracle'), 'stem', round(st['avg_ms']*1000,1),'us', round(st['tflops'],1),'TF', round(st['gbs']),'GB/s')
for f in ['cfg2_noev','b1_noev']:
    b=json.load(open(f'{O}/{f}.json')); print(f, b['ms_per_step'], b['value'])
PY
