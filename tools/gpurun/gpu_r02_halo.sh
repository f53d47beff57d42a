O=gpurun_out/r02halo; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout=600 -x -k "halo" > $O/t_halo.log 2>&1; rc=$?
tail -12 $O/t_halo.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout=900 -x -k "f16 or fp16 or full_size or half" > $O/tests.log 2>&1; rc=$?
tail -6 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
python bench.py --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --per-layer $O/pl_on_$rep.json > $O/b_on_$rep.json 2> $O/b.err || exit 42
HMV_NO_HALO=1 python bench.py --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --per-layer $O/pl_off_$rep.json > $O/b_off_$rep.json 2> $O/b.err || exit 43
done
for f in $O/b_*.json; do python -c "
import json; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'])"; done
python - <<'PY'
import json
for n in ('on_1','off_1'):
    d=json.load(open(f'gpurun_out/r02halo/pl_{n}.json'))
    print(n, [(r['layer'], r['kernel'], round(r['avg_ms']*1000)) for r in d if r['layer'] in ('layer3.1.conv2','layer2.1.conv2','layer1.1.conv2','layer3.0.conv2')])
PY
