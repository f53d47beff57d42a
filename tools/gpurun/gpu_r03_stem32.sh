O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "(full_size and f32 and not f32x3) or cfg2_full_batch" > $O/tests_stem32.log 2>&1 || { tail -30 $O/tests_stem32.log; exit 40; }
tail -1 $O/tests_stem32.log
for v in on off on off; do
if [ $v = off ]; then export HMV_NO_HS32=1; else unset HMV_NO_HS32; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --per-layer $O/per_layer_f32_stem$v.json > $O/bench_f32_stem$v.json 2> $O/bench_f32_stem$v.err || exit 56
python - <<PY
import json
d = json.load(open("gpurun_out/r03/bench_f32_stem$v.json")); print("stem32 $v f32", d["ms_per_step"])
print("   ", [(r['kernel'], round(r['avg_ms']*1e3)) for r in json.load(open("gpurun_out/r03/per_layer_f32_stem$v.json")) if r['layer']=='stem'])
PY
done
for v in on off; do
if [ $v = off ]; then export HMV_NO_HS32=1; else unset HMV_NO_HS32; fi
timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/cfg2_stem$v.json 2> $O/cfg2_stem$v.err || exit 58
python -c "import json; d=json.load(open('$O/cfg2_stem$v.json')); print('cfg2 $v', d['ms_per_step'])"
done
