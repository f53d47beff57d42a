O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stream32" > $O/tests_s32.log 2>&1 || { tail -40 $O/tests_s32.log; exit 40; }
tail -1 $O/tests_s32.log
for v in on off on off; do
if [ $v = on ]; then export HMV_STREAM32_K256=1; else unset HMV_STREAM32_K256; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --per-layer $O/per_layer_f32_k256$v.json > $O/bench_f32_k256$v.json 2> $O/bench_f32_k256$v.err || exit 56
python - <<PY
import json
d = json.load(open("gpurun_out/r03/bench_f32_k256$v.json")); print("k256 $v f32", d["ms_per_step"])
rows = json.load(open("gpurun_out/r03/per_layer_f32_k256$v.json"))
print("   ", " ".join(f"{r['layer'].replace('layer','l')}:{r['avg_ms']*1e3:.0f}" for r in rows if r['layer'].startswith('layer3') and r['layer'].endswith('conv3')))
PY
done
