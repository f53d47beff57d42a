O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stream32 or (full_size and f32 and not f32x3)" > $O/tests_s32.log 2>&1 || { tail -40 $O/tests_s32.log; exit 40; }
tail -1 $O/tests_s32.log
for v in on off; do
if [ $v = off ]; then export HMV_NO_STREAM32=1; else unset HMV_NO_STREAM32; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --per-layer $O/per_layer_f32_s32$v.json > $O/bench_f32_s32$v.json 2> $O/bench_f32_s32$v.err || exit 56
python - <<PY
import json
d = json.load(open("gpurun_out/r03/bench_f32_s32$v.json")); print("stream32 $v f32", d["ms_per_step"])
rows = json.load(open("gpurun_out/r03/per_layer_f32_s32$v.json"))
print("   ", " ".join(f"{r['layer'].replace('layer','l')}:{r['avg_ms']*1e3:.0f}" for r in rows if r['layer'].startswith('layer1')))
PY
done
