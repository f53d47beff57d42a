O=gpurun_out/r03; mkdir -p $O
for v in 1 2 0; do
export HMV_T256x64=$v
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 --per-layer $O/per_layer_f32_t25664_$v.json > $O/f32_t25664_$v.json 2> $O/f32_t25664_$v.err || { echo "v $v failed"; tail -3 $O/f32_t25664_$v.err; }
python - <<PY
import json
d=json.load(open("gpurun_out/r03/f32_t25664_$v.json")); print("v=$v", d["ms_per_step"])
print("   ", " ".join(f"{r['layer']}:{r['avg_ms']*1e3:.0f}({r['kernel'][15:]})" for r in json.load(open("gpurun_out/r03/per_layer_f32_t25664_$v.json")) if r['layer'].startswith('layer1') and ('conv2' in r['layer'] or r['layer']=='layer1.0.conv1')))
PY
done
