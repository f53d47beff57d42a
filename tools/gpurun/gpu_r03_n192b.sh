O=gpurun_out/r03; mkdir -p $O
for bm in 128 256 128 256; do
HMV_N192_BM=$bm timeout -k 10 400 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2 > $O/hr40_f16_bm$bm.json 2> $O/hr40_f16_bm$bm.err || { tail -5 $O/hr40_f16_bm$bm.err; exit 56; }
python - <<PY
import json
d = json.load(open("gpurun_out/r03/hr40_f16_bm$bm.json")); print("bm$bm", d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items() if "192" in k})
PY
done
