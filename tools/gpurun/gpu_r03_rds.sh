O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "rds_kernel" > $O/tests_rds.log 2>&1 || { tail -40 $O/tests_rds.log; exit 40; }
tail -1 $O/tests_rds.log
for v in on off; do
if [ $v = off ]; then export HMV_NO_RDS=1; else unset HMV_NO_RDS; fi
timeout -k 10 400 python bench.py --workload hr40 --no-cpu-baseline --steps 4 --warmup 1 > $O/hr40_f32_rds$v.json 2> $O/hr40_f32_rds$v.err || { tail -5 $O/hr40_f32_rds$v.err; exit 56; }
python - <<PY
import json
d = json.load(open("gpurun_out/r03/hr40_f32_rds$v.json")); print("rds $v hr40 f32", d["ms_per_step"])
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:6]: print("   ", k, v)
PY
done
