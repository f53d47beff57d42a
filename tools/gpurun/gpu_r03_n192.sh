O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wide_n_tile or hr" > $O/tests_n192.log 2>&1 || { tail -40 $O/tests_n192.log; exit 40; }
tail -1 $O/tests_n192.log
for i in 1 2; do
timeout -k 10 400 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2 > $O/hr40_f16_n192.json 2> $O/hr40_f16_n192.err || { tail -5 $O/hr40_f16_n192.err; exit 56; }
HMV_NO_N192=1 timeout -k 10 400 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2 > $O/hr40_f16_no192.json 2> $O/hr40_f16_no192.err || { tail -5 $O/hr40_f16_no192.err; exit 57; }
python - <<'PY'
import json
for tag in ("n192", "no192"):
    d = json.load(open(f"gpurun_out/r03/hr40_f16_{tag}.json")); print(tag, d["ms_per_step"])
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:4]: print("   ", k, v)
PY
done
