O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stream_kernel_is_bit_identical" > $O/tests_hs80.log 2>&1 || { tail -40 $O/tests_hs80.log; exit 40; }
tail -1 $O/tests_hs80.log
for n in 32 256 1024; do echo "n_img $n"; timeout -k 10 120 python tools/hs_probe.py $n 2>/dev/null | grep 80 || exit 1; done
timeout -k 10 400 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2 > $O/hr40_f16_hs80.json 2> $O/hr40_f16_hs80.err || { tail -5 $O/hr40_f16_hs80.err; exit 56; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/hr40_f16_hs80.json")); print("hs80", d["ms_per_step"])
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:6]: print("   ", k, v)
PY
