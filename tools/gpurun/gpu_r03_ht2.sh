O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tall_tile" > $O/tests_ht.log 2>&1 || { tail -40 $O/tests_ht.log; exit 40; }
tail -1 $O/tests_ht.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "f16" > $O/tests_ht_f16.log 2>&1 || { tail -40 $O/tests_ht_f16.log; exit 41; }
tail -1 $O/tests_ht_f16.log
for b in 1 2 4 8 32; do
  timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --batch $b --steps 100 --warmup 10 > $O/b${b}_ht.json 2> $O/b${b}_ht.err || exit 56
  HMV_NO_HT=1 timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --batch $b --steps 100 --warmup 10 > $O/b${b}_noht.json 2> $O/b${b}_noht.err || exit 57
done
python - <<'PY'
import json
for b in (1, 2, 4, 8, 32):
    for tag in ("ht", "noht"):
        d = json.load(open(f"gpurun_out/r03/b{b}_{tag}.json")); print("B", b, tag, d["ms_per_step"])
PY
