# round 4: the tall layers on the 16x16x32 MFMA (conv_ht<m16> + conv_m16 small tiles): op-level tests, full-size properties, then the
# remaining new tests, the copy probe and the fp16 / fp32 bench lines
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "split_precision_path or tall or small_launch" > $O/tests_m16.log 2>&1; rc=$?
grep -v Warning $O/tests_m16.log | tail -8
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 ./tools/probe/copybw > $O/probe_copy.txt 2>&1 || exit 61
cat $O/probe_copy.txt
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_m16.json > $O/bench_f16_m16.json 2> $O/bench_f16_m16.err || { tail -20 $O/bench_f16_m16.err; exit 52; }
timeout -k 10 300 python bench.py --workload cfg3 --batch 1 --dtype f16 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_b1_f16_m16.json 2> $O/bench_b1_f16.err || exit 55
python - <<'PY'
import json
for n in ("bench_f16_m16", "bench_b1_f16_m16"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
PY
