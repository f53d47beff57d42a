# round 4: the MFMA-heavy 1x1 layers on the 16x16x32 MFMA (conv_gemm8<m16> + conv_m16 1x1 tiles): op-level identity, network-level
# batch independence, fp16 fixtures, then the fp16 bench lines
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "stream_kernel_is_bit_identical or tall or small_launch or full_size_properties or chained or fp16_path_within or poisoned or hrnet_release" > $O/tests_g8m16.log 2>&1; rc=$?
grep -v Warning $O/tests_g8m16.log | tail -8
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_g8m16.json > $O/bench_f16_g8m16.json 2> $O/bench_f16_g8m16.err || { tail -20 $O/bench_f16_g8m16.err; exit 52; }
timeout -k 10 300 python bench.py --workload cfg3 --batch 1 --dtype f16 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_b1_f16_g8m16.json 2> $O/bench_b1_f16.err || exit 55
timeout -k 10 300 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline > $O/bench_hr40_f16_g8m16.json 2> $O/bench_hr40_f16.err || exit 56
python - <<'PY'
import json
for n in ("bench_f16_g8m16", "bench_b1_f16_g8m16", "bench_hr40_f16_g8m16"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
d = json.load(open("gpurun_out/r04/per_layer_f16_g8m16.json"))
PY
python tools/per_layer_table.py $O/per_layer_f16_g8m16.json > $O/per_layer_f16_g8m16.md 2>/dev/null || true
grep -n "gemm8\|conv_ht\|conv_m16" $O/per_layer_f16_g8m16.md | head -30
