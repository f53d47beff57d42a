# round 4: gemm_x3 (q/k/v projections on a four-stage ring), ff_block 32-row tiles, producer-written pairs: tests + fp16 / f32x3 / small-batch benches
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_pair_gemm or conv_kernel_random or split_precision_vs_torch or reference_fixture or fused_tail or poisoned or full_size_properties or split_precision_path or fp16_path_within or tail_on_engine or cfg2_full or random_config" > $O/tests_x3.log 2>&1; rc=$?
tail -4 $O/tests_x3.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_x3.json > $O/bench_f16_x3.json 2> $O/bench_f16_x3.err || exit 52
python tools/per_layer_table.py $O/per_layer_f16_x3.json > $O/per_layer_f16_x3.md 2>/dev/null || true
timeout -k 10 300 python bench.py --workload cfg2 --dtype f16 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_cfg2_f16_x3.json 2> $O/bench_cfg2_f16.err || exit 54
timeout -k 10 300 python bench.py --workload cfg3 --batch 1 --dtype f16 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/bench_b1_f16_x3.json 2> $O/bench_b1_f16.err || exit 55
timeout -k 10 300 python bench.py --dtype f32x3 --no-cpu-baseline > $O/bench_f32x3_x3.json 2> $O/bench_f32x3.err || exit 56
python - <<'PY'
import json
for n in ("bench_f16_x3", "bench_cfg2_f16_x3", "bench_b1_f16_x3", "bench_f32x3_x3"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
PY
grep -n "fusion" $O/per_layer_f16_x3.md | cut -c1-130
