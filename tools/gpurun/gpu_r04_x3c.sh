# round 4: gemm_x3 with column ranges per XCD (weights stay in L2) + to_out as a split-pair GEMM fed by attention rows written as pairs;
# same-box A/B against build/libhandmv_dev.so (before the tail work)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_pair_gemm or fused_tail or poisoned or full_size_properties or tail_on_engine or reference_fixture or split_precision_path or fp16_path_within or cfg2_full or attention_kernel" > $O/tests_x3c.log 2>&1; rc=$?
tail -3 $O/tests_x3c.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do
  timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_new$r.json > $O/b_new$r.json 2> $O/b_new.err || exit 52
  HMV_LIB=build/libhandmv_dev.so timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_old$r.json > $O/b_old$r.json 2> $O/b_old.err || exit 53
done
timeout -k 10 300 python bench.py --workload cfg2 --dtype f16 --no-cpu-baseline --steps 300 --warmup 30 --instrument-every 0 > $O/b_cfg2_new.json 2> $O/b_cfg2.err || exit 54
HMV_LIB=build/libhandmv_dev.so timeout -k 10 300 python bench.py --workload cfg2 --dtype f16 --no-cpu-baseline --steps 300 --warmup 30 --instrument-every 0 > $O/b_cfg2_old.json 2> $O/b_cfg2.err || exit 54
timeout -k 10 300 python bench.py --workload cfg3 --batch 1 --dtype f16 --no-cpu-baseline --steps 300 --warmup 30 --instrument-every 0 > $O/b_b1_new.json 2> $O/b_b1.err || exit 55
HMV_LIB=build/libhandmv_dev.so timeout -k 10 300 python bench.py --workload cfg3 --batch 1 --dtype f16 --no-cpu-baseline --steps 300 --warmup 30 --instrument-every 0 > $O/b_b1_old.json 2> $O/b_b1.err || exit 55
python - <<'PY'
import json
for n in ("b_new1", "b_old1", "b_new2", "b_old2", "b_cfg2_new", "b_cfg2_old", "b_b1_new", "b_b1_old"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
PY
python tools/per_layer_table.py $O/pl_new1.json > $O/pl_new1.md 2>/dev/null || true
grep -n "fusion" $O/pl_new1.md | cut -c1-110
