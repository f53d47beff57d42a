# round 4: the q / k / v projections on 256 x 256 tiles with a deep ring (gemm_x3k16, bit-identical to the fused split loop): identity tests,
# network tests, same-box A/B against build/libhandmv_nosplit.so (the committed state before it)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_pair_gemm or full_size_properties or fused_tail or poisoned or tail_on_engine or split_precision_path or fp16_path_within or reference_fixture" > $O/tests_x3k16.log 2>&1; rc=$?
tail -3 $O/tests_x3k16.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do
  timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_k16_$r.json > $O/b_k16_$r.json 2> $O/b_k16.err || exit 52
  HMV_LIB=build/libhandmv_nosplit.so timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_nok16_$r.json > $O/b_nok16_$r.json 2> $O/b_nok16.err || exit 53
done
timeout -k 10 300 python bench.py --dtype f32x3 --no-cpu-baseline > $O/b_k16_x3.json 2> $O/b_k16_x3.err || exit 54
HMV_LIB=build/libhandmv_nosplit.so timeout -k 10 300 python bench.py --dtype f32x3 --no-cpu-baseline > $O/b_nok16_x3.json 2> $O/b_nok16_x3.err || exit 55
python - <<'PY'
import json
for n in ("b_k16_1", "b_nok16_1", "b_k16_2", "b_nok16_2", "b_k16_x3", "b_nok16_x3"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d.get("launches_per_forward"))
d = json.load(open("gpurun_out/r04/pl_k16_1.json"))
for r in d:
    if "fusion" in r["layer"]: print(r["layer"], r["kernel"], round(r["avg_ms"] * 1e3, 1))
PY
