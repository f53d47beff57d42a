# round 4: gemm_x3 on eight waves; same-box A/B against the build of two commits ago (build/libhandmv_dev.so: no pairs, no gemm_x3, 16-row ff_block)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_pair_gemm or fused_tail or poisoned or full_size_properties or tail_on_engine" > $O/tests_x3b.log 2>&1; rc=$?
tail -3 $O/tests_x3b.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do
  timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_new$r.json > $O/b_new$r.json 2> $O/b_new.err || exit 52
  HMV_LIB=build/libhandmv_dev.so timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_old$r.json > $O/b_old$r.json 2> $O/b_old.err || exit 53
done
python - <<'PY'
import json
for n in ("b_new1", "b_old1", "b_new2", "b_old2"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
PY
python tools/per_layer_table.py $O/pl_new1.json > $O/pl_new1.md 2>/dev/null || true
python tools/per_layer_table.py $O/pl_old1.json > $O/pl_old1.md 2>/dev/null || true
grep -n "fusion" $O/pl_new1.md | cut -c1-110
grep -n "fusion..qkv" $O/pl_old1.md | cut -c1-110
