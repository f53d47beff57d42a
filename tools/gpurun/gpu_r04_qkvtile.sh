# round 4: which tile for the q/k/v projection GEMM of the fusion blocks (M = 5376, N = 3072, K = 544 split pairs; 128x128 today: 72 us)?
# forced-tile runs of the -DHMV_DEV_KNOBS build, per-layer tables; then the pairs change on the product build (tests + bench)
O=gpurun_out/r04; mkdir -p $O
for t in 3 5 4; do
  HMV_LIB=build/libhandmv_dev.so HMV_FORCE_TILE=$t timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --steps 10 --warmup 3 --per-layer $O/pl_tile$t.json > $O/b_tile$t.json 2> $O/b_tile$t.err || { tail -5 $O/b_tile$t.err; }
  python tools/per_layer_table.py $O/pl_tile$t.json > $O/pl_tile$t.md 2>/dev/null || true
  echo "tile $t"; grep -n "fusion.*qkv\|sample_nets\|pose_net.3" $O/pl_tile$t.md | cut -c1-120
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reference_fixture or fused_tail or poisoned or full_size_properties or split_precision or fp16_path_within or tail_on_engine" > $O/tests_pairs.log 2>&1; rc=$?
tail -4 $O/tests_pairs.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline > $O/bench_f16_pairs.json 2> $O/bench_f16_pairs.err || exit 52
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/bench_f16_pairs.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
PY
