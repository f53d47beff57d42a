# round 4: layer3.0's conv3 + downsample on the 16x16x32 MFMA too (gemm8<dual,m16> + conv_m16 dual tiles); fp16 tests + bench
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_properties or chained or fp16_path_within or poisoned or reference_fixture or hrnet_release" > $O/tests_g8dual.log 2>&1; rc=$?
tail -4 $O/tests_g8dual.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16_g8dual.json > $O/bench_f16_g8dual.json 2> $O/bench_f16_g8dual.err || { tail -20 $O/bench_f16_g8dual.err; exit 52; }
python tools/per_layer_table.py $O/per_layer_f16_g8dual.json > $O/per_layer_f16_g8dual.md 2>/dev/null || true
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/bench_f16_g8dual.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
PY
grep -n "dual\|fusion\|stem\|pose\|sample" $O/per_layer_f16_g8dual.md | head -30
