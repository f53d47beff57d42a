# round 3: chained conv3 -> conv1 launches (conv_stream.hip): identity test, the fp16 suite around it, A/B bench on one box
O=gpurun_out/r03chain; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained or stream_kernel or full_size_properties or poisoned" > $O/tests.log 2>&1; rc=$?
tail -15 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16.json > $O/bench_f16.json 2> $O/bench_f16.err || { tail -20 $O/bench_f16.err; exit 52; }
HMV_NO_CHAIN=1 timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline > $O/bench_f16_nochain.json 2> $O/bench_f16_nochain.err || exit 53
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --instrument-every 0 > $O/bench_f16_noev.json 2> $O/bench_f16_noev.err || exit 54
python - <<'PY'
import json
for n in ("bench_f16", "bench_f16_nochain", "bench_f16_noev"):
    d = json.load(open(f"gpurun_out/r03chain/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
pl = json.load(open("gpurun_out/r03chain/per_layer_f16.json"))
rows = pl if isinstance(pl, list) else pl.get("layers", pl)
for r in rows:
    lab = r.get("label", "")
    if lab.startswith("layer1") or lab.startswith("layer2.0"):
        print(lab, r.get("kernel"), round(r.get("us", r.get("ms", 0) * 1000), 1))
PY
