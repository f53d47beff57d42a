# round 4: conv_gemm8's persistent form (conv_gemm8p_f16): identity tests, then same-box A/B against build/libhandmv_htp.so (the committed state before it)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm8 or full_size_properties or stream_kernel or chained" > $O/tests_g8p.log 2>&1; rc=$?
tail -3 $O/tests_g8p.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do
  timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_g8p_$r.json > $O/b_g8p_$r.json 2> $O/b_g8p.err || exit 52
  HMV_LIB=build/libhandmv_htp.so timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_nog8p_$r.json > $O/b_nog8p_$r.json 2> $O/b_nog8p.err || exit 53
done
python - <<'PY'
import json
for n in ("b_g8p_1", "b_nog8p_1", "b_g8p_2", "b_nog8p_2"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"])
a = {r["layer"]: r for r in json.load(open("gpurun_out/r04/pl_g8p_1.json"))}
b = {r["layer"]: r for r in json.load(open("gpurun_out/r04/pl_nog8p_1.json"))}
for k, r in a.items():
    if "gemm8" in r["kernel"] or "gemm8" in b[k]["kernel"]:
        print(k, r["kernel"][-28:], round(r["avg_ms"] * 1e3, 1), round(b[k]["avg_ms"] * 1e3, 1))
PY
