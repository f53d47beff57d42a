# round 4: HRNet fuse layers' up-sampling terms as one launch (hr_fuse.hip): equivalence + fixture tests, then same-box A/B of the hr40 workload
# (bench.py --no-hr-fusion: one conv launch per term)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hrnet_fuse or hrnet_release or tail_on_engine" > $O/tests_hrfuse.log 2>&1; rc=$?
tail -5 $O/tests_hrfuse.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 8 --warmup 2 --per-layer $O/pl_hrf_f16.json > $O/b_hrf_f16.json 2> $O/b_hrf_f16.err || exit 52
timeout -k 10 300 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --no-hr-fusion --steps 8 --warmup 2 > $O/b_nohrf_f16.json 2> $O/b_nohrf_f16.err || exit 53
timeout -k 10 300 python bench.py --workload hr40 --no-cpu-baseline --steps 5 --warmup 2 --per-layer $O/pl_hrf_f32.json > $O/b_hrf_f32.json 2> $O/b_hrf_f32.err || exit 54
timeout -k 10 300 python bench.py --workload hr40 --no-cpu-baseline --no-hr-fusion --steps 5 --warmup 2 > $O/b_nohrf_f32.json 2> $O/b_nohrf_f32.err || exit 55
python - <<'PY'
import json
for n in ("b_hrf_f16", "b_nohrf_f16", "b_hrf_f32", "b_nohrf_f32"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d.get("launches_per_forward"))
for n in ("pl_hrf_f16", "pl_hrf_f32"):
    for r in json.load(open(f"gpurun_out/r04/{n}.json")):
        if "+up" in r["layer"] and ("stage4.0" in r["layer"] or "stage3.0" in r["layer"]): print(n, r["layer"], r["kernel"], round(r["avg_ms"] * 1e3, 1), round(r["gbs"]))
PY
