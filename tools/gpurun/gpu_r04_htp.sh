# round 4: conv_ht's persistent form (conv_htp_f16): identity tests, then same-box A/B against build/libhandmv_k16.so (the committed state before it)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tall_tile or full_size_properties or small_launch_tiles" > $O/tests_htp.log 2>&1; rc=$?
tail -3 $O/tests_htp.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do
  timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_htp_$r.json > $O/b_htp_$r.json 2> $O/b_htp.err || exit 52
  HMV_LIB=build/libhandmv_k16.so timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_nohtp_$r.json > $O/b_nohtp_$r.json 2> $O/b_nohtp.err || exit 53
done
python - <<'PY'
import json
from collections import defaultdict
for n in ("b_htp_1", "b_nohtp_1", "b_htp_2", "b_nohtp_2"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"])
for n in ("pl_htp_1", "pl_nohtp_1"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    t = defaultdict(float)
    for r in d: t[r["kernel"]] += r["avg_ms"]
    print(n, {k: round(v, 3) for k, v in t.items() if "conv_ht" in k or "gemm8" in k})
PY
