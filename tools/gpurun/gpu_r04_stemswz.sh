# round 4: the fp16 stem's halo image swizzled by column parity (two-way LDS bank conflicts on every fragment read before): identity tests,
# then same-box A/B of the stem launch against build/libhandmv_dev.so
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_kernel_is_bit_identical or chained or odd_frame or fp16_path_within or full_size_properties or size_gated" > $O/tests_stemswz.log 2>&1; rc=$?
tail -3 $O/tests_stemswz.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do
  timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_swz$r.json > $O/b_swz$r.json 2> $O/b_swz.err || exit 52
  HMV_LIB=build/libhandmv_dev.so timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --per-layer $O/pl_ref$r.json > $O/b_ref$r.json 2> $O/b_ref.err || exit 53
done
python - <<'PY'
import json
for n in ("b_swz1", "b_ref1", "b_swz2", "b_ref2"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("launches_per_forward"))
for n in ("pl_swz1", "pl_ref1", "pl_swz2", "pl_ref2"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    rows = d if isinstance(d, list) else d.get("launches", d.get("layers", []))
    for r in rows[:2]:
        print(n, {k: r[k] for k in r if k in ("label", "name", "us", "ms", "kernel")})
PY
