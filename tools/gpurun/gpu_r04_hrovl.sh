# round 4: HRNet four-branch modules, the last branch on a second stream beside the branch above it: tests, then same-box A/B (bench --hr-mode 1 = one stream)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hrnet or hr40 or hr64 or graph or poisoned" > $O/tests_hrovl.log 2>&1; rc=$?
tail -3 $O/tests_hrovl.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do
timeout -k 10 300 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 8 --warmup 2 > $O/b_ovl_f16_$r.json 2> $O/b_ovl_f16.err || exit 52
timeout -k 10 300 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --hr-mode 1 --steps 8 --warmup 2 > $O/b_noovl_f16_$r.json 2> $O/b_noovl_f16.err || exit 53
done
timeout -k 10 300 python bench.py --workload hr40 --no-cpu-baseline --steps 5 --warmup 2 > $O/b_ovl_f32.json 2> $O/b_ovl_f32.err || exit 54
timeout -k 10 300 python bench.py --workload hr40 --no-cpu-baseline --hr-mode 1 --steps 5 --warmup 2 > $O/b_noovl_f32.json 2> $O/b_noovl_f32.err || exit 55
timeout -k 10 300 python bench.py --workload hr40 --dtype f32x3 --no-cpu-baseline --steps 5 --warmup 2 > $O/b_ovl_x3.json 2> $O/b_ovl_x3.err || exit 56
timeout -k 10 300 python bench.py --workload hr40 --dtype f32x3 --no-cpu-baseline --hr-mode 1 --steps 5 --warmup 2 > $O/b_noovl_x3.json 2> $O/b_noovl_x3.err || exit 57
python - <<'PY'
import json
for n in ("b_ovl_f16_1", "b_noovl_f16_1", "b_ovl_f16_2", "b_noovl_f16_2", "b_ovl_f32", "b_noovl_f32", "b_ovl_x3", "b_noovl_x3"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"], d.get("launches_per_forward"))
PY
