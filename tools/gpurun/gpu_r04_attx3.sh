# round 4: attention on the fp16 matrix cores over (hi, lo) pairs (attention_x3_kernel) in the fp16-kernel modes: op-level and network tests,
# launch times beside the fp32 kernel, same-box A/B of the fp16 / f32x3 steps against build/libhandmv_noax.so (the fp32-MFMA attention there)
O=gpurun_out/r04; mkdir -p $O
rc=0

[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/att_probe.py --x3 > $O/att_x3.txt 2>&1 || { tail -5 $O/att_x3.txt; exit 61; }
timeout -k 10 200 python tools/att_probe.py > $O/att_f32.txt 2>&1 || { tail -5 $O/att_f32.txt; exit 62; }
for r in 1 2; do
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary > $O/b_ax_$r.json 2> $O/b_ax.err || exit 52
HMV_LIB=build/libhandmv_noax.so timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --no-secondary > $O/b_noax_$r.json 2> $O/b_noax.err || exit 53
done
timeout -k 10 300 python bench.py --dtype f32x3 --no-cpu-baseline --no-secondary > $O/b_ax_x3.json 2> $O/b_ax_x3.err || exit 54
HMV_LIB=build/libhandmv_noax.so timeout -k 10 300 python bench.py --dtype f32x3 --no-cpu-baseline --no-secondary > $O/b_noax_x3.json 2> $O/b_noax_x3.err || exit 55
timeout -k 10 300 python bench.py --dtype f16 --batch 1 --no-cpu-baseline --no-secondary --steps 200 --warmup 20 --instrument-every 0 > $O/b_ax_b1.json 2> $O/b_ax_b1.err || exit 56
HMV_LIB=build/libhandmv_noax.so timeout -k 10 300 python bench.py --dtype f16 --batch 1 --no-cpu-baseline --no-secondary --steps 200 --warmup 20 --instrument-every 0 > $O/b_noax_b1.json 2> $O/b_noax_b1.err || exit 57
python - <<'PY'
import json
for n in ("b_ax_1", "b_noax_1", "b_ax_2", "b_noax_2", "b_ax_x3", "b_noax_x3", "b_ax_b1", "b_noax_b1"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, d["ms_per_step"], d["value"])
PY
