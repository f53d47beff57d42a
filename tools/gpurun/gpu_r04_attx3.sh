# round 4: attention on the fp16 matrix cores over (hi, lo) pairs (attention_x3_kernel) in the fp16-kernel modes: op-level and network tests,
# launch times beside the fp32 kernel, same-box A/B of the fp16 / f32x3 / batch-1 steps against build/libhandmv_noax.so
# (python -m handmvnet_amd.build --variant noax HMV_NO_ATT_X3: the fp32-MFMA attention in every mode)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graphs.py -x -q -m gpu -k "attention or tail or fixture or fp16_path or full_size_properties or poisoned or split_precision or release_shape or graph or random_config" > $O/tests_attx3.log 2>&1; rc=$?
tail -3 $O/tests_attx3.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/att_probe.py --x3 > $O/att_x3.txt 2>&1 || { tail -5 $O/att_x3.txt; exit 61; }
timeout -k 10 200 python tools/att_probe.py > $O/att_f32.txt 2>&1 || { tail -5 $O/att_f32.txt; exit 62; }
echo x3; grep "B=" $O/att_x3.txt; echo f32; grep "B=" $O/att_f32.txt
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
