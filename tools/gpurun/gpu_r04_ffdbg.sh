# round 4: phase stamps of ff_block_kernel (dev build, HMV_FF_DBG=1) at cfg-2 and inside a cfg-3 fp16 step: where a 16-token tile's time goes;
# fused-tail tests first
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_tail or tail_on_engine or full_size_properties" > $O/tests_ff.log 2>&1; rc=$?
tail -3 $O/tests_ff.log
[ $rc -eq 0 ] || exit $rc
HMV_LIB=build/libhandmv_dev.so HMV_FF_DBG=1 timeout -k 10 200 python bench.py --workload cfg2 --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --instrument-every 0 > $O/ffdbg_cfg2.json 2> $O/ffdbg_cfg2.err || { tail -5 $O/ffdbg_cfg2.err; exit 61; }
grep "ff_block" $O/ffdbg_cfg2.err | tail -4
HMV_LIB=build/libhandmv_dev.so HMV_FF_DBG=1 timeout -k 10 200 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --instrument-every 0 > $O/ffdbg_cfg3.json 2> $O/ffdbg_cfg3.err || { tail -5 $O/ffdbg_cfg3.err; exit 62; }
grep "ff_block" $O/ffdbg_cfg3.err | tail -10
for r in 1 2; do
timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/b_ff_cfg2_$r.json 2> $O/b_ff.err || exit 52
done
python - <<'PY'
import json
for n in ("b_ff_cfg2_1", "b_ff_cfg2_2"):
    d = json.load(open(f"gpurun_out/r04/{n}.json")); print(n, d["ms_per_step"], d["value"])
PY
