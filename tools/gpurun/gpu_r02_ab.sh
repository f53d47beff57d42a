set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/r02/tests1.log 2>&1; rc=$?
tail -5 gpurun_out/r02/tests1.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
python bench.py --steps 12 --warmup 3 --per-layer gpurun_out/r02/pl_f32_tout.json --no-cpu-baseline > gpurun_out/r02/b_f32_tout.json 2> gpurun_out/r02/b_f32_tout.err || exit 11
HMV_LIB=build/libhandmv_stage.so python bench.py --steps 12 --warmup 3 --per-layer gpurun_out/r02/pl_f32_stage.json --no-cpu-baseline > gpurun_out/r02/b_f32_stage.json 2> gpurun_out/r02/b_f32_stage.err || exit 12
python bench.py --dtype f16 --steps 12 --warmup 3 --per-layer gpurun_out/r02/pl_f16_tout.json --no-cpu-baseline > gpurun_out/r02/b_f16_tout.json 2> gpurun_out/r02/b_f16_tout.err || exit 13
HMV_LIB=build/libhandmv_stage.so python bench.py --dtype f16 --steps 12 --warmup 3 --per-layer gpurun_out/r02/pl_f16_stage.json --no-cpu-baseline > gpurun_out/r02/b_f16_stage.json 2> gpurun_out/r02/b_f16_stage.err || exit 14
python bench.py --dtype f32x3 --steps 12 --warmup 3 --per-layer gpurun_out/r02/pl_x3_tout.json --no-cpu-baseline > gpurun_out/r02/b_x3_tout.json 2> gpurun_out/r02/b_x3_tout.err || exit 15
HMV_LIB=build/libhandmv_stage.so python bench.py --dtype f32x3 --steps 12 --warmup 3 --per-layer gpurun_out/r02/pl_x3_stage.json --no-cpu-baseline > gpurun_out/r02/b_x3_stage.json 2> gpurun_out/r02/b_x3_stage.err || exit 16
HMV_BENCH_SAME_DEVICE=1 HMV_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 4 --no-cpu-baseline > gpurun_out/r02/b_2rank.json 2> gpurun_out/r02/b_2rank.err || exit 17
for f in gpurun_out/r02/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d['n_gpus'], d['roofline']['kernel'], d['roofline']['bound'], d['roofline']['frac'])"; done
