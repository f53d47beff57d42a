set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/r02/tests4.log 2>&1; rc=$?
tail -8 gpurun_out/r02/tests4.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
for bu in 1 0; do
HMV_BURST=$bu python bench.py --dtype f16 --steps 12 --warmup 3 --per-layer gpurun_out/r02/pl_f16_burst$bu.json --no-cpu-baseline > gpurun_out/r02/b_f16_burst$bu.json 2> gpurun_out/r02/b_f16_burst$bu.err || exit 13
HMV_BURST=$bu python bench.py --dtype f32x3 --steps 12 --warmup 3 --per-layer gpurun_out/r02/pl_x3_burst$bu.json --no-cpu-baseline > gpurun_out/r02/b_x3_burst$bu.json 2> gpurun_out/r02/b_x3_burst$bu.err || exit 15
done
python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r02/b_f32_now.json 2> gpurun_out/r02/b_f32_now.err || exit 16
for f in gpurun_out/r02/b_*burst*.json gpurun_out/r02/b_f32_now.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d['n_gpus'], d['roofline']['kernel'], d['roofline']['bound'], d['roofline']['frac'])"; done
