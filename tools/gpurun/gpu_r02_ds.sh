mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=1200 -k "forward_matches or fp16_path or split_precision or full_size" > gpurun_out/r02/tests6.log 2>&1; rc=$?
tail -6 gpurun_out/r02/tests6.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
for nf in 0 1; do
  if [ $nf = 1 ]; then export HMV_NO_DSFUSE=1; else unset HMV_NO_DSFUSE; fi
  python bench.py --steps 12 --warmup 3 --per-layer gpurun_out/r02/pl_f32_nofuse$nf.json --no-cpu-baseline > gpurun_out/r02/b_f32_nofuse$nf.json 2> gpurun_out/r02/b_f32_nofuse$nf.err || exit 13
  python bench.py --dtype f16 --steps 12 --warmup 3 --per-layer gpurun_out/r02/pl_f16_nofuse$nf.json --no-cpu-baseline > gpurun_out/r02/b_f16_nofuse$nf.json 2> gpurun_out/r02/b_f16_nofuse$nf.err || exit 14
done
for f in gpurun_out/r02/b_*nofuse*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['bound'], d['roofline']['frac'])"; done
