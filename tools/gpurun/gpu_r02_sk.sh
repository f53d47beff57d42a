mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q --timeout=1200 > gpurun_out/r02/tests7.log 2>&1; rc=$?
tail -6 gpurun_out/r02/tests7.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
for ns in 0 1; do
  if [ $ns = 1 ]; then export HMV_NO_SPLITK=1; else unset HMV_NO_SPLITK; fi
  python bench.py --workload cfg2 --steps 200 --warmup 20 --instrument-every 0 --per-layer gpurun_out/r02/pl_cfg2_nosk$ns.json --no-cpu-baseline > gpurun_out/r02/b_cfg2_nosk$ns.json 2> gpurun_out/r02/b_cfg2_nosk$ns.err || exit 13
done
unset HMV_NO_SPLITK
python bench.py --workload cfg3 --batch 1 --steps 200 --warmup 20 --instrument-every 0 --no-cpu-baseline > gpurun_out/r02/b_b1.json 2> gpurun_out/r02/b_b1.err || exit 14
for f in gpurun_out/r02/b_cfg2_nosk*.json gpurun_out/r02/b_b1.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['bound'], d['roofline']['frac']); print({k:(v['ms_per_step'],v['tflops']) for k,v in d['kernels'].items()})"; done
