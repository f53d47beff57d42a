set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/r02/tests2.log 2>&1; rc=$?
tail -15 gpurun_out/r02/tests2.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest rc=$rc: stopping"; exit $rc; fi
HMV_BENCH_DTYPE=f16 python tools/conv_sweep.py 256 1,2,3,4,5,6,9 > gpurun_out/r02/sweep_f16_tout.txt 2>&1 || exit 21
tail -25 gpurun_out/r02/sweep_f16_tout.txt
