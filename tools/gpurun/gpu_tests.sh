mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q --timeout=1200 > gpurun_out/r02/tests5.log 2>&1; rc=$?
tail -25 gpurun_out/r02/tests5.log
exit $rc
