# round 3: the persistent weight-stationary 1x1 kernel -- parity (bit-identity with conv_igemm), then A/B timing on one box
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stream_kernel" > $O/stream_tests.log 2>&1; rc=$?
tail -15 $O/stream_tests.log
[ $rc -eq 0 ] || exit $rc
HMV_BENCH_DTYPE=f16 timeout -k 10 300 python tools/stream_probe.py > $O/stream_probe_on.txt 2>&1 || { tail $O/stream_probe_on.txt; exit 61; }
HMV_BENCH_DTYPE=f16 HMV_NO_STREAM=1 timeout -k 10 300 python tools/stream_probe.py > $O/stream_probe_off.txt 2>&1 || { tail $O/stream_probe_off.txt; exit 62; }
HMV_BENCH_DTYPE=f16 timeout -k 10 300 python tools/stream_probe.py > $O/stream_probe_on2.txt 2>&1
cat $O/stream_probe_on.txt $O/stream_probe_off.txt $O/stream_probe_on2.txt
