O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stream_kernel" > $O/tests_c1.log 2>&1 || { tail -30 $O/tests_c1.log; exit 40; }
tail -1 $O/tests_c1.log
HMV_BENCH_DTYPE=f16 timeout -k 10 300 python tools/stream_probe.py 256 20 | tail -4
HMV_NO_STREAM=1 HMV_BENCH_DTYPE=f16 timeout -k 10 300 python tools/stream_probe.py 256 20 | tail -4
HMV_BENCH_DTYPE=f16 timeout -k 10 300 python tools/stream_probe.py 256 20 | tail -4
