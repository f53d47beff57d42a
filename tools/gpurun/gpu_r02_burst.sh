python -m pytest tests/test_gpu_parity.py -q -x --timeout=900 -k "conv_kernel or forward_matches or split_precision" > gpurun_out/r02_tests3.log 2>&1; rc=$?; tail -4 gpurun_out/r02_tests3.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
python tools/stagger_probe.py > gpurun_out/r02_burst.txt 2>&1; cat gpurun_out/r02_burst.txt
for bu in 0 1; do echo "== f16 l3 conv3 burst=$bu"; HMV_BURST=$bu HMV_BENCH_DTYPE=f16 python tools/timeline.py 256,32,256,1024,1,1,0,1 5 2>&1 | head -6; done
for bu in 0 1; do echo "== f32 l3 conv3 burst=$bu"; HMV_BURST=$bu python tools/timeline.py 256,32,256,1024,1,1,0,1 5 2>&1 | head -6; done
