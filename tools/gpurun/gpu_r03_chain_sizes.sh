# round 3: the cross-layer launches at other frame sizes (ragged pooled blocks, partial pixel tiles) + the size smoke
O=gpurun_out/r03sizes; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained" > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/size_smoke.py > $O/size_smoke.txt 2>&1 || { tail -20 $O/size_smoke.txt; exit 61; }
tail -12 $O/size_smoke.txt
