# round 4, first call: the new parity tests (HRNet-w40 release shape, tail on the engine's own tokens, size rows, tightened fp16
# full-size checks) with their printed numbers, then the read + write copy probe
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "hrnet_release_shape or tail_on_engine_tokens or size_gated or full_size_properties or fp16_path_within or split_precision_path or reference_fixture" > $O/new_tests.log 2>&1; rc=$?
grep -v Warning $O/new_tests.log | tail -15
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 ./tools/probe/copybw > $O/probe_copy.txt 2>&1 || exit 61
cat $O/probe_copy.txt
