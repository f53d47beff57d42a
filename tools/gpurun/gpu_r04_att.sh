# round 4: attention kernel (next-chunk operand prefetch; two waves per workgroup from three query blocks up): op-level tests, launch times
# against build/libhandmv_prev.so (the kernel before it)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention or fused_tail or tail_on_engine or full_size_properties" > $O/tests_att.log 2>&1; rc=$?
tail -3 $O/tests_att.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/att_probe.py > $O/att_new.txt 2>&1 || { tail -5 $O/att_new.txt; exit 61; }
HMV_LIB=build/libhandmv_prev.so timeout -k 10 200 python tools/att_probe.py > $O/att_old.txt 2>&1 || { tail -5 $O/att_old.txt; exit 62; }
echo new; grep "B=" $O/att_new.txt; echo old; grep "B=" $O/att_old.txt
