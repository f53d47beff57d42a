# round 4: MFMA-shape A/B of conv_ht (32x32x16 vs 16x16x32 fp16 MFMA), op-level correctness first, then per-kernel durations from
# rocprofv3 --kernel-trace --stats (dense random and ReLU-sparse operands, one shape per process); then the remaining new tests + copy probe
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tall_tile" > $O/tests_ht.log 2>&1; rc=$?
tail -5 $O/tests_ht.log
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
for sp in 0 1; do for sh in 0 1; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/shape_${sp}_${sh} -o st -- python3 $GRAFT_REPO_ROOT/tools/mfma_shape_probe.py 8 $sp $sh > $GRAFT_REPO_ROOT/$O/shape_${sp}_${sh}.log 2>&1 || exit 62
done; done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for d in sorted(glob.glob("gpurun_out/r04/shape_*_*")):
    if not d.endswith((".log",)):
        for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "conv_ht" in r["Name"]:
                    print(d.split("/")[-1], r["Name"][:60], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "hrnet_release_shape or size_gated or full_size_properties or fp16_path_within or split_precision_path" > $O/new_tests2.log 2>&1; rc=$?
grep -v Warning $O/new_tests2.log | tail -8
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 ./tools/probe/copybw > $O/probe_copy.txt 2>&1 || exit 61
cat $O/probe_copy.txt
