# round 4: rocprofv3 passes of the bench command on the final code (kernel stats; then PMC in separate passes, never combined with other traces)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04prof; mkdir -p $O
B="python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f32 -- $B > $O/bench_under_rocprof_f32.json 2> $O/stats_f32.err || exit 31
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f16 -- $B --dtype f16 > $O/bench_under_rocprof_f16.json 2> $O/stats_f16.err || exit 32
echo "stats done"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
for dt in f32 f16; do
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq_$dt -- $B --dtype $dt > $O/pmc_sq_$dt.json 2> $O/pmc_sq_$dt.err || exit 33
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$dt -- $B --dtype $dt > $O/pmc_fetch_$dt.json 2> $O/pmc_fetch_$dt.err || exit 34
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$dt -- $B --dtype $dt > $O/pmc_write_$dt.json 2> $O/pmc_write_$dt.err || exit 35
  python3 tools/pmc_summary.py r04_$dt $O/sq_$dt $O/fetch_$dt $O/write_$dt --dtype=$dt > $O/pmc_print_$dt.txt 2>&1 || { cat $O/pmc_print_$dt.txt; exit 36; }
  echo "pmc $dt done"
done
cp profiles/r04_f32_pmc_summary.json profiles/r04_f16_pmc_summary.json profiles/pmc_traffic.json $O/
for dt in f32 f16; do f=$(find $O/stats_$dt -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$dt.csv; done
# keep only the small summaries
find $O -name "*counter_collection.csv" -size +5M -delete
find $O -name "*kernel_trace.csv" -size +5M -delete
cat $O/pmc_print_f16.txt
