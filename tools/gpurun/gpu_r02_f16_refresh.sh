# fp16 records after the halo mode: bench + per-layer, kernel stats, PMC passes (each its own run)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f16; mkdir -p $O
python bench.py --dtype f16 --steps 20 --warmup 5 --per-layer $O/per_layer_f16.json --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err || exit 42
python bench.py --dtype f16 --steps 20 --warmup 5 --instrument-every 0 --no-cpu-baseline > $O/bench_f16_noev.json 2> $O/bench_f16_noev.err || exit 43
B="python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --dtype f16"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f16 -- $B > $O/bench_under_rocprof_f16.json 2> $O/stats_f16.err || exit 32
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq_f16 -- $B > $O/pmc_sq_f16.json 2> $O/pmc_sq_f16.err || exit 33
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_f16 -- $B > $O/pmc_fetch_f16.json 2> $O/pmc_fetch_f16.err || exit 34
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_f16 -- $B > $O/pmc_write_f16.json 2> $O/pmc_write_f16.err || exit 35
python3 tools/pmc_summary.py r02_f16 $O/sq_f16 $O/fetch_f16 $O/write_f16 --dtype=f16 > $O/pmc_print_f16.txt 2>&1 || exit 36
cp profiles/r02_f16_pmc_summary.json profiles/pmc_traffic.json $O/
find $O -name "*counter_collection.csv" -size +20M -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
head -6 $O/pmc_print_f16.txt | cut -c1-400
for f in bench_f16 bench_f16_noev bench_under_rocprof_f16; do python3 -c "
import json; d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; done
