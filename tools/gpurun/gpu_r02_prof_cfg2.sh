cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02prof_cfg2; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload cfg2 --steps 100 --warmup 10 --instrument-every 0 --no-cpu-baseline > $O/bench.json 2> $O/err.txt || exit 31
f=$(ls -t $O/stats/*/*_kernel_stats.csv | head -1); cp $f $O/kernel_stats.csv; cat $O/kernel_stats.csv | cut -c1-170
find $O -name "*kernel_trace.csv" -size +20M -delete
