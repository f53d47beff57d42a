python -m pytest tests -m gpu -q --timeout=1200 > gpurun_out/r02/tests9.log 2>&1; rc=$?
tail -4 gpurun_out/r02/tests9.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02prof_att; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/err.txt || exit 31
f=$(ls -t $O/stats/*/*_kernel_stats.csv | head -1); grep -i "attention\|layernorm\|maxpool" $f | cut -c1-140
python -c "import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'])"
python bench.py --workload cfg2 --steps 200 --warmup 20 --instrument-every 0 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2', d['ms_per_step'])"
find $O -name "*kernel_trace.csv" -size +20M -delete
