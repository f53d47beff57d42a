O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "reference_fixture or matches_oracle or poisoned or (full_size and f32) or non_square" > $O/tests_sa.log 2>&1 || { tail -30 $O/tests_sa.log; exit 40; }
tail -1 $O/tests_sa.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sa -o sa -- python3 bench.py --dtype f16 --no-cpu-baseline --steps 6 --warmup 2 --instrument-every 0 > $O/prof_sa.json 2> $O/prof_sa.err || exit 60
grep -a "soft_argmax" $O/prof_sa/sa_kernel_stats.csv | cut -c1-140
find gpurun_out/r03 -name "*kernel_trace.csv" -size +5M -delete
