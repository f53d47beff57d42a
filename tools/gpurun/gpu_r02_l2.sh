cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l2; mkdir -p $O
B="python3 bench.py --dtype f16 --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/p1 -- $B > $O/p1.json 2> $O/p1.err || exit 31
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum --output-format csv -d $O/p2 -- $B > $O/p2.json 2> $O/p2.err || exit 32
python3 tools/l2_hits.py $O/p1 $O/p2 > $O/l2.txt 2>&1
head -30 $O/l2.txt
find $O -name "*.csv" -size +20M -delete
