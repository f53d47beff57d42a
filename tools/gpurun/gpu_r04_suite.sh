# round 4: the whole GPU suite + smoke() (what the driver runs at round end)
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; rc=$?
tail -6 $O/suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 41; }
tail -2 $O/smoke.log
