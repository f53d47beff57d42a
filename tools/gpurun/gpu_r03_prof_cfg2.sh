# per-kernel times of the launch-bound configs, fused tail on / off (rocprofv3 kernel trace)
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in fused unfused; do
  if [ $v = unfused ]; then export HMV_NO_FFFUSE=1 HMV_NO_CHEBFUSE=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2_$v -o cfg2 -- python3 bench.py --workload cfg2 --no-cpu-baseline --steps 50 --warmup 5 --instrument-every 0 > $O/prof_cfg2_$v.json 2> $O/prof_cfg2_$v.err || { tail -5 $O/prof_cfg2_$v.err; exit 60; }
done
python3 - <<'PY'
import csv, glob
for v in ("fused", "unfused"):
    f = glob.glob(f"gpurun_out/r03/prof_cfg2_{v}/**/*kernel_stats.csv", recursive=True)
    print(v, f)
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"  total kernel time {tot/1e6/55:.3f} ms per forward")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
        print(f"  {r['Name'][:90]:90s} calls {int(r['Calls'])//55:3d}/fwd avg {float(r['AverageNs'])/1e3:7.1f} us  {float(r['TotalDurationNs'])/1e6/55*1e3:7.1f} us/fwd")
PY
find gpurun_out/r03 -name "*kernel_trace.csv" -size +5M -delete
