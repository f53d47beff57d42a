# round 3: where a conv_hs block's time goes (in-kernel phase stamps of workgroup 0's first wave)
O=gpurun_out/r03hsphase; mkdir -p $O
HMV_BENCH_CLOCK=1 HMV_BENCH_PHASES=1 HMV_BENCH_DTYPE=f16 timeout -k 10 200 python tools/hs_probe.py > $O/phases.txt 2>&1 || { tail -5 $O/phases.txt; exit 51; }
cat $O/phases.txt
