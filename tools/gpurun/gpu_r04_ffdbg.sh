# round 4: phase stamps of ff_block_kernel (dev build, HMV_FF_DBG=1) at cfg-2 and batch 1: where a 16-token tile's 40 us go
O=gpurun_out/r04; mkdir -p $O
HMV_LIB=build/libhandmv_dev.so HMV_FF_DBG=1 timeout -k 10 200 python bench.py --workload cfg2 --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --instrument-every 0 > $O/ffdbg_cfg2.json 2> $O/ffdbg_cfg2.err || { tail -5 $O/ffdbg_cfg2.err; exit 61; }
grep "ff_block" $O/ffdbg_cfg2.err | tail -12
HMV_LIB=build/libhandmv_dev.so HMV_FF_DBG=1 timeout -k 10 200 python bench.py --dtype f16 --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --instrument-every 0 > $O/ffdbg_cfg3.json 2> $O/ffdbg_cfg3.err || { tail -5 $O/ffdbg_cfg3.err; exit 62; }
grep "ff_block" $O/ffdbg_cfg3.err | tail -12
