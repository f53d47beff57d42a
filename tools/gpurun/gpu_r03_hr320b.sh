O=gpurun_out/r03; mkdir -p $O
for t in 0 1 2 3 9 base; do
if [ $t = base ]; then unset HMV_FORCE_TILE; else export HMV_FORCE_TILE=$t; fi
timeout -k 10 400 python bench.py --workload hr40 --no-cpu-baseline --steps 3 --warmup 1 --per-layer $O/per_layer_hr40_f32b_t$t.json > $O/hr40_f32b_t$t.json 2> $O/hr40_f32b_t$t.err || { tail -5 $O/hr40_f32b_t$t.err; exit 56; }
done
