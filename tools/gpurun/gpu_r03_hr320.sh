O=gpurun_out/r03; mkdir -p $O
for t in 9 0 1; do
HMV_FORCE_TILE=$t timeout -k 10 400 python bench.py --workload hr40 --no-cpu-baseline --steps 3 --warmup 1 --per-layer $O/per_layer_hr40_f32_t$t.json > $O/hr40_f32_t$t.json 2> $O/hr40_f32_t$t.err || { tail -5 $O/hr40_f32_t$t.err; exit 56; }
done
timeout -k 10 400 python bench.py --workload hr40 --no-cpu-baseline --steps 3 --warmup 1 --per-layer $O/per_layer_hr40_f32_base.json > $O/hr40_f32_base.json 2> $O/hr40_f32_base.err || exit 57
