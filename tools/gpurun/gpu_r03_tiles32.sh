O=gpurun_out/r03; mkdir -p $O
for t in 2 3 4 5 6 7 8 10 base; do
if [ $t = base ]; then unset HMV_FORCE_TILE; else export HMV_FORCE_TILE=$t; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 --per-layer $O/per_layer_f32_t$t.json > $O/f32_t$t.json 2> $O/f32_t$t.err || { echo "tile $t failed"; tail -3 $O/f32_t$t.err; }
done
