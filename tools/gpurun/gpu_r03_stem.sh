O=gpurun_out/r03; mkdir -p $O
for t in 9 2 0 base; do
if [ $t = base ]; then unset HMV_FORCE_TILE HMV_FORCE_TILE_ALL; else export HMV_FORCE_TILE=$t HMV_FORCE_TILE_ALL=1; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 --per-layer $O/per_layer_f32_all_t$t.json > $O/f32_all_t$t.json 2> $O/f32_all_t$t.err || { echo "tile $t failed"; tail -3 $O/f32_all_t$t.err; }
done
