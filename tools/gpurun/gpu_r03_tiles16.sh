O=gpurun_out/r03; mkdir -p $O
for t in 1 2 3 4 5 9 base; do
if [ $t = base ]; then unset HMV_FORCE_TILE; else export HMV_FORCE_TILE=$t; fi
timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline --steps 6 --warmup 2 --per-layer $O/per_layer_f16_t$t.json > $O/f16_t$t.json 2> $O/f16_t$t.err || { echo "tile $t failed"; tail -3 $O/f16_t$t.err; }
done
