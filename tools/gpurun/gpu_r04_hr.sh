# round 4: per-layer tables of the HRNet-w40 workload (fp16 and fp32) to see where its 340 launches spend their time
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 300 python bench.py --workload hr40 --dtype f16 --no-cpu-baseline --steps 8 --warmup 2 --per-layer $O/pl_hr40_f16.json > $O/b_hr40_f16.json 2> $O/b_hr40_f16.err || exit 52
timeout -k 10 300 python bench.py --workload hr40 --no-cpu-baseline --steps 5 --warmup 2 --per-layer $O/pl_hr40_f32.json > $O/b_hr40_f32.json 2> $O/b_hr40_f32.err || exit 53
python tools/per_layer_table.py $O/pl_hr40_f16.json > $O/pl_hr40_f16.md
python tools/per_layer_table.py $O/pl_hr40_f32.json > $O/pl_hr40_f32.md
wc -l $O/pl_hr40_f16.md
