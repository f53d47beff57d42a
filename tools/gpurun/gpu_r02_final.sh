# the round's record runs (copied into profiles/ afterwards)
O=gpurun_out/r02final; mkdir -p $O
python bench.py --steps 20 --warmup 5 --per-layer $O/per_layer_f32.json > $O/bench.json 2> $O/bench.err || exit 41
python bench.py --dtype f16 --steps 20 --warmup 5 --per-layer $O/per_layer_f16.json --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err || exit 42
python bench.py --dtype f32x3 --steps 20 --warmup 5 --per-layer $O/per_layer_f32x3.json --no-cpu-baseline > $O/bench_f32x3.json 2> $O/bench_f32x3.err || exit 43
python bench.py --workload cfg2 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err || exit 44
python bench.py --workload cfg2 --dtype f16 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_f16.json 2> $O/bench_cfg2_f16.err || exit 45
python bench.py --workload hr40 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_hr40.json 2> $O/bench_hr40.err || exit 46
python bench.py --workload hr40 --dtype f16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_hr40_f16.json 2> $O/bench_hr40_f16.err || exit 47
python bench.py --workload hr40 --dtype f32x3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_hr40_f32x3.json 2> $O/bench_hr40_f32x3.err || exit 48
python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err || exit 49
python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_sustained_1000steps.json 2> $O/bench_sustained.err || exit 51
HMV_BENCH_SAME_DEVICE=1 HMV_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 4 --no-cpu-baseline > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank.err || exit 50
for f in $O/bench*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); r=d['roofline']; print(d['dtype'], d['ms_per_step'], d['value'], d['n_gpus'], r['kernel'], r['bound'], r['achieved'], r['unit'], r['frac'], (d.get('cpu_baseline') or {}).get('value'), d.get('parity_rel_l2_vs_oracle'))"; done
