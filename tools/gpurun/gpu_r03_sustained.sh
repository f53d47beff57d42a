O=gpurun_out/r03; mkdir -p $O
timeout -k 10 400 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-secondary --instrument-every 0 > $O/bench_sustained_1000steps.json 2> $O/sust32.err || exit 56
python -c "import json; d=json.load(open('$O/bench_sustained_1000steps.json')); print('f32 1000 steps', d['ms_per_step'], d['value'])"
timeout -k 10 400 python bench.py --dtype f16 --steps 3000 --warmup 50 --no-cpu-baseline --instrument-every 0 > $O/bench_f16_sustained_3000steps.json 2> $O/sust16.err || exit 57
python -c "import json; d=json.load(open('$O/bench_f16_sustained_3000steps.json')); print('f16 3000 steps', d['ms_per_step'], d['value'])"
timeout -k 10 100 python bench.py --dtype f16 --steps 20 --warmup 3 --no-cpu-baseline --instrument-every 0 > $O/bench_f16_short_after.json 2> $O/short16.err || exit 58
python -c "import json; d=json.load(open('$O/bench_f16_short_after.json')); print('f16 20 steps right after', d['ms_per_step'])"
