O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size and f32 and not f32x3" > $O/tests_w8.log 2>&1 || { tail -30 $O/tests_w8.log; exit 40; }
tail -1 $O/tests_w8.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > $O/f32_w8b.json 2> $O/f32_w8b.err || exit 56
python -c "import json; d=json.load(open('$O/f32_w8b.json')); print('f32', d['ms_per_step'], d['value'])"
done
