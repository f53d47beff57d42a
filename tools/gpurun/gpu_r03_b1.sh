O=gpurun_out/r03; mkdir -p $O
for i in 1 2; do
timeout -k 10 300 python bench.py --workload cfg3 --batch 1 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/b1_new.json 2> $O/b1_new.err || exit 57
python -c "import json; d=json.load(open('$O/b1_new.json')); print('b1', d['ms_per_step'])"
timeout -k 10 300 python bench.py --workload cfg3 --batch 2 --no-cpu-baseline --steps 100 --warmup 20 --instrument-every 0 > $O/b2_new.json 2> $O/b2_new.err || exit 57
python -c "import json; d=json.load(open('$O/b2_new.json')); print('b2', d['ms_per_step'])"
done
timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/cfg2_new.json 2> $O/cfg2_new.err || exit 58
python -c "import json; d=json.load(open('$O/cfg2_new.json')); print('cfg2', d['ms_per_step'])"
timeout -k 10 300 python bench.py --workload cfg1 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0 > $O/cfg1_new.json 2> $O/cfg1_new.err || exit 58
python -c "import json; d=json.load(open('$O/cfg1_new.json')); print('cfg1', d['ms_per_step'])"
