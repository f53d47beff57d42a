# round 4: the bench records of the round, one box
O=gpurun_out/r04final; mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; exit 70; }; python -c "
import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['value'], d['dtype'], d.get('launches_per_forward'), d['roofline']['kernel'], d['roofline']['frac'], d['forward'])"; }
run bench_default
run bench_f16 --dtype f16 --no-cpu-baseline --per-layer $O/per_layer_f16.json
run bench_f16_no_events --dtype f16 --no-cpu-baseline --instrument-every 0
run bench_f32_per_layer --no-cpu-baseline --no-secondary --per-layer $O/per_layer_f32.json
run bench_f32x3 --dtype f32x3 --no-cpu-baseline --per-layer $O/per_layer_f32x3.json
run bench_cfg2 --workload cfg2 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0
run bench_cfg2_f16 --workload cfg2 --dtype f16 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0
run bench_b1 --workload cfg3 --batch 1 --no-cpu-baseline --steps 200 --warmup 20 --instrument-every 0
run bench_cfg1 --workload cfg1 --steps 200 --warmup 20 --instrument-every 0
run bench_cfg3_lq --fusion cross_attn_learnable_query --no-cpu-baseline --steps 10 --warmup 2
run bench_cfg3_lq_f16 --fusion cross_attn_learnable_query --dtype f16 --no-cpu-baseline --steps 10 --warmup 2
run bench_hr40 --workload hr40 --no-cpu-baseline --steps 6 --warmup 2
run bench_hr40_f16 --workload hr40 --dtype f16 --no-cpu-baseline --steps 10 --warmup 2
run bench_hr40_f32x3 --workload hr40 --dtype f32x3 --no-cpu-baseline --steps 6 --warmup 2
HMV_BENCH_SAME_DEVICE=1 HMV_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank_rehearsal.err || { echo "FAILED 2rank"; tail -5 $O/bench_2rank_rehearsal.err; exit 71; }
python -c "
import json; d=json.load(open('$O/bench_2rank_rehearsal.json')); print('2rank', d['ms_per_step'], d['value'], d['n_gpus'], d['communicator'])"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?
tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 72; }
tail -2 $O/smoke.log
