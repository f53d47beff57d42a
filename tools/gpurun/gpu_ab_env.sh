# usage: bash tools/gpu_ab_env.sh ENVVAR [bench args...]: same-box A/B of one environment switch (unset vs =1), interleaved twice
V=$1; shift
O=gpurun_out/r02ab; mkdir -p $O
for rep in 1 2; do
  for on in 0 1; do
    if [ $on = 1 ]; then export $V=1; else unset $V; fi
    python bench.py --steps 12 --warmup 3 --no-cpu-baseline "$@" > $O/${V}_${on}_$rep.json 2> $O/${V}_${on}_$rep.err || exit 13
  done
done
unset $V
for f in $O/${V}_*.json; do echo $f $(python -c "import json; d=json.load(open('$f')); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"); done
