O=gpurun_out/r03; mkdir -p $O
for i in 1 2; do
  timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline > $O/ab_ht_$i.json 2> $O/ab_ht_$i.err || exit 56
  HMV_NO_HT=1 timeout -k 10 300 python bench.py --dtype f16 --no-cpu-baseline > $O/ab_noht_$i.json 2> $O/ab_noht_$i.err || exit 57
done
python - <<'PY'
import json
for i in (1, 2):
    for tag in ("ht", "noht"):
        d = json.load(open(f"gpurun_out/r03/ab_{tag}_{i}.json")); print(tag, i, d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
