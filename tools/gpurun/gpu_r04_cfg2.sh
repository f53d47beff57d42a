# round 4: per-launch table of cfg-2 (ResNet-18, B8 V4, fp32) and of batch-1 r50
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 300 python bench.py --workload cfg2 --no-cpu-baseline --steps 100 --warmup 10 --instrument-every 1 --per-layer $O/pl_cfg2_f32.json > $O/b_cfg2_f32.json 2> $O/b_cfg2_f32.err || exit 52
timeout -k 10 300 python bench.py --workload cfg3 --batch 1 --no-cpu-baseline --steps 100 --warmup 10 --instrument-every 1 --per-layer $O/pl_b1_f32.json > $O/b_b1_f32.json 2> $O/b_b1_f32.err || exit 53
python - <<'PY'
import json
for n in ("pl_cfg2_f32", "pl_b1_f32"):
    d = json.load(open(f"gpurun_out/r04/{n}.json"))
    print(n, len(d), round(sum(r["avg_ms"] for r in d) * 1e3), "us in conv/GEMM launches")
    for r in d: print(f'  {r["layer"]:34s} {r["kernel"]:44s} {r["avg_ms"]*1e3:6.1f} us  {r["gflop"]:6.2f} GF  {r["tflops"]:6.1f} TF')
PY
