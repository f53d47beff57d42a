# round 4: PMC pass on the hr40 fp16 workload to see what the fused fuse-layer launch spends its cycles on
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04hrf; mkdir -p $O
B="python3 bench.py --workload hr40 --dtype f16 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $O/sq1 -- $B > $O/b1.json 2> $O/sq1.err || exit 33
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVES --output-format csv -d $O/sq2 -- $B > $O/b2.json 2> $O/sq2.err || exit 34
python3 - <<'PY'
import csv, glob, collections
for d in ("sq1", "sq2"):
    f = glob.glob(f"gpurun_out/r04hrf/{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "hr_fuse" not in k and "conv_hs_f16<3, 3, 5" not in k: continue
        key = k[:60] + " grid=" + r.get("Grid_Size", "")
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
    for key, v in acc.items():
        print(d, key)
        for c, x in v.items(): print("    ", c, x / cnt[(key, c)])
PY
find $O -name "*.csv" -size +3M -delete
