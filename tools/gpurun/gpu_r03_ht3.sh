O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python tools/ht_probe.py > $O/probe_ht.txt 2>&1 || { cat $O/probe_ht.txt; exit 30; }
cat $O/probe_ht.txt
bash tools/gpurun/gpu_r03_prof.sh
