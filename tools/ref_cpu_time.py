#!/usr/bin/env python3
"""Times the REAL reference (PyTorch on the CPU, /root/reference imported through tests/golden/ref_harness.py) on the
shapes bench.py's cpu_baseline samples, in the BUILD container, and writes profiles/ref_pytorch_cpu.json.  bench.py
quotes that figure beside the C port it times on the GPU box's host cores (the reference itself cannot travel there).

    python tools/ref_cpu_time.py            # build container only (needs /root/reference)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, q)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ref_harness import build_reference_model  # noqa: E402
from handmvnet_amd.spec import config_from_params  # noqa: E402
from handmvnet_amd.synth import synth_inputs, synth_state_dict  # noqa: E402

sys.path.insert(0, ROOT)
import bench  # noqa: E402

SHAPES = {"cfg3": 256, "cfg2": 256, "hr40": 256, "cfg1": 128}   # cfg1 = BASELINE configs[0]: HO3D_HandMvNet.yaml, B1 x V4 x 128x128


def main():
    out = {"where": f"build container, torch {torch.__version__} CPU, {torch.get_num_threads()} threads "
                    f"({os.cpu_count()} CPUs visible)",
           "note": "the reference's own HandMvNet.forward (eval mode, random init) through PyTorch on the CPU; "
                   "B=1 multi-view sample, the same shape bench.py's cpu_baseline times the C port on",
           "shapes": {}}
    for wl, size in SHAPES.items():
        bt, ch, V, B, _ = bench.WORKLOADS[wl]
        tp, mp, dp = bench.params(bt, ch, V, 1, size)
        cfg = config_from_params(tp, mp, dp)
        sd = synth_state_dict(cfg, 1)
        model = build_reference_model(tp, mp, dp, sd)
        x, bbox, intr = synth_inputs(cfg, 1, 4242, size)
        xt, bt_, it = torch.from_numpy(x), torch.from_numpy(bbox), torch.from_numpy(intr)
        cam = {"intrinsic": it, "extrinsic": torch.zeros(1, V, 4, 4)}
        with torch.no_grad():
            model(xt, bt_, cam)
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < 12.0:
                model(xt, bt_, cam)
                n += 1
            el = time.perf_counter() - t0
        out["shapes"][f"{bt}_B1_V{V}_{size}"] = {"value": round(n * V / el, 2), "unit": "frames/s", "cores": torch.get_num_threads(),
                                                "ms_per_forward": round(el / n * 1e3, 1), "forwards": n}
        print(wl, out["shapes"][f"{bt}_B1_V{V}_{size}"], flush=True)
    with open(os.path.join(ROOT, "profiles", "ref_pytorch_cpu.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
