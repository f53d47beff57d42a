#!/usr/bin/env python3
"""Host (CPU) time per forward with eager launches vs hipGraph replay: enqueue N forwards without synchronising and
divide the CPU time spent inside the calls.  Development tool (run on the GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, params  # noqa: E402
from handmvnet_amd import HandMvNet  # noqa: E402
from handmvnet_amd.spec import config_from_params  # noqa: E402
from handmvnet_amd.synth import synth_inputs, synth_state_dict  # noqa: E402


def main():
    for wl, batch in (("cfg2", 8), ("cfg3", 1), ("cfg3", 32)):
        bt, ch, V, B, size = WORKLOADS[wl]
        tp, mp, dp = params(bt, ch, V, batch, size)
        cfg = config_from_params(tp, mp, dp)
        m = HandMvNet(tp, mp, dp)
        m.load_state_dict(synth_state_dict(cfg, 1))
        m.to("cuda").eval()
        x, bbox, intr = synth_inputs(cfg, batch, 5, size)
        xt, bb, cam = torch.from_numpy(x).cuda(), torch.from_numpy(bbox).cuda(), {"intrinsic": torch.from_numpy(intr).cuda()}
        for graphs in (False, True):
            m.use_graphs(graphs)
            for _ in range(6):
                out = m(xt, bb, cam)
            torch.cuda.synchronize()
            n = 50
            t0 = time.perf_counter()
            cpu0 = time.process_time()
            for _ in range(n):
                out = m(xt, bb, cam)
            cpu = (time.process_time() - cpu0) / n
            enq = (time.perf_counter() - t0) / n
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n
            print(f"{wl} B={batch}: graphs={graphs}: host enqueue {enq * 1e3:.3f} ms/forward (cpu {cpu * 1e3:.3f} ms), "
                  f"wall {wall * 1e3:.3f} ms/forward, stats {m.graph_stats()}", flush=True)


if __name__ == "__main__":
    main()
