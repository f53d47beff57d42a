# experiment: hr40_lq in the fp16 mode, fuse-layer fusion on / off: token distance, tail error against the f64 oracle on the engine's own tokens
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from helpers import load_case, rel_l2
from handmvnet_amd import HandMvNet
from oracle.oracle import Oracle
from test_gpu_parity import _run
for name in ("hr40_lq",):
    cfg, (tp, mp, dp), sd, (x, bbox, intr), fx = load_case(name)
    o = Oracle(cfg, sd, "f64")
    o32 = Oracle(cfg, sd, "f32")
    for mode in ("f16", "f32"):
        m = HandMvNet(tp, mp, dp); m.load_state_dict(sd)
        if mode == "f16": m.half()
        res = {}
        for on in (True, False):
            m.set_hr_fusion(on)
            got = _run(m, x, bbox, intr)
            ref = o.fuse_tokens(got["tokens"])
            res[on] = got
            r32 = o32.fuse_tokens(got["tokens"])
            print("   oracle f32 vs f64 on these tokens: fused", rel_l2(r32["fused"], ref["fused"]), "cam", rel_l2(r32["joints_cam"], ref["joints_cam"]),
                  " |tokens| max", float(np.abs(got["tokens"]).max()), "rms", float(np.sqrt((got["tokens"] ** 2).mean())))
            tk = got["tokens"].astype(np.float64)
            for eps in (1e-7, 1e-6):
                rng = np.random.default_rng(0)
                pert = o.fuse_tokens((tk * (1 + eps * rng.standard_normal(tk.shape))).astype(np.float32))
                print("   f64 tail, tokens perturbed by", eps, "-> fused moves", rel_l2(pert["fused"], ref["fused"]))
            print(name, mode, "fusion", on, "tail fused err", rel_l2(got["fused"], ref["fused"]), "cam", rel_l2(got["joints_cam"], ref["joints_cam"]),
                  "tokens vs fixture", rel_l2(got["tokens"], fx["tokens_full"]) if "tokens_full" in fx else None)
        print(name, mode, "tokens on vs off", rel_l2(res[True]["tokens"], res[False]["tokens"]), "feat0", rel_l2(res[True]["feat0"], res[False]["feat0"]),
              "fused on vs off", rel_l2(res[True]["fused"], res[False]["fused"]))
