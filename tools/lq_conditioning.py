#!/usr/bin/env python3
"""How ill-conditioned is the learnable-query fusion on HRNet-w40 tokens?  (build container only: imports /root/reference)

VERDICT r2 item 4: the engine's `fused` was 2e-2 from the f64 oracle on an HRNet-w40 4-level learnable-query configuration where
the fp32 oracle was 1.8e-3.  This script measures, ON THE REFERENCE'S OWN MODULE IN FLOAT64, by how much a relative perturbation
of the token matrix is amplified into the fusion output and into joints_cam: MultiHeadAttentionLearnableQuery has no LayerNorm
around the attention (layers.py:240-301), the activations grow to several hundred and the softmax turns near one-hot, so the map
tokens -> fused has a large Lipschitz constant.  An fp32 implementation whose tokens carry 1e-5 relative rounding noise (any GPU
summation order) then lands amplification x 1e-5 from the exact result, whatever its kernels do.

    python tools/lq_conditioning.py > profiles/r03_probe_lq_hr40.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, q)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ref_harness  # noqa: E402
from cases import CASES, case_params  # noqa: E402
from handmvnet_amd.spec import config_from_params  # noqa: E402
from handmvnet_amd.synth import synth_inputs, synth_state_dict  # noqa: E402


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    for name in ("hr40_lq", "r50_lq", "r18_lq_wocam", "hr40_tiny"):
        spec = CASES[name]
        tp, mp, dp = case_params(spec)
        cfg = config_from_params(tp, mp, dp)
        sd = synth_state_dict(cfg, spec["wseed"])
        model = ref_harness.build_reference_model(tp, mp, dp, sd)
        x, bbox, intr = synth_inputs(cfg, spec["B"], spec["iseed"], spec["size"])
        st = {}
        hk = [model.joints_late_fusion.register_forward_pre_hook(lambda m, i: st.__setitem__("tokens", i[0].detach().clone())),
              model.joints_late_fusion.register_forward_hook(lambda m, i, o: st.__setitem__("fused", o.detach().clone()))]
        with torch.no_grad():
            out = model(torch.from_numpy(x), torch.from_numpy(bbox), {"intrinsic": torch.from_numpy(intr)})
        for h in hk:
            h.remove()
        tokens32, fused32 = st["tokens"], st["fused"]
        fus = model.joints_late_fusion.double()
        for mod in fus.modules():   # plain float attributes (the PE table) do not follow .double()
            for k, v in list(vars(mod).items()):
                if torch.is_tensor(v) and v.dtype == torch.float32:
                    setattr(mod, k, v.double())
        is_lq = mp["fusion"] == "cross_attn_learnable_query"
        call = (lambda t: fus(t)) if is_lq else (lambda t: fus(t, add_pos="sin" in mp["pos_enc"]))
        with torch.no_grad():
            t64 = tokens32.double()
            f64 = call(t64)
            print(f"{name}: fusion={mp['fusion']}  d={cfg.feat_dim}  |tokens| max {t64.abs().max():.3g}  |fused| max {f64.abs().max():.3g}")
            print(f"  reference fp32 fused vs the same module in float64 on the same tokens: {rel(fused32.double(), f64):.3e}")
            g = torch.Generator().manual_seed(7)
            for eps in (1e-7, 1e-6, 1e-5):
                amps = []
                for _ in range(3):
                    noise = torch.randn(t64.shape, generator=g, dtype=torch.float64)
                    tp_ = t64 * (1.0 + eps * noise)
                    amps.append(rel(call(tp_), f64) / rel(tp_, t64))
                print(f"  relative token perturbation {eps:.0e} (3 draws): d(fused) / d(tokens) = " + ", ".join(f"{a:8.1f}" for a in amps))
        print(f"  joints_cam of the fp32 reference: max |.| {out['joints_cam'].abs().max():.3g}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
