"""Prints the fp16 path's deviation from the reference fixtures for every case (development tool, GPU box).
Set HMV_PROBE_MODE=f32x3 to probe the split-precision path instead."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")): sys.path.insert(0, q)
from helpers import load_case, rel_l2
from handmvnet_amd import HandMvNet
from cases import CASES
import json
NOISE = json.load(open(os.path.join(ROOT, "tests", "golden", "fp16_noise.json")))["cases"]
for name in CASES:
    cfg, (tp, mp, dp), sd, (x, bbox, intr), fx = load_case(name)
    m = HandMvNet(tp, mp, dp); m.load_state_dict(sd); (m.float32x3() if os.environ.get("HMV_PROBE_MODE") == "f32x3" else m.half()); m.capture_stages(True)
    dev = torch.device("cuda:0")
    out = m(torch.from_numpy(x).to(dev), torch.from_numpy(bbox).to(dev), {"intrinsic": torch.from_numpy(intr).to(dev)})
    torch.cuda.synchronize()
    cam = out["joints_cam"].cpu().numpy(); crop = out["joints_crop_img"].cpu().numpy()
    coords = m.read_stage("coords_hm").cpu().numpy(); feat = m.read_stage("feat0").cpu().numpy().reshape(-1)
    dc = np.abs(coords - fx["coords_hm"]); 
    hm = out["heatmap"].cpu().numpy().reshape(-1)
    print(f"{name:22s} joints_cam rel {rel_l2(cam, fx['joints_cam']):.3e}  feat0 rel {rel_l2(feat[fx['feat0_idx']], fx['feat0_val']):.3e}  hm rel {rel_l2(hm[fx['heatmap_idx']], fx['heatmap_val']):.3e}  coords: max {dc.max():.3f} px, >0.5px: {(dc>0.5).mean()*100:.1f}%  median {np.median(dc):.4f}  | floor: {NOISE.get(name)}")
