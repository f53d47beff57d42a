import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
from helpers import load_case, rel_l2
from handmvnet_amd import HandMvNet
from cases import CASES
for name in CASES:
    cfg, (tp, mp, dp), sd, (x, bbox, intr), fx = load_case(name)
    m = HandMvNet(tp, mp, dp); m.load_state_dict(sd); m.half(); m.capture_stages(True)
    dev = torch.device("cuda:0")
    out = m(torch.from_numpy(x).to(dev), torch.from_numpy(bbox).to(dev), {"intrinsic": torch.from_numpy(intr).to(dev)})
    torch.cuda.synchronize()
    cam = out["joints_cam"].cpu().numpy(); crop = out["joints_crop_img"].cpu().numpy()
    coords = m.read_stage("coords_hm").cpu().numpy(); feat = m.read_stage("feat0").cpu().numpy().reshape(-1)
    dc = np.abs(coords - fx["coords_hm"]); 
    hm = out["heatmap"].cpu().numpy().reshape(-1)
    print(f"{name:22s} joints_cam rel {rel_l2(cam, fx['joints_cam']):.3e}  feat0 rel {rel_l2(feat[fx['feat0_idx']], fx['feat0_val']):.3e}  hm rel {rel_l2(hm[fx['heatmap_idx']], fx['heatmap_val']):.3e}  coords: max {dc.max():.3f} px, >0.5px: {(dc>0.5).mean()*100:.1f}%  median {np.median(dc):.4f}")
