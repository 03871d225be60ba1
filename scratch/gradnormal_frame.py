"""A full 800x800 frame in the two depth-gradient normal modes (fitted checkpoint): finite outputs, frame time against the eps-normal
mode (one density-gradient query per sample = 2 trunk-equivalents instead of 4 offset queries), and how close the three normals are."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest as C
import torch
from ibl_nerf_amd import renderer as R, checkpoint as ck
f = np.load(C.GOLDEN + "/fitted_ckpt.npz")
sdc, sdf = ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])
lut = C.load_lut_rgb()
H = W = 800
foc = 0.5 * W / np.tan(0.5 * np.deg2rad(60.0))
K = np.array([[foc, 0, W / 2], [0, foc, H / 2], [0, 0, 1]], np.float32)
c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
normals = {}
for mode in ("normal_map_from_depth_gradient_epsilon", "normal_map_from_depth_gradient", "normal_map_from_depth_gradient_direction"):
    r = R.Renderer(64, 128, normal_mode=mode)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    ro, rd = r.get_rays(H, W, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    res = r.render_rays(ro, rd, 0.5, 8.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = r.render_rays(ro, rd, 0.5, 8.0); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad = {k: int((~torch.isfinite(v)).sum()) for k, v in res.items() if v.dtype.is_floating_point and int((~torch.isfinite(v)).sum())}
    normals[mode] = res["target_normal_map"].cpu().numpy()
    print("%-45s %.2f s/frame  %.3g rays/s  non-finite: %s  fallbacks %d" % (mode, dt, H * W / dt, bad or "none", r.range_fallbacks))
a = normals["normal_map_from_depth_gradient_epsilon"]
for m in ("normal_map_from_depth_gradient", "normal_map_from_depth_gradient_direction"):
    cosang = np.clip((a * normals[m]).sum(-1), -1, 1)
    ang = np.degrees(np.arccos(cosang))
    print("angle between the eps-normal and %-42s median %.2f deg, 90 %% %.2f deg, 99 %% %.2f deg" % (m, np.median(ang), np.percentile(ang, 90), np.percentile(ang, 99)))
