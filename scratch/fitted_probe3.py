"""Where does f16_mxfp6 lose the coarse-pass reflected query of fitted_insert?  (teacher-forced on the recorded points)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import *
import iblnerf_oracle as O
from ibl_nerf_amd import renderer as R
np.set_printoptions(precision=4, linewidth=220, suppress=True)
g, sdc, sdf, gt, edit = load_golden("fitted_insert")
lut = load_lut_rgb()
for prec in ("f16_mxfp6", "f16x3"):
    r = R.Renderer(64, 128, max_rays_per_launch=4096, mlp_precision=prec)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    pts, dirs, ref = g["q_c_refl_pts"], g["q_c_refl_dirs"], g["q_c_refl_raw"]
    raw = r.network_query(pts, dirs, 0).cpu().numpy()
    err = np.abs(raw - ref)
    i = np.unravel_index(np.argmax(err[..., 0]), err[..., 0].shape)
    print(prec, "max sigma abs err %.3e at ray %d sample %d  pt %s  ref sigma %.4f got %.4f |pt| max %.2f" % (err[..., 0].max(), i[0], i[1], pts[i], ref[i][0], raw[i][0], np.abs(pts).max()))
    print(prec, "per-channel max abs err", err.reshape(-1, 18).max(0))
    res = {k: v.cpu().numpy() for k, v in r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0, gt, **edit).items()}
    e = np.abs(res["reflected_radiance_map0"] - g["out__reflected_radiance_map0"]).max(-1)
    j = int(np.argmax(e))
    print(prec, "worst ray", j, "rf0 err", e[j], "got", res["reflected_radiance_map0"][j], "ref", g["out__reflected_radiance_map0"][j])
    print("   ref refl sigma along that ray:", ref[j, :, 0][:24])
    print("   got refl sigma (teacher pts): ", raw[j, :, 0][:24])
    print("   depth0 got/ref", res["depth_map0"][j], g["out__depth_map0"][j], "normal0", res["target_normal_map0"][j], g["out__target_normal_map0"][j], "mask", gt["object_insert_mask"][j, 0] * 255)
    print("   refl pts[0..2]", pts[j, :3], "dir", dirs[j])
