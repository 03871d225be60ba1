"""Indices of the rays of fitted_wide whose normal is worst under f16x3_main and f16_mxfp6 (GPU) -> gpurun_out/worst_rays.npy"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import *
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
g, sdc, sdf, gt, edit = load_golden("fitted_wide")
tot = np.zeros(1024)
for prec in ("f16x3_main", "f16_mxfp6", "bf16x3"):
    r = R.Renderer(64, 128, max_rays_per_launch=4096, mlp_precision=prec)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    res = r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0)
    e = np.abs(res["target_normal_map"].cpu().numpy() - g["out__target_normal_map"]).max(-1)
    print(prec, "worst", np.argsort(-e)[:10], e[np.argsort(-e)[:10]])
    tot = np.maximum(tot, e / e.max())
idx = np.argsort(-tot)[:96]
np.save(os.path.join(ROOT, "gpurun_out", "worst_rays.npy"), idx)
print(idx)
