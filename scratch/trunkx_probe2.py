"""Coarse-grid offsets on the mixed TRUNK form too?  (IBLNERF_X_COARSE=1)  normal0 / normal on fitted fixtures"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import *
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
for name in ("fitted_wide", "fitted_plain", "fitted_insert"):
    g, sdc, sdf, gt, edit = load_golden(name)
    r = R.Renderer(64, 128, max_rays_per_launch=4096, mlp_precision="f16x3_mxfp6x")
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    res = {k: v.cpu().numpy() for k, v in r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0, gt, **edit).items()}
    for k in ("target_normal_map", "target_normal_map0"):
        e = np.abs(res[k] - g["out__" + k]).max(-1)
        print("X_COARSE=%s %-14s %-20s max %.2e p99.9 %.2e p99 %.2e" % (os.environ.get("IBLNERF_X_COARSE", "0"), name, k, e.max(), np.percentile(e, 99.9), np.percentile(e, 99)))
