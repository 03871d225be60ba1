"""Indices of the rays of fitted_wide whose DIRECT channels are worst under f16_mxfp6 / bf16x3 (GPU) -> gpurun_out/worst_rays_direct.npy"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import *
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
g, sdc, sdf, gt, edit = load_golden("fitted_wide")
KEYS = ["depth_map", "acc_map", "albedo_map", "roughness_map", "irradiance_map", "radiance_map", "radiance_map_3"]
tot = np.zeros(1024)
for prec in ("f16_mxfp6", "bf16x3", "f16x3_mxfp6x"):
    r = R.Renderer(64, 128, max_rays_per_launch=4096, mlp_precision=prec)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    res = r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0)
    for k in KEYS:
        ref = g["out__" + k]
        e = np.abs(res[k].cpu().numpy().reshape(ref.shape) - ref).reshape(1024, -1).max(-1) / np.abs(ref).max()
        print(prec, k, "max %.2e" % e.max(), "floor %.2e" % float(g["floor__" + k]), np.argsort(-e)[:4])
        if prec != "f16x3_mxfp6x":
            tot = np.maximum(tot, e / e.max())
idx = np.argsort(-tot)[:96]
np.save(os.path.join(ROOT, "gpurun_out", "worst_rays_direct.npy"), idx)
