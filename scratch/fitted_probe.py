"""Prints the HIP path's errors on the fitted fixtures (no asserts): per-channel MLP stage errors, end-to-end map errors,
teacher-forced pass errors.  Run on the GPU box: python scratch/fitted_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as CT
from conftest import *
import iblnerf_oracle as O
from ibl_nerf_amd import renderer as R
import torch
lut = load_lut_rgb()
np.set_printoptions(precision=1, linewidth=250)
for prec in (sys.argv[1:] or ("bf16x3", "f16_mxfp6", "f16_mixed", "f16x3", "f16x3_mxfp6")):
    g, sdc, sdf, gt, edit = load_golden("fitted_plain")
    r = R.Renderer(64, 128, max_rays_per_launch=4096, mlp_precision=prec)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    for p, which in (("c", 0), ("f", 1)):
        raw = r.network_query(g["q_%s_main_pts" % p], g["q_%s_main_dirs" % p], which).cpu().numpy()
        ref = g["q_%s_main_raw" % p]
        refl = r.network_query(g["q_%s_refl_pts" % p], g["q_%s_refl_dirs" % p], which).cpu().numpy()
        sig = r.network_query(g["q_%s_eps_pts" % p], None, which).cpu().numpy()
        print(prec, p, "main rel", np.array([rel_linf(raw[..., c], ref[..., c]) for c in range(18)]))
        print(prec, p, "main abs", np.abs(raw - ref).reshape(-1, 18).max(0), "ref absmax", np.abs(ref).reshape(-1, 18).max(0))
        print(prec, p, "refl rel", np.array([rel_linf(refl[..., c], g["q_%s_refl_raw" % p][..., c]) for c in range(18)]))
        print(prec, p, "eps sigma rel %.1e abs %.1e" % (rel_linf(sig, g["q_%s_eps_sigma" % p]), np.abs(sig - g["q_%s_eps_sigma" % p]).max()))
        o = O.network_query(sdc if which == 0 else sdf, g["q_%s_main_pts" % p][:16], g["q_%s_main_dirs" % p][:16])
        print(prec, p, "oracle main abs", np.abs(o - ref[:16]).reshape(-1, 18).max(0))
    for name in FITTED_FIXTURES:
        g, sdc, sdf, gt, edit = load_golden(name)
        res = {k: v.cpu().numpy() for k, v in r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0, gt, **edit).items()}
        print(prec, name, "fallbacks", r.range_fallbacks, " ".join("%s %.1e" % (k.replace("_map", "").replace("reflected", "rf").replace("radiance", "rad"), rel_linf(res[k], g["out__" + k])) for k in res))
