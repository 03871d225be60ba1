"""Per-ray anatomy of the end-to-end error on fitted_plain (no asserts)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import *
import iblnerf_oracle as O
from ibl_nerf_amd import renderer as R
import torch
lut = load_lut_rgb()
np.set_printoptions(precision=2, linewidth=250, suppress=False)
g, sdc, sdf, gt, edit = load_golden("fitted_plain")
n = g["rays_o"].shape[0]
zc = O.coarse_z(0.5, 8.0, 64, n)
zmid = 0.5 * (zc[:, 1:] + zc[:, :-1])
zref = np.sort(np.concatenate([zc, g["pdf_samples"]], -1), -1)
for prec in (sys.argv[1:] or ("bf16x3", "f16_mxfp6", "f16x3", "f16x3_mxfp6")):
    r = R.Renderer(64, 128, max_rays_per_launch=4096, mlp_precision=prec)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    res = {k: v.cpu().numpy() for k, v in r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0).items()}
    zs = r.sample_pdf(zmid, res["weights0"][:, 1:-1], 128).cpu().numpy()
    zs_teacher = r.sample_pdf(g["pdf_bins"], g["pdf_weights"], 128).cpu().numpy()
    zf = np.sort(np.concatenate([zc, zs], -1), -1)
    dz = np.abs(zf - zref)
    w0err = np.abs(res["weights0"] - g["out__weights0"]).max(-1)
    derr = np.abs(res["depth_map"] - g["out__depth_map"]) / np.abs(g["out__depth_map"]).max()
    werr = np.abs(res["weights"] - g["out__weights"]).max(-1) / np.abs(g["out__weights"]).max()
    aerr = np.abs(res["albedo_map"] - g["out__albedo_map"]).max(-1)
    nerr = np.abs(res["target_normal_map"] - g["out__target_normal_map"]).max(-1)
    print(prec, "teacher-forced sample_pdf max dz %.1e" % np.abs(zs_teacher - g["pdf_samples"]).max())
    print(prec, "rays with a z sample off by > 1e-3:", int((dz.max(-1) > 1e-3).sum()), "of", n)
    order = np.argsort(-derr)[:12]
    print("ray  depth_err  weights_err  albedo_err  normal_err  w0_err  max_dz  n_dz>1e-3  n_dz>1e-4  ref_max_w0")
    for i in order:
        print("%3d  %.1e   %.1e   %.1e   %.1e   %.1e  %.1e  %3d %3d  %.3f" % (i, derr[i], werr[i], aerr[i], nerr[i], w0err[i], dz[i].max(), (dz[i] > 1e-3).sum(), (dz[i] > 1e-4).sum(), g["out__weights0"][i].max()))
    print("median depth err %.1e  90th pct %.1e  max %.1e" % (np.median(derr), np.percentile(derr, 90), derr.max()))
    # the oracle (fp32 numpy) on the same rays for comparison
    if prec == "bf16x3":
        ora = O.render_rays(sdc, sdf, g["rays_o"], g["rays_d"], 0.5, 8.0, lut)
        oderr = np.abs(ora["depth_map"] - g["out__depth_map"]) / np.abs(g["out__depth_map"]).max()
        print("oracle: median depth err %.1e  90th %.1e  max %.1e" % (np.median(oderr), np.percentile(oderr, 90), oderr.max()))
