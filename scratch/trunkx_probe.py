"""The fast kernel's mixed TRUNK form (VAR_TRUNK_X): stage error on the fitted checkpoint, speed, end-to-end normal on 1 024 rays.
IBLNERF_X_USER=1 python scratch/trunkx_probe.py"""
import os, sys
os.environ["IBLNERF_X_USER"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import *
from ibl_nerf_amd import renderer as R, checkpoint as ck
import torch
lut = load_lut_rgb()
g, sdc, sdf, gt, edit = load_golden("fitted_plain")
for prec in ("f16x3_mxfp6x", "f16_mxfp6", "f16x3"):
    r = R.Renderer(64, 128, max_rays_per_launch=4096, mlp_precision=prec)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    for p, which in (("c", 0), ("f", 1)):
        sig = r.network_query(g["q_%s_eps_pts" % p], None, which).cpu().numpy()
        e = np.abs(sig - g["q_%s_eps_sigma" % p])
        print("%-13s %s offset-query sigma: abs err max %.2e rms %.2e  (finite %s)" % (prec, p, e.max(), np.sqrt((e ** 2).mean()), np.isfinite(sig).all()))
    # host packer vs device packer for the mixed form
    if prec == "f16x3_mxfp6x":
        rd_ = R.Renderer(64, 128, max_rays_per_launch=64, mlp_precision=prec)
        rd_.load_weights(1, {k: torch.from_numpy(v).cuda() for k, v in sdf.items()})
        a = r.network_query(g["q_f_eps_pts"][:64], None, 1); b = rd_.network_query(g["q_f_eps_pts"][:64], None, 1)
        print("device-packed == host-packed:", bool(torch.equal(a, b)))
        for n in (1, 31, 33, 127, 129):
            pts = torch.rand((n, 7, 3), device="cuda") * 6 - 3
            x = r.network_query(pts, None, 1)
            y = R.Renderer(64, 128, max_rays_per_launch=64, mlp_precision="f16x3")
            y.load_weights(1, sdf)
            print("ragged n=%d max |x - f16x3| %.2e" % (n, float((x - y.network_query(pts, None, 1)).abs().max())))
    N, S = 65536, 256
    pts = torch.rand((N, S, 3), device="cuda") * 8 - 4
    r.network_query(pts, None, 1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); r.network_query(pts, None, 1); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    x0 = r.network_query(pts[:2048], None, 1); x1 = r.network_query(pts[:2048], None, 1)
    print("%-13s TRUNK %d x %d: %.2f ms   repeatable %s" % (prec, N, S, sorted(ts)[2], bool(torch.equal(x0, x1))), flush=True)
gw, sdc, sdf, _, _ = load_golden("fitted_wide")
for prec in ("f16x3_mxfp6x", "f16x3_mxfp6", "f16x3_main"):
    r = R.Renderer(64, 128, max_rays_per_launch=4096, mlp_precision=prec)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    res = {k: v.cpu().numpy() for k, v in r.render_rays(gw["rays_o"], gw["rays_d"], 0.5, 8.0).items()}
    e = np.abs(res["target_normal_map"] - gw["out__target_normal_map"]).max(-1)
    d = np.abs(res["depth_map"] - gw["out__depth_map"]) / np.abs(gw["out__depth_map"]).max()
    print("%-13s fitted_wide normal: max %.2e p99.9 %.2e p99 %.2e | depth max %.1e | prefiltered %.1e | fallbacks %d" % (
        prec, e.max(), np.percentile(e, 99.9), np.percentile(e, 99), d.max(), rel_linf(res["prefiltered_reflected_map"], gw["out__prefiltered_reflected_map"]), r.range_fallbacks))
