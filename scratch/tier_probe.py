"""Offset tiers (k_importance) as the intermediate between TRUNK_X and three f16 products for the fine offsets: normal metric vs SAFE and frame time, per tau."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg
_pkg.load()
import torch
import bench as Bn
from ibl_nerf_amd import dist as D, renderer as R, binding as B

def posed_c2w():
    g = np.load(os.path.join(ROOT, "tests", "golden", "fitted_posed4k.npz"))
    return np.asarray(g["c2w"], dtype=np.float32)[:3, :4]

K, c2w0 = Bn.camera()
for kind, c2w, tag in (("fitted", posed_c2w(), "posed"), ("fitted2", c2w0, "frontal"), ("fitted3", c2w0, "frontal"), ("fitted3", posed_c2w(), "posed")):
    sdc, sdf = Bn.load_checkpoint(kind)
    r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x")
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(Bn.load_lut())
    ro, rd = r.get_rays(800, 800, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    probe = D.frame_probe_for_call(r, 800, 800, K, c2w, 0.5, 8.0)
    r.decide_route(probe["rays_o"], probe["rays_d"], 0.5, 8.0)
    r._set_routing(B.ROUTE_FINE_MAIN_PRECISE | B.ROUTE_FINE_OFFSETS_PRECISE)
    safe, _, _ = r._render(probe["rays_o"], probe["rays_d"], 0.5, 8.0, None, {})
    r._set_routing(B.ROUTE_FINE_MAIN_PRECISE)
    line = "%s %s:" % (kind, tag)
    for tau in (0.0, 1e-4, 2e-5, 5e-6, 1e-6):
        B.check(r.ctx, r.lib.iblnerf_set_offset_tier_threshold(r.ctx, tau))
        out, _, _ = r._render(probe["rays_o"], probe["rays_d"], 0.5, 8.0, None, {})
        r.render_rays(ro[:131072], rd[:131072], 0.5, 8.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render_rays(ro, rd, 0.5, 8.0)
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
        k = "target_normal_map"
        e = (out[k].double() - safe[k].double()).abs().amax(-1) / safe[k].double().abs().amax()
        line += "  [tau %.0e %.0f ms: p999 %.1e share %.1e]" % (tau, ms, float(torch.quantile(e.cpu(), 0.999)), float((e > 1e-3).double().mean()))
    print(line, flush=True)
