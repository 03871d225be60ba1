import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg
_pkg.load()
import torch
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R
import test_gpu_fitted as TF
from test_gpu_parity import make_renderer
lut = load_lut_rgb()
g, sdc, sdf, _, _ = load_golden("fitted_launch16k")
sd = TF.cancelling_network(g, sdc)
n = 8192
ro, rd = g["rays_o"][:n], g["rays_d"][:n]
good = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=4096, mlp_precision="f16x3_mxfp6x")
good.render_rays(ro, rd, 0.5, 8.0)
r = make_renderer(R, g, sd, sdf, lut, max_rays_per_launch=4096, mlp_precision="f16x3_mxfp6x", query_routing=("coarse_density_15slot", "no_offset_tiers"))
r.set_route(good.route)
res, bits, trip = r._render(torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda(), 0.5, 8.0, None, {}, want_trips=True)
print("bits", bits, "marked", int(trip.sum()), "sel", r.last_selection())
whole = make_renderer(R, g, sd, sdf, lut, max_rays_per_launch=4096, mlp_precision="f16x3_mxfp6x", query_routing=("coarse_density_all_points",)).render_rays(ro, rd, 0.5, 8.0)
for k in ("depth_map", "weights", "target_normal_map", "depth_map0", "weights0"):
    e = ((res[k] - whole[k]).abs().reshape(n, -1).amax(-1) / whole[k].abs().max())
    print(k, "max %.2e" % float(e.max()), "rays>1e-4:", int((e > 1e-4).sum()), "of which marked:", int(((e > 1e-4) & (trip > 0)).sum()))
