"""One frame under an imposed route and a pinned table, for rocprofv3 --kernel-trace --stats:  python scratch/trace_frame.py <checkpoint> <fast|tiered|safe> [tau_main]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
_pkg.load()
import torch
import bench as Bn
from ibl_nerf_amd import dist as D, renderer as R, binding as B
kind, table = sys.argv[1], sys.argv[2]
K, c2w = Bn.camera()
sdc, sdf = Bn.load_checkpoint(kind)
r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x")
r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(Bn.load_lut())
ro, rd = r.get_rays(800, 800, K, c2w)
ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
probe = D.frame_probe_for_call(r, 800, 800, K, c2w, 0.5, 8.0)
r.decide_route(probe["rays_o"], probe["rays_d"], 0.5, 8.0)
if len(sys.argv) > 3:
    B.check(r.ctx, r.lib.iblnerf_set_tier_thresholds(r.ctx, 0.0, float(sys.argv[3])))
r._set_routing({"fast": 0, "tiered": B.ROUTE_FINE_TIERS, "safe": B.ROUTE_FINE_MAIN_PRECISE | B.ROUTE_FINE_OFFSETS_PRECISE}[table])
for _ in range(2):
    r.render_rays(ro, rd, 0.5, 8.0)
torch.cuda.synchronize()
print("done", r.describe_route())
