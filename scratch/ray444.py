import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg
_pkg.load()
import torch
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R, binding as B
from test_gpu_parity import make_renderer
lut = load_lut_rgb()
g, sdc, sdf, gt, edit = load_golden("fitted3_posed4k")
out = {}
for name, bits in (("fast", 0), ("tiered", B.ROUTE_FINE_TIERS), ("safe", B.ROUTE_FINE_MAIN_PRECISE | B.ROUTE_FINE_OFFSETS_PRECISE)):
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384, mlp_precision="f16x3_mxfp6x", query_routing=bits)
    out[name] = r.render_rays(g["rays_o"], g["rays_d"], float(g["near"]), float(g["far"]), gt, **edit)
ref = torch.from_numpy(g["out__albedo_map"]).cuda()
for name in out:
    e = (out[name]["albedo_map"] - ref).abs().amax(-1) / ref.abs().max()
    top = torch.topk(e, 3)
    print(name, "albedo worst rays", top.indices.tolist(), ["%.2e" % v for v in top.values.tolist()], "e[444] %.2e" % float(e[444]))
w = {k: out[k]["weights"][444].cpu().numpy() for k in out}
i = np.argsort(-w["safe"])[:12]
print("ray 444 heaviest samples", sorted(i.tolist()))
for k in w:
    print(k, np.array2string(w[k][sorted(i.tolist())], precision=5))
print("sum |tiered - safe| weights", float(np.abs(w["tiered"] - w["safe"]).sum()), "fast", float(np.abs(w["fast"] - w["safe"]).sum()))
