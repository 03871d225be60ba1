import os, sys, json
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
for name in sys.argv[1:]:
    g, sdc, sdf, gt, edit = load_golden(name)
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384)
    r.render_rays(g["rays_o"], g["rays_d"], float(g["near"]), float(g["far"]), gt, **edit)
    p = r.policy
    print(name, p["decision"], "FAST:", p["triggers"], "TIERED:", p.get("triggers_tiered"), {k: "%.1e" % v["p999"] for k, v in (p.get("metrics_tiered") or {}).items()}, "trip_bits", getattr(r, "trip_bits", 0), "esc", r.probe_escalations)
