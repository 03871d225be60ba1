"""Round 5: what Renderer.calibrate measures (FAST vs SAFE on the same context) on each fixture's rays: per map p99.9 and the share of rays above 1e-3."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
for name in ("fitted_launch16k", "fitted_posed4k", "fitted2_launch4k", "fitted2_posed4k", "fitted3_launch4k", "fitted3_posed4k"):
    g, sdc, sdf, gt, edit = load_golden(name)
    r = R.Renderer(64, 128, max_rays_per_launch=16384)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    n = g["rays_o"].shape[0]
    ro, rd = torch.from_numpy(g["rays_o"]).cuda(), torch.from_numpy(g["rays_d"]).cuda()
    for idx in (torch.linspace(0, n - 1, 4096).long(), torch.arange(2048), torch.arange(n - 2048, n), torch.arange(0, n, 2)[:4096]):
        r.policy = None
        p = r.calibrate(ro[idx.cuda()].contiguous(), rd[idx.cuda()].contiguous(), 0.5, 8.0)
        print(name, len(idx), p["decision"], {k: ("%.1e" % v["p999"], "%.1e" % v["above_1e-3"]) for k, v in p["metrics"].items()}, p["triggers"], flush=True)
