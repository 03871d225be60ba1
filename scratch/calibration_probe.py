"""What the load-time calibration (Renderer.calibrate, mlp_precision="auto") measures on each launch-scale fixture: FAST vs SAFE routing of one context
on up to 4 096 of the fixture's rays, and on other subsets (robustness of the decision against the choice of rays).   python scratch/calibration_probe.py"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from test_gpu_parity import make_renderer
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
for name in sys.argv[1:] or ["fitted_launch16k", "fitted_posed4k", "fitted_edit_cfg4", "fitted_insert_cfg5", "fitted2_launch4k", "fitted2_posed4k", "plain_g10", "plain_g16"]:
    g, sdc, sdf, gt, edit = load_golden(name)
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384)
    n = g["rays_o"].shape[0]
    ro, rd = torch.from_numpy(g["rays_o"]).cuda(), torch.from_numpy(g["rays_d"]).cuda()
    for label, idx in (("strided 4096", torch.linspace(0, n - 1, min(n, 4096)).long()), ("first 1024", torch.arange(min(n, 1024))), ("last 1024", torch.arange(max(n - 1024, 0), n)),
                       ("every 7th", torch.arange(0, n, 7))):
        idx = idx.cuda()
        r.policy = None
        p = r.calibrate(ro[idx].contiguous(), rd[idx].contiguous(), float(g["near"]), float(g["far"]))
        print("%-18s %-13s %4d rays -> %-4s %s %s" % (name, label, p["rays"], p["decision"], "  ".join("%s %.1e/%.2f%%" % (k.replace("_map", "").replace("target_", ""), v["p999"], 100 * v["above_1e-3"]) for k, v in p["metrics"].items()),
                                                  p["triggers"]), flush=True)
