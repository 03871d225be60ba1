"""Round 5: is a looser transmittance threshold a real gain?  Alternating A/B on one box: frame time and refined share per setting."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R, binding as B
lut = load_lut_rgb()
g, sdc, sdf, gt, edit = load_golden("fitted_launch16k")
f_ = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
Kc = np.array([[f_, 0, 400], [0, f_, 400], [0, 0, 1]], dtype=np.float32)
c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
settings = {"r4": (1e-8, 1e-10, 1e-12), "own": (1e-8, 1e-10, 0.0), "chunk10": (1e-8, 1e-10, 1e-10)}
rs = {}
for k, tm in settings.items():
    r = R.Renderer(64, 128, max_rays_per_launch=65536)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    B.check(r.ctx, r.lib.iblnerf_set_select_tmin(r.ctx, *tm))
    rs[k] = r
o, d = rs["r4"].get_rays(800, 800, Kc, c2w)
o, d = o.reshape(-1, 3), d.reshape(-1, 3)
for r in rs.values():
    r.render_rays(o, d, 0.5, 8.0)
torch.cuda.synchronize()
times = {k: [] for k in rs}
for rep in range(5):
    for k, r in rs.items():
        torch.cuda.synchronize(); t0 = time.time()
        r.render_rays(o, d, 0.5, 8.0); r.render_rays(o, d, 0.5, 8.0)
        torch.cuda.synchronize()
        times[k].append((time.time() - t0) / 2 * 1e3)
for k, r in rs.items():
    sel = r.last_selection()
    print("%-8s %s  frame ms %s  median %.1f  refined %.4f" % (k, settings[k], ["%.0f" % t for t in times[k]], np.median(times[k]), sel[0] / max(sel[1], 1)))
