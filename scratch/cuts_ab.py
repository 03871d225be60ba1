"""Round 5: where to cut the z-chunked estimates (iblnerf_set_chunk_cuts): alternating A/B of frame times on one box; results do not depend on the cuts."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R, binding as B
lut = load_lut_rgb()
name = sys.argv[1] if len(sys.argv) > 1 else "fitted_launch16k"
g, sdc, sdf, gt, edit = load_golden(name)
f_ = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
Kc = np.array([[f_, 0, 400], [0, f_, 400], [0, 0, 1]], dtype=np.float32)
c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
settings = {"built-in": (0, 0, 0, 0), "fine 96/144": (96, 144, 0, 0), "fine 120/156": (120, 156, 0, 0), "fine 128/160": (128, 160, 0, 0), "fine 112/152": (112, 152, 0, 0),
            "refl 16/32": (0, 0, 16, 32), "refl 24/44": (0, 0, 24, 44), "refl 8/24": (0, 0, 8, 24)}
rs = {}
for k, cu in settings.items():
    r = R.Renderer(64, 128, max_rays_per_launch=65536)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    B.check(r.ctx, r.lib.iblnerf_set_chunk_cuts(r.ctx, *cu))
    rs[k] = r
o, d = rs["built-in"].get_rays(800, 800, Kc, c2w)
o, d = o.reshape(-1, 3), d.reshape(-1, 3)
ref = None
for k, r in rs.items():
    out = r.render_rays(o, d, 0.5, 8.0)
    if ref is None:
        ref = out
    else:
        worst = max(float((out[m] - ref[m]).abs().max() / ref[m].abs().max().clamp_min(1e-30)) for m in out if torch.isfinite(ref[m]).all())
        print("%-14s max relative difference of any map against the built-in cuts: %.1e" % (k, worst), flush=True)
torch.cuda.synchronize()
times = {k: [] for k in rs}
for rep in range(4):
    for k, r in rs.items():
        torch.cuda.synchronize(); t0 = time.time()
        r.render_rays(o, d, 0.5, 8.0); r.render_rays(o, d, 0.5, 8.0)
        torch.cuda.synchronize()
        times[k].append((time.time() - t0) / 2 * 1e3)
for k, r in rs.items():
    print("%-14s frame ms %s  median %.1f  executed flops %.4g" % (k, ["%.0f" % t for t in times[k]], np.median(times[k]), r.last_executed_flops()))
