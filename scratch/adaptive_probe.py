"""How many rays NEED the precise kernels?  Per-ray error of the all-fast mode (f16_mxfp6) against the reference on the 16 384-ray fixture,
beside the default mode's: fractions of rays above thresholds.    python scratch/adaptive_probe.py"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from test_gpu_parity import make_renderer, to_np
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
g, sdc, sdf, gt, edit = load_golden("fitted_launch16k")
res = {}
for mode in ("f16_mxfp6", "f16x3_mxfp6x", "f16x3"):
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384, mlp_precision=mode)
    res[mode] = to_np(r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0))
def per_ray(a, key):
    ref = g["out__" + key].astype(np.float64)
    return np.abs(a[key].astype(np.float64).reshape(ref.shape) - ref).reshape(len(ref), -1).max(-1) / np.abs(ref).max()
for key, ths in (("depth_map0", (1e-6, 1e-5, 1e-4)), ("target_normal_map0", (5e-5, 2e-4, 1e-3)), ("depth_map", (1e-5, 1e-4, 1e-3)), ("target_normal_map", (1e-4, 3e-4, 1e-3)), ("albedo_map", (1e-5, 1e-4, 1e-3))):
    for mode in res:
        e = per_ray(res[mode], key)
        print("%-20s %-14s " % (key, mode) + "  ".join("frac(e>%.0e)=%.4f" % (t, (e > t).mean()) for t in ths) + "  p50 %.1e p90 %.1e p99 %.1e max %.1e" % tuple(np.percentile(e, [50, 90, 99, 100])), flush=True)
# how are the fast mode's errors distributed against the coarse weights' peak (a cheap conditioning indicator)?
w0 = res["f16x3"]["weights0"]; peak = w0.max(-1); e = per_ray(res["f16_mxfp6"], "target_normal_map0")
for lo, hi in ((0, 0.5), (0.5, 0.8), (0.8, 0.95), (0.95, 0.999), (0.999, 1.01)):
    m = (peak >= lo) & (peak < hi)
    print("coarse peak weight in [%.3f, %.3f): %5.1f %% of rays, fast-mode normal0 error p50 %.1e p99 %.1e max %.1e" % (lo, hi, 100 * m.mean(), *(np.percentile(e[m], [50, 99, 100]) if m.any() else (0, 0, 0))))
