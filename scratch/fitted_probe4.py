"""Per-ray error distribution on the 1024-ray fitted fixture, every mode, against the reference (no asserts)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import *
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
g, sdc, sdf, gt, edit = load_golden("fitted_wide")
keys = ["depth_map", "albedo_map", "roughness_map", "irradiance_map", "weights", "target_normal_map", "prefiltered_reflected_map", "color_map", "depth_map0", "target_normal_map0"]
def per_ray(a, b):
    a = a.reshape(a.shape[0], -1).astype(np.float64); b = b.reshape(b.shape[0], -1).astype(np.float64)
    return np.abs(a - b).max(-1) / np.abs(b).max()
print("floor(ref fp64 vs fp32):", " ".join("%s %.1e" % (k.replace("_map", ""), float(g["floor__" + k])) for k in keys))
for prec in (sys.argv[1:] or ("f16x3_mxfp6", "f16x3", "f16_mxfp6", "bf16x3")):
    r = R.Renderer(64, 128, max_rays_per_launch=4096, mlp_precision=prec)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    res = {k: v.cpu().numpy() for k, v in r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0).items()}
    print(prec, "fallbacks", r.range_fallbacks)
    for k in keys:
        e = per_ray(res[k], g["out__" + k])
        print("   %-28s max %.1e  p99.9 %.1e  p99 %.1e  p90 %.1e  median %.1e   (x floor: %.1f)" % (k, e.max(), np.percentile(e, 99.9), np.percentile(e, 99), np.percentile(e, 90), np.median(e), e.max() / float(g["floor__" + k])))
