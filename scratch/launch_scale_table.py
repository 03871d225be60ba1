"""Compact parity table at launch scale: HIP path vs the reference's float32 render on the four launch-scale fixtures, per precision mode, beside the
reference's own per-ray sensitivity (float64-vs-float32 and one-ulp nudges).  Writes the table to stdout (committed as profiles/r03_parity/launch_scale_stats.txt)."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from test_gpu_parity import make_renderer, to_np
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
KEYS = ["depth_map", "albedo_map", "roughness_map", "irradiance_map", "radiance_map", "weights", "target_normal_map", "target_normal_map0", "n_dot_v_map",
        "prefiltered_reflected_map", "specular_map", "color_map", "color_map0"]
q = lambda a: "%.1e %.1e %.1e %.1e" % (np.nanmedian(a), np.nanpercentile(a, 99), np.nanpercentile(a, 99.9), np.nanmax(a))
print("per-ray error = max over a map's channels of |x - reference float32| / max|reference|;  columns: median  99%  99.9%  worst;  n>1e-3 = rays above 1e-3")
for name in (sys.argv[1:] or ("fitted_launch16k", "fitted_edit_cfg4", "fitted_insert_cfg5", "fitted_posed4k")):
    g, sdc, sdf, gt, edit = load_golden(name)
    we = int(g["weights_every"])
    print("\n== %s (%d rays)" % (name, len(g["rays_o"])))
    rows = {}
    for mode in ("f16x3_mxfp6x", "f16x3_mxfp6", "f16x3", "f16_mxfp6", "bf16x3"):
        r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384, mlp_precision=mode)
        res = to_np(r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0, gt, **edit))
        for k in KEYS:
            ref = g["out__" + k].astype(np.float64)
            got = res[k][::we] if k.startswith("weights") else res[k]
            e = np.nanmax(np.abs(got.astype(np.float64).reshape(ref.shape) - ref).reshape(len(ref), -1), -1) / max(np.nanmax(np.abs(ref)), 1e-30)
            rows.setdefault(k, []).append("%-13s %s  n>1e-3 %d" % (mode, q(e), int((e > 1e-3).sum())))
        del r
    for k in KEYS:
        f = g["floorray__" + k].astype(np.float64)
        nd = g["nudgeray__" + k].astype(np.float64) if "nudgeray__" + k in g.files else None
        if nd is not None and "branchray__" + k in g.files:
            nd = np.maximum(nd, g["branchray__" + k].astype(np.float64))
        if k.startswith("weights"):
            f = f[::we]; nd = None if nd is None else nd[::we]
        print("%s" % k)
        print("   reference f64-f32 %s  n>1e-3 %d" % (q(f), int((f > 1e-3).sum())) + ("" if nd is None else "   | one-ulp nudge / threshold branch %s" % q(nd)))
        for row in rows[k]:
            print("   " + row)
