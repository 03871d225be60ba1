"""Launch-scale parity statistics on the GPU: the HIP renderer against the reference's own render of 16 384 / 4 096 / 4 096 seeded pixels
of the fitted checkpoint (fixtures fitted_launch16k, fitted_edit_cfg4, fitted_insert_cfg5), with the reference's float64-vs-float32
per-ray difference beside it.    python scratch/launch_scale_probe.py [mode ...]"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from test_gpu_parity import make_renderer, to_np, DIRECT, DERIVED
from ibl_nerf_amd import renderer as R

modes = sys.argv[1:] or ["f16x3_mxfp6x", "f16x3_mxfp6"]
lut = load_lut_rgb()
for name in ("fitted_launch16k", "fitted_edit_cfg4", "fitted_insert_cfg5"):
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", name + ".npz")):
        continue
    g, sdc, sdf, gt, edit = load_golden(name)
    we = int(g["weights_every"])
    for mode in modes:
        r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384, mlp_precision=mode)
        res = to_np(r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0, gt, **edit))
        print("== %s  %s  (%d rays, fallbacks %d)" % (name, mode, len(g["rays_o"]), r.range_fallbacks))
        for k in [k[5:] for k in g.files if k.startswith("out__")]:
            ref = g["out__" + k].astype(np.float64)
            got = res[k].astype(np.float64)
            fr = g["floorray__" + k].astype(np.float64)
            if k.startswith("weights"):
                got, fr = got[::we], fr[::we]
            scale = max(np.nanmax(np.abs(ref)), 1e-30)
            e = np.nanmax(np.abs(got.reshape(ref.shape) - ref).reshape(len(ref), -1), -1) / scale
            q = lambda a: "max %.1e p99.9 %.1e p99 %.1e med %.1e" % (np.nanmax(a), np.nanpercentile(a, 99.9), np.nanpercentile(a, 99), np.nanmedian(a))
            print("%-36s gpu: %s | ref64-32: %s | ratio max %.1f p99.9 %.1f p99 %.1f" % (
                k, q(e), q(fr), np.nanmax(e) / max(np.nanmax(fr), 1e-30), np.nanpercentile(e, 99.9) / max(np.nanpercentile(fr, 99.9), 1e-30),
                np.nanpercentile(e, 99) / max(np.nanpercentile(fr, 99), 1e-30)), flush=True)
        del r
