"""Worst rays of a launch-scale fixture per precision mode: index, per-channel error, the reference's own f64-vs-f32 on that ray.
    python scratch/worst_ray_probe.py fitted_edit_cfg4 depth_map [mode ...]"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from test_gpu_parity import make_renderer, to_np
from ibl_nerf_amd import renderer as R
name, key = sys.argv[1], sys.argv[2]
modes = sys.argv[3:] or ["f16x3_mxfp6x", "f16x3", "bf16x3", "f16_mxfp6"]
lut = load_lut_rgb()
g, sdc, sdf, gt, edit = load_golden(name)
ref = g["out__" + key].astype(np.float64); scale = np.abs(ref).max()
fr = g["floorray__" + key]
out = {}
for mode in modes:
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384, mlp_precision=mode)
    res = to_np(r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0, gt, **edit))
    e = np.abs(res[key].astype(np.float64).reshape(ref.shape) - ref).reshape(len(ref), -1).max(-1) / scale
    out[mode] = e
    top = np.argsort(-e)[:6]
    print("%-14s %s worst rays: %s" % (mode, key, "  ".join("#%d e=%.1e (ref f64-f32 %.1e; z_std %.2e, ref %s)" % (i, e[i], fr[i], res["z_std"][i], np.array2string(ref[i], precision=4)) for i in top)), flush=True)
    print("   n(e>1e-3)=%d  n(e>2e-4)=%d   n(floor>1e-3/8)=%d n(floor>2e-4/8)=%d   n(e > max(2e-4, 8 floor_ray))=%d" % (
        (e > 1e-3).sum(), (e > 2e-4).sum(), (fr > 1e-3 / 8).sum(), (fr > 2e-4 / 8).sum(), (e > np.maximum(2e-4, 8 * fr)).sum()))
np.save(os.path.join(ROOT, "gpurun_out", "worst_%s_%s.npy" % (name, key)), np.stack([out[m] for m in modes]))
