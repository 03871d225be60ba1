"""Round 5: the exact-fp32 trunk (csrc/trunk_fp32_kernel.hip) against the C restatement's fp32 network on the coarse grid's points, and what it does to the coarse weights
(against the reference's own, fixture fitted_launch16k) and to the frame time."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R
import iblnerf_oracle as O, iblnerf_cpu as OC
g, sdc, sdf, gt, edit = load_golden("fitted_launch16k")
lut = load_lut_rgb()
we = int(g["weights_every"])
res = {}
for label, routing in (("fp32", ()), ("15slot", ("coarse_density_15slot",))):
    r = R.Renderer(64, 128, max_rays_per_launch=16384, mlp_precision="f16x3_mxfp6x", query_routing=routing)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    m = r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0)
    torch.cuda.synchronize()
    print(label, r.describe_route().splitlines()[1])
    res[label] = {k: v.cpu().numpy() for k, v in m.items()}
    e = np.abs(res[label]["weights0"][::we] - g["out__weights0"]).max(-1)
    print(label, "coarse weights vs reference: p50 %.2e p99 %.2e p99.9 %.2e max %.2e" % tuple(np.percentile(e, [50, 99, 99.9, 100])))
    for k in ("target_normal_map", "depth_map", "albedo_map", "target_normal_map0", "depth_map0"):
        ref = g["out__" + k].astype(np.float64).reshape(len(e) * we, -1)
        er = np.abs(res[label][k].astype(np.float64).reshape(ref.shape) - ref).max(-1) / np.abs(ref).max()
        print("   %-20s rays above 1e-3: %d  p99.9 %.1e  max %.1e" % (k, (er > 1e-3).sum(), np.percentile(er, 99.9), er.max()))
    ro, rd = r.get_rays(800, 800, np.array([[692.8203, 0, 400], [0, 692.8203, 400], [0, 0, 1]], dtype=np.float32), np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32))
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    r2 = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x", query_routing=routing)
    r2.load_weights(0, sdc); r2.load_weights(1, sdf); r2.load_lut(lut)
    r2.render_rays(ro, rd, 0.5, 8.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        r2.render_rays(ro, rd, 0.5, 8.0)
    torch.cuda.synchronize()
    print(label, "frame: %.0f rays/s" % (2 * 640000 / (time.perf_counter() - t0)), "slot units per ray %.3g" % (r2.last_slot_units() / 640000), flush=True)
    del r, r2
