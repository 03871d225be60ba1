"""The TIERED table (IBLNERF_ROUTE_FINE_TIERS) against FAST and SAFE: probe metrics vs SAFE and frame time, per checkpoint x camera; tau_main swept."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg
_pkg.load()
import torch
import bench as Bn
from ibl_nerf_amd import dist as D, renderer as R, binding as B

def posed_c2w():
    g = np.load(os.path.join(ROOT, "tests", "golden", "fitted_posed4k.npz"))
    return np.asarray(g["c2w"], dtype=np.float32)[:3, :4]

K, c2w0 = Bn.camera()
cases = (("fitted", c2w0, "frontal"), ("fitted", posed_c2w(), "posed"), ("fitted2", c2w0, "frontal"), ("fitted2", posed_c2w(), "posed"))
if "--holdout" in sys.argv:
    cases = (("fitted3", c2w0, "frontal"), ("fitted3", posed_c2w(), "posed"))
for kind, c2w, tag in cases:
    sdc, sdf = Bn.load_checkpoint(kind)
    r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x")
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(Bn.load_lut())
    ro, rd = r.get_rays(800, 800, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    probe = D.frame_probe_for_call(r, 800, 800, K, c2w, 0.5, 8.0)
    r.decide_route(probe["rays_o"], probe["rays_d"], 0.5, 8.0)
    def run(bits, tau_main=0.0, tau_off=0.0, frame=True):
        B.check(r.ctx, r.lib.iblnerf_set_tier_thresholds(r.ctx, tau_off, tau_main))
        r._set_routing(bits)
        out, _, _ = r._render(probe["rays_o"], probe["rays_d"], 0.5, 8.0, None, {})
        ms = 0.0
        if frame:
            r.render_rays(ro[:131072], rd[:131072], 0.5, 8.0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r.render_rays(ro, rd, 0.5, 8.0)
            torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
        return out, ms
    safe, ms_safe = run(B.ROUTE_FINE_MAIN_PRECISE | B.ROUTE_FINE_OFFSETS_PRECISE)
    fast, ms_fast = run(0)
    line = "%s %s: fast %.0f ms, safe %.0f ms;" % (kind, tag, ms_fast, ms_safe)
    for tm in (1e-2, 1e-3, 1e-4):
        out, ms = run(B.ROUTE_FINE_TIERS, tau_main=tm)
        trig = []
        for k, lim in r.CAL_LIMITS.items():
            x, y = out[k].double().reshape(4096, -1), safe[k].double().reshape(4096, -1)
            e = (x - y).abs().nan_to_num(0.0).amax(-1) / y.abs().nan_to_num(0.0).amax().clamp_min(1e-30)
            trig.append("%s %.0e%s" % (k.replace("_map", "")[:6], float(torch.quantile(e.cpu(), 0.999)), "!" if float(torch.quantile(e.cpu(), 0.999)) > lim or float((e > 1e-3).double().mean()) > r.CAL_MAX_SHARE_ABOVE_1E3.get(k, 1.0) else ""))
        line += "  [tiered tau_main %.0e %.0f ms: %s]" % (tm, ms, " ".join(trig))
    print(line, flush=True)
