"""Steady-state cost of the per-checkpoint measurements (route probe, FAST/SAFE calibration) against one frame: what a per-CALL decision would add."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
pkg = _pkg.load()
import torch
import bench as Bn
from ibl_nerf_amd import dist as D, renderer as R

for kind in ("fitted", "fitted2", "fitted3"):
    sdc, sdf = Bn.load_checkpoint(kind)
    K, c2w = Bn.camera()
    r = R.Renderer(64, 128)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(Bn.load_lut())
    ro, rd = r.get_rays(800, 800, K, c2w)
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    pro, prd = D.frame_probe(r, 800, 800, K, c2w)
    def t(f, n=3):
        torch.cuda.synchronize(); best = 1e9
        for _ in range(n):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, 1e3 * (time.perf_counter() - t0))
        return best
    D.calibrate_on_frame(r, 800, 800, K, c2w, 0.5, 8.0)
    r.render_rays(ro, rd, 0.5, 8.0)
    def route():
        r.route = None; r.set_route(None); r.decide_route(pro, prd, 0.5, 8.0)
    def cal():
        r.calibrate(pro, prd, 0.5, 8.0)
    print(kind, "decision", r.policy["decision"], "route ms %.1f" % t(route), "calibrate (2 renders of 4096) ms %.1f" % t(cal),
          "render 4096 ms %.1f" % t(lambda: r.render_rays(pro, prd, 0.5, 8.0)), "frame ms %.1f" % t(lambda: r.render_rays(ro, rd, 0.5, 8.0), 2), "trips", r.trips, flush=True)
