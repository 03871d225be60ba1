"""Does the stream pair engage, and what does it buy, per checkpoint / table?"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as Bn, _pkg
pkg = _pkg.load()
from ibl_nerf_amd import renderer as R, dist as D
torch.cuda.set_device(0)
lut = Bn.load_lut(); K, c2w = Bn.camera()
for kind, mode in (("fitted2", "auto"), ("fitted", "f16x3_mxfp6"), ("fitted", "auto")):
    sdc, sdf = Bn.load_checkpoint(kind)
    r = R.Renderer(64, 128, max_rays_per_launch=327680, mlp_precision=mode)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    ro, rd = r.get_rays(800, 800, K, c2w); ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    probe = D.frame_probe_for_call(r, 800, 800, K, c2w, Bn.NEAR, Bn.FAR)
    for pair in (False, True, False, True):
        r.pair_streams = pair
        for _ in range(2): r.render_rays(ro, rd, Bn.NEAR, Bn.FAR, probe=probe)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): r.render_rays(ro, rd, Bn.NEAR, Bn.FAR, probe=probe)
        torch.cuda.synchronize()
        print(kind, mode, "pair" if pair else "one ", "%.1f ms" % ((time.perf_counter() - t0) / 3 * 1e3), "pair_last", r._pair_last, "decision", (r.policy or {}).get("decision"),
              "act_scale", {k: len(v) for k, v in r._act_scale.items()}, "rescales", r.range_rescales, "fallbacks", r.range_fallbacks, "trips", r.trips, flush=True)
    del r
