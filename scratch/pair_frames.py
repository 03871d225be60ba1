import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as Bn, _pkg
pkg = _pkg.load()
from ibl_nerf_amd import renderer as R, dist as D
torch.cuda.set_device(0)
lut = Bn.load_lut(); K, c2w = Bn.camera()
for kind in ("fitted2", "fitted3"):
    sdc, sdf = Bn.load_checkpoint(kind)
    r = R.Renderer(64, 128, max_rays_per_launch=327680)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    ro, rd = r.get_rays(800, 800, K, c2w); ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    probe = D.frame_probe_for_call(r, 800, 800, K, c2w, Bn.NEAR, Bn.FAR)
    ts = []
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render_rays(ro, rd, Bn.NEAR, Bn.FAR, probe=probe)
        torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    print(kind, ts, "trips", r.trips, "pair_last", r._pair_last, flush=True)
