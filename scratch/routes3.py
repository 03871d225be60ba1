"""Round 5: the measured route of each fitted checkpoint on the bench frame's probe, the frame rate under it, tripwire events."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _pkg; _pkg.load()
from conftest import load_lut_rgb
from ibl_nerf_amd import renderer as R, checkpoint as ck, dist as D
lut = load_lut_rgb()
fl = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
for which in ("fitted", "fitted2", "fitted3"):
    f = np.load(os.path.join(ROOT, "tests", "golden", which + "_ckpt.npz"))
    r = R.Renderer(64, 128)
    r.load_weights(0, ck.blob_to_state_dict(f["coarse"])); r.load_weights(1, ck.blob_to_state_dict(f["fine"])); r.load_lut(lut)
    pol = D.calibrate_on_frame(r, 800, 800, K, c2w, 0.5, 8.0)
    ro, rd = r.get_rays(800, 800, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    r.render_rays(ro, rd, 0.5, 8.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        r.render_rays(ro, rd, 0.5, 8.0)
    torch.cuda.synchronize()
    print(which, pol["decision"], "%.0f rays/s" % (2 * 640000 / (time.perf_counter() - t0)), "trips", r.trips, r.get_route(), flush=True)
