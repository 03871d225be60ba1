"""Times the per-view test.py flow at 800x800: render only vs render + export (D2H, mapping, to8b, PNG)."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg; _pkg.load()
from ibl_nerf_amd import checkpoint as ck, export as E, model as M
import bench

class DS:
    far = 8.0
    def __init__(self, n):
        c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
        self.poses = torch.from_numpy(np.stack([c2w] * n)).cuda()
    def get_resized_normal_albedo(self, f, i):
        return {}

tmp = tempfile.mkdtemp(); os.makedirs(tmp + "/exp")
_, kw, *_ = M.create_IBLNeRF(M.default_args(basedir=tmp))
kw["network_fn"].load_state_dict(ck.synthetic_state_dict(0)); kw["network_fine"].load_state_dict(ck.synthetic_state_dict(1))
kw.update(near=0.5, far=8.0, brdf_lut=torch.from_numpy(bench.load_lut()))
K, _ = bench.camera()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ds = DS(n)
E.render_decomp_path(DS(1), (800, 800, float(K[0, 0])), None, 1024, kw, savedir=None, render_factor=1, approximate_radiance=True)
torch.cuda.synchronize()
for label, sd in (("render + host mapping, no files", None), ("render + export PNGs", tmp + "/png")):
    t0 = time.perf_counter()
    E.render_decomp_path(ds, (800, 800, float(K[0, 0])), None, 1024, kw, savedir=sd, render_factor=1, approximate_radiance=True)
    torch.cuda.synchronize()
    print("%-36s %.3f s/view" % (label, (time.perf_counter() - t0) / n), flush=True)
