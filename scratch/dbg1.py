import sys, os, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0,'tests')
import _pkg; pkg = _pkg.load()
import iblnerf_oracle as O
from ibl_nerf_amd import renderer as R, checkpoint as ck
sd = ck.synthetic_state_dict(0)
r = R.Renderer(64, 0, max_rays_per_launch=64)
r.load_weights(0, sd)
rng = np.random.RandomState(0)
pts = rng.uniform(-2, 2, (4, 32, 3)).astype(np.float32)
dirs = rng.uniform(-1, 1, (4, 3)).astype(np.float32)
ref_s = O.network_query(sd, pts, None)
got_s = r.network_query(pts, None, 0).cpu().numpy()
print('trunk: ref', ref_s.reshape(-1)[:6], 'got', got_s.reshape(-1)[:6])
print('trunk max err', np.abs(ref_s-got_s).max())
ref = O.network_query(sd, pts, dirs)
got = r.network_query(pts, dirs, 0).cpu().numpy()
print('full ref', ref[0,0], '\n got', got[0,0])
print('full max err per channel', np.abs(ref-got).reshape(-1,18).max(0))
