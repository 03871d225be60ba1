"""Round 5: k_trunk_fp32 on 350 000 points, timed (HIP events through torch), and checked against float64."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_lut_rgb
from ibl_nerf_amd import renderer as R, checkpoint as ck
f = np.load(os.path.join(ROOT, "tests", "golden", "fitted_ckpt.npz"))
sdc = ck.blob_to_state_dict(f["coarse"])
r = R.Renderer(64, 128, max_rays_per_launch=4096)
r.load_weights(0, sdc); r.load_lut(load_lut_rgb())
pts = (torch.rand((350000, 3), generator=torch.Generator().manual_seed(1)) * 4 - 2).cuda()
out = r.trunk_density_fp32(pts, 0)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10):
    out = r.trunk_density_fp32(pts, 0)
ev[1].record(); torch.cuda.synchronize()
print("k_trunk_fp32: %.3f ms per 350 000 points" % (ev[0].elapsed_time(ev[1]) / 10))
import iblnerf_oracle as O
ref = O.network_query({k: v.astype(np.float64) for k, v in sdc.items()}, pts[:4096].cpu().numpy()[:, None, :].astype(np.float32), None)[:, 0, 0]
print("max |fp32 kernel - oracle| on 4096 points: %.2e" % np.abs(out[:4096].cpu().numpy() - ref).max())
