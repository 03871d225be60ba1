"""How small are the live rows of dL/d raw in a 4 096-ray training step?  Share of the nonzero rows below a relative threshold (candidates for dropping)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as Bn, _pkg
pkg = _pkg.load()
from ibl_nerf_amd import renderer as R, training as T
import train_loss as TL
torch.cuda.set_device(0)
sdc, sdf = Bn.load_checkpoint("fitted")
lut = torch.from_numpy(Bn.load_lut()).cuda()
K, _ = Bn.camera(); fl = float(K[0, 0]); H = W = 800
n = 4096
nets = Bn._trainable_module(sdc).cuda(), Bn._trainable_module(sdf).cuda()
kw = dict(network_fn=nets[0], network_fine=nets[1], N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=0.0, brdf_lut=lut, lut_coefficient="F",
          gamma_correct=True, correct_depth_for_prefiltered_radiance_infer=True, epsilon=0.01, use_radiance_linear=False, lindisp=False, near=Bn.NEAR, far=Bn.FAR,
          target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient_epsilon", max_rays_per_launch=4096)
rng = np.random.RandomState(0)
pix = rng.permutation(H * W)[:n]
i, j = (pix % W).astype(np.float32), (pix // W).astype(np.float32)
d = np.stack([(i - W / 2) / fl, -(j - H / 2) / fl, -np.ones_like(i)], -1).astype(np.float32)
rays = torch.from_numpy(np.stack([np.zeros_like(d), d], 0)).cuda()
tg = {k: torch.from_numpy(v).cuda() for k, v in TL.targets(rng, n).items()}
orig = T.network_backward_live
def spy(r, pts, rd, draw, which):
    rows = draw.reshape(-1, draw.shape[-1])
    mag = rows.abs().amax(-1)
    mx = float(mag.max())
    nz = int((mag > 0).sum())
    print("pass %d: %d points, nonzero rows %d (%.3f); of those below rel thr:" % (which, rows.shape[0], nz, nz / rows.shape[0]),
          {t: round(float(((mag > 0) & (mag < t * mx)).sum()) / max(nz, 1), 3) for t in (1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3)})
    return orig(r, pts, rd, draw, which)
T.network_backward_live = spy
res = R.render_decomp(H, W, K, chunk=n, rays=rays, gt_values={}, approximate_radiance=True, **kw)
TL.total_loss(torch, res, tg, True).backward()
