"""A whole training step on the fused path (render_decomp with trainable modules -> losses of train.py -> backward -> Adam), timed.
    python scratch/train_step_fused.py [n_rays ...]"""
import sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg; _pkg.load()
from conftest import load_lut_rgb
from torch_ref import RefShaped
import train_loss as TL
from ibl_nerf_amd import checkpoint as ck, renderer as R
f = np.load(os.path.join(ROOT, "tests", "golden", "fitted_ckpt.npz"))
sdc, sdf = ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])
lut = torch.from_numpy(load_lut_rgb()).cuda()
fl = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
for n in [int(a) for a in sys.argv[1:]] or [512, 1024, 4096]:
    nets = RefShaped(sdc).cuda(), RefShaped(sdf).cuda()
    for net in nets:
        net.coarse_radiance_number = 3
    opt = torch.optim.Adam([p for net in nets for p in net.parameters()], lr=5e-4)
    kw = dict(network_fn=nets[0], network_fine=nets[1], N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=0.0, brdf_lut=lut, lut_coefficient="F",
              gamma_correct=True, correct_depth_for_prefiltered_radiance_infer=True, epsilon=0.01, use_radiance_linear=False, lindisp=False, near=0.5, far=8.0,
              target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient_epsilon", max_rays_per_launch=max(n, 1024))
    rng = np.random.RandomState(0)
    pix = rng.permutation(640000)[:n]
    i, j = (pix % 800).astype(np.float32), (pix // 800).astype(np.float32)
    d = np.stack([(i - 400) / fl, -(j - 400) / fl, -np.ones_like(i)], -1).astype(np.float32)
    rays = torch.from_numpy(np.stack([np.zeros_like(d), d], 0)).cuda()
    tg = {k: torch.from_numpy(v).cuda() for k, v in TL.targets(rng, n).items()}
    for approx in (False, True):
        ts = []
        for it in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = R.render_decomp(800, 800, K, chunk=n, rays=rays, gt_values={}, approximate_radiance=approx, **kw)
            t1 = time.perf_counter()
            loss = TL.total_loss(torch, res, tg, approx)
            opt.zero_grad(); loss.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
            opt.step(); torch.cuda.synchronize(); t3 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1, t3 - t2, float(loss.detach())))
        a = np.array(ts[3:])
        print("%5d rays  approximate_radiance=%-5s  render %.2f ms  loss+backward %.2f ms  Adam %.2f ms  = %.2f ms/step   loss %.4f -> %.4f" % (
            n, approx, 1e3 * a[:, 0].mean(), 1e3 * a[:, 1].mean(), 1e3 * a[:, 2].mean(), 1e3 * a[:, :3].sum(1).mean(), ts[0][3], ts[-1][3]), flush=True)
