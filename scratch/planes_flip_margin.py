"""Round 5 (build container only: imports the reference): in the reference's own warm-up step of train_step_planes.npz, how close do the stochastic fine samples of each ray
sit to a bin boundary of the inverse CDF?  (The HIP path's z_std differs on ray 22 by 4e-3 = one sample in another bin, and f.positions_linears.1's gradient by 4e-3.)"""
import os, sys, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), ROOT]
import make_golden as MG
torch, R, M, Hh = MG.import_reference()
lut = MG.load_lut(torch)
G = np.load(os.path.join(ROOT, "tests", "golden", "train_step_planes.npz"))
tmp = tempfile.mkdtemp()
try:
    kw, _, *_ = M.create_IBLNeRF(MG.reference_args(tmp, 128))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
sd_c, sd_f = MG.fitted_state_dicts()
kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
kw["network_fine"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
kw.update(near=torch.from_numpy(G["near"]), far=torch.from_numpy(G["far"]), pytest=True, brdf_lut=lut)
cap = {}
pdf0 = R.sample_pdf
def spy(bins, weights, N, det=False, pytest=False):
    out = pdf0(bins, weights, N, det=det, pytest=pytest)
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    np.random.seed(0)
    u = torch.Tensor(np.random.rand(*(list(cdf.shape[:-1]) + [N])))
    cap.update(cdf=cdf.detach().numpy().astype(np.float64), u=u.numpy().astype(np.float64))
    return out
R.sample_pdf = spy
K = np.array([[692.8203, 0, 400], [0, 692.8203, 400], [0, 0, 1]], dtype=np.float32)
rays = torch.from_numpy(np.stack([G["rays_o"], G["rays_d"]], 0))
with torch.no_grad():
    res = R.render_decomp(800, 800, K, chunk=64, rays=rays, gt_values={}, approximate_radiance=False, **kw, **MG.EDIT_KEYS_OFF)
R.sample_pdf = pdf0
print("z_std matches the fixture:", float(np.abs(res["z_std"].numpy() - G["warmup__out__z_std"]).max()))
d = np.abs(cap["u"][:, :, None] - cap["cdf"][:, None, :]).min(-1)          # [rays, N]: distance of each u to the nearest cdf entry
per_ray = d.min(-1)
order = np.argsort(per_ray)
print("rays by smallest |u - cdf| (in units of 2^-24 = half an ulp of 1):")
for r in order[:6]:
    print("  ray %2d  %.3g  = %.1f x 2^-24" % (r, per_ray[r], per_ray[r] * 2 ** 24))
