"""Round 5: dL/d env of iblnerf_ray_outputs_backward_env against torch autograd through training._ray_outputs, on a step's own inputs (both passes)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _pkg; _pkg.load()
from conftest import GOLDEN, rel_linf, load_lut_rgb
import test_gpu_training as TT
import train_loss as TL
from ibl_nerf_amd import renderer as R, training as T
lut = load_lut_rgb()
GI = np.load(os.path.join(GOLDEN, "train_step_incident.npz"))
nets, kw, K, rays = TT._setup(GI, lut, "full")
kw["use_gradient_for_incident_radiance"] = True
orig = R.Renderer.ray_outputs_backward
def spy(self, maps, upstream, n_dot_v=None, env=None, depth0=1.0, gt=None, want_denv=False):
    out = orig(self, maps, upstream, n_dot_v, env, depth0, gt, want_denv)
    if want_denv:
        dx, denv = out
        with torch.enable_grad():
            x = maps.detach().double().requires_grad_(True)
            e = env.detach().double().reshape(-1, 4, 3).requires_grad_(True)
            consts = dict(n_dot_v=n_dot_v.double(), env=e, lut=torch.from_numpy(lut).cuda().double(), depth0=depth0)
            outs = T._ray_outputs(x, consts, T._flags(self), None)
            pairs = [(outs[k], g) for k, g in upstream.items() if g is not None and k in outs]
            gx, ge = torch.autograd.grad([o for o, _ in pairs], [x, e], [g.reshape(o.shape).double() for o, g in pairs])
        print("dx %.2e   denv %.2e   |denv| %.2e" % (rel_linf(dx.cpu().numpy(), gx.cpu().numpy()), rel_linf(denv.cpu().numpy(), ge.cpu().numpy()), float(ge.abs().max())))
    return out
R.Renderer.ray_outputs_backward = spy
res = R.render_decomp(800, 800, K, chunk=int(GI["chunk"]), rays=rays, gt_values={}, approximate_radiance=True, **kw)
TL.total_loss(torch, res, {k[8:]: GI[k] for k in GI.files if k.startswith("target__")}, True).backward()
