"""Round 5: inside a step with use_gradient_for_incident_radiance — each pass's reflected-ray backward (g2) against float64 autograd on the step's own inputs."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _pkg; _pkg.load()
from conftest import GOLDEN, rel_linf, load_lut_rgb
import torch.nn.functional as F
import test_gpu_training as TT
import train_loss as TL
from torch_ref import RefShaped, torch_query
from ibl_nerf_amd import renderer as R, training as T, checkpoint as ck
lut = load_lut_rgb()
GI = np.load(os.path.join(GOLDEN, "train_step_incident.npz"))
f = np.load(os.path.join(GOLDEN, "fitted_ckpt.npz"))
sds = ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])
nets, kw, K, rays = TT._setup(GI, lut, "full")
kw["use_gradient_for_incident_radiance"] = True
stash = {}
o_cdb, o_nb = R.Renderer.composite_direct_backward, R.Renderer.network_backward
def cdb(self, raw, z, rd, dm, dw=None, full=False):
    out = o_cdb(self, raw, z, rd, dm, dw, full)
    if full:
        stash.update(z=z.clone(), rd=rd.clone(), dm=dm.clone())
    return out
def nb(self, pts, vd, draw, which=0, grad_scale=None):
    out = o_nb(self, pts, vd, draw, which, grad_scale)
    if stash:
        with torch.enable_grad():
            net = RefShaped(sds[which]).double().cuda()
            raw = torch_query(pts.double(), vd.double(), net)
            z64 = stash["z"].double()
            dists = torch.cat([z64[:, 1:] - z64[:, :-1], torch.full_like(z64[:, :1], 1e10)], -1) * torch.norm(stash["rd"].double()[:, None, :], dim=-1)
            alpha = 1.0 - torch.exp(-F.relu(raw[..., 0]) * dists)
            w = alpha * torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
            env = torch.sum(w[..., None] * torch.sigmoid(raw[..., 6:18]), -2)
            (env * stash["dm"][:, 7:19].double()).sum().backward()
        rep = {nme: rel_linf(out[1][nme].cpu().numpy(), p.grad.cpu().numpy()) for nme, p in net.named_parameters() if float(p.grad.abs().max()) > 0}
        top = sorted(rep.items(), key=lambda kv: -kv[1])[:4]
        print("reflected backward of network", which, "scale", getattr(self, "last_grad_scale", None), " worst:", ", ".join("%s %.1e" % kv for kv in top))
        stash.clear()
    return out
R.Renderer.composite_direct_backward, R.Renderer.network_backward = cdb, nb
res = R.render_decomp(800, 800, K, chunk=int(GI["chunk"]), rays=rays, gt_values={}, approximate_radiance=True, **kw)
TL.total_loss(torch, res, {k[8:]: GI[k] for k in GI.files if k.startswith("target__")}, True).backward()
