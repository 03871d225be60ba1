"""Round 5: the reflected-ray part of use_gradient_for_incident_radiance on its own — network_query -> composite_direct_backward(full) -> network_backward on reflected rays
of the fitted checkpoint against torch autograd (float64) through the same arithmetic (raw2outputs_simple: every map on the live weights)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _pkg; _pkg.load()
from conftest import GOLDEN, rel_linf, load_lut_rgb
import torch.nn.functional as F
from torch_ref import RefShaped, torch_query
from ibl_nerf_amd import renderer as R, checkpoint as ck
lut = load_lut_rgb()
G = np.load(os.path.join(GOLDEN, "train_step_incident.npz"))
f = np.load(os.path.join(GOLDEN, "fitted_ckpt.npz"))
sds = ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])
r = R.Renderer(64, 128, max_rays_per_launch=256)
r.load_weights(0, sds[0]); r.load_weights(1, sds[1]); r.load_lut(lut)
ro, rd = torch.from_numpy(G["rays_o"]).cuda(), torch.from_numpy(G["rays_d"]).cuda()
n = ro.shape[0]
gen = torch.Generator().manual_seed(5)
for which, sfx in ((0, "0"), (1, "")):
    nrm = torch.from_numpy(G["full__out__target_normal_map" + sfx]).cuda()
    dep = torch.from_numpy(G["full__out__target_depth_map" + sfx]).cuda()
    xs = (ro + rd * dep[:, None]).contiguous()
    rdir = (rd - 2 * torch.sum(nrm * rd, -1, keepdim=True) * nrm).contiguous()
    zc = (0.5 + 7.5 * torch.linspace(0, 1, 64)).expand(n, 64).contiguous().cuda()
    rpts = (xs[:, None, :] + rdir[:, None, :] * zc[:, :, None]).contiguous()
    denv = torch.randn((n, 12), generator=gen).cuda()
    rraw = r.network_query(rpts, rdir, which)
    dm = torch.zeros((n, 19), device="cuda"); dm[:, 7:19] = denv
    rdraw = r.composite_direct_backward(rraw, zc, rdir, dm, None, full=True)
    _, g2 = r.network_backward(rpts, rdir, rdraw, which)
    net = RefShaped(sds[which]).double().cuda()
    raw = torch_query(rpts.double(), rdir.double(), net)
    z64 = zc.double()
    dists = torch.cat([z64[:, 1:] - z64[:, :-1], torch.full_like(z64[:, :1], 1e10)], -1) * torch.norm(rdir.double()[:, None, :], dim=-1)
    alpha = 1.0 - torch.exp(-F.relu(raw[..., 0]) * dists)
    w = alpha * torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    env = torch.sum(w[..., None] * torch.sigmoid(raw[..., 6:18]), -2)
    (env * denv.double()).sum().backward()
    print("network", which, " raw vs f64 %.1e" % rel_linf(rraw.cpu().numpy(), raw.detach().cpu().numpy()))
    rep = {nme: rel_linf(g2[nme].cpu().numpy(), p.grad.cpu().numpy()) for nme, p in net.named_parameters() if float(p.grad.abs().max()) > 0}
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1])[:6]:
        print("   %-40s %.2e" % (k, v))
