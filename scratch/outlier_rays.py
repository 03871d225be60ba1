"""Round 5: WHY do a few rays of a launch sit above 1e-3 on the normal where the fp32 C restatement does not?

For the rays of a launch-scale fixture whose HIP normal is off by more than 1e-3 against the reference: the pipeline stage by stage against the numpy / C oracle
(fp32): coarse weights, fine z, fine main weights, and the four offset depths recomputed from the HIP kernels' own densities under each product scheme
(three f16 products, mixed trunk, 15-slot) at the HIP path's own fine samples — composited on the host with the oracle's fp32 arithmetic.  Which stage moves the normal?
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg
_pkg.load()
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R, binding as B
import iblnerf_oracle as O
import iblnerf_cpu as OC

name = sys.argv[1] if len(sys.argv) > 1 else "fitted_launch16k"
g, sdc, sdf, gt, edit = load_golden(name)
lut = load_lut_rgb()
F32 = np.float32


def mk(**kw):
    r = R.Renderer(64, 128, max_rays_per_launch=16384, **kw)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    return r


r = mk()
ro, rd = g["rays_o"], g["rays_d"]
res = {k: v.cpu().numpy() for k, v in r.render_rays(ro, rd, 0.5, 8.0).items()}
print("policy", r.policy["decision"], "route", r.route, flush=True)
ref_n = g["out__target_normal_map"]
err = np.abs(res["target_normal_map"] - ref_n).max(-1)
err_d = np.abs(res["depth_map"] - g["out__depth_map"]) / np.abs(g["out__depth_map"]).max()
bad = np.flatnonzero(err > 1e-3)
print("rays above 1e-3 on the normal:", bad, err[bad], "depth err there:", err_d[bad], flush=True)
worst = np.argsort(-err)[:12]
sel = np.unique(np.concatenate([bad, worst]))
o, d = ro[sel], rd[sel]
# the oracle, stage by stage
st = {}
ora = O.render_rays(sdc, sdf, o, d, 0.5, 8.0, lut, 64, 128, stages=st)
print("oracle normal vs reference on these rays:", np.abs(ora["target_normal_map"] - ref_n[sel]).max(-1), flush=True)
zf_o = st["z_fine"]
# HIP stages through the taps of a tapped call (main queries whole-batch; same kernels per query class)
n = len(sel)
taps = B.Taps()
buf = dict(zc=torch.empty((n, 64), device="cuda"), zf=torch.empty((n, 192), device="cuda"), rc=torch.empty((n, 64, 18), device="cuda"), rf=torch.empty((n, 192, 18), device="cuda"))
taps.d_z_coarse, taps.d_z_fine, taps.d_raw_coarse, taps.d_raw_fine = (buf[k].data_ptr() for k in ("zc", "zf", "rc", "rf"))
tap = {k: v.cpu().numpy() for k, v in r.render_rays(o, d, 0.5, 8.0, taps=taps).items()}
torch.cuda.synchronize()
zf_h = buf["zf"].cpu().numpy()
print("tapped call's normal vs the frame call's:", np.abs(tap["target_normal_map"] - res["target_normal_map"][sel]).max(-1))
print("fine z: HIP vs oracle, max |dz| per ray:", np.abs(zf_h - zf_o).max(-1))
print("coarse weights: HIP vs oracle max:", np.abs(tap["weights0"] - ora["weights0"]).max(-1))
print("fine weights: HIP vs oracle max:", np.abs(tap["weights"] - ora["weights"]).max(-1))


def normal_from(sig4, z, o, d):
    eps = F32(0.01)
    up0 = np.broadcast_to(np.array([0, 1, 0], dtype=F32), d.shape)
    right = O.cross(d, up0)
    up = O.cross(right, d)
    dists = O.ray_dists(z, d)
    N = o.shape[0]
    D = [np.sum(O.alpha_weights(sig4[s], dists) * z, -1, dtype=F32) for s in range(4)]
    dx = (F32(2) * eps * right + (D[0] - D[1])[:, None] * d).astype(F32)
    dy = (F32(2) * eps * up + (D[2] - D[3])[:, None] * d).astype(F32)
    return O.normalize(O.cross(dx, dy)), np.stack(D, 0)


def offset_points(z, o, d):
    eps = F32(0.01)
    up0 = np.broadcast_to(np.array([0, 1, 0], dtype=F32), d.shape)
    right = O.cross(d, up0)
    up = O.cross(right, d)
    pts = (o[:, None, :] + d[:, None, :] * z[:, :, None]).astype(F32)
    offs = [eps * right, -(eps * right), eps * up, -(eps * up)]
    return np.stack([(pts + f[:, None, :]).astype(F32) for f in offs], 0)       # [4, n, S, 3]


for zname, z in (("HIP's fine z", zf_h), ("oracle's fine z", zf_o)):
    P = offset_points(z, o, d)
    sig_c = np.stack([OC.network_query(sdf, P[v], None)[..., 0] for v in range(4)], 0)
    nrm, D_c = normal_from(sig_c, z, o, d)
    print("--- offsets at %s ---" % zname)
    print("  C fp32 densities  : normal err vs reference", np.abs(nrm - ref_n[sel]).max(-1))
    for label, kw in (("f16x3 (3 products)", dict(mlp_precision="f16x3")), ("mixed trunk", dict(mlp_precision="f16x3_mxfp6x", query_routing="user_trunk_mixed")),
                      ("15-slot", dict(mlp_precision="f16x3_mxfp6x", query_routing="user_trunk_p")), ("bf16x3", dict(mlp_precision="bf16x3"))):
        rq = mk(**kw)
        sig = np.stack([rq.network_query(torch.from_numpy(P[v]).cuda(), None, which=1)[..., 0].cpu().numpy() for v in range(4)], 0)
        nrm_h, D_h = normal_from(sig, z, o, d)
        print("  %-18s: normal err vs reference" % label, np.abs(nrm_h - ref_n[sel]).max(-1), " max |sigma - C|", np.abs(sig - sig_c).max(), " max |D - D_C|", np.abs(D_h - D_c).max(0))
        del rq
print("rays:", sel)
