"""Ray #1017 of fitted_edit_cfg4: stage by stage on the GPU against the oracle (which agrees with the reference on this ray)."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg; _pkg.load()
import iblnerf_oracle as O
from conftest import load_golden, load_lut_rgb
from test_gpu_parity import make_renderer, to_np
from ibl_nerf_amd import renderer as R
g, sdc, sdf, gt, edit = load_golden("fitted_edit_cfg4")
lut = load_lut_rgb()
sel = [1017, 1016, 1018, 1019]
st = {}
ro, rd = g["rays_o"][sel], g["rays_d"][sel]
ref = O.render_rays(sdc, sdf, ro, rd, 0.5, 8.0, lut, stages=st)
zf, zs = st["z_fine"], st["z_samples"]
pts_f = (ro[:, None, :] + rd[:, None, :] * zf[..., None]).astype(np.float32)
raw_o = O.network_query(sdf, pts_f, rd)
zc = O.coarse_z(0.5, 8.0, 64, len(sel))
pts_c = (ro[:, None, :] + rd[:, None, :] * zc[..., None]).astype(np.float32)
rawc_o = O.network_query(sdc, pts_c, rd)
for mode in ("f16x3", "f16_mxfp6", "bf16x3"):
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=64, mlp_precision=mode)
    rawc = r.network_query(pts_c, rd, 0).cpu().numpy()
    raw = r.network_query(pts_f, rd, 1).cpu().numpy()
    ec, ef = np.abs(rawc[..., 0] - rawc_o[..., 0]), np.abs(raw[..., 0] - raw_o[..., 0])
    print(mode, "sigma abs err coarse max per ray", ec.max(-1), "fine", ef.max(-1), " at sample", ef.argmax(-1), "sigma there", raw_o[np.arange(4), ef.argmax(-1), 0])
    mid = 0.5 * (zc[:, 1:] + zc[:, :-1])
    zs_g = r.sample_pdf(mid, ref["weights0"][:, 1:-1], 128).cpu().numpy()
    print("   sample_pdf on the oracle's weights0: max |dz|", np.abs(zs_g - zs).max(-1))
    res = to_np(r.render_rays(ro, rd, 0.5, 8.0))
    print("   render: depth err", np.abs(res["depth_map"] - ref["depth_map"]) / 7.6, "weights err", np.abs(res["weights"] - ref["weights"]).max(-1), "weights0 err", np.abs(res["weights0"] - ref["weights0"]).max(-1), "z_std", res["z_std"] - ref["z_std"])
    w, wo = res["weights"][0], ref["weights"][0]
    j = np.argsort(-np.abs(w - wo))[:6]
    print("   ray 1017 weights differ most at", [(int(k), float(w[k]), float(wo[k]), float(zf[0, k])) for k in j])
