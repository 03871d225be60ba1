"""Round 5: the selection's transmittance thresholds (iblnerf_set_select_tmin) — frame time, refined samples and parity of the 65 536-ray launch fixture per setting."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R, binding as B
lut = load_lut_rgb()
name = sys.argv[1] if len(sys.argv) > 1 else "fitted_launch64k"
g, sdc, sdf, gt, edit = load_golden(name)
f_ = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
Kc = np.array([[f_, 0, 400], [0, f_, 400], [0, 0, 1]], dtype=np.float32)
c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
settings = [(1e-8, 1e-10, 1e-12), (1e-6, 1e-8, 1e-10), (1e-5, 1e-7, 1e-9), (1e-4, 1e-6, 1e-8), (1e-3, 1e-5, 1e-7), (1e-4, 1e-8, 1e-10), (1e-8, 1e-6, 1e-8)]
col = json.load(open(os.path.join(ROOT, "tests", "golden", "c_restatement_column.json"))).get(name, {})
for tm in settings:
    r = R.Renderer(64, 128, max_rays_per_launch=65536)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    B.check(r.ctx, r.lib.iblnerf_set_select_tmin(r.ctx, *tm))
    o, d = r.get_rays(800, 800, Kc, c2w)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    if "rays_o" in g.files:
        ro, rd = torch.from_numpy(g["rays_o"]).cuda(), torch.from_numpy(g["rays_d"]).cuda()
    else:
        idx = torch.as_tensor(g["pix"], device=o.device)
        ro, rd = o[idx].contiguous(), d[idx].contiguous()
    res = {k: v.cpu().numpy() for k, v in r.render_rays(ro, rd, 0.5, 8.0).items()}
    sel = r.last_selection()
    rep = {}
    for k in ("depth_map", "target_normal_map", "albedo_map", "roughness_map", "irradiance_map", "n_dot_v_map"):
        ref = g["out__" + k]
        e = np.abs(res[k].reshape(ref.shape) - ref).reshape(len(ref), -1).max(-1) / np.abs(ref).max()
        rep[k[:6]] = "%d %.1e %.1e" % (int((e > 1e-3).sum()), np.percentile(e, 99), np.percentile(e, 99.9))
    # frame time
    r.render_rays(o, d, 0.5, 8.0); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(2):
        r.render_rays(o, d, 0.5, 8.0)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / 2 * 1e3
    print("tmin %s  frame %.0f ms (%.0f k rays/s)  refined %.3f  decision %s | >1e-3, p99, p99.9: %s" % (tm, ms, 640000 / ms, sel[0] / max(sel[1], 1), (r.policy or {}).get("decision"), rep), flush=True)
    del r
    torch.cuda.empty_cache()
