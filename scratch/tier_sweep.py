"""Round 5: the offset tiers' threshold — the fast table on the 65 536-ray launch fixture: rays above 1e-3 on the normal, share of flagged samples, frame rate."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R, binding as B
g, sdc, sdf, gt, edit = load_golden("fitted_launch64k")
lut = load_lut_rgb()
for tau in (0.0,):
    r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x")
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    B.check(r.ctx, r.lib.iblnerf_set_offset_tier_threshold(r.ctx, float(tau)))
    f_ = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
    o, d = r.get_rays(800, 800, np.array([[f_, 0, 400], [0, f_, 400], [0, 0, 1]], dtype=np.float32), np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32))
    idx = torch.as_tensor(g["pix"], device=o.device)
    ro, rd = o.reshape(-1, 3)[idx].contiguous(), d.reshape(-1, 3)[idx].contiguous()
    m = {k: v.cpu().numpy() for k, v in r.render_rays(ro, rd, 0.5, 8.0).items()}
    slots = r.last_slot_units() / 65536
    out = []
    for k in ("target_normal_map", "n_dot_v_map", "depth_map"):
        ref = g["out__" + k].astype(np.float64).reshape(65536, -1)
        e = np.abs(m[k].astype(np.float64).reshape(ref.shape) - ref).max(-1) / np.abs(ref).max()
        out.append("%s >1e-3: %d p99.9 %.1e max %.1e" % (k[:8], (e > 1e-3).sum(), np.percentile(e, 99.9), e.max()))
    fo, fd = o.reshape(-1, 3), d.reshape(-1, 3)
    r.render_rays(fo, fd, 0.5, 8.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        r.render_rays(fo, fd, 0.5, 8.0)
    torch.cuda.synchronize()
    print("tau %.0e: %s | slot units/ray %.3g | frame %.0f rays/s" % (tau, "; ".join(out), slots, 2 * 640000 / (time.perf_counter() - t0)), flush=True)
    del r
