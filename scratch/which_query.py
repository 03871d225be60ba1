"""Round 5: which query of the fast table leaves the normal outliers of the 65 536-ray launch?  fast / fine offsets precise / fine main precise / both (= safe)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R, binding as B
g, sdc, sdf, gt, edit = load_golden("fitted_launch64k")
lut = load_lut_rgb()
bad = {}
for label, routing in (("fast", ()), ("offsets precise", ("fine_offsets_precise",)), ("main precise", ("fine_main_precise",)), ("both", ("fine_offsets_precise", "fine_main_precise")),
                       ("fast, no tiers", ("no_offset_tiers",)), ("fast, estimate_all", ("offsets_estimate_all",))):
    r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x", query_routing=routing)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    f_ = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
    o, d = r.get_rays(800, 800, np.array([[f_, 0, 400], [0, f_, 400], [0, 0, 1]], dtype=np.float32), np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32))
    idx = torch.as_tensor(g["pix"], device=o.device)
    m = {k: v.cpu().numpy() for k, v in r.render_rays(o.reshape(-1, 3)[idx].contiguous(), d.reshape(-1, 3)[idx].contiguous(), 0.5, 8.0).items()}
    ref = g["out__target_normal_map"].astype(np.float64)
    e = np.abs(m["target_normal_map"].astype(np.float64) - ref).max(-1)
    bad[label] = set(np.flatnonzero(e > 1e-3).tolist())
    print("%-22s normal >1e-3: %d  rays %s  errors %s" % (label, len(bad[label]), sorted(bad[label]), np.round(e[sorted(bad[label])], 4)), flush=True)
    del r
