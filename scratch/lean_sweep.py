"""Round 5 experiment (needs the iblnerf_set_offsets_lean hook of commit 5f6f323+1, since removed): with placement and own-selection errors gone, does the offset copies' predicted range tolerate cheaper forms?  level 0 = default, 1 = fine copies'
predicted range on the plain f16 + 2 fp6 trunk (6 slots instead of 7.5), 2 = also the coarse copies' on the mixed trunk form (7.5 instead of 12)."""
import os, sys, time, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R, binding as B
lut = load_lut_rgb()
for name in ("fitted_launch64k", "fitted_posed4k", "fitted2_launch4k"):
    g, sdc, sdf, gt, edit = load_golden(name)
    for level in (0, 1, 2):
        r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x")
        r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
        r.lib.iblnerf_set_offsets_lean.argtypes = [C.c_void_p, C.c_int]
        r.lib.iblnerf_set_offsets_lean(r.ctx, level)
        if "rays_o" in g.files:
            ro, rd = torch.from_numpy(g["rays_o"]).cuda(), torch.from_numpy(g["rays_d"]).cuda()
        else:
            f_ = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
            o, d = r.get_rays(800, 800, np.array([[f_, 0, 400], [0, f_, 400], [0, 0, 1]], dtype=np.float32), np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32))
            idx = torch.as_tensor(g["pix"], device=o.device)
            ro, rd = o.reshape(-1, 3)[idx].contiguous(), d.reshape(-1, 3)[idx].contiguous()
        m = {k: v.cpu().numpy() for k, v in r.render_rays(ro, rd, 0.5, 8.0).items()}
        n = ro.shape[0]
        slots = r.last_slot_units() / n
        out = []
        for k in ("target_normal_map", "target_normal_map0", "depth_map"):
            ref = g["out__" + k].astype(np.float64).reshape(n, -1)
            e = np.abs(m[k].astype(np.float64).reshape(ref.shape) - ref).max(-1) / np.abs(ref).max()
            out.append("%s >1e-3: %d p99.9 %.1e max %.1e" % (k[7:], (e > 1e-3).sum(), np.percentile(e, 99.9), e.max()))
        rate = ""
        if name == "fitted_launch64k":
            fo, fd = o.reshape(-1, 3), d.reshape(-1, 3)
            r.render_rays(fo, fd, 0.5, 8.0); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                r.render_rays(fo, fd, 0.5, 8.0)
            torch.cuda.synchronize()
            rate = "| frame %.0f rays/s" % (2 * 640000 / (time.perf_counter() - t0))
        print("%s level %d: %s | slot units/ray %.3g %s" % (name, level, "; ".join(out), slots, rate), flush=True)
        del r
