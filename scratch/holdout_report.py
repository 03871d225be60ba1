"""Round 5: every launch-scale fixture on a DEFAULT-constructed renderer — decision, route, tripwire events, rays above 1e-3 per map (next to the C restatement's), and
whether the STRICT rules of tests/test_gpu_launch_scale.py hold.  The fitted3_* fixtures are the hold-out: nothing was tuned on them."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from ibl_nerf_amd import renderer as R
import test_gpu_launch_scale as LS
lut = load_lut_rgb()
col = {}
p = os.path.join(ROOT, "tests", "golden", "c_restatement_column.json")
if os.path.exists(p):
    col = json.load(open(p))
names = sys.argv[1:] or ["fitted_launch16k", "fitted_edit_cfg4", "fitted_insert_cfg5", "fitted_posed4k", "fitted2_launch4k", "fitted2_posed4k", "fitted3_launch4k", "fitted3_posed4k", "fitted_launch64k"]
for name in names:
    g, sdc, sdf, gt, edit = load_golden(name)
    for prec in ("auto",):
        r = R.Renderer(64, 128, max_rays_per_launch=65536 if "64k" in name else 16384, mlp_precision=prec)
        r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
        if "rays_o" in g.files:
            ro, rd = g["rays_o"], g["rays_d"]
        else:
            f_ = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
            o, d = r.get_rays(800, 800, np.array([[f_, 0, 400], [0, f_, 400], [0, 0, 1]], dtype=np.float32), np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32))
            idx = torch.as_tensor(g["pix"], device=o.device)
            ro, rd = o.reshape(-1, 3)[idx].contiguous(), d.reshape(-1, 3)[idx].contiguous()
        res = {k: v.cpu().numpy() for k, v in r.render_rays(ro, rd, float(g["near"]) if "near" in g.files else 0.5, float(g["far"]) if "far" in g.files else 8.0, gt, **edit).items()}
        rep = {}
        try:
            LS.check_against_fixture(res, g, rep, rules=LS.STRICT)
            verdict = "STRICT ok"
        except AssertionError as e:
            verdict = "STRICT FAILS: " + str(e)[:300].replace("\n", " ")
        dec = (r.policy or {}).get("decision")
        print("%s [%s -> %s] trips %d fallbacks %d route %s" % (name, prec, dec, r.trips, r.range_fallbacks, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (r.route or {}).items()}))
        print("   ", verdict)
        for k in ("depth_map", "albedo_map", "roughness_map", "irradiance_map", "target_normal_map", "n_dot_v_map", "weights", "target_normal_map0", "n_dot_v_map0", "weights0") + tuple(LS.REFLECTED) + tuple(x + "0" for x in LS.REFLECTED):
            if "out__" + k not in g.files:
                continue
            we = int(g["weights_every"]) if "weights_every" in g.files else 1
            e = LS.per_ray(res[k][::we] if k.startswith("weights") else res[k], g["out__" + k])
            c = col.get(name, {}).get(k)
            print("    %-20s >1e-3: %3d   p99.9 %.1e  max %.1e   | C restatement: %s" % (k, int((e > 1e-3).sum()), np.nanpercentile(e, 99.9), np.nanmax(e),
                                                                                    ("%d  p99.9 %.1e max %.1e" % (c["above_1e-3"], c["p999"], c["max"])) if c else "-"), flush=True)
        del r
