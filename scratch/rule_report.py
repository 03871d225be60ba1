"""Every violation of the launch-scale rules (tests/test_gpu_launch_scale.py) for a fixture, per precision mode / routing — not only the first.
    python scratch/rule_report.py <fixture> [<fixture> ...]"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from test_gpu_parity import make_renderer, to_np, DIRECT
import test_gpu_launch_scale as LS
from ibl_nerf_amd import renderer as R, binding as B
lut = load_lut_rgb()
for name in sys.argv[1:]:
    g, sdc, sdf, gt, edit = load_golden(name)
    we = int(g["weights_every"])
    MODES = (("default", {}), ("default, coarse density on 22-bit operands (round 3)", dict(query_routing=B.ROUTE_COARSE_MAIN_22BIT)),
             ("fine main precise", dict(query_routing=B.ROUTE_FINE_MAIN_PRECISE)), ("f16x3_mxfp6", dict(mlp_precision="f16x3_mxfp6")),
             ("f16x3_mxfp6 + fine main precise", dict(mlp_precision="f16x3_mxfp6", query_routing=B.ROUTE_FINE_MAIN_PRECISE)), ("f16x3", dict(mlp_precision="f16x3")))
    if os.environ.get("RULE_MODES"):
        MODES = tuple(m for m in MODES if m[0] in os.environ["RULE_MODES"].split("|"))
    for label, kw in MODES:
        r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384, **kw)
        res = to_np(r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0, gt, **edit))
        print("\n== %s  [%s]   NaN maps: %s / reference: %s" % (name, label, [k for k in res if not np.isfinite(res[k]).all()], [k[5:] for k in g.files if k.startswith("out__") and not np.isfinite(g[k]).all()]))
        for sfx in ("", "0"):
            for k in DIRECT + ["diffuse_map"] + LS.NORMAL_LIKE:
                key = k + sfx
                got, f = res[key], LS.ray_floor(g, key)
                if k == "weights":
                    got, f = got[::we], f[::we]
                e = LS.per_ray(got, g["out__" + key])
                nl = k in LS.NORMAL_LIKE
                base = 1e-3 if nl else 5e-4
                bad, worse = e > np.maximum(base, 8 * f), e > np.maximum(2e-3 if nl else 1e-3, 16 * f)
                p999, bound = float(np.nanpercentile(e, 99.9)), max(1e-3 if (nl or k == "weights") else 2e-4, 1.5 * float(np.nanpercentile(f, 99.9)))
                flags = []
                if bad.sum() > max(1, len(e) // 2000): flags.append("8x-own rule: %d rays (allowed %d)" % (bad.sum(), max(1, len(e) // 2000)))
                if worse.any(): flags.append("16x-own rule: %d rays, worst %.1e (own %.1e)" % (worse.sum(), np.nanmax(e[worse]), f[worse][np.nanargmax(e[worse])]))
                if (e > 1e-3).sum() > (f > 1e-3 / 8).sum(): flags.append(">1e-3: %d rays, reference flags %d" % ((e > 1e-3).sum(), (f > 1e-3 / 8).sum()))
                if p999 > bound: flags.append("p99.9 %.1e > %.1e" % (p999, bound))
                if flags or key in ("depth_map", "albedo_map", "target_normal_map", "weights", "target_normal_map0", "weights0"):
                    print("   %-22s p99 %.1e p99.9 %.1e max %.1e  >1e-3: %d (ref flags %d; with the 22-bit-parameter column %d) | %s" % (
                        key, np.nanpercentile(e, 99), p999, np.nanmax(e), (e > 1e-3).sum(), (f > 1e-3 / 8).sum(),
                        (LS.ray_floor(g, key, True)[::we if k == "weights" else 1] > 1e-3 / 8).sum(), "; ".join(flags) or "ok"))
        for k in LS.REFLECTED:
            for sfx in ("", "0"):
                e, f = LS.per_ray(res[k + sfx], g["out__" + k + sfx]), LS.ray_floor(g, k + sfx)
                for q in (50, 99, 99.9):
                    b = max(LS.DIST_FACTOR * float(np.nanpercentile(f, q)), LS.DIST_FLOOR[q])
                    if float(np.nanpercentile(e, q)) > b:
                        print("   %-22s p%s %.1e > %.1e" % (k + sfx, q, np.nanpercentile(e, q), b))
        del r
