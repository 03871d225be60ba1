"""Which half of the TIERED table does a checkpoint x camera need?  On the frame's probe (4 096 strided rays): FAST, main-query tiers only, offset tiers only and
TIERED against SAFE under the calibration limits; then the frame time under each of them (imposed)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as Bn, _pkg
pkg = _pkg.load()
from ibl_nerf_amd import renderer as R, binding as B
torch.cuda.set_device(0)
lut = Bn.load_lut()
K, c2w = Bn.camera()
BIG = 1e9
for kind in sys.argv[1:] or ["fitted", "fitted2", "fitted3"]:
    sdc, sdf = Bn.load_checkpoint(kind)
    r = R.Renderer(64, 128, max_rays_per_launch=327680)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    ro, rd = r.get_rays(800, 800, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    n = ro.shape[0]
    idx = torch.linspace(0, n - 1, 4096, device=ro.device).long()
    pro, prd = ro[idx].contiguous(), rd[idx].contiguous()
    r.decide_route(pro, prd, Bn.NEAR, Bn.FAR)
    r.policy = {"decision": "experiment", "imposed": True}
    def probe(routing, taus):
        B.check(r.ctx, r.lib.iblnerf_set_tier_thresholds(r.ctx, *taus))
        r._set_routing(routing)
        return r._render(pro, prd, Bn.NEAR, Bn.FAR, None, {}, on_range="raise")[0]
    safe = probe(r.SAFE_ROUTING, (0., 0.))
    cfgs = {"fast": (0, (0., 0.)), "main_only": (r.TIERED_ROUTING, (BIG, 0.)), "offsets_only": (r.TIERED_ROUTING, (0., BIG)), "tiered": (r.TIERED_ROUTING, (0., 0.)),
            "safe": (r.SAFE_ROUTING, (0., 0.))}
    for name, (routing, taus) in cfgs.items():
        x = probe(routing, taus)
        trig = []
        for k, lim in r.CAL_LIMITS.items():
            a, b = x[k].double().reshape(4096, -1), safe[k].double().reshape(4096, -1)
            e = (a - b).abs().nan_to_num(0.0).amax(-1) / b.abs().nan_to_num(0.0).amax().clamp_min(1e-30)
            p999, share = float(torch.quantile(e.cpu(), 0.999)), float((e > 1e-3).double().mean())
            if p999 > lim: trig.append("%s p99.9 %.1e > %.1e" % (k, p999, lim))
            if share > r.CAL_MAX_SHARE_ABOVE_1E3.get(k, 1.0): trig.append("%s %.2f%% > 1e-3" % (k, 100 * share))
        B.check(r.ctx, r.lib.iblnerf_set_tier_thresholds(r.ctx, *taus)); r._set_routing(routing)
        r._render(ro, rd, Bn.NEAR, Bn.FAR, None, {}, on_range="raise"); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2): r._render(ro, rd, Bn.NEAR, Bn.FAR, None, {}, on_range="raise")
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 2 * 1e3
        print("%-8s %-13s frame %.1f ms  %s" % (kind, name, ms, "HOLDS" if not trig else trig), flush=True)
