"""TRUNK network_query: f16_mxfp6 vs bf16x3, time and agreement (dev loop for mlp_kernel_mx.hip)."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import _pkg; _pkg.load()
from ibl_nerf_amd import renderer as R, checkpoint as ck
sd = ck.synthetic_state_dict(0)
N, S = 65536, 128
pts = torch.rand((N, S, 3), device='cuda') * 8 - 4
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = {}
for prec in ("bf16x3", "f16_mxfp6"):
    r = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision=prec); r.load_weights(0, sd)
    ms = t(lambda: r.network_query(pts, None, 0))
    out[prec] = r.network_query(pts[:4096], None, 0)
    print("%-10s TRUNK %.2f ms  %.1f Mpts/s  alg %.0f TFLOP/s (frac %.3f)" % (prec, ms, N*S/ms/1e3, N*S*982528/ms/1e9, N*S*982528/ms/1e9/2500), flush=True)
print("max |mx - bf16x3| = %.2e" % float((out["bf16x3"] - out["f16_mxfp6"]).abs().max()))
