"""TRUNK and FULL network_query timing per precision (8.4 M points): per-slot efficiency of the head layers."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import _pkg; _pkg.load()
from ibl_nerf_amd import renderer as R, checkpoint as ck
sd = ck.synthetic_state_dict(0)
N, S = 65536, 128
pts = torch.rand((N, S, 3), device='cuda') * 8 - 4
dirs = torch.rand((N, 3), device='cuda') * 2 - 1
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for prec, slots in (("bf16x3", (60 * 48, 97 * 48)), ("f16_mxfp6", (60 * 24, 98 * 24))):
    r = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision=prec); r.load_weights(0, sd)
    mt, mf = t(lambda: r.network_query(pts, None, 0)), t(lambda: r.network_query(pts, dirs, 0))
    g = N * S / 128 / 256
    print("%-10s TRUNK %.2f ms (%.1f ns/slot, alg %.0f TF)   FULL %.2f ms (%.1f ns/slot, alg %.0f TF)" % (
        prec, mt, mt * 1e6 / (g * slots[0]), N * S * 982528 / mt / 1e9, mf, mf * 1e6 / (g * slots[1]), N * S * 1591552 / mf / 1e9))
