"""End-to-end normal error on the 1 024-ray fitted fixture when the FINE grid's offset queries run the mixed trunk form with their LAST layers
as the plain f16 product alone (emulation, CPU, numpy; everything else f16x3).  python scratch/prec_probe_mixed.py [n_rays]"""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import _pkg; _pkg.load()
import iblnerf_oracle as O
from conftest import load_golden, load_lut_rgb

def f16(a): return a.astype(np.float16).astype(np.float32)
def q6(a, block=32):
    sh = a.shape; K = sh[-1]; pad = (-K) % block
    x = np.pad(a.astype(np.float64), [(0, 0)] * (a.ndim - 1) + [(0, pad)]).reshape(sh[:-1] + (-1, block))
    mx = np.abs(x).max(-1, keepdims=True)
    s = 2.0 ** np.where(mx > 0, np.floor(np.log2(np.maximum(mx, 1e-300))) - 2, 0.0)
    v = x / s
    m, e = np.frexp(v); normal = np.ldexp(np.rint(m * 16) / 16, e)
    q = np.where(np.abs(v) >= 1.0, normal, np.rint(v * 8) / 8)
    return (np.clip(q, -7.5, 7.5) * s).reshape(sh[:-1] + (-1,))[..., :K]

HEADS = ("sigma_linear", "roughness_linear", "albedo_linear", "irradiance_linear", "radiance_linear", "additional_radiance_linear")
FAST = False          # the current query runs the fast scheme ...
PRECISE = set()       # ... except these layers
PLAIN = set()         # ... and these run the f16 main product alone (2^-11)
def lin(sd, name, x):
    W, b = sd[name + ".weight"], sd[name + ".bias"]
    if name.startswith(HEADS): return (x @ W.T + b).astype(np.float32)
    x64 = lambda a: a.astype(np.float64)
    Wh, Xh = f16(W), f16(x)
    if not FAST or name in PRECISE:
        Wl, Xl = f16(W - Wh), f16(x - Xh)
        return (x64(Xh) @ x64(Wh).T + x64(Xl) @ x64(Wh).T + x64(Xh) @ x64(Wl).T + b).astype(np.float32)
    if name in PLAIN:
        return (x64(Xh) @ x64(Wh).T + b).astype(np.float32)
    Wl, Xl = W - Wh, x - Xh
    return (x64(Xh) @ x64(Wh).T + q6(Xl) @ q6(Wh).T + q6(Xh) @ q6(Wl).T + b).astype(np.float32)
O._lin = lin
_nq = O.network_query
N_OFFSET = [0]
def network_query(sd, pts, viewdirs):
    global FAST
    if viewdirs is None:
        N_OFFSET[0] += 1
        FAST = N_OFFSET[0] % 2 == 0      # render_rays issues the coarse pass's offset query first, then the fine pass's: only the fine one is fast
    else:
        FAST = False
    out = _nq(sd, pts, viewdirs)
    FAST = False
    return out
O.network_query = network_query

g, sdc, sdf, gt, edit = load_golden("fitted_wide")
# the 96 rays whose normal is worst on the GPU under the coarser modes (scratch/worst_rays.py) + 96 others: the full 1 024 take 25 min per row
import os
rsel = np.load("gpurun_out/worst_rays.npy") if os.path.exists("gpurun_out/worst_rays.npy") else np.arange(96)
rsel = np.concatenate([rsel, np.setdiff1d(np.arange(1024), rsel)[:96]])
lut = load_lut_rgb()
L = ["positions_linears.%d" % i for i in range(8)]
for label, sel, plain in (("layers 0-1 precise (shipped)", L[:2], []), ("... layer 7 plain f16", L[:2], L[7:]), ("... layers 6-7 plain", L[:2], L[6:]),
                          ("... layers 4-7 plain", L[:2], L[4:]), ("... layers 2-7 plain", L[:2], L[2:])):
    PRECISE = set(sel); PLAIN = set(plain); N_OFFSET[0] = 0
    res = O.render_rays(sdc, sdf, g["rays_o"][rsel], g["rays_d"][rsel], 0.5, 8.0, lut)
    e = np.abs(res["target_normal_map"] - g["out__target_normal_map"][rsel]).max(-1)
    d = np.abs(res["depth_map"] - g["out__depth_map"][rsel]) / np.abs(g["out__depth_map"]).max()
    print("%-30s normal: max %.2e  2nd %.2e  10th %.2e  median %.2e   depth max %.1e" % (label, e.max(), np.sort(e)[-2], np.sort(e)[-10], np.median(e), d.max()), flush=True)
