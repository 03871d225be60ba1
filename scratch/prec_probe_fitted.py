"""Operand-precision probe on the FITTED checkpoint (numpy emulation on the oracle, CPU): which product scheme holds which
error on a network with surfaces.  Schemes emulate the kernels: hidden-layer GEMMs in the scheme, N=1/3 heads in fp32.
  python scratch/prec_probe_fitted.py [n_rays]"""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, "."); sys.path.insert(0, "scratch")
import _pkg; _pkg.load()
import iblnerf_oracle as O
from conftest import load_golden, rel_linf, load_lut_rgb
lut = load_lut_rgb()

def f16(a): return a.astype(np.float16).astype(np.float32)
def bf16(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)
def q6(a, block=32):   # e2m3 with a power-of-two scale per 32 k-elements, block max in [4, 8)
    sh = a.shape; K = sh[-1]; pad = (-K) % block
    x = np.pad(a.astype(np.float64), [(0, 0)] * (a.ndim - 1) + [(0, pad)]).reshape(sh[:-1] + (-1, block))
    mx = np.abs(x).max(-1, keepdims=True)
    s = 2.0 ** np.where(mx > 0, np.floor(np.log2(np.maximum(mx, 1e-300))) - 2, 0.0)
    v = x / s
    m, e = np.frexp(v); normal = np.ldexp(np.rint(m * 16) / 16, e)
    q = np.where(np.abs(v) >= 1.0, normal, np.rint(v * 8) / 8)
    return (np.clip(q, -7.5, 7.5) * s).reshape(sh[:-1] + (-1,))[..., :K]

HEADS = ("sigma_linear", "roughness_linear", "albedo_linear", "irradiance_linear", "radiance_linear", "additional_radiance_linear")
MODE = "fp32"
def lin(sd, name, x):
    W, b = sd[name + ".weight"], sd[name + ".bias"]
    if MODE == "fp32" or name.startswith(HEADS):
        return (x @ W.T + b).astype(np.float32)
    x64 = lambda a: a.astype(np.float64)
    if MODE in ("bf16x3", "f16x3", "f16x4", "bf16x6"):
        r = bf16 if MODE.startswith("bf16") else f16
        Wh, Xh = r(W), r(x); Wl, Xl = r(W - Wh), r(x - Xh)
        y = x64(Xh) @ x64(Wh).T + x64(Xl) @ x64(Wh).T + x64(Xh) @ x64(Wl).T
        if MODE == "f16x4": y += x64(Xl) @ x64(Wl).T
        if MODE == "bf16x6":
            Wm, Xm = r(W - Wh - Wl), r(x - Xh - Xl)
            y += x64(Xl) @ x64(Wl).T + x64(Xm) @ x64(Wh).T + x64(Xh) @ x64(Wm).T
        return (y + b).astype(np.float32)
    if MODE == "f16+fp6":
        Wh, Xh = f16(W), f16(x); Wl, Xl = W - Wh, x - Xh
        return (x64(Xh) @ x64(Wh).T + q6(Xl) @ q6(Wh).T + q6(Xh) @ q6(Wl).T + b).astype(np.float32)
    if MODE == "f16":
        return (x64(f16(x)) @ x64(f16(W)).T + b).astype(np.float32)
    raise ValueError(MODE)
O._lin = lin

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
g, sdc, sdf, gt, edit = load_golden("fitted_plain")
ro, rd = g["rays_o"][:n], g["rays_d"][:n]
KEYS = ["depth_map", "weights", "albedo_map", "roughness_map", "irradiance_map", "target_normal_map", "prefiltered_reflected_map", "color_map"]
print("%-9s sigma_abs(main,f) " % "mode" + " ".join("%-9s" % k.replace("_map", "")[:9] for k in KEYS) + "  | ray13 depth")
for mode in ("fp32", "bf16x3", "f16+fp6", "f16x3", "f16x4", "bf16x6", "f16"):
    MODE = mode
    raw = O.network_query(sdf, g["q_f_main_pts"][:24], g["q_f_main_dirs"][:24])
    sig_abs = np.abs(raw[..., 0] - g["q_f_main_raw"][:24, :, 0]).max()
    res = O.render_rays(sdc, sdf, ro, rd, 0.5, 8.0, lut)
    errs = [rel_linf(res[k], g["out__" + k][:n]) for k in KEYS]
    d13 = abs(res["depth_map"][13] - g["out__depth_map"][13]) / np.abs(g["out__depth_map"]).max() if n > 13 else 0
    print("%-9s %.1e           " % (mode, sig_abs) + " ".join("%.1e  " % e for e in errs) + "  | %.1e" % d13, flush=True)
