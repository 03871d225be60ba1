"""Would the main queries keep their parity with the HEAD layers (feature / albedo- / irradiance-feature / views / additional-radiance-
feature) in the fast scheme and only the eight trunk layers in f16x3?  Emulation on fitted_plain (CPU, numpy)."""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import _pkg; _pkg.load()
import iblnerf_oracle as O
from conftest import load_golden, load_lut_rgb, rel_linf
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
FAST_HEAD_LAYERS = False
def lin(sd, name, x):
    W, b = sd[name + ".weight"], sd[name + ".bias"]
    if name.startswith(HEADS): return (x @ W.T + b).astype(np.float32)
    x64 = lambda a: a.astype(np.float64)
    Wh, Xh = f16(W), f16(x)
    if name.startswith("positions_linears") or not FAST_HEAD_LAYERS:
        Wl, Xl = f16(W - Wh), f16(x - Xh)
        return (x64(Xh) @ x64(Wh).T + x64(Xl) @ x64(Wh).T + x64(Xh) @ x64(Wl).T + b).astype(np.float32)
    Wl, Xl = W - Wh, x - Xh
    return (x64(Xh) @ x64(Wh).T + q6(Xl) @ q6(Wh).T + q6(Xh) @ q6(Wl).T + b).astype(np.float32)
O._lin = lin
g, sdc, sdf, gt, edit = load_golden("fitted_plain")
lut = load_lut_rgb()
KEYS = ["depth_map", "weights", "albedo_map", "roughness_map", "irradiance_map", "radiance_map", "radiance_map_3", "target_normal_map", "prefiltered_reflected_map", "color_map", "albedo_map0", "irradiance_map0"]
for fast in (False, True):
    FAST_HEAD_LAYERS = fast
    res = O.render_rays(sdc, sdf, g["rays_o"], g["rays_d"], 0.5, 8.0, lut)
    raw = O.network_query(sdf, g["q_f_main_pts"][:24], g["q_f_main_dirs"][:24])
    print("head layers %-8s raw abs err per channel %s" % ("fast" if fast else "precise", np.array2string(np.abs(raw - g["q_f_main_raw"][:24]).reshape(-1, 18).max(0), precision=1)))
    print("   ", " ".join("%s %.1e" % (k.replace("_map", ""), rel_linf(res[k], g["out__" + k])) for k in KEYS), flush=True)
