"""Precision probe for the fp16-main + low-precision-correction product schemes (f16+fp6 is the shipped one; the two
five-slot variants f16W+fp6X / fp6W+f16X drop one residual product):
  y = f16(W)·f16(X)  +  fp8(Wh)·fp8(Xl) + fp8(Wl)·fp8(Xh)      (fp32 accumulate)
with Xh = f16(X), Xl = X - Xh (|Xl| <= 2^-12 |X|), fp8 = e4m3 with an MX-style power-of-two scale per
block of 32 K-elements (per weight row / per point).  Emulated in numpy on the oracle."""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import _pkg; _pkg.load()
import iblnerf_oracle as O
from conftest import load_golden, golden_flags, rel_linf
from PIL import Image
lut = np.ascontiguousarray((np.asarray(Image.open("tests/golden/ibl_brdf_lut.png").convert("RGB"), dtype=np.float32) / np.float32(255)).transpose(2, 0, 1))

def f16(a):
    return a.astype(np.float16).astype(np.float32)

def round_sig(a, bits):
    """round to `bits` significant bits (RNE), no exponent limits"""
    m, e = np.frexp(a.astype(np.float64))
    return np.ldexp(np.rint(m * 2.0 ** bits) / 2.0 ** bits, e)

def fp8_block(a, block=32, sig=4, emin=-6, top=7, vmax=448.0):
    """e4m3 with a shared power-of-two scale per `block` consecutive elements of the last axis:
    scale chosen so the block max lands in [128, 256); values below 2^emin (after scaling) are subnormal."""
    sh = a.shape
    K = sh[-1]
    pad = (-K) % block
    x = np.pad(a.astype(np.float64), [(0, 0)] * (a.ndim - 1) + [(0, pad)])
    x = x.reshape(sh[:-1] + (-1, block))
    mx = np.abs(x).max(-1, keepdims=True)
    e = np.where(mx > 0, np.floor(np.log2(np.maximum(mx, 1e-300))) - top, 0.0)
    s = 2.0 ** e
    v = x / s
    mag = np.abs(v)
    normal = round_sig(v, sig)
    sub = np.rint(v / 2.0 ** (emin - (sig - 1))) * 2.0 ** (emin - (sig - 1))   # subnormal grid: 2^-9 for e4m3
    q = np.where(mag >= 2.0 ** emin, normal, sub)
    q = np.clip(q, -vmax, vmax)
    return (q * s).reshape(sh[:-1] + (-1,))[..., :K]

MODE = "fp32"
def lin(sd, name, x):
    W, b = sd[name + ".weight"], sd[name + ".bias"]
    if MODE == "fp32":
        return (x @ W.T + b).astype(np.float32)
    Wh, Xh = f16(W), f16(x)
    Wl, Xl = (W - Wh).astype(np.float32), (x - Xh).astype(np.float32)
    main = Xh.astype(np.float64) @ Wh.T.astype(np.float64)
    if MODE == "f16_only":
        return (main + b).astype(np.float32)
    if MODE == "f16x3":
        return (main + Xl.astype(np.float64) @ Wh.T + Xh.astype(np.float64) @ f16(Wl).T + b).astype(np.float32)
    if MODE == "f16W+fp6X":        # five slots: f16(W) f16(X) + fp6(W) fp6(X - f16 X); the weights only to f16 (a fixed perturbation)
        q6 = lambda a: fp8_block(a, sig=4, emin=0, top=2, vmax=7.5)
        return (main + q6(Xl) @ q6(Wh).T + b).astype(np.float32)
    if MODE == "fp6W+f16X":        # five slots the other way: activations only to f16 (per-point noise)
        q6 = lambda a: fp8_block(a, sig=4, emin=0, top=2, vmax=7.5)
        return (main + q6(Xh) @ q6(Wl).T + b).astype(np.float32)
    fmt = {"f16+fp8": dict(sig=4, emin=-6, top=7, vmax=448.0),       # e4m3
           "f16+bf8": dict(sig=3, emin=-14, top=14, vmax=57344.0),   # e5m2
           "f16+fp6": dict(sig=4, emin=0, top=2, vmax=7.5),          # e2m3: block max in [4, 8)
           "f16+fp4": dict(sig=2, emin=0, top=2, vmax=6.0)}[MODE]    # e2m1
    q = lambda a: fp8_block(a, **fmt)
    corr = q(Xl) @ q(Wh).T + q(Xh) @ q(Wl).T
    return (main + corr + b).astype(np.float32)

O._lin = lin
# hybrid modes "A|B": trunk-only queries (the eps-normal's offset points) in scheme A, queries with view directions (main, reflected) in B
_nq = O.network_query
def network_query(sd, pts, viewdirs):
    global MODE
    if "|" not in HYBRID:
        return _nq(sd, pts, viewdirs)
    global N_VIEW_QUERIES
    parts = HYBRID.split("|")
    if viewdirs is None:
        MODE = parts[0]
    else:                       # "A|B|keepcoarse": the coarse pass's main query (the first one with directions: it places the fine samples) stays in A
        MODE = parts[0] if (len(parts) > 2 and N_VIEW_QUERIES == 0) else parts[1]
        N_VIEW_QUERIES += 1
    return _nq(sd, pts, viewdirs)
O.network_query = network_query
HYBRID = ""
N_VIEW_QUERIES = 0
name = sys.argv[1] if len(sys.argv) > 1 else "plain_g10"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f16+fp8", "f16+bf8", "f16+fp6", "f16+fp4"]
g, sdc, sdf, gt, edit = load_golden(name)
keys = ["albedo_map", "roughness_map", "irradiance_map", "radiance_map", "depth_map", "target_normal_map", "prefiltered_reflected_map", "specular_map", "color_map", "weights"]
for MODE in modes:
    HYBRID = MODE
    N_VIEW_QUERIES = 0
    res = O.render_rays(sdc, sdf, g["rays_o"][:n], g["rays_d"][:n], float(g["near"]), float(g["far"]), lut, 64, int(g["n_importance"]),
                        {k: v[:n] for k, v in gt.items()} if gt else gt, edit, {}, golden_flags(g))
    print("%-18s" % HYBRID, " ".join("%s=%.1e" % (k.replace("_map", ""), rel_linf(res[k], g["out__" + k][:n])) for k in keys), flush=True)
