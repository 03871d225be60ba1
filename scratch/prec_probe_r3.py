"""Round 3 precision ladder (CPU emulation, numpy): which product scheme does each query class need on the fitted checkpoint?

One query class at a time runs an emulated scheme, per layer one of
    p   three f16 products on hi/lo splits                      Wh Xh + Wh Xl + Wl Xh                 12 slots
    f   one f16 product + two block-scaled fp6 residual products Wh Xh + q6(Wh) q6(Xl) + q6(Wl) q6(Xh)  6 slots
    a   activations precise, weight residual in fp6              Wh Xh + Wh Xl + q6(Wl) q6(Xh)           9 slots
    w   weights precise, activation residual in fp6              Wh Xh + Wl Xh + q6(Wh) q6(Xl)           9 slots
every other query is the fp32 oracle (which stands for f16x3: 2^-22).  Rays: the worst-conditioned of fitted_launch16k by the
reference's own float64-vs-float32 difference + a random set.  Error is taken against the reference's float32 render.
    python scratch/prec_probe_r3.py <class> <n_rays> <scheme> [<scheme> ...]      class: offc | offf | mainc | mainf
    scheme: 8 letters for positions_linears.0-7 (heads fp32), e.g. ppffffff; for the main classes 8+1: the last letter = all head-side layers
"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
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
ACTIVE = None      # scheme string of the query being evaluated, or None = fp32
_lin0 = O._lin
def lin(sd, name, x):
    if ACTIVE is None or name.startswith(HEADS):
        return _lin0(sd, name, x)
    if name.startswith("positions_linears."):
        s = ACTIVE[int(name.split(".")[1])]
    else:
        s = ACTIVE[8] if len(ACTIVE) > 8 else "p"
    W, b = sd[name + ".weight"], sd[name + ".bias"]
    x64 = lambda a: a.astype(np.float64)
    Wh, Xh = f16(W), f16(x)
    acc = x64(Xh) @ x64(Wh).T
    Wl, Xl = W - Wh, x - Xh
    if s == "F":
        # F: three f16 products + THREE block-scaled fp6 products for everything at the 2^-22 level: Wl Xl, W3 Xh, Wh X3 (W3, X3 = what two f16 halves
        #    leave of the fp32 value) — 15 slots; every operand is then represented to ~2^-25
        Xl16, Wl16 = f16(Xl), f16(Wl)
        X3, W3 = (Xl - Xl16).astype(np.float32), (Wl - Wl16).astype(np.float32)
        acc = x64(Xh) @ x64(Wh).T + x64(Xl16) @ x64(Wh).T + x64(Xh) @ x64(Wl16).T
        acc += q6(Xl16) @ q6(Wl16).T + q6(Xh) @ q6(W3).T + q6(X3) @ q6(Wh).T
        return (acc + b).astype(np.float32)
    if s in "eE":
        # e: EXACT weights (a three-way split of W: the third term costs one block-scaled slot per 64 k), activations as f16 pairs: W Xh + Wh Xl
        # E: the same with the activation residual's product against the exact W too (W Xh + W Xl)
        Xl_ = x64(f16(Xl))
        acc = x64(Xh) @ x64(W).T + Xl_ @ (x64(W) if s == "E" else x64(Wh)).T
        return (acc + b).astype(np.float32)
    if s in "sSq":
        # s: three f16 products whose residual operands are converted at a 2^12 larger exponent (no f16-denormal loss in the lo parts: what
        #    scaling W and X per layer by powers of two would give); S: the same for the activation residual only; q: s plus the fourth product Wl Xl
        sc = np.float32(4096.0)
        Xl_ = x64(f16(Xl * sc)) / 4096.0 if s in "sSq" else x64(f16(Xl))
        Wl_ = x64(f16(Wl * sc)) / 4096.0 if s in "sq" else x64(f16(Wl))
        acc += Xl_ @ x64(Wh).T + x64(Xh) @ Wl_.T
        if s == "q":
            acc += Xl_ @ Wl_.T
        return (acc + b).astype(np.float32)
    acc += (x64(f16(Xl)) @ x64(Wh).T) if s in "pa" else (q6(Xl) @ q6(Wh).T)
    acc += (x64(Xh) @ x64(f16(Wl)).T) if s in "pw" else (q6(Xh) @ q6(Wl).T)
    return (acc + b).astype(np.float32)
O._lin = lin
_nq = O.network_query
COUNT = {"off": 0, "main": 0}
CLASS, SCHEME = None, None
def network_query(sd, pts, viewdirs):
    global ACTIVE
    # render_rays order per pass: main, offsets, reflected; coarse pass first
    if viewdirs is None:
        COUNT["off"] += 1
        cls = "offc" if COUNT["off"] % 2 == 1 else "offf"
    elif pts.shape[1] == 64 and COUNT["main"] % 4 in (1, 3):
        COUNT["main"] += 1; cls = "refl"
    else:
        cls = "mainc" if COUNT["main"] % 4 == 0 else "mainf"
        COUNT["main"] += 1
    ACTIVE = SCHEME if cls == CLASS else None
    out = _nq(sd, pts, viewdirs)
    ACTIVE = None
    return out
O.network_query = network_query

CLASS, n_rays = sys.argv[1], int(sys.argv[2])
FIXTURE = os.environ.get("PROBE_FIXTURE", "fitted_launch16k")
g, sdc, sdf, gt, edit = load_golden(FIXTURE)
fl = np.maximum.reduce([g["floorray__" + k] for k in ("target_normal_map0", "target_normal_map", "depth_map", "depth_map0")])
order = np.argsort(-fl)
rsel = np.concatenate([order[:n_rays // 2], np.random.RandomState(0).permutation(order[n_rays // 2:])[:n_rays - n_rays // 2]])
if os.environ.get("PROBE_RAYS"):      # explicit ray ids first (e.g. the rays a GPU run flagged)
    extra = np.array([int(t) for t in os.environ["PROBE_RAYS"].split(",")])
    rsel = np.concatenate([extra, np.setdiff1d(rsel, extra)[:n_rays - len(extra)]])
lut = load_lut_rgb()
keys = ("weights0", "target_normal_map0", "target_normal_map", "depth_map0", "depth_map", "albedo_map", "roughness_map")
print("class %s, %d rays (the %d worst-conditioned of 16 384 + random); reference's own f64-vs-f32 on them: %s" % (
    CLASS, n_rays, n_rays // 2, "  ".join("%s %.1e" % (k, g["floorray__" + k][rsel].max()) for k in keys)), flush=True)
SCHEME = None; COUNT["off"] = COUNT["main"] = 0; CLASS_, CLASS = CLASS, "none"
BASE = O.render_rays(sdc, sdf, g["rays_o"][rsel], g["rays_d"][rsel], 0.5, 8.0, lut)      # the fp32 oracle on the same rays (weights0: the fixtures keep every k-th row only)
CLASS = CLASS_
for scheme in sys.argv[3:]:
    SCHEME = scheme; COUNT["off"] = COUNT["main"] = 0
    t0 = time.time()
    res = O.render_rays(sdc, sdf, g["rays_o"][rsel], g["rays_d"][rsel], 0.5, 8.0, lut)
    row = []
    for k in keys:
        ref = (BASE[k] if k.startswith("weights") else g["out__" + k][rsel]).astype(np.float64)
        e = np.abs(res[k].astype(np.float64).reshape(ref.shape) - ref).reshape(len(ref), -1).max(-1) / np.abs(g["out__" + k]).max()
        row.append("%s %.1e/%.1e [%d>1e-3]" % (k.replace("target_", "").replace("_map", ""), e.max(), np.sort(e)[-max(2, len(e) // 50)], int((e > 1e-3).sum())))
    slots = sum({"p": 12, "f": 6, "a": 9, "w": 9, "s": 12, "S": 12, "q": 16, "e": 13, "E": 14, "F": 15}[c] for c in scheme[:8]) / 8
    print("%-10s %4.1f slots  (max / 98%%)  %s   [%.0f s]" % (scheme, slots, "  ".join(row), time.time() - t0), flush=True)
    if os.environ.get("PROBE_RAYS"):
        ref = g["out__target_normal_map"][rsel].astype(np.float64)
        e = np.abs(res["target_normal_map"].astype(np.float64) - ref).max(-1)
        print("           normal error on the listed rays:", " ".join("%.1e" % v for v in e[:len(extra)]), flush=True)
