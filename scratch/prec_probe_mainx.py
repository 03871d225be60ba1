"""Direct-channel error on the worst rays of the 1 024-ray fitted fixture when the FINE pass's MAIN query runs the fast scheme with its first
k trunk layers in f16x3 (emulation, CPU, numpy).  The rest as the default mode: coarse main / coarse offsets f16x3, fine offsets fast with
layers 0-1 precise, reflected queries fast.  python scratch/prec_probe_mainx.py"""
import sys, os, numpy as np
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
FAST = False
PRECISE = set()
PLAIN_ALL = False     # the fast query runs the f16 main product alone (2^-11): what a 4-slot f16 block would compute
def lin(sd, name, x):
    W, b = sd[name + ".weight"], sd[name + ".bias"]
    if name.startswith(HEADS) and not (HEADS_FAST and (CURRENT_IS_FINE_MAIN[0] or CURRENT_IS_REFL[0])): return (x @ W.T + b).astype(np.float32)
    x64 = lambda a: a.astype(np.float64)
    Wh, Xh = f16(W), f16(x)
    if not FAST or name in PRECISE:
        Wl, Xl = f16(W - Wh), f16(x - Xh)
        return (x64(Xh) @ x64(Wh).T + x64(Xl) @ x64(Wh).T + x64(Xh) @ x64(Wl).T + b).astype(np.float32)
    if PLAIN_ALL and CURRENT_IS_FINE_MAIN[0]:
        return (x64(Xh) @ x64(Wh).T + b).astype(np.float32)
    Wl, Xl = W - Wh, x - Xh
    return (x64(Xh) @ x64(Wh).T + q6(Xl) @ q6(Wh).T + q6(Xh) @ q6(Wl).T + b).astype(np.float32)
O._lin = lin
_nq = O.network_query
L = ["positions_linears.%d" % i for i in range(8)]
N_DIR, N_OFF = [0], [0]
CURRENT_IS_FINE_MAIN = [False]
CURRENT_IS_REFL = [False]
HEADS_FAST = "--heads" in sys.argv      # the N = 1/3 head products of the fast queries on the matrix core too (f16 + fp6 forms of their input)
MAIN_PRECISE = None      # None: fine main query in f16x3; else the set of its precise layers
def network_query(sd, pts, viewdirs):
    global FAST, PRECISE
    if viewdirs is None:
        N_OFF[0] += 1
        FAST, PRECISE = N_OFF[0] % 2 == 0, set(L[:2])          # coarse offsets precise, fine offsets mixed
    else:
        N_DIR[0] += 1
        q = (N_DIR[0] - 1) % 4                                   # 0 coarse main, 1 coarse reflected, 2 fine main, 3 fine reflected
        CURRENT_IS_FINE_MAIN[0] = q == 2
        CURRENT_IS_REFL[0] = q in (1, 3)
        if q in (1, 3): FAST, PRECISE = True, set()
        elif q == 2 and MAIN_PRECISE is not None: FAST, PRECISE = True, MAIN_PRECISE
        else: FAST = False
    out = _nq(sd, pts, viewdirs)
    FAST = False
    CURRENT_IS_FINE_MAIN[0] = False
    CURRENT_IS_REFL[0] = False
    return out
O.network_query = network_query

g, sdc, sdf, gt, edit = load_golden("fitted_wide")
rsel = np.load("gpurun_out/worst_rays_direct.npy")[:int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64]
lut = load_lut_rgb()
KEYS = ["depth_map", "weights", "albedo_map", "roughness_map", "irradiance_map", "radiance_map", "radiance_map_3", "target_normal_map"]
for label, sel, plain in (("fine main all fast (shipped)", set(), False), ("fine main plain f16 (2^-11)", set(), True)) if "--plain" in sys.argv else \
        (("fine main + reflected: heads on the matrix core too", set(), False),) if HEADS_FAST else \
        tuple((a, b, False) for a, b in (("fine main f16x3", None), ("fine main fast, L0-1 precise", set(L[:2])), ("L0-2", set(L[:3])), ("L0-3", set(L[:4])), ("all fast (shipped)", set()))):
    MAIN_PRECISE = sel; PLAIN_ALL = plain; N_DIR[0] = 0; N_OFF[0] = 0
    res = O.render_rays(sdc, sdf, g["rays_o"][rsel], g["rays_d"][rsel], 0.5, 8.0, lut)
    out = []
    for k in KEYS:
        ref = g["out__" + k]
        e = np.abs(res[k].reshape(ref[rsel].shape) - ref[rsel]).max() / np.abs(ref).max()
        out.append("%s %.1e (%.1fx)" % (k.replace("_map", ""), e, e / float(g["floor__" + k])))
    print("%-30s %s" % (label, "  ".join(out)), flush=True)
