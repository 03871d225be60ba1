"""Which layers carry the sigma error on the fitted checkpoint?  Emulation (CPU, numpy): f16 + MX-fp6 products everywhere except a
set of trunk layers evaluated as f16x3; density of the fine pass's offset-query points of fitted_plain against the reference.
    python scratch/prec_probe_layers.py"""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import _pkg; _pkg.load()
import iblnerf_oracle as O
from conftest import load_golden

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

PRECISE = set()
def lin(sd, name, x):
    W, b = sd[name + ".weight"], sd[name + ".bias"]
    if name.startswith("sigma_linear"): return (x @ W.T + b).astype(np.float32)      # fp32 VALU head
    x64 = lambda a: a.astype(np.float64)
    Wh, Xh = f16(W), f16(x)
    if name in PRECISE:
        Wl, Xl = f16(W - Wh), f16(x - Xh)
        return (x64(Xh) @ x64(Wh).T + x64(Xl) @ x64(Wh).T + x64(Xh) @ x64(Wl).T + b).astype(np.float32)
    Wl, Xl = W - Wh, x - Xh
    return (x64(Xh) @ x64(Wh).T + q6(Xl) @ q6(Wh).T + q6(Xh) @ q6(Wl).T + b).astype(np.float32)
O._lin = lin
g, sdc, sdf, gt, edit = load_golden("fitted_plain")
pts, ref = g["q_f_eps_pts"][:96], g["q_f_eps_sigma"][:96]
L = ["positions_linears.%d" % i for i in range(8)]
cases = [("none (all f16+fp6)", [])] + [("only layer %d" % i, [L[i]]) for i in range(8)] + \
        [("first %d" % k, L[:k]) for k in (2, 3, 4, 6)] + [("layers 0,5 (encoding inputs)", [L[0], L[5]]), ("layers 0-2 + 5", L[:3] + [L[5]]), ("all", L)]
for label, sel in cases:
    PRECISE = set(sel)
    e = np.abs(O.network_query(sdf, pts, None) - ref)
    print("%-30s slots/eval %.1f   sigma abs err max %.2e  rms %.2e" % (label, (6 * (8 - len(sel)) + 12 * len(sel)) / 8, e.max(), np.sqrt((e ** 2).mean())), flush=True)

# --- would scaling the raw-coordinate slots of the encoding (|x| up to 8, next to sin / cos <= 1 in one fp6 block) help the MX scheme?
print("--- f16 + fp6 everywhere, raw coordinate slots of the encoding scaled by 2^-k (weights by 2^k: exact)")
_embed = O.embed
for k in (0, 2, 3, 4):
    PRECISE = set()
    def embed_scaled(x, n_freqs, k=k):
        e = _embed(x, n_freqs).copy()
        e[..., :3] *= np.float32(2.0 ** -k)
        return e
    O.embed = embed_scaled
    sd2 = dict(sdf)
    for name in ("positions_linears.0.weight", "positions_linears.5.weight"):
        w = sdf[name].copy(); w[:, :3] *= np.float32(2.0 ** k); sd2[name] = w
    e = np.abs(O.network_query(sd2, pts, None) - ref)
    print("k=%d  sigma abs err max %.2e  rms %.2e" % (k, e.max(), np.sqrt((e ** 2).mean())), flush=True)
O.embed = _embed
