"""Precision probe for a cheaper product scheme (NOT shipped): main term bf16(W)·bf16(X) in float, the two
correction terms W_hi·X_lo + W_lo·X_hi on int8 limbs with a per-row (W) / per-point (X) power-of-two-related
scale, accumulated in int32.  Emulated in numpy on the oracle; compared with the reference goldens."""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import _pkg; _pkg.load()
import iblnerf_oracle as O
from conftest import load_golden, golden_flags, rel_linf
from PIL import Image
lut = np.ascontiguousarray((np.asarray(Image.open("tests/golden/ibl_brdf_lut.png").convert("RGB"), dtype=np.float32) / np.float32(255)).transpose(2, 0, 1))

def bf16(a):
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)

MODE = "bf16x3"
def lin(sd, name, x):
    W, b = sd[name + ".weight"], sd[name + ".bias"]
    Wh, Xh = bf16(W), bf16(x)
    Wl, Xl = (W - Wh).astype(np.float32), (x - Xh).astype(np.float32)
    if MODE == "fp32":
        return (x @ W.T + b).astype(np.float32)
    if MODE == "bf16x3":
        return (Xh.astype(np.float64) @ Wh.T.astype(np.float64) + Xl.astype(np.float64) @ Wh.T + Xh.astype(np.float64) @ bf16(Wl).T + b).astype(np.float32)
    # int8 corrections
    sXh = np.maximum(np.abs(Xh).max(-1, keepdims=True), 1e-30) / 127.0
    sWh = np.maximum(np.abs(Wh).max(-1, keepdims=True), 1e-30) / 127.0
    if MODE == "int8_pow2":      # scales rounded UP to powers of two (exact rescale, one shared accumulator)
        sXh = 2.0 ** np.ceil(np.log2(sXh)); sWh = 2.0 ** np.ceil(np.log2(sWh))
    sXl, sWl = sXh / 256.0, sWh / 256.0
    q = lambda a, s: np.clip(np.rint(a / s), -127, 127)
    corr = (q(Xl, sXl) @ q(Wh, sWh).T + q(Xh, sXh) @ q(Wl, sWl).T) * (sXh * sWh.T / 256.0)
    return (Xh.astype(np.float64) @ Wh.T.astype(np.float64) + corr + b).astype(np.float32)

O._lin = lin
name = sys.argv[1] if len(sys.argv) > 1 else "plain_g10"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
g, sdc, sdf, gt, edit = load_golden(name)
keys = ["albedo_map", "roughness_map", "irradiance_map", "radiance_map", "depth_map", "target_normal_map", "prefiltered_reflected_map", "specular_map", "color_map", "weights"]
for MODE in ("fp32", "bf16x3", "int8", "int8_pow2"):
    res = O.render_rays(sdc, sdf, g["rays_o"][:n], g["rays_d"][:n], float(g["near"]), float(g["far"]), lut, 64, int(g["n_importance"]),
                        {k: v[:n] for k, v in gt.items()} if gt else gt, edit, {}, golden_flags(g))
    print("%-10s" % MODE, " ".join("%s=%.1e" % (k.replace("_map", ""), rel_linf(res[k], g["out__" + k][:n])) for k in keys), flush=True)
