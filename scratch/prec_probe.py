"""Precision probe: effect of rounding the WEIGHTS (not activations) to k bits on the oracle's maps,
relative L-inf vs the golden reference outputs.  Activations stay fp32."""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import _pkg; _pkg.load()
import iblnerf_oracle as O
from conftest import load_golden, golden_flags, rel_linf
from PIL import Image
lut = np.ascontiguousarray((np.asarray(Image.open("tests/golden/ibl_brdf_lut.png").convert("RGB"), dtype=np.float32) / np.float32(255)).transpose(2, 0, 1))

def round_bits(a, keep):  # keep = explicit mantissa bits
    if keep == "fp16":
        return a.astype(np.float16).astype(np.float32)
    if keep >= 23: return a
    u = a.view(np.uint32).astype(np.uint64)
    sh = 23 - keep
    u = (u + (1 << (sh - 1)) - 1 + ((u >> sh) & 1)) >> sh << sh
    return u.astype(np.uint32).view(np.float32)

name = sys.argv[1] if len(sys.argv) > 1 else "plain_g10"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
g, sdc, sdf, gt, edit = load_golden(name)
keys = ["albedo_map", "roughness_map", "irradiance_map", "radiance_map", "depth_map", "target_normal_map", "prefiltered_reflected_map", "specular_map", "color_map", "weights"]
for mode in (23, "fp16", 13, 10, 7):
    for trunk_only in (False,):
        rc = {k: (round_bits(v, mode) if k.endswith("weight") else v) for k, v in sdc.items()}
        rf = {k: (round_bits(v, mode) if k.endswith("weight") else v) for k, v in sdf.items()}
        res = O.render_rays(rc, rf, g["rays_o"][:n], g["rays_d"][:n], float(g["near"]), float(g["far"]), lut, 64, int(g["n_importance"]),
                            {k: v[:n] for k, v in gt.items()} if gt else gt, edit, {}, golden_flags(g))
        print(mode, " ".join("%s=%.1e" % (k.replace("_map", ""), rel_linf(res[k], g["out__" + k][:n])) for k in keys), flush=True)
