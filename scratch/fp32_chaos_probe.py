"""A fourth yardstick for rays the three reference yardsticks do not explain: the SAME fp32 formulas in two roundings — the C restatement with fused multiply-adds
in its dense layers (AVX-512 build) and without (baseline build).  A ray whose output moves between the two by as much as the HIP path is off is chaotic under ANY
change of rounding; one that does not, is not.     python scratch/fp32_chaos_probe.py <fixture> [n_worst]"""
import os, sys, subprocess, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
name = sys.argv[1]
if len(sys.argv) > 2 and sys.argv[2] == "--child":
    import iblnerf_cpu as OC
    from conftest import load_golden, load_lut_rgb
    g, sdc, sdf, gt, edit = load_golden(name)
    sel = np.load(sys.argv[3])
    if "rays_o" in g.files:
        ro, rd = g["rays_o"][sel], g["rays_d"][sel]
    else:
        fl = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
        K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
        fo, fd = OC.get_rays(800, 800, K, np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32))
        ro, rd = fo.reshape(-1, 3)[g["pix"][sel]], fd.reshape(-1, 3)[g["pix"][sel]]
    r = OC.render_rays(sdc, sdf, ro, rd, 0.5, 8.0, load_lut_rgb())
    np.savez(sys.argv[4], isa=np.array(OC.isa()), **{k: r[k] for k in ("depth_map", "target_normal_map", "albedo_map")})
    sys.exit(0)
from conftest import load_golden
import test_gpu_launch_scale as LS
g, *_ = load_golden(name)
e = np.load(os.path.join(ROOT, "gpurun_out", "ray_errors_%s.npz" % name))
rows = []
for key, base in (("target_normal_map", 2e-3), ("depth_map", 1e-3), ("albedo_map", 1e-3)):
    f = LS.ray_floor(g, key)
    for mode in ("f16x3_mxfp6x", "f16x3"):
        err = e[mode + "__" + key]
        bad = np.flatnonzero(err > np.maximum(base, 16 * f))
        for i in bad:
            rows.append((key, mode, int(i), float(err[i]), float(f[i])))
sel = np.array(sorted({r[2] for r in rows}), dtype=np.int64)
print("%s: %d rays beyond max(1e-3 | 2e-3, 16x their own reference sensitivity) in some mode: %s" % (name, len(sel), sel.tolist()))
if len(sel) == 0:
    sys.exit(0)
# add their neighbours in the list so that the dense-layer blocks are not degenerate, then render in both roundings
with tempfile.TemporaryDirectory() as td:
    np.save(td + "/sel.npy", sel)
    outs = {}
    for isa in ("avx512", "base"):
        subprocess.run([sys.executable, __file__, name, "--child", td + "/sel.npy", td + "/%s.npz" % isa], env=dict(os.environ, IBLNERF_CPU_ISA=isa), check=True)
        outs[isa] = np.load(td + "/%s.npz" % isa)
    assert str(outs["avx512"]["isa"]) == "avx512" and str(outs["base"]["isa"]) == "base"
    for key in ("target_normal_map", "depth_map", "albedo_map"):
        ref = g["out__" + key][sel].astype(np.float64); scale = np.abs(g["out__" + key]).max()
        pr = lambda a, b: np.abs(a.astype(np.float64).reshape(ref.shape) - b.astype(np.float64).reshape(ref.shape)).reshape(len(sel), -1).max(-1) / scale
        hip = {m: e[m + "__val__" + key][sel] for m in ("f16x3_mxfp6x", "f16x3")}
        print("\n%s   ray: reference's own sensitivity | HIP default, HIP f16x3 vs reference | C(fma) vs reference, C(no fma) vs reference | C(fma) vs C(no fma) | HIP f16x3 vs C(fma)" % key)
        f = LS.ray_floor(g, key)[sel]
        a, b = outs["avx512"][key], outs["base"][key]
        for j, i in enumerate(sel):
            print("  %6d: %.1e | %.1e %.1e | %.1e %.1e | %.1e | %.1e" % (i, f[j], pr(hip["f16x3_mxfp6x"], ref)[j], pr(hip["f16x3"], ref)[j], pr(a, ref)[j], pr(b, ref)[j], pr(a, b)[j], pr(hip["f16x3"], a)[j]))
