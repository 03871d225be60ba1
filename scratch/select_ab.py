"""The 15-slot density on the relevant coarse samples only (k_select_points) against on all of them: agreement of every map and the frame time, on both fitted checkpoints.
    python scratch/select_ab.py"""
import sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_lut_rgb
from ibl_nerf_amd import renderer as R, checkpoint as ck
lut = load_lut_rgb()
fl = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
for which in ("fitted", "fitted2", "synthetic"):
    if which == "synthetic":
        sdc, sdf = ck.synthetic_state_dict(0), ck.synthetic_state_dict(1)
    else:
        f = np.load(os.path.join(ROOT, "tests", "golden", which + "_ckpt.npz"))
        sdc, sdf = ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])
    out = {}
    for label, routing in (("selected", ()), ("est whole", ("estimates_whole",)), ("all points", ("coarse_density_all_points",))):
        r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x", query_routing=routing)
        r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
        ro, rd = r.get_rays(800, 800, K, c2w)
        ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        r.render_rays(ro[:65536], rd[:65536], 0.5, 8.0)
        torch.cuda.synchronize(); t0 = time.time()
        m = r.render_rays(ro, rd, 0.5, 8.0)
        torch.cuda.synchronize(); dt = time.time() - t0
        out[label] = m
        print("%-10s %-10s %.3f s/frame  selection %s" % (which, label, dt, r.last_selection()), flush=True)
    for a, b in ((out["selected"], out["all points"]), (out["selected"], out["est whole"])):
        worst = {k: float(((a[k] - b[k]).abs().nan_to_num(0).amax() / b[k].abs().nan_to_num(0).amax().clamp_min(1e-30))) for k in a}
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
        same = sum(bool(torch.equal(a[k].nan_to_num(7), b[k].nan_to_num(7))) for k in a)
        print("   maps bit-identical: %d of %d; largest relative differences: %s" % (same, len(a), "  ".join("%s %.1e" % kv for kv in top)), flush=True)
        print("   " + "  ".join("%s %.1e" % (k, worst[k]) for k in ("target_normal_map0", "depth_map0", "weights0", "albedo_map0", "target_normal_map", "depth_map", "weights", "color_map")), flush=True)
