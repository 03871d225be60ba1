"""How many of the FINE pass's 192 samples per ray carry a weight at all (the question behind extending k_select_points to the fine grid)?
    python scratch/fine_relevance.py"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_lut_rgb
from ibl_nerf_amd import renderer as R, checkpoint as ck
lut = load_lut_rgb()
fl = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
for which in ("fitted", "fitted2"):
    f = np.load(os.path.join(ROOT, "tests", "golden", which + "_ckpt.npz"))
    sdc, sdf = ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])
    r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x")
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    ro, rd = r.get_rays(800, 800, K, c2w)
    idx = torch.randperm(640000, device=ro.device)[:65536]
    m = r.render_rays(ro.reshape(-1, 3)[idx].contiguous(), rd.reshape(-1, 3)[idx].contiguous(), 0.5, 8.0)
    w = m["weights"]
    for thr in (0.0, 1e-12, 1e-8, 1e-6, 1e-4):
        print(which, "fine weights > %g: %.4f of %d samples" % (thr, float((w > thr).float().mean()), w.numel()), flush=True)
    first = (w > 0).float().argmax(1).float()
    last = 191 - (w > 1e-8).float().flip(1).argmax(1).float()
    print(which, "window (first weight > 0 .. last weight > 1e-8): mean length %.1f of 192" % float((last - first + 1).clamp_min(0).mean()), flush=True)
    w0 = m["weights0"]
    print(which, "coarse weights > 0: %.4f   > 1e-8: %.4f" % (float((w0 > 0).float().mean()), float((w0 > 1e-8).float().mean())), flush=True)
