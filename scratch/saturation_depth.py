"""Where along the sample index do rays saturate?  (the question behind skipping the ESTIMATES of samples behind saturation by estimating in z-chunks)"""
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
    r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x")
    r.load_weights(0, ck.blob_to_state_dict(f["coarse"])); r.load_weights(1, ck.blob_to_state_dict(f["fine"])); r.load_lut(lut)
    ro, rd = r.get_rays(800, 800, K, c2w)
    idx = torch.randperm(640000, device=ro.device)[:65536]
    m = r.render_rays(ro.reshape(-1, 3)[idx].contiguous(), rd.reshape(-1, 3)[idx].contiguous(), 0.5, 8.0)
    for name, w in (("fine", m["weights"].double()), ("coarse", m["weights0"].double())):
        S = w.shape[1]
        cum = w.cumsum(1)
        sat = cum >= 1 - 1e-6                                  # transmittance behind sample k below 1e-6 (as far as float32 weights can tell)
        first = torch.where(sat.any(1), sat.float().argmax(1), torch.full((w.shape[0],), S, device=w.device))
        print(which, name, "rays that saturate at all: %.3f; samples behind saturation: %.3f of all; by quarter of the index range: %s" % (
            float(sat.any(1).float().mean()), float((S - first.clamp(max=S)).float().mean() / S),
            ["%.2f" % float((first <= q * S // 4).float().mean()) for q in (1, 2, 3)]), flush=True)
