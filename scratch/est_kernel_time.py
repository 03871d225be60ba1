"""One default frame under rocprofv3 --kernel-trace --stats: per-kernel time of the estimate (mxk16 TRUNK) and list kernels."""
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
f = np.load(os.path.join(ROOT, "tests", "golden", "fitted_ckpt.npz"))
r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x")
r.load_weights(0, ck.blob_to_state_dict(f["coarse"])); r.load_weights(1, ck.blob_to_state_dict(f["fine"])); r.load_lut(lut)
ro, rd = r.get_rays(800, 800, K, c2w)
for _ in range(2):
    r.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), 0.5, 8.0)
torch.cuda.synchronize()
