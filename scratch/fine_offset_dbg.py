"""Which rule of k_select_points moves the fine normal?  (debug: IBL_DBG_MARGIN / IBL_DBG_TMIN override the constants)"""
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
sdc, sdf = ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])
def render(routing):
    r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x", query_routing=routing)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    ro, rd = r.get_rays(800, 800, K, c2w)
    m = r.render_rays(ro.reshape(-1, 3)[:262144], rd.reshape(-1, 3)[:262144], 0.5, 8.0)
    return m, r.last_selection()
m, sel = render(tuple(sys.argv[1:]))
torch.save({k: m[k].cpu() for k in ("target_normal_map", "depth_map", "z_std", "weights")}, sys.argv[0] + ".%s.pt" % os.environ.get("TAG", "x"))
print(os.environ.get("TAG"), sel)
