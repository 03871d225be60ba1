import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from test_gpu_parity import make_renderer
from test_gpu_launch_scale import _frame_rays, per_ray, ray_floor
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
g, sdc, sdf, _, _ = load_golden("fitted_launch64k")
out = {}
for mode in ("f16x3_mxfp6x", "f16x3_mxfp6"):
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=65536, mlp_precision=mode)
    ro, rd = _frame_rays(r)
    idx = torch.as_tensor(g["pix"], device=rd.device)
    m = r.render_rays(ro[idx].contiguous(), rd[idx].contiguous(), 0.5, 8.0)
    for k in [k[5:] for k in g.files if k.startswith("out__")]:
        out[mode + "__" + k] = per_ray(m[k].cpu().numpy(), g["out__" + k]).astype(np.float32)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "launch64k_errors.npz"), **out)
