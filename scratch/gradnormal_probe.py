"""Where does the gradient-normal error on a fog fixture come from?  GPU density gradient -> oracle normal (teacher forcing) for the
coarse pass of gradnormal_g10 / graddir_g10, per precision."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest as C
import iblnerf_oracle as O
from test_gpu_parity import make_renderer
import torch
from ibl_nerf_amd import renderer as R
lut = C.load_lut_rgb()
for name in ["gradnormal_g10", "graddir_g10"]:
    g, sdc, sdf, gt, edit = C.load_golden(name)
    o, d = g["rays_o"], g["rays_d"]
    z = O.coarse_z(float(g["near"]), float(g["far"]), 64, o.shape[0])
    pts = (o[:, None, :] + d[:, None, :] * z[:, :, None]).astype(np.float32)
    so, go = O.density_gradient(sdc, pts.reshape(-1, 3))
    direction = name == "graddir_g10"
    n_ref = g["normal_raw_c"]
    for prec in ["f16x3", "bf16x3"]:
        r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=4096, mlp_precision=prec)
        s, gr = r.density_gradient(pts, 0)
        s, gr = s.cpu().numpy(), gr.cpu().numpy()
        print(name, prec, "sigma err", np.abs(s.reshape(-1) - so).max(), "grad err rel", np.abs(gr.reshape(-1, 3) - go).max() / np.abs(go).max())
        n1 = O.normal_from_depth_gradient(sdc, o, d, z, direction=direction, sigma_grad=(s, gr))
        print("    GPU grads -> oracle normal vs ref:", C.rel_linf(n1, n_ref))
        res = r.render_rays(o, d, float(g["near"]), float(g["far"]), gt, **edit)
        n2 = res["target_normal_map0"].cpu().numpy()
        print("    GPU render coarse normal vs ref:", C.rel_linf(n2, g["out__target_normal_map0"]), " vs teacher-forced oracle:", C.rel_linf(n2, n1))
        worst = np.abs(n2 - g["out__target_normal_map0"]).max(-1)
        print("    worst rays", np.argsort(worst)[-3:], worst[np.argsort(worst)[-3:]])
