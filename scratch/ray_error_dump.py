"""Per-ray errors of the HIP path against a launch-scale fixture, per mode, dumped for CPU-side analysis (gpurun_out/ray_errors_<fixture>.npz).
    python scratch/ray_error_dump.py <fixture> [...]"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_golden, load_lut_rgb
from test_gpu_parity import make_renderer
from test_gpu_launch_scale import _frame_rays, per_ray
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
for name in sys.argv[1:]:
    g, sdc, sdf, gt, edit = load_golden(name)
    out = {}
    for mode in ("f16x3_mxfp6x", "f16x3"):
        r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=65536, mlp_precision=mode)
        if "rays_o" in g.files:
            ro, rd = torch.from_numpy(g["rays_o"]).cuda(), torch.from_numpy(g["rays_d"]).cuda()
        else:
            fo, fd = _frame_rays(r)
            idx = torch.as_tensor(g["pix"], device=fd.device)
            ro, rd = fo[idx].contiguous(), fd[idx].contiguous()
        m = r.render_rays(ro, rd, 0.5, 8.0, gt, **edit)
        for k in ("depth_map", "target_normal_map", "albedo_map", "depth_map0", "target_normal_map0"):
            if "out__" + k in g.files:
                out[mode + "__" + k] = per_ray(m[k].cpu().numpy(), g["out__" + k]).astype(np.float32)
                out[mode + "__val__" + k] = m[k].cpu().numpy()
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "ray_errors_%s.npz" % name), **out)
    print(name, {k: float(v.max()) for k, v in out.items() if "__val__" not in k})
