"""Launch time of the density-gradient query (forward + backward chain) against the trunk-only query, same points, both precise kernels."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest as C
import torch
from ibl_nerf_amd import renderer as R
from test_gpu_parity import make_renderer
g, sdc, sdf, gt, edit = C.load_golden("fitted_gradnormal")
lut = C.load_lut_rgb()
n = 64000 * 64
pts = (torch.rand((n, 3), device="cuda") * 4 - 2).contiguous()
for prec in ["f16x3", "bf16x3"]:
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=64, mlp_precision=prec)
    for name, fn in (("trunk", lambda: r.network_query(pts[None], None, 1)), ("trunk+grad", lambda: r.density_gradient(pts, 1))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("%-8s %-11s %.2f ms per %d points  (%.1f ns/point)" % (prec, name, dt * 1e3, n, dt / n * 1e9))
