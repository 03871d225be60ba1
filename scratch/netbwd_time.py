"""Whole-network backward (iblnerf_network_backward): stage times on a large batch (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest as C
import torch
from ibl_nerf_amd import renderer as R, checkpoint as ck
sd = ck.blob_to_state_dict(np.load(C.GOLDEN + "/fitted_ckpt.npz")["coarse"])
r = R.Renderer(64, 0, max_rays_per_launch=64)
r.load_weights(0, sd)
N, S = 4096, 192        # four training steps of 1 024 rays
pts = (torch.rand((N, S, 3), device="cuda") * 3 - 1.5).contiguous()
dirs = (torch.rand((N, 3), device="cuda") * 2 - 1).contiguous()
draw = (torch.rand((N, S, 18), device="cuda") * 2 - 1).contiguous()
for name, fn in (("forward (full query)", lambda: r.network_query(pts, dirs, 0)), ("whole backward", lambda: r.network_backward(pts, dirs, draw, 0, grad_scale=16.0))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-24s %.2f ms per %d points (%.2f ns/point)" % (name, dt * 1e3, N * S, dt / (N * S) * 1e9))
