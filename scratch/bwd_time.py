"""Trunk backward (iblnerf_trunk_backward): error per parameter tensor against the reference's autograd fixture, and the time of its
stages on a large batch (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest as C
import torch
from ibl_nerf_amd import renderer as R, checkpoint as ck
g = np.load(C.GOLDEN + "/trunk_backward.npz")
for tag in ("g10", "fit"):
    sd = ck.synthetic_state_dict(60, 1.0) if tag == "g10" else ck.blob_to_state_dict(np.load(C.GOLDEN + "/fitted_ckpt.npz")["coarse"])
    r = R.Renderer(64, 0, max_rays_per_launch=64)
    r.load_weights(0, sd)
    s, dp, grads = r.trunk_backward(g[tag + "__pts"], g[tag + "__dsigma"], 0)
    print(tag, "scale 2^%d" % int(np.log2(r.last_grad_scale)), " ".join("%s %.1e" % (k.replace("positions_linears.", "L").replace("weight", "w").replace("bias", "b"), C.rel_linf(v.cpu().numpy(), g[tag + "__grad__" + k])) for k, v in grads.items()))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024 * 192 * 8
pts = (torch.rand((n, 3), device="cuda") * 3 - 1.5).contiguous()
c = (torch.rand((n,), device="cuda") * 2 - 1).contiguous()
for name, fn in (("forward (trunk query)", lambda: r.network_query(pts[None], None, 0)), ("forward + dgrad", lambda: r.density_gradient(pts, 0)),
                 ("forward + dgrad + stash + wgrad", lambda: r.trunk_backward(pts, c, 0, grad_scale=16.0))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-34s %.2f ms per %d points (%.2f ns/point)" % (name, dt * 1e3, n, dt / n * 1e9))
