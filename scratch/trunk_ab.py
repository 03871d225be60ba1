"""TRUNK query timing of the default mode's two offset-query kernels for one library build: python scratch/trunk_ab.py [lib.so]"""
import sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); import _pkg; _pkg.load()
from ibl_nerf_amd import binding as B, checkpoint as ck, renderer as R
lib = sys.argv[1] if len(sys.argv) > 1 else B.LIB_PATH
B.load_library(lib)
f = np.load(os.path.join(ROOT, "tests", "golden", "fitted_ckpt.npz"))
sd = ck.blob_to_state_dict(f["fine"])
N, S = 65536, 256
pts = (torch.rand((N, S, 3), device="cuda") * 3 - 1.5).contiguous()
for name, kw in (("mixed TRUNK (mxk<5>)", dict(mlp_precision="f16x3_mxfp6x", query_routing="user_trunk_mixed")), ("f16x3 TRUNK", dict(mlp_precision="f16x3")),
                 ("fast TRUNK (mxk<1>)", dict(mlp_precision="f16_mxfp6"))):
    r = R.Renderer(64, 128, max_rays_per_launch=64, **kw)
    r.load_weights(0, sd); r.load_weights(1, sd)
    r.network_query(pts, None, 1); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r.network_query(pts, None, 1); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print("%-28s %-22s %.2f ms (min %.2f) [%s]" % (os.path.basename(lib), name, np.median(ts), min(ts), " ".join("%.1f" % t for t in ts)), flush=True)
