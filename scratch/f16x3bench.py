"""TRUNK network_query timing of one precision mode (A/B harness: IBLNERF_LIB=scratch/lib_f16x3_<tag>.so python scratch/f16x3bench.py)."""
import sys, os, torch
sys.path.insert(0, '.')
import _pkg; _pkg.load()
from ibl_nerf_amd import renderer as R, checkpoint as ck
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
N, S = 65536, 256
pts = torch.rand((N, S, 3), device='cuda') * 8 - 4
r = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision=prec); r.load_weights(0, ck.synthetic_state_dict(0))
r.network_query(pts, None, 0); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); r.network_query(pts, None, 0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[2]
print("%-28s %-8s TRUNK %.2f ms  alg %.0f TFLOP/s (frac %.3f)  [%s]" % (os.environ.get("IBLNERF_LIB", "product"), prec, ms, N*S*982528/ms/1e9, N*S*982528/ms/1e9/2500, " ".join("%.1f" % t for t in ts)), flush=True)
