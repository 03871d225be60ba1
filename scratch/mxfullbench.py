"""FULL network_query timing of the fast kernel (A/B harness: IBLNERF_LIB=scratch/lib_mx_scalarfma.so python scratch/mxfullbench.py)."""
import sys, os, torch
sys.path.insert(0, '.')
import _pkg; _pkg.load()
from ibl_nerf_amd import renderer as R, checkpoint as ck
N, S = 65536, 128
pts = torch.rand((N, S, 3), device='cuda') * 8 - 4
dirs = torch.rand((N, 3), device='cuda') * 2 - 1
r = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision="f16_mxfp6"); r.load_weights(0, ck.synthetic_state_dict(0))
out = r.network_query(pts, dirs, 0); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); r.network_query(pts, dirs, 0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[2]
print("%-30s f16_mxfp6 FULL %.2f ms  alg %.0f TFLOP/s (frac %.3f)  checksum %.6f [%s]" % (os.environ.get("IBLNERF_LIB", "product"), ms, N*S*1591552/ms/1e9, N*S*1591552/ms/1e9/2500, float(out.double().abs().mean()), " ".join("%.1f" % t for t in ts)), flush=True)
