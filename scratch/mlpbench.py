import sys, os, numpy as np, torch, time
sys.path.insert(0, '.')
import _pkg; pkg = _pkg.load()
from ibl_nerf_amd import renderer as R, checkpoint as ck
sd = ck.synthetic_state_dict(0)
r = R.Renderer(64, 0, max_rays_per_launch=64)
r.load_weights(0, sd)
N, S = 65536, 128       # 8.4M points = 256 groups per CU... (65536*128/128/256 = 256 groups per WG)
pts = torch.rand((N, S, 3), device='cuda') * 8 - 4
dirs = torch.rand((N, 3), device='cuda') * 2 - 1
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms_t = t(lambda: r.network_query(pts, None, 0))
ms_f = t(lambda: r.network_query(pts, dirs, 0))
P = N * S
print("lib=%s" % os.environ.get("IBLNERF_LIB", "default"))
print("TRUNK %.2f ms  %.1f Mpts/s  alg %.0f TFLOP/s  (mfma-issue %.0f TF, %.1f%% of 2500)" % (ms_t, P/ms_t/1e3, P*982528/ms_t/1e9, P*60*48*32768/32/ms_t/1e9, P*60*48*32768/32/ms_t/1e9/25))
print("FULL  %.2f ms  %.1f Mpts/s  alg %.0f TFLOP/s  (mfma-issue %.0f TF, %.1f%% of 2500)" % (ms_f, P/ms_f/1e3, P*1591552/ms_f/1e9, P*(97*48-48)*32768/32/ms_f/1e9, P*(97*48-48)*32768/32/ms_f/1e9/25))
