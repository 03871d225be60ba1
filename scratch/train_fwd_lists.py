"""How much would a training step's forward gain from lists?  4 096 rays of the fitted checkpoint: the tapped every-sample forward (what a step runs today) against the
eager call under a route (lists), GPU time by events; and the step's host time (is the step host-bound?)."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as Bn
import _pkg
pkg = _pkg.load()
from ibl_nerf_amd import renderer as R
import train_loss as TL
torch.cuda.set_device(0)
sdc, sdf = Bn.load_checkpoint("fitted")
lut = Bn.load_lut()
K, _ = Bn.camera(); fl = float(K[0, 0]); H, W = Bn.H, Bn.W
n = 4096
rng = np.random.RandomState(0)
pix = rng.permutation(H * W)[:n]
i, j = (pix % W).astype(np.float32), (pix // W).astype(np.float32)
d = np.stack([(i - W / 2) / fl, -(j - H / 2) / fl, -np.ones_like(i)], -1).astype(np.float32)
ro, rd = torch.zeros((n, 3), device="cuda"), torch.from_numpy(d).cuda()

def ev(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(reps): fn()
    b.record(); host = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, 1e3 * host

def mk(**kw):
    rr = R.Renderer(64, 128, max_rays_per_launch=4096, **kw)
    rr.load_weights(0, sdc); rr.load_weights(1, sdf); rr.load_lut(lut)
    return rr
r = mk()
print("eager, per-call decisions (lists):      gpu %.2f ms, host issue %.2f ms" % ev(lambda: r.render_rays(ro, rd, Bn.NEAR, Bn.FAR)))
r.decide_route(ro, rd, Bn.NEAR, Bn.FAR)
print("eager, imposed route (lists):           gpu %.2f ms, host issue %.2f ms" % ev(lambda: r.render_rays(ro, rd, Bn.NEAR, Bn.FAR)), r.route)
t_rand, u = torch.rand((n, 64), device="cuda"), torch.rand((n, 128), device="cuda")
print("sampled, imposed route (lists, whole repeat on trips): gpu %.2f ms, host %.2f ms" % ev(lambda: r.render_rays(ro, rd, Bn.NEAR, Bn.FAR, draws=(t_rand, u))))
rl = mk(range_check="lazy")
from ibl_nerf_amd import binding as B
def tapped(rr):
    taps = B.Taps()
    e = lambda *s: torch.empty(s, device="cuda")
    sv = dict(zc=e(n, 64), zf=e(n, 192), rawc=e(n, 64, 18), rawf=e(n, 192, 18), envc=e(n, 4, 3), envf=e(n, 4, 3))
    taps.d_z_coarse, taps.d_z_fine, taps.d_raw_coarse, taps.d_raw_fine = (sv[k].data_ptr() for k in ("zc", "zf", "rawc", "rawf"))
    taps.d_env_coarse, taps.d_env_fine = sv["envc"].data_ptr(), sv["envf"].data_ptr()
    return rr.render_rays(ro, rd, Bn.NEAR, Bn.FAR, draws=(t_rand, u), taps=taps), sv
print("tapped, lazy, every sample (today):     gpu %.2f ms, host issue %.2f ms" % ev(lambda: tapped(rl)))
print("tapped, eager ctx, imposed route (offsets + reflected on lists): gpu %.2f ms, host %.2f ms" % ev(lambda: tapped(r)))
