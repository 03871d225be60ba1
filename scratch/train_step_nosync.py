"""Upper bound of what the per-call range checks cost the fused training path: scratch/train_step_timing.py's fused step with
Renderer.out_of_range stubbed out (no device synchronisation per call)."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import _pkg; _pkg.load()
from ibl_nerf_amd import checkpoint as ck, model as M, renderer as R
from torch_ref import RefShaped, torch_query
torch.manual_seed(0)
nets = [RefShaped(ck.synthetic_state_dict(i)).cuda() for i in (0, 1)]
opt = torch.optim.Adam([p for n in nets for p in n.parameters()], lr=5e-4)
N = 512
def step(q):
    loss = 0.0
    for net, S in ((nets[0], 64), (nets[1], 192)):
        pts = torch.rand(N, S, 3, device='cuda') * 4 - 2
        dirs = torch.rand(N, 3, device='cuda') * 2 - 1
        raw = q(pts, dirs, net)
        with torch.no_grad():
            eps = q(torch.rand(4 * N, S, 3, device='cuda') * 4 - 2, None, net)
            refl = q(torch.rand(N, 64, 3, device='cuda') * 4 - 2, dirs, net)
        loss = loss + raw.square().mean() + 0.0 * (eps.mean() + refl.mean())
    opt.zero_grad(); loss.backward(); opt.step()
def timeit(q, n=10):
    for _ in range(3): step(q)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(q)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
q = M.training_network_query_fn(torch_query, fused_trunk_backward=True)
a = timeit(q)
R.Renderer.out_of_range = lambda self: False
_tb = R.Renderer.trunk_backward
R.Renderer.trunk_backward = lambda self, pts, ds, which=0, grad_scale=None, features=False: _tb(self, pts, ds, which, 16.0, features)
b = timeit(q)
print("fused training step: %.1f ms; without the per-call range checks and the max|upstream| read-back: %.1f ms" % (a, b))
