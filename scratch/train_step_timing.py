"""Query pattern of one training step of the shipped config (N_rand = 512 rays, 64 + 128 samples; train.py:286-297):
gradient-carrying main queries + no-grad eps-normal / reflected queries, all on the PyTorch module (as the
reference does) vs the no-grad ones on the fused kernel (model.training_network_query_fn)."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import _pkg; _pkg.load()
from ibl_nerf_amd import checkpoint as ck, model as M
from torch_ref import RefShaped, torch_query
torch.manual_seed(0)
nets = [RefShaped(ck.synthetic_state_dict(i)).cuda() for i in (0, 1)]
opt = torch.optim.Adam([p for n in nets for p in n.parameters()], lr=5e-4)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
def step(q):
    loss = 0.0
    for net, S in ((nets[0], 64), (nets[1], 192)):
        pts = torch.rand(N, S, 3, device='cuda') * 4 - 2
        dirs = torch.rand(N, 3, device='cuda') * 2 - 1
        raw = q(pts, dirs, net)                                   # main query: carries gradients
        with torch.no_grad():
            eps = q(torch.rand(4 * N, S, 3, device='cuda') * 4 - 2, None, net)          # 4 offset copies, trunk only
            refl = q(torch.rand(N, 64, 3, device='cuda') * 4 - 2, dirs, net)            # reflected ray, 64 coarse z
        loss = loss + raw.square().mean() + 0.0 * (eps.mean() + refl.mean())
    opt.zero_grad(); loss.backward(); opt.step()
def timeit(q, n=10):
    for _ in range(3): step(q)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(q)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a = timeit(torch_query)
b = timeit(M.training_network_query_fn(torch_query))
c = timeit(M.training_network_query_fn(torch_query, fused_trunk_backward=True))
# ... and with the compositing between the raw rows and ray-sized maps in the step (losses on radiance, albedo, depth, weights)
from torch_ref import composite_direct
def step2(q, comp):
    loss = 0.0
    for net, S in ((nets[0], 64), (nets[1], 192)):
        d = torch.rand(N, 3, device='cuda') * 2 - 1
        z = torch.sort(torch.rand(N, S, device='cuda') * 4 + 0.5, -1)[0]
        pts = d[:, None] * z[..., None]
        maps, w = comp(q(pts, d, net), z, d)
        with torch.no_grad():
            eps = q(torch.rand(4 * N, S, 3, device='cuda') * 4 - 2, None, net)
            refl = q(torch.rand(N, 64, 3, device='cuda') * 4 - 2, d, net)
        loss = loss + maps[0].square().mean() + maps[1].square().mean() + 0.1 * maps[2].square().mean() + 0.01 * w.square().mean() + 0.0 * (eps.mean() + refl.mean())
    opt.zero_grad(); loss.backward(); opt.step()
def timeit2(q, comp, n=10):
    for _ in range(3): step2(q, comp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step2(q, comp)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def comp_torch(raw, z, d):
    m, w = composite_direct(raw, z, d)
    return (m[:, 7:10], m[:, 2:5], m[:, 0]), w
def comp_fused(raw, z, d):
    m, w = M.fused_composite(raw, z, d)
    return (m["radiance_map"], m["albedo_map"], m["depth_map"]), w
a2 = timeit2(torch_query, comp_torch)
c2 = timeit2(M.training_network_query_fn(torch_query, fused_trunk_backward=True), comp_fused)
print("queries of one training step: all PyTorch fp32 %.1f ms; no-grad queries on the fused kernel %.1f ms (%.2fx); "
      "+ the gradient-carrying queries fused in both directions (the whole network) %.1f ms (%.2fx)" % (a, b, a / b, c, a / c))
print("the same with the compositing and losses on ray-sized maps in the step: all PyTorch %.1f ms; network and compositing fused in both directions %.1f ms (%.2fx)" % (a2, c2, a2 / c2))
