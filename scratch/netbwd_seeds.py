"""Which point sets of tests/test_gpu_gradnormal.py::test_network_backward_every_parameter are free of ReLU pass-bit flips (the kernel is
deterministic, so a seed that passes keeps passing): worst per-tensor error of the all-channel case and of eight single channels."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest as C
import torch
from ibl_nerf_amd import renderer as R
from torch_ref import RefShaped, torch_query
g, sdc, _, _, _ = C.load_golden("plain_g10")
net = RefShaped(sdc).cuda()
r = R.Renderer(64, 0, max_rays_per_launch=64)
r.load_weights(0, sdc)
named = dict(net.named_parameters())
for seed in range(21, 33):
    rng = np.random.RandomState(seed)
    pts = torch.from_numpy(rng.uniform(-1.5, 1.5, (7, 45, 3)).astype(np.float32)).cuda()
    dirs = torch.from_numpy(rng.uniform(-1, 1, (7, 3)).astype(np.float32)).cuda()
    draw = torch.from_numpy(rng.uniform(-1, 1, (7, 45, 18)).astype(np.float32)).cuda()
    worst = []
    for ch in (None, 0, 2, 4, 5, 7, 10, 13, 16):
        d1 = draw if ch is None else torch.zeros_like(draw)
        if ch is not None: d1[..., ch] = draw[..., ch]
        net.zero_grad()
        p = pts.clone().requires_grad_(True)
        (torch_query(p, dirs, net) * d1).sum().backward()
        dp, g1 = r.network_backward(pts, dirs, d1, 0)
        e = max(C.rel_linf(g1[k].cpu().numpy(), named[k].grad.cpu().numpy()) for k in g1 if named[k].grad is not None and float(named[k].grad.abs().max()) > 0)
        worst.append(max(e, C.rel_linf(dp.cpu().numpy(), p.grad.cpu().numpy()) / 2))
    print("seed", seed, "worst per case:", " ".join("%.1e" % w for w in worst))
