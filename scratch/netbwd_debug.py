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
rng = np.random.RandomState(21)
pts = torch.from_numpy(rng.uniform(-1.5, 1.5, (7, 45, 3)).astype(np.float32)).cuda()
dirs = torch.from_numpy(rng.uniform(-1, 1, (7, 3)).astype(np.float32)).cuda()
draw = torch.from_numpy(rng.uniform(-1, 1, (7, 45, 18)).astype(np.float32)).cuda()
r = R.Renderer(64, 0, max_rays_per_launch=64)
r.load_weights(0, sdc)
named = dict(net.named_parameters())
for ch in list(range(18)):
    d1 = torch.zeros_like(draw); d1[..., ch] = draw[..., ch]
    net.zero_grad()
    p = pts.clone().requires_grad_(True)
    (torch_query(p, dirs, net) * d1).sum().backward()
    dp, g1 = r.network_backward(pts, dirs, d1, 0)
    bad = {}
    for k in g1:
        ref = named[k].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            if float(g1[k].abs().max()) != 0.0: bad[k] = "nonzero %.1e" % float(g1[k].abs().max())
        else:
            e = C.rel_linf(g1[k].cpu().numpy(), ref.cpu().numpy())
            if e > 1e-3: bad[k] = "%.1e" % e
    print("ch %2d dpts %.1e  bad: %s" % (ch, C.rel_linf(dp.cpu().numpy(), p.grad.cpu().numpy()), bad))
print("---- scale sweep, channel 13 and all channels")
for ch in (13, None):
    d1 = torch.zeros_like(draw)
    if ch is None: d1 = draw
    else: d1[..., ch] = draw[..., ch]
    net.zero_grad()
    (torch_query(pts, dirs, net) * d1).sum().backward()
    for sc in (2.0 ** 4, 2.0 ** 10, 2.0 ** 14, 2.0 ** 18):
        try:
            dp, g1 = r.network_backward(pts, dirs, d1, 0, grad_scale=sc)
        except FloatingPointError as e:
            print(ch, sc, "overflow"); continue
        print(ch, "scale 2^%d" % int(np.log2(sc)), {k.replace("positions_linears.", "L"): "%.1e" % C.rel_linf(g1[k].cpu().numpy(), named[k].grad.cpu().numpy()) for k in ("positions_linears.0.weight", "positions_linears.1.weight", "positions_linears.2.weight", "positions_linears.7.weight")},
              "max|dW1| %.2e" % float(named["positions_linears.1.weight"].grad.abs().max()))
print("---- determinism and comparison with the trunk-only path (channel 0)")
d1 = torch.zeros_like(draw); d1[..., 0] = draw[..., 0]
net.zero_grad(); (torch_query(pts, dirs, net) * d1).sum().backward()
outs = [r.network_backward(pts, dirs, d1, 0, grad_scale=1024.0) for _ in range(3)]
for k in ("positions_linears.0.weight", "positions_linears.1.weight", "positions_linears.1.bias", "positions_linears.2.weight"):
    a, b, c = (o[1][k].cpu().numpy() for o in outs)
    print(k, "run-to-run", np.abs(a - b).max(), np.abs(a - c).max(), " vs torch %.1e" % C.rel_linf(a, named[k].grad.cpu().numpy()))
_, _, gt = r.trunk_backward(pts.reshape(-1, 3), d1[..., 0].reshape(-1), 0, grad_scale=1024.0)
for k in ("positions_linears.0.weight", "positions_linears.1.weight", "positions_linears.1.bias"):
    print(k, "trunk-only path vs torch %.1e" % C.rel_linf(gt[k].cpu().numpy(), named[k].grad.cpu().numpy()))
# which rows of dW1 are wrong?
a = outs[0][1]["positions_linears.1.weight"].cpu().numpy(); ref = named["positions_linears.1.weight"].grad.cpu().numpy()
err = np.abs(a - ref)
print("dW1 error by row block of 32:", [float("%.1e" % err[32*t:32*t+32].max()) for t in range(8)], " by col block:", [float("%.1e" % err[:, 32*t:32*t+32].max()) for t in range(8)])
b1 = outs[0][1]["positions_linears.1.bias"].cpu().numpy(); rb = named["positions_linears.1.bias"].grad.cpu().numpy()
print("db1 error by block:", [float("%.1e" % np.abs(b1 - rb)[32*t:32*t+32].max()) for t in range(8)], "scale", np.abs(rb).max())
