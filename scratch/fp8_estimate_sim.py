"""Feasibility study (CPU, torch): what would a density estimate on fp8 / fp6 matrix products look like?
Trunk of the fitted checkpoints in fp32 vs the same with both operands of every layer rounded to
e4m3 (MX block scale per 32 along K) / e2m3 / e3m2; on the samples of a fixture's rays (main ray, coarse grid + a dense grid)."""
import sys, importlib.util, numpy as np, torch
sys.path.insert(0, 'tests')
spec = importlib.util.spec_from_file_location('ck', 'ibl-nerf_amd/checkpoint.py')
ck = importlib.util.module_from_spec(spec); spec.loader.exec_module(ck)
from torch_ref import RefShaped, embed

torch.set_num_threads(8)

def q_block(x, fmt):
    """round x [..., K] to fmt with one power-of-two scale per block of 32 along K (MX)"""
    K = x.shape[-1]
    pad = (-K) % 32
    xp = torch.nn.functional.pad(x, (0, pad))
    b = xp.reshape(*xp.shape[:-1], -1, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    emax = {'e4m3': 8, 'e2m3': 2, 'e3m2': 4}[fmt]          # largest binade exponent of the format
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    y = b / scale
    if fmt == 'e4m3':
        y = y.clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32)
    else:
        mant, emin, vmax = {'e2m3': (3, 0, 7.5), 'e3m2': (2, -2, 28.0)}[fmt]
        y = y.clamp(-vmax, vmax)
        e = torch.floor(torch.log2(y.abs().clamp_min(1e-30))).clamp_min(emin)
        step = torch.exp2(e - mant)
        y = torch.round(y / step) * step
    return (y * scale).reshape(*xp.shape)[..., :K]

def trunk(net, e_pts, fmt=None, f16=False):
    h = e_pts
    for i, l in enumerate(net.positions_linears):
        w, x = l.weight, h
        if fmt: w, x = q_block(w, fmt), q_block(x, fmt)
        elif f16: w, x = w.half().float(), x.half().float()
        h = torch.relu(x @ w.t() + l.bias)
        if i == 4: h = torch.cat([e_pts, h], -1)
    w, x = net.sigma_linear.weight, h
    if fmt: w, x = q_block(w, fmt), q_block(x, fmt)
    elif f16: w, x = w.half().float(), x.half().float()
    return (x @ w.t() + net.sigma_linear.bias)[..., 0]

for ckname, fx in (('fitted_ckpt', 'fitted_posed4k'), ('fitted2_ckpt', 'fitted2_posed4k'), ('fitted3_ckpt', 'fitted3_posed4k')):
    c = np.load(f'tests/golden/{ckname}.npz'); g = np.load(f'tests/golden/{fx}.npz')
    ro, rd = torch.tensor(g['rays_o']), torch.tensor(g['rays_d'])
    if ro.ndim == 1: ro = ro.expand_as(rd)
    near, far = float(g['near']), float(g['far'])
    idx = torch.arange(0, rd.shape[0], 8)
    ro, rd = ro[idx], rd[idx]
    for which in ('fine',):
        net = RefShaped({k: v for k, v in ck.blob_to_state_dict(c[which]).items()})
        z = torch.linspace(near, far, 192)
        pts = ro[:, None, :] + rd[:, None, :] * z[None, :, None]
        e = embed(pts.reshape(-1, 3), 10)
        with torch.no_grad():
            ref = trunk(net, e)
            print(ckname, which, 'points', ref.numel(), 'raw density quantiles', np.quantile(ref.numpy(), [0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99]).round(2))
            empt = ref < -0.5
            print('   share clearly empty (< -0.5) %.3f' % empt.float().mean().item())
            for name, kw in (('f16', dict(f16=True)), ('e4m3', dict(fmt='e4m3')), ('e2m3', dict(fmt='e2m3')), ('e3m2', dict(fmt='e3m2'))):
                est = trunk(net, e, **kw)
                err = (est - ref)
                a = err.abs().numpy()
                # margin needed: the largest over-/under-estimate near zero (|ref| < 20)
                zone = (ref.abs() < 20).numpy()
                print(f'   {name}: |err| p50 {np.quantile(a,0.5):.3g} p99 {np.quantile(a,0.99):.3g} p99.99 {np.quantile(a,0.9999):.3g} max {a.max():.3g}; in |ref|<20: max {a[zone].max():.3g} p99.9 {np.quantile(a[zone],0.999):.3g}')
                for M in (2, 4, 8, 16):
                    sure = (est < -M)
                    print(f'        M={M}: judged empty on this estimate {sure.float().mean().item():.3f} of all = {(sure & empt).float().sum().item() / max(1, empt.float().sum().item()):.3f} of the empty; wrong (ref > -0.5) {(sure & ~empt).sum().item()}')
