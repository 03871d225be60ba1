"""A plain-PyTorch module with the reference IBLNeRF's parameter names, registration order and forward
arithmetic (nerf_models/ibl_nerf.py:45-72, :154-210), written for the tests: it plays the role of the
module train.py / test.py hand over as `network_fn`, and of the reference's autograd query path."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class AuxShaped(nn.Module):
    """Parameter holder with the schema of the reference's PositionMLP (src/networks/MLP.py:6-30): positions_linears.0-7 + out_linears."""

    def __init__(self, out_ch, sd=None):
        super().__init__()
        W, ch = 256, 63
        self.positions_linears = nn.ModuleList([nn.Linear(ch, W)] + [nn.Linear(W + ch if i == 4 else W, W) for i in range(7)])
        self.out_linears = nn.Linear(W, out_ch)
        if sd is not None:
            self.load_state_dict({k: torch.as_tensor(v).clone() for k, v in sd.items()})

    def forward(self, e_pts):
        h = e_pts
        for i, l in enumerate(self.positions_linears):
            h = F.relu(l(h))
            if i == 4:
                h = torch.cat([e_pts, h], -1)
        return self.out_linears(h)


class RefShaped(nn.Module):
    def __init__(self, sd=None, arch=(8, 256, 10, 4)):
        """arch = (netdepth, netwidth, multires, multires_views) (ibl_nerf.py:14-60 with skips = [4])"""
        super().__init__()
        D, W, ch, chv = arch[0], arch[1], 3 + 6 * arch[2], 3 + 6 * arch[3]
        self.positions_linears = nn.ModuleList([nn.Linear(ch, W)] + [nn.Linear(W + ch if i == 4 else W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(chv + W, W)])
        self.feature_linear = nn.Linear(W, W)
        self.sigma_linear = nn.Linear(W, 1)
        self.albedo_feature_linear = nn.Linear(W, W // 2)
        self.albedo_linear = nn.Linear(W // 2, 3)
        self.roughness_linear = nn.Linear(W, 1)
        self.irradiance_feature_linear = nn.Linear(W, W // 2)
        self.irradiance_linear = nn.Linear(W // 2, 1)
        self.radiance_linear = nn.Linear(W, 3)
        self.additional_radiance_feature_linear = nn.ModuleList([nn.Linear(W, W // 2) for _ in range(3)])
        self.additional_radiance_linear = nn.ModuleList([nn.Linear(W // 2, 3) for _ in range(3)])
        if sd is not None:
            self.load_state_dict({k: torch.as_tensor(v).clone() for k, v in sd.items()})

    def forward(self, e_pts, e_dirs=None):
        h = e_pts
        for i, l in enumerate(self.positions_linears):
            h = F.relu(l(h))
            if i == 4:
                h = torch.cat([e_pts, h], -1)
        sigma = self.sigma_linear(h)
        if e_dirs is None:
            return sigma
        albedo = self.albedo_linear(F.relu(self.albedo_feature_linear(h)))
        rough = self.roughness_linear(h)
        irr = self.irradiance_linear(F.relu(self.irradiance_feature_linear(h)))
        h2 = F.relu(self.views_linears[0](torch.cat([self.feature_linear(h), e_dirs], -1)))
        ret = [sigma, albedo, rough, irr, self.radiance_linear(h2)]
        for fl, ol in zip(self.additional_radiance_feature_linear, self.additional_radiance_linear):
            ret.append(ol(F.relu(fl(h2))))
        return torch.cat(ret, -1)


def embed(x, n_freqs):
    out = [x]
    for k in range(n_freqs):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


def torch_query(inputs, viewdirs, net):
    """run_network (ibl_nerf.py:236-252) in plain PyTorch: the autograd-carrying query path."""
    flat = inputs.reshape(-1, 3)
    e = embed(flat, 10)
    if viewdirs is None:
        return net(e).reshape(*inputs.shape[:-1], 1)
    d = viewdirs[:, None].expand(inputs.shape).reshape(-1, 3)
    return net(e, embed(d, 4)).reshape(*inputs.shape[:-1], 18)


def composite_direct(raw, z_vals, rays_d):
    """The compositing of raw2outputs (ibl_nerf_renderer.py:203-206, 241-259, 281-318) in plain torch, for the autograd comparison:
    -> (maps [n, 19] = [depth, acc, albedo3, roughness, irradiance, radiance3, radiance_1..3], weights [n, S])."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1) * torch.norm(rays_d[..., None, :], dim=-1)
    alpha = 1.0 - torch.exp(-F.relu(raw[..., 0]) * dists)
    w = alpha * torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    cols = [torch.sum(w * z_vals, -1, keepdim=True), torch.sum(w, -1, keepdim=True)]
    act = torch.sigmoid(raw[..., 1:18])                                            # sigmoid on every channel (use_radiance_linear=False)
    wd = w.detach()       # albedo, roughness, irradiance and the coarse radiances are composited with weights_detached (:246, :282-315)
    cols += [torch.sum(wd[..., None] * act[..., 0:5], -2), torch.sum(w[..., None] * act[..., 5:8], -2), torch.sum(wd[..., None] * act[..., 8:17], -2)]
    return torch.cat(cols, -1), w
