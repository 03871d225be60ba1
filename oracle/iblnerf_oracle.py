"""CPU ORACLE (test infrastructure, NOT the product) — numpy fp32 restatement of IBL-NeRF's
forward/inference hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / reported baseline.  The shipped path is the HIP library behind
include/iblnerf.h; nothing under `ibl-nerf_amd/` imports `oracle/`.

Parity status: PINNED.  The reference ships no golden vectors, so the pins are
outputs of the reference itself, run in the build container by tests/golden/make_golden.py and
committed as tests/golden/*.npz; tests/test_oracle_golden.py checks every function below against
them.

Every function cites the reference lines it restates (paths relative to /root/reference/src).
All arithmetic is float32 unless the reference's CPU kernels accumulate wider (cumsum/cumprod:
ATen's CPU scan kernels accumulate float tensors in double and round each prefix to float).
Operation ORDER is kept where rounding matters downstream (positions are multiplied by up to 2^9
inside the positional encoding): `o + d*z` is a rounded multiply followed by a rounded add.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
GAMMA = 2.2          # nerf_models/ibl_nerf_renderer.py:22
EPS_SRGB = 1e-12     # nerf_models/ibl_nerf_renderer.py:23


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------
def torch_linspace(start, end, steps):
    """float32 `torch.linspace` as ATen's CPU kernel computes it: symmetric fill from both ends,
    `start + step*i` for i < steps/2 and `end - step*(steps-1-i)` above, each evaluated as ONE fused
    multiply-add (verified bit-for-bit against torch.linspace in tests).  The fma is emulated in
    float64, which holds the 24-bit x small-int product and the sum exactly before the one rounding."""
    start, end = F32(start), F32(end)
    if steps == 1:
        return np.array([start], dtype=F32)
    step = np.float64(F32((end - start) / F32(steps - 1)))
    idx = np.arange(steps)
    lo = (np.float64(start) + step * idx).astype(F32)
    hi = (np.float64(end) - step * (steps - 1 - idx)).astype(F32)
    return np.where(idx < steps // 2, lo, hi).astype(F32)


def sigmoid(x):
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def relu(x):
    return np.maximum(x, F32(0))


def rgb_to_srgb(x):
    """nerf_models/ibl_nerf_renderer.py:26-27."""
    return np.power(x + F32(EPS_SRGB), F32(1.0 / GAMMA), dtype=F32)


def normalize(v, eps=1e-12):
    """torch.nn.functional.normalize(dim=-1)."""
    n = np.sqrt(np.sum(v * v, -1, keepdims=True, dtype=F32), dtype=F32)
    return (v / np.maximum(n, F32(eps))).astype(F32)


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1).astype(F32)


# --------------------------------------------------------------------------------------------
# A.1 rays — nerf_models/nerf_renderer_helper.py:36-45
# --------------------------------------------------------------------------------------------
def get_rays(H, W, K, c2w):
    K = np.asarray(K, dtype=F32)
    c2w = np.asarray(c2w, dtype=F32)
    i = np.broadcast_to(torch_linspace(0, W - 1, W)[None, :], (H, W))   # column index
    j = np.broadcast_to(torch_linspace(0, H - 1, H)[:, None], (H, W))   # row index
    dirs = np.stack([(i - K[0, 2]) / K[0, 0], -(j - K[1, 2]) / K[1, 1], -np.ones_like(i)], -1).astype(F32)
    rays_d = np.sum(dirs[..., None, :] * c2w[:3, :3], -1, dtype=F32)
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape).astype(F32)
    return rays_o, rays_d


# --------------------------------------------------------------------------------------------
# A.3 positional encoding — nerf_models/positional_embedder.py:4-52
# --------------------------------------------------------------------------------------------
def embed(x, n_freqs):
    x = np.asarray(x, dtype=F32)
    out = [x]
    for k in range(n_freqs):
        f = F32(2.0 ** k)           # 2.**linspace(0, L-1, L): exact powers of two
        out.append(np.sin(x * f, dtype=F32))
        out.append(np.cos(x * f, dtype=F32))
    return np.concatenate(out, -1).astype(F32)


# --------------------------------------------------------------------------------------------
# A.4 MLP — nerf_models/ibl_nerf.py:154-210 (forward_not_freezed)
# --------------------------------------------------------------------------------------------
def _lin(sd, name, x):
    return (x @ sd[name + ".weight"].T + sd[name + ".bias"]).astype(F32)


COLOR_INDEPENDENT = False   # is_color_independent_to_direction (ibl_nerf.py:192): tests set it per fixture


def mlp_forward(sd, e_pts, e_dirs=None, color_independent=None):
    """e_pts [P,63]; e_dirs [P,27] or None.  Returns [P,18], or [P,1] (sigma) if e_dirs is None.
    Any IBLNeRF(D, W, input_ch, input_ch_views, skips=[4]) (ibl_nerf.py:14-60): the depth is read off the state dict, the widths off its shapes."""
    if color_independent is None:
        color_independent = COLOR_INDEPENDENT
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("positions_linears.") and k.endswith(".weight"))
    h = e_pts
    for i in range(depth):
        h = relu(_lin(sd, "positions_linears.%d" % i, h))
        if i == 4 and depth > 5:
            h = np.concatenate([e_pts, h], -1)                         # :168 skip order [x63, h] (skips = [4], ibl_nerf.py:265; for D = 5 the reference's own forward fails)
    sigma = _lin(sd, "sigma_linear", h)
    if e_dirs is None:
        return sigma                                                   # :175-176 early return
    albedo = _lin(sd, "albedo_linear", relu(_lin(sd, "albedo_feature_linear", h)))
    rough = _lin(sd, "roughness_linear", h)
    irr = _lin(sd, "irradiance_linear", relu(_lin(sd, "irradiance_feature_linear", h)))
    if color_independent:                                              # :192: the radiance heads read the trunk output
        h2 = h
    else:
        feat = _lin(sd, "feature_linear", h)                           # no activation (:193)
        h2 = relu(_lin(sd, "views_linears.0", np.concatenate([feat, e_dirs], -1)))   # :194-197
    ret = [sigma, albedo, rough, irr, _lin(sd, "radiance_linear", h2)]
    for k in range(3):                                                 # taken from h2 (:202-206)
        f = relu(_lin(sd, "additional_radiance_feature_linear.%d" % k, h2))
        ret.append(_lin(sd, "additional_radiance_linear.%d" % k, f))
    return np.concatenate(ret, -1).astype(F32)


def network_query(sd, pts, viewdirs):
    """`network_query_fn` = run_network, nerf_models/ibl_nerf.py:236-252.
    pts [N,S,3]; viewdirs [N,3] (expanded over the S samples) or None."""
    pts = np.asarray(pts, dtype=F32)
    N, S, _ = pts.shape
    n_freq = (sd["positions_linears.0.weight"].shape[1] - 3) // 6                                 # multires (10 in every shipped config)
    e = embed(pts.reshape(-1, 3), n_freq)
    if viewdirs is None:
        return mlp_forward(sd, e).reshape(N, S, 1)
    n_freq_views = (sd["views_linears.0.weight"].shape[1] - sd["views_linears.0.weight"].shape[0] - 3) // 6     # multires_views (4)
    d = np.broadcast_to(np.asarray(viewdirs, dtype=F32)[:, None, :], pts.shape).reshape(-1, 3)
    return mlp_forward(sd, e, embed(d, n_freq_views)).reshape(N, S, 18)


def position_mlp_query(aux_sd, pts):
    """network_query_fn(pts, None, <PositionMLP>) — src/networks/MLP.py:19-30: the trunk (ReLU after every layer, skip
    after layer 4) then out_linears, no activation.  pts [N,S,3] -> [N,S,out_ch]."""
    pts = np.asarray(pts, dtype=F32)
    N, S, _ = pts.shape
    depth = 1 + max(int(k.split(".")[1]) for k in aux_sd if k.startswith("positions_linears.") and k.endswith(".weight"))      # (any D / W / multires: read off the state dict)
    e = embed(pts.reshape(-1, 3), (aux_sd["positions_linears.0.weight"].shape[1] - 3) // 6)
    h = e
    for i in range(depth):
        h = relu(_lin(aux_sd, "positions_linears.%d" % i, h))
        if i == 4 and depth > 5:
            h = np.concatenate([e, h], -1)
    return _lin(aux_sd, "out_linears", h).reshape(N, S, -1).astype(F32)


def position_direction_mlp_query(sd, pts, viewdirs):
    """network_query_fn(inputs, viewdirs, depth_mlp) for a PositionDirectionMLP (src/networks/MLP.py:32-74; run_network
    ibl_nerf.py:236-252): pts [N,S,3], viewdirs [N,3] -> [N,S,out_ch].  Trunk as IBLNeRF's, feature_linear without activation,
    cat([feature, embedded dirs]) through four ReLU view layers of width 128, final_linear."""
    N, S = pts.shape[:2]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("positions_linears.") and k.endswith(".weight"))
    n_view = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("views_linears.") and k.endswith(".weight"))
    width = sd["positions_linears.0.weight"].shape[0]
    e_p = embed(pts.reshape(-1, 3), (sd["positions_linears.0.weight"].shape[1] - 3) // 6)
    e_d = embed(np.repeat(viewdirs[:, None, :], S, 1).reshape(-1, 3), (sd["views_linears.0.weight"].shape[1] - width - 3) // 6)
    h = e_p
    for i in range(depth):
        h = relu(_lin(sd, "positions_linears.%d" % i, h))
        if i == 4 and depth > 5:
            h = np.concatenate([e_p, h], -1)
    h = np.concatenate([_lin(sd, "feature_linear", h), e_d], -1)
    for i in range(n_view):
        h = relu(_lin(sd, "views_linears.%d" % i, h))
    return _lin(sd, "final_linear", h).reshape(N, S, -1)


# --------------------------------------------------------------------------------------------
def density_gradient(sd, pts):
    """What autograd returns for d raw[..., 0] / d pts through run_network with viewdirs=None (ibl_nerf.py:236-252 -> positional_embedder.py:4-52
    -> ibl_nerf.py:154-176), written out: the trunk forward keeping every ReLU's pass mask, then the chain
    dZ(l-1) = (dZ(l) W(l)) * [Z(l-1) > 0], the skip layer's encoding columns, and d sin(f x) = f cos(f x), d cos(f x) = -f sin(f x).
    pts [P,3] -> (sigma [P], grad [P,3]), float32 like the reference's backward."""
    pts = np.asarray(pts, dtype=F32).reshape(-1, 3)
    e = embed(pts, 10)
    h, masks = e, []
    for i in range(8):
        z = _lin(sd, "positions_linears.%d" % i, h)
        masks.append(z > 0)
        h = relu(z)
        if i == 4:
            h = np.concatenate([e, h], -1)
    sigma = _lin(sd, "sigma_linear", h)[:, 0]
    g = np.broadcast_to(sd["sigma_linear.weight"].astype(F32), (pts.shape[0], 256)) * masks[7]     # dZ(7)
    g_enc = np.zeros_like(e)
    for i in range(7, 0, -1):
        g = (g.astype(F32) @ sd["positions_linears.%d.weight" % i]).astype(F32)                      # d / d input of layer i
        if i == 5:                                                                                    # input = [x63, h4] (:168)
            g_enc = g_enc + g[:, :63]
            g = g[:, 63:]
        g = g * masks[i - 1]
    g_enc = (g_enc + (g.astype(F32) @ sd["positions_linears.0.weight"]).astype(F32)).astype(F32)
    grad = g_enc[:, 0:3].copy()
    for k in range(10):
        f = F32(2.0 ** k)
        s_, c_ = e[:, 3 + 6 * k:6 + 6 * k], e[:, 6 + 6 * k:9 + 6 * k]
        grad = grad + f * (g_enc[:, 3 + 6 * k:6 + 6 * k] * c_ - g_enc[:, 6 + 6 * k:9 + 6 * k] * s_)
    return sigma.astype(F32), grad.astype(F32)


def trunk_backward(sd, pts, dsigma):
    """What loss.backward() leaves in .grad of the trunk's parameters for L with dL/d sigma_p = dsigma_p on the trunk-only query
    (train.py:479-481 through ibl_nerf.py:236-252, 154-176), written out: dW(l) = dZ(l)^T X(l-1), db(l) = sum_p dZ(l).
    Returns (sigma [P], dL/dpts [P,3], {name: gradient})."""
    pts = np.asarray(pts, dtype=F32).reshape(-1, 3)
    up = np.asarray(dsigma, dtype=np.float64).reshape(-1, 1)
    e = embed(pts, 10)
    h, masks, ins = e, [], []
    for i in range(8):
        ins.append(h)
        z = _lin(sd, "positions_linears.%d" % i, h)
        masks.append(z > 0)
        h = relu(z)
        if i == 4:
            h = np.concatenate([e, h], -1)
    sigma = _lin(sd, "sigma_linear", h)[:, 0]
    grads = {"sigma_linear.weight": (up * h).sum(0, keepdims=True).astype(F32), "sigma_linear.bias": up.sum(0).astype(F32)}
    g = up * sd["sigma_linear.weight"].astype(np.float64) * masks[7]
    g_enc = np.zeros(e.shape)
    for i in range(7, -1, -1):
        grads["positions_linears.%d.weight" % i] = (g.T @ ins[i].astype(np.float64)).astype(F32)
        grads["positions_linears.%d.bias" % i] = g.sum(0).astype(F32)
        gi = g @ sd["positions_linears.%d.weight" % i].astype(np.float64)
        if i == 5:
            g_enc = g_enc + gi[:, :63]
            gi = gi[:, 63:]
        if i == 0:
            g_enc = g_enc + gi
        else:
            g = gi * masks[i - 1]
    grad = g_enc[:, 0:3].copy()
    for k in range(10):
        f = 2.0 ** k
        s_, c_ = e[:, 3 + 6 * k:6 + 6 * k], e[:, 6 + 6 * k:9 + 6 * k]
        grad = grad + f * (g_enc[:, 3 + 6 * k:6 + 6 * k] * c_ - g_enc[:, 6 + 6 * k:9 + 6 * k] * s_)
    return sigma.astype(F32), grad.astype(F32), grads


def network_backward(sd, pts, viewdirs, draw):
    """What loss.backward() leaves in .grad of EVERY parameter for L with dL/d raw = draw on the full query network_query_fn(pts, viewdirs, fn)
    (train.py:479-481 through ibl_nerf.py:236-252, 154-210), written out layer by layer.  pts [N,S,3], viewdirs [N,3], draw [N,S,18]
    -> (dL/dpts [N,S,3], {name: gradient}).  float64 inside."""
    pts = np.asarray(pts, dtype=F32)
    N, S, _ = pts.shape
    d = np.asarray(draw, dtype=np.float64).reshape(N * S, 18)
    e = embed(pts.reshape(-1, 3), 10)
    ed = embed(np.broadcast_to(np.asarray(viewdirs, dtype=F32)[:, None, :], pts.shape).reshape(-1, 3), 4)
    W = lambda n: sd[n + ".weight"].astype(np.float64)
    grads = {}

    def lin_bwd(name, x, dz):                         # dz = dL/d output of the layer (after its activation's mask) -> dL/d input
        grads[name + ".weight"] = (dz.T @ x.astype(np.float64)).astype(F32)
        grads[name + ".bias"] = dz.sum(0).astype(F32)
        return dz @ W(name)

    # forward, keeping inputs and masks
    h, ins, masks = e, [], []
    for i in range(8):
        ins.append(h)
        z = _lin(sd, "positions_linears.%d" % i, h)
        masks.append(z > 0)
        h = relu(z)
        if i == 4:
            h = np.concatenate([e, h], -1)
    h7 = h
    fa, fi = _lin(sd, "albedo_feature_linear", h7), _lin(sd, "irradiance_feature_linear", h7)
    feat = _lin(sd, "feature_linear", h7)
    xin = np.concatenate([feat, ed], -1)
    zv = _lin(sd, "views_linears.0", xin)
    h2 = relu(zv)
    fk = [_lin(sd, "additional_radiance_feature_linear.%d" % k, h2) for k in range(3)]
    # backward of the heads
    dh2 = lin_bwd("radiance_linear", h2, d[:, 6:9])
    for k in range(3):
        df = lin_bwd("additional_radiance_linear.%d" % k, relu(fk[k]), d[:, 9 + 3 * k:12 + 3 * k]) * (fk[k] > 0)
        dh2 = dh2 + lin_bwd("additional_radiance_feature_linear.%d" % k, h2, df)
    dxin = lin_bwd("views_linears.0", xin, dh2 * (zv > 0))
    dh7 = lin_bwd("feature_linear", h7, dxin[:, :256])
    dh7 = dh7 + lin_bwd("sigma_linear", h7, d[:, 0:1]) + lin_bwd("roughness_linear", h7, d[:, 4:5])
    dh7 = dh7 + lin_bwd("albedo_feature_linear", h7, lin_bwd("albedo_linear", relu(fa), d[:, 1:4]) * (fa > 0))
    dh7 = dh7 + lin_bwd("irradiance_feature_linear", h7, lin_bwd("irradiance_linear", relu(fi), d[:, 5:6]) * (fi > 0))
    # the trunk
    g = dh7 * masks[7]
    g_enc = np.zeros(e.shape)
    for i in range(7, -1, -1):
        gi = lin_bwd("positions_linears.%d" % i, ins[i], g)
        if i == 5:
            g_enc = g_enc + gi[:, :63]
            gi = gi[:, 63:]
        if i == 0:
            g_enc = g_enc + gi
        else:
            g = gi * masks[i - 1]
    grad = g_enc[:, 0:3].copy()
    for k in range(10):
        f = 2.0 ** k
        s_, c_ = e[:, 3 + 6 * k:6 + 6 * k], e[:, 6 + 6 * k:9 + 6 * k]
        grad = grad + f * (g_enc[:, 3 + 6 * k:6 + 6 * k] * c_ - g_enc[:, 6 + 6 * k:9 + 6 * k] * s_)
    return grad.reshape(N, S, 3).astype(F32), grads


# A.5 compositing — ibl_nerf_renderer.py:203-206, 241-245 (and :44-52, normal_from_depth.py:160-170)
# --------------------------------------------------------------------------------------------
def ray_dists(z_vals, rays_d):
    d = z_vals[..., 1:] - z_vals[..., :-1]
    d = np.concatenate([d, np.full_like(d[..., :1], F32(1e10))], -1)
    nrm = np.sqrt(np.sum(rays_d * rays_d, -1, dtype=F32), dtype=F32)[..., None]
    return (d * nrm).astype(F32)


def alpha_weights(sigma_raw, dists):
    alpha = (F32(1) - np.exp(-relu(sigma_raw) * dists, dtype=F32)).astype(F32)
    one_m = (F32(1) - alpha + F32(1e-10)).astype(F32)
    # ATen's CPU cumprod accumulates float tensors in double, rounding each prefix to float.
    T = np.cumprod(np.concatenate([np.ones_like(one_m[:, :1]), one_m], -1).astype(np.float64), -1).astype(F32)[:, :-1]
    return (alpha * T).astype(F32)


# --------------------------------------------------------------------------------------------
# A.6 fine sampling — nerf_models/nerf_renderer_helper.py:91-134 with det=True
# --------------------------------------------------------------------------------------------
def aten_sum_lastdim(x):
    """torch.sum(x, -1, keepdim=True) of a contiguous float32 [N, n] array, bit for bit: ATen's CPU cascade sum (aten/src/ATen/native/cpu/
    SumKernel.cpp) as it runs for 8-float vectors — vectorized_inner_sum: the row is read as n // 8 vectors; row_sum keeps 4 interleaved
    partial vectors (vector j goes to partial j % 4 while j < 4 * (n // 32), later vectors to partial 0; the cascade levels of multi_row_sum
    only fill from 16 groups = 512 elements on), folds them ((p0 + p1) + p2) + p3; the scalar tail x[8 * (n // 8):] is summed sequentially
    from 0 and the 8 vector lanes are then added to it in order.  Pinned against torch.sum in tests/test_oracle_golden.py (n = 30 .. 255).
    It matters because sample_pdf's `denom < 1e-5` test sits one fp32 ulp from the denominator of an empty bin (see sample_pdf)."""
    x = np.ascontiguousarray(x, dtype=F32)
    N, n = x.shape
    assert n < 512, "the cascade levels of ATen's sum start at 512 elements: not restated"
    vs = n // 8
    g = vs // 4
    v = x[:, :8 * vs].reshape(N, vs, 8)
    p = np.zeros((N, 4, 8), F32)
    for i in range(g):
        p = (p + v[:, 4 * i:4 * i + 4]).astype(F32)
    for j in range(4 * g, vs):
        p[:, 0] = (p[:, 0] + v[:, j]).astype(F32)
    P = (((p[:, 0] + p[:, 1]).astype(F32) + p[:, 2]).astype(F32) + p[:, 3]).astype(F32)
    fin = np.zeros((N,), F32)
    for t in range(8 * vs, n):
        fin = (fin + x[:, t]).astype(F32)
    for c in range(8):
        fin = (fin + P[:, c]).astype(F32)
    return fin[:, None]


def sample_pdf(bins, weights, n_samples, u=None):
    """u: [N, n_samples] uniform draws of the det=False branch (:103), None = det=True (u = linspace).
    The row sum follows torch.sum's own order (aten_sum_lastdim): an empty bin's cdf step is 1e-5 / sum = 9.994e-6 +- one ulp of the
    cdf, i.e. 167 or 168 ulps where the `denom < 1e-5` test (:128-129) flips — the last bit of the sum decides whether such a bin's
    samples collapse onto its edge."""
    bins = np.asarray(bins, dtype=F32)
    w = (np.asarray(weights, dtype=F32) + F32(1e-5)).astype(F32)
    pdf = (w / aten_sum_lastdim(w.reshape(-1, w.shape[-1])).reshape(w.shape[:-1] + (1,))).astype(F32)
    cdf = np.cumsum(pdf.astype(np.float64), -1).astype(F32)           # double accumulate (ATen CPU)
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)
    u = np.broadcast_to(torch_linspace(0, 1, n_samples), cdf.shape[:-1] + (n_samples,)) if u is None else np.asarray(u, dtype=F32)
    nb = cdf.shape[-1]
    inds = np.stack([np.searchsorted(cdf[r], u[r], side="right") for r in range(cdf.shape[0])])
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, nb - 1)
    c0, c1 = np.take_along_axis(cdf, below, -1), np.take_along_axis(cdf, above, -1)
    b0, b1 = np.take_along_axis(bins, below, -1), np.take_along_axis(bins, above, -1)
    den = (c1 - c0).astype(F32)
    den = np.where(den < F32(1e-5), F32(1), den)
    t = ((u - c0) / den).astype(F32)
    return (b0 + t * (b1 - b0)).astype(F32)


# --------------------------------------------------------------------------------------------
# A.7 epsilon normal — nerf_models/normal_from_depth.py:139-183
# --------------------------------------------------------------------------------------------
def normal_from_depth_eps(sd, rays_o, rays_d, z_vals, eps=0.01, return_depths=False, sigma=None):
    eps = F32(eps)
    up0 = np.broadcast_to(np.array([0, 1, 0], dtype=F32), rays_d.shape)
    right = cross(rays_d, up0)
    up = cross(right, rays_d)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]).astype(F32)
    offs = [eps * right, -(eps * right), eps * up, -(eps * up)]        # pts +- eps*v (:151-154)
    new_pts = np.concatenate([(pts + o[:, None, :]).astype(F32) for o in offs], 0)
    raw = network_query(sd, new_pts, None)[..., 0] if sigma is None else sigma   # sigma: teacher-forced query result [4N,S]
    dists = ray_dists(z_vals, rays_d)
    N = rays_o.shape[0]
    D = [np.sum(alpha_weights(raw[s * N:(s + 1) * N], dists) * z_vals, -1, dtype=F32) for s in range(4)]
    dx = (F32(2) * eps * right + (D[0] - D[1])[:, None] * rays_d).astype(F32)
    dy = (F32(2) * eps * up + (D[2] - D[3])[:, None] * rays_d).astype(F32)
    n = normalize(cross(dx, dy))
    return (n, np.stack(D, 0)) if return_depths else n


def normal_from_depth_direction_eps(sd, rays_o, rays_d, z_vals, eps=0.005, sigma=None):
    """nerf_models/normal_from_depth.py:55-100: depths along four rays whose (normalised) directions are tilted by
    +-eps*right / +-eps*up, sampled at the centre ray's z values; normal from the four end points."""
    eps = F32(eps)
    up0 = np.broadcast_to(np.array([0, 1, 0], dtype=F32), rays_d.shape)
    right = cross(rays_d, up0)
    up = cross(right, rays_d)
    new_d = [normalize((rays_d + eps * right).astype(F32)), normalize((rays_d - eps * right).astype(F32)),
             normalize((rays_d + eps * up).astype(F32)), normalize((rays_d - eps * up).astype(F32))]       # :64-67
    pts = np.concatenate([(rays_o[:, None, :] + d[:, None, :] * z_vals[:, :, None]).astype(F32) for d in new_d], 0)
    raw = network_query(sd, pts, None)[..., 0] if sigma is None else sigma
    dists = ray_dists(z_vals, rays_d)                                                                     # the centre ray's (:77-79)
    N = rays_o.shape[0]
    D = [np.sum(alpha_weights(raw[s * N:(s + 1) * N], dists) * z_vals, -1, dtype=F32) for s in range(4)]
    pos = [(rays_o + D[s][:, None] * new_d[s]).astype(F32) for s in range(4)]                            # :88-91
    return normalize(cross((pos[0] - pos[1]).astype(F32), (pos[2] - pos[3]).astype(F32)))


def depth_gradient_wrt_density(sigma_raw, dists, z_vals):
    """d depth_map / d raw[..., 0] of depth = sum_s w_s z_s (normal_from_depth.py:39-45 / :124-130), i.e. autograd through relu, exp,
    cumprod and the sum:  d depth / d alpha_s = T_s z_s - (sum_{i>s} w_i z_i) / (1 - alpha_s + 1e-10);  d alpha / d raw = dist exp(-raw dist) [raw > 0].
    float64 inside (the reference's float32 backward differs from it by its own rounding)."""
    sr = np.maximum(np.asarray(sigma_raw, dtype=np.float64), 0.0)
    di = np.asarray(dists, dtype=np.float64)
    alpha = 1.0 - np.exp(-sr * di)
    om = 1.0 - alpha + 1e-10
    T = np.cumprod(np.concatenate([np.ones_like(om[:, :1]), om], -1), -1)[:, :-1]
    wz = alpha * T * np.asarray(z_vals, dtype=np.float64)
    suffix = np.concatenate([np.cumsum(wz[:, ::-1], -1)[:, ::-1][:, 1:], np.zeros_like(wz[:, :1])], -1)   # sum over i > s
    d_alpha = T * z_vals - suffix / om
    return d_alpha * di * np.exp(-sr * di) * (np.asarray(sigma_raw) > 0)


def normal_from_depth_gradient(sd, rays_o, rays_d, z_vals, direction=False, sigma_grad=None):
    """nerf_models/normal_from_depth.py:102-137 (position: the ray origin moves by a*right + b*up) and :16-52 (direction: the ray
    direction becomes a*right + b*up + sqrt(1 - a^2 - b^2) d), autograd of depth_map with respect to (a, b) at 0:
        d depth / d a = sum_s (d depth / d raw_s) (grad raw_s . right) [z_s],     normal = normalize(right dx + up dy - d).
    sigma_grad: teacher-forced (sigma [N,S], grad [N,S,3]) of the density-gradient query."""
    up0 = np.broadcast_to(np.array([0, 1, 0], dtype=F32), rays_d.shape)
    right = cross(rays_d, up0)
    up = cross(right, rays_d)
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]).astype(F32)
    N, S = z_vals.shape
    if sigma_grad is None:
        sigma, grad = density_gradient(sd, pts.reshape(-1, 3))
        sigma, grad = sigma.reshape(N, S), grad.reshape(N, S, 3)
    else:
        sigma, grad = sigma_grad
    G = depth_gradient_wrt_density(sigma, ray_dists(z_vals, rays_d), z_vals)
    if direction:
        G = G * z_vals
    dx = np.sum(G * np.sum(grad.astype(np.float64) * right[:, None, :], -1), -1)
    dy = np.sum(G * np.sum(grad.astype(np.float64) * up[:, None, :], -1), -1)
    g = right * dx[:, None] + up * dy[:, None]
    return normalize((g - rays_d).astype(F32))


# --------------------------------------------------------------------------------------------
# A.9 pieces
# --------------------------------------------------------------------------------------------
def lut_fetch(lut, n_dot_v, rough):
    """F.grid_sample(bilinear, zeros padding, align_corners=True), ibl_nerf_renderer.py:418-421.
    lut [3,512,512] (row = roughness, col = n.v).  Returns [N,3]."""
    Hh, Ww = lut.shape[1:]
    gx = (F32(2) * n_dot_v - F32(1)).astype(F32)
    gy = (F32(2) * rough - F32(1)).astype(F32)
    x = (((gx + F32(1)) / F32(2)) * F32(Ww - 1)).astype(F32)
    y = (((gy + F32(1)) / F32(2)) * F32(Hh - 1)).astype(F32)
    x0, y0 = np.floor(x), np.floor(y)
    out = np.zeros((x.shape[0], 3), dtype=F32)
    for dx_, dy_ in ((0, 0), (1, 0), (0, 1), (1, 1)):
        xi, yi = x0 + dx_, y0 + dy_
        wx = (x0 + F32(1) - x) if dx_ == 0 else (x - x0)
        wy = (y0 + F32(1) - y) if dy_ == 0 else (y - y0)
        ok = (xi >= 0) & (xi <= Ww - 1) & (yi >= 0) & (yi <= Hh - 1)
        xc, yc = np.clip(xi, 0, Ww - 1).astype(np.int64), np.clip(yi, 0, Hh - 1).astype(np.int64)
        v = lut[:, yc, xc].T
        out += np.where(ok[:, None], v * (wx * wy).astype(F32)[:, None], F32(0)).astype(F32)
    return out


def fresnel_schlick_roughness(cos_theta, F0, rough):
    """nerf_models/microfacet.py:8-12."""
    c, r = cos_theta[:, None], rough[:, None]
    F1 = np.maximum(F32(1) - r, F0) - F0
    return (F0 + F1 * np.power(np.clip(F32(1) - c, 0, 1), F32(5), dtype=F32)).astype(F32)


def composite_reflected(raw, z_vals, dirs, radiance_f=None):
    """raw2outputs_simple, ibl_nerf_renderer.py:38-68 -> [N,4,3] (radiance, coarse radiance 1..3)."""
    radiance_f = radiance_f or sigmoid
    w = alpha_weights(raw[..., 0], ray_dists(z_vals, dirs))
    maps = [np.sum(w[..., None] * radiance_f(raw[..., 6 + 3 * k:9 + 3 * k]), -2, dtype=F32) for k in range(4)]
    return np.stack(maps, 1).astype(F32)


def decode_masks(mask_img, n_obj):
    """ibl_nerf_renderer.py:223-228 / :233-238: object q <=> 9(q+1)/255 < m < 11(q+1)/255."""
    m = mask_img[:, 0]
    masks = [np.logical_and(F32(11 * (q + 1) / 255.) > m, m > F32(9 * (q + 1) / 255.)) for q in range(n_obj)]
    return masks, m > 0


# --------------------------------------------------------------------------------------------
# one pass: raw2outputs, ibl_nerf_renderer.py:153-527 (approximate_radiance=True, shipped flags)
# --------------------------------------------------------------------------------------------
def raw2outputs(sd, rays_o, rays_d, z_vals, z_const, near, far, lut, gt=None, edit=None, stages=None, flags=None, aux=None,
                teacher=None, noise=None):
    """flags: use_radiance_linear (radiance_f = ReLU + Reinhard LDR map, :30-35, :192-197, :480-483),
    lut_coefficient ('F' | 'F0', :433-438), gamma_correct (default True as in the shipped configs),
    epsilon (default 0.01, :358-361), correct_depth_for_prefiltered_radiance_infer (default True, :455-461),
    target_normal_map_for_radiance_calculation ('normal_map_from_depth_gradient_epsilon' | 'ground_truth' |
    'normal_map_from_depth_gradient_direction_epsilon' with epsilon_direction, default 0.005; :348-375),
    depth_map_from_ground_truth / calculate_{albedo,roughness,irradiance}_from_gt (:251-252, :320-330): the target map is
    the gt_values row and no longer aliases the network's map, so edits stop showing in depth_map / disp / the mip level.
    aux: {'albedo_mlp' | 'roughness_mlp' | 'irradiance_mlp': PositionMLP state dict} (:291-303): their samples replace the main
    network's before compositing; an irradiance_mlp's go through sigmoid whatever radiance_f is.
    teacher: {'raw' [N,S,18], 'sigma_offsets' [4N,S], 'refl_raw' [N,Sc,18]} — recorded results of the three network queries
    (:201, normal_from_depth.py:158, :445) used INSTEAD of evaluating `sd` (teacher forcing: the pass downstream of the MLP
    in isolation).
    noise: [N,S] added to the density before compositing (raw_noise_std > 0, :208-216, :242), already multiplied by the std."""
    teacher = teacher or {}
    gt = gt or {}
    edit = edit or {}
    flags = flags or {}
    linear = bool(flags.get("use_radiance_linear", False))
    radiance_f = relu if linear else sigmoid
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]).astype(F32)       # :200
    raw = teacher["raw"] if "raw" in teacher else network_query(sd, pts, rays_d)            # :201 (rays_d, not viewdirs)
    dists = ray_dists(z_vals, rays_d)
    masks, mask_all = None, None
    if edit.get("edit_intrinsic"):
        assert edit["num_edit_objects"] > 0
        masks, mask_all = decode_masks(gt["edit_intrinsic_mask"], edit["num_edit_objects"])
    elif edit.get("insert_object"):
        assert edit["num_insert_objects"] > 0
        masks, mask_all = decode_masks(gt["object_insert_mask"], edit["num_insert_objects"])
    sig_raw = raw[..., 0] if noise is None else (raw[..., 0] + np.asarray(noise, dtype=F32)).astype(F32)
    w = alpha_weights(sig_raw, dists)                                                       # :241-245
    depth = np.sum(w * z_vals, -1, dtype=F32)                                               # :249
    tdepth = depth                                                                          # :250 (one array: aliases depth_map)
    if flags.get("depth_map_from_ground_truth", False):
        tdepth = gt["depth"][:, 0].astype(F32).copy()                                       # :251-252
    if edit.get("edit_intrinsic") and edit.get("edit_depth"):
        tdepth[mask_all] = gt["edit_depth"][:, 0][mask_all]                                 # :253-254
    if edit.get("insert_object"):
        tdepth[mask_all] = gt["object_insert_depth"][:, 0][mask_all]                        # :255-256
    acc = np.sum(w, -1, dtype=F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        disp = (F32(1) / np.maximum(F32(1e-10), depth / acc)).astype(F32)                   # :258
    x_surface = (rays_o + rays_d * tdepth[:, None]).astype(F32)                             # :262
    albedo = np.sum(w[..., None] * sigmoid(raw[..., 1:4]), -2, dtype=F32)                   # :281-282
    rough = np.sum(w * sigmoid(raw[..., 4]), -1, dtype=F32)                                 # :284-285
    irr = np.sum(w * radiance_f(raw[..., 5]), -1, dtype=F32)[:, None]                       # :287-288, :328
    aux = aux or {}
    if aux.get("albedo_mlp") is not None:                                                   # :291-294
        albedo = np.sum(w[..., None] * sigmoid(position_mlp_query(aux["albedo_mlp"], pts)[..., 0:3]), -2, dtype=F32)
    if aux.get("roughness_mlp") is not None:                                                # :296-299
        rough = np.sum(w * sigmoid(position_mlp_query(aux["roughness_mlp"], pts)[..., 0]), -1, dtype=F32)
    if aux.get("irradiance_mlp") is not None:                                               # :301-304
        irr = np.sum(w * sigmoid(position_mlp_query(aux["irradiance_mlp"], pts)[..., 0]), -1, dtype=F32)[:, None]
    rough_net = rough                                                                       # :324 (aliases unless from gt)
    if flags.get("calculate_albedo_from_gt", False):
        albedo = gt["albedo"][:, :3].astype(F32).copy()                                     # :321-322
    if flags.get("calculate_roughness_from_gt", False):
        rough = gt["roughness"][:, 0].astype(F32).copy()                                    # :325-326
    if flags.get("calculate_irradiance_from_gt", False):
        irr = gt["irradiance"][:, :3].astype(F32).copy()                                    # :329-330
    rad = [np.sum(w[..., None] * radiance_f(raw[..., 6 + 3 * k:9 + 3 * k]), -2, dtype=F32) for k in range(4)]

    inferred_normal = None
    if flags.get("infer_normal", False):                                                    # :267-276 (per-sample form)
        if flags.get("infer_normal_at_surface", False):                                     # :268-271: one query at x_surface
            inferred_normal = (F32(2) * sigmoid(position_mlp_query(aux["normal_mlp"], x_surface[:, None, :])[:, 0]) - F32(1)).astype(F32)
        else:
            inferred_normal = np.sum(w[..., None] * (F32(2) * sigmoid(position_mlp_query(aux["normal_mlp"], pts)) - F32(1)), -2, dtype=F32)
    nmode = flags.get("target_normal_map_for_radiance_calculation", "normal_map_from_depth_gradient_epsilon")
    if nmode == "inferred_normal_map":
        normal = inferred_normal.copy()                                                     # :372-373, used as it is
    elif nmode == "ground_truth":
        normal = normalize(F32(2) * gt["normal"] - F32(1))                                  # :370-371
    elif nmode == "normal_map_from_depth_gradient_epsilon":
        normal = normal_from_depth_eps(sd, rays_o, rays_d, z_vals, eps=float(flags.get("epsilon", 0.01)),
                                       sigma=teacher.get("sigma_offsets"))                  # :358-361
    elif nmode == "normal_map_from_depth_gradient_direction_epsilon":
        normal = normal_from_depth_direction_eps(sd, rays_o, rays_d, z_vals, eps=float(flags.get("epsilon_direction", 0.005)),
                                                 sigma=teacher.get("sigma_offsets"))        # :366-369
    elif nmode in ("normal_map_from_depth_gradient", "normal_map_from_depth_gradient_direction"):   # :354-357, :362-365 (gradients enabled)
        normal = normal_from_depth_gradient(sd, rays_o, rays_d, z_vals, direction=nmode.endswith("direction"),
                                            sigma_grad=teacher.get("sigma_grad"))
    elif nmode in ("normal_map_from_sigma_gradient", "normal_map_from_sigma_gradient_surface"):
        raise NameError("get_normal_from_sigma_gradient")                                   # :349-353: the import is commented out (:15)
    else:
        raise ValueError(nmode)                                                             # :374-375
    if stages is not None:
        stages["normal_raw"] = normal.copy()
    if edit.get("edit_intrinsic"):                                                          # :378-398
        if edit.get("edit_normal"):
            normal[mask_all] = normalize(F32(2) * gt["edit_normal"] - F32(1))[mask_all]
        if edit.get("edit_albedo"):
            if edit.get("edit_albedo_by_img"):
                albedo[mask_all] = gt["edit_albedo"][mask_all]
            else:
                lst = np.asarray(edit["editing_target_albedo_list"], dtype=F32)
                for q in range(edit["num_edit_objects"]):
                    albedo[masks[q]] = lst[3 * q:3 * q + 3]
        if edit.get("edit_roughness"):
            if edit.get("edit_roughness_by_img"):                                           # :394-395: the FIRST masked row of this call's rays, for all of them
                if mask_all.any():
                    rough[mask_all] = np.asarray(gt["edit_roughness"], dtype=F32).reshape(len(rough), -1)[mask_all][0][0]
            else:
                for q, r in enumerate(edit["editing_target_roughness_list"]):
                    rough[masks[q]] = F32(r)
    elif edit.get("insert_object"):                                                         # :400-410
        normal[mask_all] = normalize(F32(2) * gt["object_insert_normal"] - F32(1))[mask_all]
        al = np.asarray(edit["inserting_target_albedo_list"], dtype=F32)
        for q in range(edit["num_insert_objects"]):
            rough[masks[q]] = F32(edit["inserting_target_roughness_list"][q])
            if edit["inserting_target_irradiance_list"][q] > 0:
                irr[masks[q]] = F32(edit["inserting_target_irradiance_list"][q])
            albedo[masks[q]] = al[3 * q:3 * q + 3]

    ndv = np.clip(np.sum(-rays_d * normal, -1, dtype=F32), 0, 1).astype(F32)                # :412-413
    env = lut_fetch(lut, ndv, rough)                                                        # :418-421
    metal = (F32(1) - rough)[:, None]
    F0 = (F32(0.04) * (F32(1) - metal) + albedo * metal).astype(F32)                        # :424-427
    fres = fresnel_schlick_roughness(ndv, F0, rough)
    lutc = flags.get("lut_coefficient", "F")
    if lutc not in ("F", "F0"):
        raise ValueError(lutc)                                                              # :437-438
    spec_coef = ((fres if lutc == "F" else F0) * env[:, 0:1] + env[:, 1:2]).astype(F32)     # :433-436
    refl_d = (rays_d - F32(2) * np.sum(normal * rays_d, -1, keepdims=True, dtype=F32) * normal).astype(F32)
    refl_pts = (x_surface[:, None, :] + refl_d[:, None, :] * z_const[:, :, None]).astype(F32)   # :440
    refl_raw = teacher["refl_raw"] if "refl_raw" in teacher else network_query(sd, refl_pts, refl_d)   # :445
    pref_maps = composite_reflected(refl_raw, z_const, refl_d, radiance_f)                  # :446-448
    depth_0 = ((np.asarray(far, dtype=F32) + np.asarray(near, dtype=F32)) * F32(0.5)).astype(F32)   # :456
    if depth_0.ndim:
        depth_0 = depth_0.reshape(-1)                                                       # per-ray planes [N, 1]: depth_0[..., 0] (:458)
    if flags.get("correct_depth_for_prefiltered_radiance_infer", True):
        level = np.clip(rough_net * depth / depth_0, 0, 1).astype(F32)                      # :458-459 (roughness_map)
    else:
        level = rough_net.astype(F32)                                                       # :461
    i1 = np.clip((level * F32(3)).astype(np.int64), 0, 3)                                   # :464-465
    i2 = np.clip(i1 + 1, 0, 3)
    rem = ((level * F32(3)) - i1.astype(F32))[:, None].astype(F32)
    ar = np.arange(pref_maps.shape[0])
    pref = ((F32(1) - rem) * pref_maps[ar, i1] + rem * pref_maps[ar, i2]).astype(F32)       # :468-470
    diffuse = ((F32(1) - fres) * (F32(1) - metal) * albedo * irr).astype(F32)               # :472
    specular = (spec_coef * pref).astype(F32)
    color = (diffuse + specular).astype(F32)
    if stages is not None:
        stages.update(raw=raw, refl_raw=refl_raw, pref_maps=pref_maps, env=env, refl_d=refl_d,
                      x_surface=x_surface, level=level)
    gam = rgb_to_srgb if flags.get("gamma_correct", True) else (lambda x: x)
    ldr = (lambda x: (x / (x + F32(1))).astype(F32)) if linear else (lambda x: x)           # tonemap_reinherd :30-31
    g = lambda x: gam(ldr(x))                                                               # output_f :487
    res = {"color_map": g(color), "radiance_map": g(rad[0])}
    for k in range(3):
        res["radiance_map_%d" % (k + 1)] = g(rad[k + 1])
    for k in range(3):
        res["reflected_coarse_radiance_map_%d" % (k + 1)] = g(pref_maps[:, k + 1])
    res.update({
        "irradiance_map": g(irr), "reflected_radiance_map": g(pref_maps[:, 0]),
        "prefiltered_reflected_map": g(pref), "albedo_map": gam(albedo), "roughness_map": rough,
        "specular_map": g(specular), "diffuse_map": g(diffuse), "n_dot_v_map": ndv,
        "target_normal_map": normal, "disp_map": disp, "acc_map": acc, "depth_map": depth,
        "target_depth_map": tdepth, "weights": w})
    if inferred_normal is not None:
        res["inferred_normal_map"] = inferred_normal                                        # :517
    return res


# --------------------------------------------------------------------------------------------
# render_rays / render_decomp — ibl_nerf_renderer.py:629-732, :759-813 (perturb=0, raw_noise_std=0)
# --------------------------------------------------------------------------------------------
def coarse_z(near, far, n_samples, n_rays, lindisp=False):
    """near / far: scalars, or one plane per ray as [n_rays, 1] arrays (render_decomp :802-805 broadcasts them against rays_d[..., :1])."""
    t = torch_linspace(0, 1, n_samples)
    near, far = np.asarray(near, dtype=F32), np.asarray(far, dtype=F32)
    if lindisp:                                                                            # :674
        z = (F32(1) / (F32(1) / near * (F32(1) - t) + F32(1) / far * t)).astype(F32)
    else:
        z = (near * (F32(1) - t) + far * t).astype(F32)                                    # :672
    return np.broadcast_to(z, (n_rays, n_samples)).copy()


def pytest_uniform(n, m):
    """np.random.seed(0); np.random.rand(n, m) -> float32: the draws of the reference's `pytest` branches
    (ibl_nerf_renderer.py:686-690, nerf_renderer_helper.py:106-113)."""
    state = np.random.get_state()
    np.random.seed(0)
    a = np.random.rand(n, m).astype(F32)
    np.random.set_state(state)
    return a


def render_rays(sd_coarse, sd_fine, rays_o, rays_d, near, far, lut, n_samples=64, n_importance=128,
                gt=None, edit=None, stages=None, flags=None, aux=None, t_rand=None, u=None, noise_c=None, noise_f=None):
    """t_rand [N, n_samples] / u [N, n_importance]: the uniform draws of perturb > 0 (:678-692 stratified jitter, :703 stochastic
    fine samples); None = the deterministic test-time path.  noise_c [N, n_samples] / noise_f [N, n_samples + n_importance]: the
    density noise of raw_noise_std > 0 per pass (already multiplied by the std)."""
    rays_o = np.ascontiguousarray(rays_o, dtype=F32)
    rays_d = np.ascontiguousarray(rays_d, dtype=F32)
    N = rays_o.shape[0]
    flags = flags or {}
    z = coarse_z(near, far, n_samples, N, bool(flags.get("lindisp", False)))
    if t_rand is not None:                                                                 # :678-692
        mids = (F32(0.5) * (z[:, 1:] + z[:, :-1])).astype(F32)
        upper = np.concatenate([mids, z[:, -1:]], -1)
        lower = np.concatenate([z[:, :1], mids], -1)
        z = (lower + (upper - lower) * np.asarray(t_rand, dtype=F32)).astype(F32)
    st_c = {} if stages is not None else None
    res = raw2outputs(sd_coarse, rays_o, rays_d, z, z, near, far, lut, gt, edit, st_c, flags, aux, noise=noise_c)
    if n_importance > 0:
        mids = (F32(0.5) * (z[:, 1:] + z[:, :-1])).astype(F32)                             # :701
        zs = sample_pdf(mids, res["weights"][:, 1:-1], n_importance, u)                    # :702-703
        zf = np.sort(np.concatenate([z, zs], -1), -1)                                      # :707
        st_f = {} if stages is not None else None
        fine = raw2outputs(sd_fine if sd_fine is not None else sd_coarse, rays_o, rays_d, zf, z,
                           near, far, lut, gt, edit, st_f, flags, aux, noise=noise_f)
        for k, v in res.items():
            fine[k + "0"] = v                                                              # :712-713
        res = fine
        res["z_std"] = np.std(zs, -1, dtype=F32)                                           # :718
        if stages is not None:
            stages.update(c=st_c, f=st_f, z_samples=zs, z_fine=zf)
    elif stages is not None:
        stages.update(c=st_c)
    if (aux or {}).get("depth_mlp") is not None:                                           # infer_depth, :722-726
        viewdirs = (rays_d / np.sqrt(np.sum(rays_d * rays_d, -1, keepdims=True, dtype=F32))).astype(F32)   # :795
        out = position_direction_mlp_query(aux["depth_mlp"], rays_o[:, None, :], viewdirs)
        res["inferred_depth_map"] = relu(out[..., 0]).reshape(-1)
    return res


def render_decomp(H, W, K, sd_coarse, sd_fine, lut, near, far, rays=None, c2w=None, chunk=1024,
                  n_samples=64, n_importance=128, gt_values=None, flags=None, **edit):
    """ibl_nerf_renderer.py:759-813.  Exactly one of rays ([2,N,3]) / c2w ([3,4])."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, K, c2w)
    else:
        rays_o, rays_d = rays
    sh = rays_d.shape
    ro, rd = np.reshape(rays_o, (-1, 3)).astype(F32), np.reshape(rays_d, (-1, 3)).astype(F32)
    gt_values = gt_values or {}
    outs = {}
    for i in range(0, ro.shape[0], chunk):                                                 # batchify_rays :735-756
        gt = {k: np.array(v[i:i + chunk], dtype=F32) for k, v in gt_values.items()}
        r = render_rays(sd_coarse, sd_fine, ro[i:i + chunk], rd[i:i + chunk], near, far, lut,
                        n_samples, n_importance, gt, edit, None, flags)
        for k, v in r.items():
            outs.setdefault(k, []).append(v)
    return {k: np.concatenate(v, 0).reshape(sh[:-1] + v[0].shape[1:]) for k, v in outs.items()}
