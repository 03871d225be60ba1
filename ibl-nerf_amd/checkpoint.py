"""Checkpoint schema of the reference `IBLNeRF` module and host-side helpers.

Mirrors the state-dict layout of `src/nerf_models/ibl_nerf.py:14-75` (key order is the
module registration order, verified against the reference in this container) and the
`.tar` dict written by `src/train.py:180-191` / read by `src/nerf_models/ibl_nerf.py:345-378`.

Nothing here computes on the hot path: it turns a state dict into the flat fp32 "blob"
(`[out,in]` row-major weight then bias, per layer, in `SCHEMA` order) that
`iblnerf_upload_weights` (include/iblnerf.h) consumes.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np

# (key prefix, out_features, in_features) in the reference's registration order.
SCHEMA = (
    [("positions_linears.0", 256, 63)]
    + [("positions_linears.%d" % i, 256, 256) for i in (1, 2, 3, 4)]
    + [("positions_linears.5", 256, 319)]
    + [("positions_linears.%d" % i, 256, 256) for i in (6, 7)]
    + [
        ("views_linears.0", 256, 283),
        ("feature_linear", 256, 256),
        ("sigma_linear", 1, 256),
        ("albedo_feature_linear", 128, 256),
        ("albedo_linear", 3, 128),
        ("roughness_linear", 1, 256),
        ("irradiance_feature_linear", 128, 256),
        ("irradiance_linear", 1, 128),
        ("radiance_linear", 3, 256),
    ]
    + [("additional_radiance_feature_linear.%d" % i, 128, 256) for i in range(3)]
    + [("additional_radiance_linear.%d" % i, 3, 128) for i in range(3)]
)
N_PARAMS = sum(o * i + o for _, o, i in SCHEMA)  # 798 994
assert N_PARAMS == 798994


def synthetic_state_dict(seed: int, gain: float = 1.0, sigma_bias: float = 0.3) -> "OrderedDict[str, np.ndarray]":
    """Deterministic stand-in for a trained checkpoint (no checkpoint ships with the reference).

    Every weight and bias ~ U(-g/sqrt(fan_in), +g/sqrt(fan_in)) — PyTorch's `nn.Linear`
    default is g = 1 — from numpy's legacy `RandomState(seed)` stream (bit-stable across
    numpy versions/platforms), then `sigma_linear.bias = sigma_bias` so density is non-zero
    (SURVEY.md §8 d).
    """
    rng = np.random.RandomState(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, out_f, in_f in SCHEMA:
        bound = gain / np.sqrt(in_f)
        sd[name + ".weight"] = rng.uniform(-bound, bound, size=(out_f, in_f)).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-bound, bound, size=(out_f,)).astype(np.float32)
    sd["sigma_linear.bias"][:] = np.float32(sigma_bias)
    return sd


# the 15 wide activations of IBLNeRF.forward in the order iblnerf_layer_ranges reports them (include/iblnerf.h)
ACTIVATIONS = tuple("positions_linears.%d" % i for i in range(8)) + ("feature_linear", "albedo_feature_linear", "irradiance_feature_linear", "views_linears.0") \
    + tuple("additional_radiance_feature_linear.%d" % i for i in range(3))


def scale_activations(sd, t, color_independent=False):
    """The same network with activation `name` held as t[name] * (its value): an EXACT reparametrisation when every t is a power of two
    (relu(t x) = t relu(x) for t > 0; a power-of-two factor changes no mantissa) — each producing layer's weight and bias take t_out / t_in
    per input column block and t_out, each consumer's columns 1 / t_in, the N = 1 / 3 heads included, so the raw outputs are unchanged.
    `t`: {activation name (ACTIVATIONS): factor}, missing = 1.  Used by the renderer's f16 range policy: a checkpoint whose activations leave
    the f16 range (65504) is rescaled until they fit instead of being given to the 2^-17 bf16x3 kernels.  Works on numpy arrays and torch tensors."""
    g = lambda k: float(t.get(k, 1.0))
    out = OrderedDict((k, v.copy() if isinstance(v, np.ndarray) else v.clone()) for k, v in sd.items())

    def produce(name, t_out, col_blocks):
        """weight columns [c0, c1) scaled by t_out / t_in for each (c0, c1, t_in); bias by t_out"""
        w = out[name + ".weight"]
        for c0, c1, t_in in col_blocks:
            w[:, c0:c1] *= np.float32(t_out / t_in)
        out[name + ".bias"] *= np.float32(t_out)

    def consume(name, t_in):
        out[name + ".weight"] *= np.float32(1.0 / t_in)

    P = "positions_linears.%d"
    produce(P % 0, g(P % 0), [(0, 63, 1.0)])
    for l in (1, 2, 3, 4, 6, 7):
        produce(P % l, g(P % l), [(0, 256, g(P % (l - 1)))])
    produce(P % 5, g(P % 5), [(0, 63, 1.0), (63, 319, g(P % 4))])                      # cat([x63, h]) (ibl_nerf.py:168)
    t7 = g(P % 7)
    consume("sigma_linear", t7)
    consume("roughness_linear", t7)
    produce("albedo_feature_linear", g("albedo_feature_linear"), [(0, 256, t7)])
    consume("albedo_linear", g("albedo_feature_linear"))
    produce("irradiance_feature_linear", g("irradiance_feature_linear"), [(0, 256, t7)])
    consume("irradiance_linear", g("irradiance_feature_linear"))
    t2 = t7                                                                              # what the radiance heads read
    if not color_independent:
        produce("feature_linear", g("feature_linear"), [(0, 256, t7)])
        produce("views_linears.0", g("views_linears.0"), [(0, 256, g("feature_linear")), (256, 283, 1.0)])   # cat([feature, dir27]) (:194)
        t2 = g("views_linears.0")
    consume("radiance_linear", t2)
    for k in range(3):
        n = "additional_radiance_feature_linear.%d" % k
        produce(n, g(n), [(0, 256, t2)])
        consume("additional_radiance_linear.%d" % k, g(n))
    return out


def arch_schema(D=8, W=256, multires=10, multires_views=4):
    """(name, out, in) of IBLNeRF(D, W, input_ch = 3 + 6 multires, input_ch_views = 3 + 6 multires_views, skips=[4], coarse_radiance_number=3) in registration order
    (ibl_nerf.py:45-72; create_IBLNeRF fixes skips = [4], :265: positions_linears.5 takes cat([x, h]) when it exists, i.e. for D >= 6)."""
    ch, chv = 3 + 6 * multires, 3 + 6 * multires_views
    sch = [("positions_linears.0", W, ch)] + [("positions_linears.%d" % (i + 1), W, W + ch if i == 4 else W) for i in range(D - 1)]
    sch += [("views_linears.0", W, chv + W), ("feature_linear", W, W), ("sigma_linear", 1, W), ("albedo_feature_linear", W // 2, W), ("albedo_linear", 3, W // 2),
            ("roughness_linear", 1, W), ("irradiance_feature_linear", W // 2, W), ("irradiance_linear", 1, W // 2), ("radiance_linear", 3, W)]
    sch += [("additional_radiance_feature_linear.%d" % i, W // 2, W) for i in range(3)] + [("additional_radiance_linear.%d" % i, 3, W // 2) for i in range(3)]
    return tuple(sch)


def arch_of(sd):
    """(D, W, multires, multires_views) of an IBLNeRF state dict, from its shapes."""
    shp = lambda k: tuple(int(v) for v in sd[k].shape)
    W, ch = shp("positions_linears.0.weight")
    D = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("positions_linears.") and k.endswith(".weight"))
    chv = shp("views_linears.0.weight")[1] - W
    if (ch - 3) % 6 or (chv - 3) % 6:
        raise ValueError("not a positional-encoding width: input_ch %d, input_ch_views %d" % (ch, chv))
    return D, W, (ch - 3) // 6, (chv - 3) // 6


SHIPPED_ARCH = (8, 256, 10, 4)


def is_member_of_built(arch):
    """Can IBLNeRF(D, W, multires, multires_views) be embedded exactly in the built 8 x 256 / 10 / 4 architecture (embed_architecture)?"""
    D, W, L, Lv = arch
    return 1 <= D <= 8 and D != 5 and 2 <= W <= 256 and 0 <= L <= 10 and 0 <= Lv <= 4


def is_generic_arch(arch):
    """An architecture the fused kernels cannot hold but csrc/generic_mlp.hip can (iblnerf_upload_weights_arch): layer by layer in exact fp32, every sample evaluated."""
    D, W, L, Lv = arch
    return not is_member_of_built(arch) and 1 <= D <= 32 and D != 5 and 2 <= W <= 4096 and W % 2 == 0 and 0 <= L <= 24 and 0 <= Lv <= 24


def weights_checksum(sd):
    """blob_checksum of a state dict of ANY architecture: of the built-shape blob it is uploaded as when it is a member of the built architecture (what the fixtures
    have always recorded), of its own flattening (arch_blob) otherwise."""
    arch = arch_of(sd)
    return blob_checksum(state_dict_to_blob(embed_architecture(sd))) if is_member_of_built(arch) else blob_checksum(arch_blob(sd, arch))


def arch_blob(sd, arch=None):
    """An IBLNeRF state dict of ANY architecture flattened in registration order (arch_schema): what iblnerf_upload_weights_arch takes.  Validates names and shapes."""
    arch = arch or arch_of(sd)
    parts = []
    for name, o, i in arch_schema(*arch):
        w, b = _to_numpy(sd[name + ".weight"]), _to_numpy(sd[name + ".bias"])
        if tuple(w.shape) != (o, i) or tuple(b.shape) != (o,):
            raise ValueError("state dict does not follow the IBLNeRF%s schema: %s is %s / %s, expected (%d, %d) / (%d,)" % (arch, name, tuple(w.shape), tuple(b.shape), o, i, o))
        parts += [w.reshape(-1), b.reshape(-1)]
    extra = set(sd) - {n + t for n, _, _ in arch_schema(*arch) for t in (".weight", ".bias")}
    if extra:
        raise ValueError("state dict holds tensors outside the IBLNeRF%s schema: %s" % (arch, sorted(extra)[:4]))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


def synthetic_arch_state_dict(seed, arch, gain=1.0, sigma_bias=0.3):
    """synthetic_state_dict for IBLNeRF(*arch): same stream discipline (U(-g / sqrt(fan_in), g / sqrt(fan_in)) per tensor in registration order)."""
    rng = np.random.RandomState(seed)
    sd = OrderedDict()
    for name, out_f, in_f in arch_schema(*arch):
        bound = gain / np.sqrt(in_f)
        sd[name + ".weight"] = rng.uniform(-bound, bound, size=(out_f, in_f)).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-bound, bound, size=(out_f,)).astype(np.float32)
    sd["sigma_linear.bias"][:] = np.float32(sigma_bias)
    return sd


def _set_identity(w, n, c0):
    """w[u, c0 + u] = 1 for u < n in ONE operation (a numpy array or a torch tensor on any device): a Python loop is n tiny kernel launches per identity layer on a
    device tensor, and a training step uploads both networks every step."""
    if hasattr(w, "fill_diagonal_"):
        w[:n, c0:c0 + n].fill_diagonal_(1.0)
    else:
        np.fill_diagonal(w[:n, c0:c0 + n], 1.0)


def embed_architecture(sd):
    """A SMALLER IBLNeRF as a member of the built architecture (D = 8, W = 256, multires 10 / 4): the same function, exactly.
      narrower layers (W < 256, W // 2 < 128): zero rows and columns — an absent unit is a unit that outputs relu(0) = 0 and is read with weight 0;
      fewer frequencies (multires < 10, multires_views < 4): the encoding [x, sin 2^0 x, cos 2^0 x, sin 2^1 x, ...] of the smaller network is a PREFIX of the built one's
        (positional_embedder.py:21-34): zero columns for the frequencies it does not have;
      fewer trunk layers (D < 8): identity layers behind the last one — the trunk's activations are >= 0 after their ReLU, so relu(I h + 0) = h; positions_linears.5
        = [0 | I] when the smaller network has no skip layer of its own (D < 6; D = 5 does not exist in the reference: its forward concatenates behind the last layer).
    Products with 0 and 1 and sums with 0 are exact in every product scheme of the kernels and in fp32, so the embedded network's outputs are the smaller network's to
    the scheme's own precision.  Not embeddable (ValueError): D > 8 or D == 5, W > 256, multires > 10, multires_views > 4.  numpy arrays or torch tensors in, same kind out
    (the reference's key order).  A state dict of the built shape is returned as it is."""
    D, W, L, Lv = arch_of(sd)
    if (D, W, L, Lv) == SHIPPED_ARCH:
        return sd
    if D > 8 or D == 5 or D < 1 or W > 256 or W < 2 or L > 10 or Lv > 4 or L < 0 or Lv < 0:
        raise ValueError("IBLNeRF(D=%d, W=%d, multires=%d, multires_views=%d) is not a member of the built architecture (D <= 8 and != 5, W <= 256, multires <= 10, "
                         "multires_views <= 4)" % (D, W, L, Lv))
    small = arch_schema(D, W, L, Lv)
    for name, o, i in small:
        if tuple(int(v) for v in sd[name + ".weight"].shape) != (o, i):
            raise ValueError("%s.weight has shape %s, IBLNeRF(D=%d, W=%d, ...) has (%d, %d)" % (name, tuple(sd[name + ".weight"].shape), D, W, o, i))
    like = sd["positions_linears.0.weight"]
    zeros = (lambda *sh: np.zeros(sh, dtype=np.float32)) if isinstance(like, np.ndarray) else (lambda *sh: like.new_zeros(sh))
    ch, chv, H = 3 + 6 * L, 3 + 6 * Lv, W // 2
    out = OrderedDict()
    for name, o, i in SCHEMA:
        out[name + ".weight"], out[name + ".bias"] = zeros(o, i), zeros(o)

    def put(name, rows, blocks):
        """blocks: (destination column, source column, width)"""
        w = sd[name + ".weight"].detach() if hasattr(sd[name + ".weight"], "detach") else sd[name + ".weight"]
        for dc, sc, n in blocks:
            out[name + ".weight"][:rows, dc:dc + n] = w[:, sc:sc + n]
        out[name + ".bias"][:rows] = sd[name + ".bias"].detach() if hasattr(sd[name + ".bias"], "detach") else sd[name + ".bias"]

    P = "positions_linears.%d"
    put(P % 0, W, [(0, 0, ch)])
    for l in range(1, 8):
        if l < D:
            put(P % l, W, [(0, 0, ch), (63, ch, W)] if l == 5 else [(0, 0, W)])
        else:      # an identity layer: relu(h) = h for h >= 0
            c0 = 63 if l == 5 else 0
            _set_identity(out[(P % l) + ".weight"], W, c0)
    put("views_linears.0", W, [(0, 0, W), (256, W, chv)])
    put("feature_linear", W, [(0, 0, W)])
    for name, rows, width in (("sigma_linear", 1, W), ("albedo_feature_linear", H, W), ("albedo_linear", 3, H), ("roughness_linear", 1, W),
                              ("irradiance_feature_linear", H, W), ("irradiance_linear", 1, H), ("radiance_linear", 3, W)):
        put(name, rows, [(0, 0, width)])
    for k in range(3):
        put("additional_radiance_feature_linear.%d" % k, H, [(0, 0, W)])
        put("additional_radiance_linear.%d" % k, 3, [(0, 0, H)])
    return out


def unembed_gradients(grads, arch):
    """The gradients of embed_architecture's input from those of its output: the sub-blocks the smaller network's parameters were written to."""
    D, W, L, Lv = arch
    ch, chv, H = 3 + 6 * L, 3 + 6 * Lv, W // 2
    out = OrderedDict()
    for name, o, i in arch_schema(D, W, L, Lv):
        g = grads[name + ".weight"]
        if name == "positions_linears.5":
            w = g[:o, :ch].new_zeros((o, i)) if hasattr(g, "new_zeros") else np.zeros((o, i), np.float32)
            w[:, :ch], w[:, ch:] = g[:o, :ch], g[:o, 63:63 + W]
        elif name == "views_linears.0":
            w = g[:o, :W].new_zeros((o, i)) if hasattr(g, "new_zeros") else np.zeros((o, i), np.float32)
            w[:, :W], w[:, W:] = g[:o, :W], g[:o, 256:256 + chv]
        else:
            w = g[:o, :i]
        out[name + ".weight"], out[name + ".bias"] = w, grads[name + ".bias"][:o]
    return out


# Auxiliary PositionMLPs (src/networks/MLP.py:6-30, ibl_nerf.py:312-326): the main network's trunk shape + out_linears
AUX_OUT_CH = {"albedo_mlp": 3, "roughness_mlp": 1, "irradiance_mlp": 1, "normal_mlp": 3}
TRUNK_SCHEMA = tuple(e for e in SCHEMA if e[0].startswith("positions_linears."))


def _trunk_schema(D, W, multires):
    ch = 3 + 6 * multires
    return tuple([("positions_linears.0", W, ch)] + [("positions_linears.%d" % (i + 1), W, W + ch if i == 4 else W) for i in range(D - 1)])


def synthetic_position_mlp(seed: int, out_ch: int, gain: float = 1.0, arch=None) -> "OrderedDict[str, np.ndarray]":
    """A seeded PositionMLP state dict (D=8, W=256, input_ch=63, skips=[4]; arch = (D, W, multires[, ...]) for a smaller one) in nn.Module registration order."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    schema = TRUNK_SCHEMA + (("out_linears", out_ch, 256),) if arch is None else _trunk_schema(*arch[:3]) + (("out_linears", out_ch, arch[1]),)
    for name, o, i in schema:
        sd[name + ".weight"] = (rs.randn(o, i) * gain * np.sqrt(2.0 / i)).astype(np.float32)
        sd[name + ".bias"] = (rs.randn(o) * 0.1).astype(np.float32)
    return sd


def posdir_schema(out_ch: int = 1):
    """(name, out, in) of the nn.Linear layers of a PositionDirectionMLP (src/networks/MLP.py:32-49; D=8, W=256, input_ch=63,
    input_ch_views=27, skips=[4]) in registration order — the depth_mlp of infer_depth (ibl_nerf.py:293-297)."""
    return TRUNK_SCHEMA + (("feature_linear", 256, 256), ("views_linears.0", 128, 283), ("views_linears.1", 128, 128),
                           ("views_linears.2", 128, 128), ("views_linears.3", 128, 128), ("final_linear", out_ch, 128))


def synthetic_position_direction_mlp(seed: int, out_ch: int = 1, gain: float = 1.0, arch=None) -> "OrderedDict[str, np.ndarray]":
    """A seeded PositionDirectionMLP state dict in nn.Module registration order (arch = (D, W, multires, multires_views) for a smaller one: D // 2 view layers of W // 2)."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    schema = posdir_schema(out_ch)
    if arch is not None:
        D, W, L, Lv = arch
        schema = _trunk_schema(D, W, L) + (("feature_linear", W, W), ("views_linears.0", W // 2, W + 3 + 6 * Lv)) + \
            tuple(("views_linears.%d" % l, W // 2, W // 2) for l in range(1, max(D // 2, 1))) + (("final_linear", out_ch, W // 2),)
    for name, o, i in schema:
        sd[name + ".weight"] = (rs.randn(o, i) * gain * np.sqrt(2.0 / i)).astype(np.float32)
        sd[name + ".bias"] = (rs.randn(o) * 0.1).astype(np.float32)
    return sd


def _trunk_arch(sd):
    """(D, W, multires) of a state dict's positions_linears.* (PositionMLP / PositionDirectionMLP / IBLNeRF)"""
    W, ch = (int(v) for v in sd["positions_linears.0.weight"].shape)
    D = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("positions_linears.") and k.endswith(".weight"))
    if (ch - 3) % 6:
        raise ValueError("not a positional-encoding width: input_ch %d" % ch)
    return D, W, (ch - 3) // 6


def _embed_trunk(sd, out, D, W, ch):
    """positions_linears.0-7 of the built shape from a smaller trunk (embed_architecture's rules: zero units, zero frequency columns, identity layers behind the last one)"""
    if D > 8 or D == 5 or D < 1 or W > 256 or W < 2 or ch > 63:
        raise ValueError("a trunk of D=%d, W=%d, input_ch=%d is not a member of the built architecture (D <= 8 and != 5, W <= 256, multires <= 10)" % (D, W, ch))
    P = "positions_linears.%d"
    val = lambda k: sd[k].detach() if hasattr(sd[k], "detach") else sd[k]
    for l in range(8):
        w, b = out[(P % l) + ".weight"], out[(P % l) + ".bias"]
        if l < D:
            src = val((P % l) + ".weight")
            want = (W, ch) if l == 0 else ((W, W + ch) if l == 5 else (W, W))
            if tuple(int(v) for v in src.shape) != want:
                raise ValueError("%s.weight has shape %s, expected %s" % (P % l, tuple(src.shape), want))
            if l == 0:
                w[:W, :ch] = src
            elif l == 5:
                w[:W, :ch], w[:W, 63:63 + W] = src[:, :ch], src[:, ch:]
            else:
                w[:W, :W] = src
            b[:W] = val((P % l) + ".bias")
        else:
            c0 = 63 if l == 5 else 0
            _set_identity(w, W, c0)


def embed_position_mlp(sd):
    """A smaller PositionMLP (networks/MLP.py:6-30: albedo_mlp / roughness_mlp / irradiance_mlp / normal_mlp under netdepth / netwidth / multires below the built
    8 / 256 / 10) as the member of the built shape that computes the same function (embed_architecture's rules).  A built-shape dict is returned as it is."""
    D, W, L = _trunk_arch(sd)
    if (D, W, L) == (8, 256, 10):
        return sd
    out_ch = int(sd["out_linears.weight"].shape[0])
    if tuple(int(v) for v in sd["out_linears.weight"].shape) != (out_ch, W):
        raise ValueError("out_linears.weight has shape %s, expected (%d, %d)" % (tuple(sd["out_linears.weight"].shape), out_ch, W))
    like = sd["positions_linears.0.weight"]
    zeros = (lambda *sh: np.zeros(sh, dtype=np.float32)) if isinstance(like, np.ndarray) else (lambda *sh: like.new_zeros(sh))
    out = OrderedDict()
    for name, o, i in TRUNK_SCHEMA + (("out_linears", out_ch, 256),):
        out[name + ".weight"], out[name + ".bias"] = zeros(o, i), zeros(o)
    _embed_trunk(sd, out, D, W, 3 + 6 * L)
    val = lambda k: sd[k].detach() if hasattr(sd[k], "detach") else sd[k]
    out["out_linears.weight"][:, :W] = val("out_linears.weight")
    out["out_linears.bias"][:] = val("out_linears.bias")
    return out


def embed_position_direction_mlp(sd):
    """A smaller PositionDirectionMLP (networks/MLP.py:32-74: the depth_mlp of infer_depth) inside the built shape: the trunk as above; feature_linear and the view
    layers (W // 2 wide, D // 2 of them) with zero units, zero direction frequencies and — for D < 8 — identity view layers behind the last one (their inputs are
    >= 0 after the ReLU); final_linear padded."""
    D, W, L = _trunk_arch(sd)
    H = W // 2
    chv = int(sd["views_linears.0.weight"].shape[1]) - W
    n_view = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("views_linears.") and k.endswith(".weight"))
    if (D, W, L, chv, n_view) == (8, 256, 10, 27, 4):
        return sd
    if (chv - 3) % 6 or chv > 27 or chv < 3 or n_view != max(D // 2, 1) or n_view > 4:
        raise ValueError("not a PositionDirectionMLP inside the built shape: D=%d, W=%d, input_ch_views=%d, %d view layers" % (D, W, chv, n_view))
    out_ch = int(sd["final_linear.weight"].shape[0])
    like = sd["positions_linears.0.weight"]
    zeros = (lambda *sh: np.zeros(sh, dtype=np.float32)) if isinstance(like, np.ndarray) else (lambda *sh: like.new_zeros(sh))
    out = OrderedDict()
    for name, o, i in posdir_schema(out_ch):
        out[name + ".weight"], out[name + ".bias"] = zeros(o, i), zeros(o)
    _embed_trunk(sd, out, D, W, 3 + 6 * L)
    val = lambda k: sd[k].detach() if hasattr(sd[k], "detach") else sd[k]
    out["feature_linear.weight"][:W, :W] = val("feature_linear.weight")
    out["feature_linear.bias"][:W] = val("feature_linear.bias")
    w0 = val("views_linears.0.weight")
    out["views_linears.0.weight"][:H, :W], out["views_linears.0.weight"][:H, 256:256 + chv] = w0[:, :W], w0[:, W:]
    out["views_linears.0.bias"][:H] = val("views_linears.0.bias")
    for l in range(1, 4):
        if l < n_view:
            out["views_linears.%d.weight" % l][:H, :H] = val("views_linears.%d.weight" % l)
            out["views_linears.%d.bias" % l][:H] = val("views_linears.%d.bias" % l)
        else:
            _set_identity(out["views_linears.%d.weight" % l], H, 0)
    out["final_linear.weight"][:, :H] = val("final_linear.weight")
    out["final_linear.bias"][:] = val("final_linear.bias")
    return out


def posdir_blob(sd) -> np.ndarray:
    """A PositionDirectionMLP state dict -> the flat fp32 blob of iblnerf_upload_posdir_mlp (weight [out,in] row-major, then bias,
    layer after layer).  Validates names, order and shapes."""
    if "positions_linears.0.weight" in sd and "views_linears.0.weight" in sd and "final_linear.weight" in sd:
        sd = embed_position_direction_mlp(sd)      # (a smaller network: the member of the built shape that computes the same function)
    out_ch = int(_to_numpy(sd["final_linear.bias"]).shape[0]) if "final_linear.bias" in sd else 1
    want = [n + sfx for n, _, _ in posdir_schema(out_ch) for sfx in (".weight", ".bias")]
    if list(sd.keys()) != want:
        raise KeyError("not a PositionDirectionMLP state dict (D=8, W=256, skips=[4]): %s" % list(sd.keys())[:4])
    parts = []
    for name, o, i in posdir_schema(out_ch):
        w, b = _to_numpy(sd[name + ".weight"]), _to_numpy(sd[name + ".bias"])
        if w.shape != (o, i) or b.shape != (o,):
            raise ValueError("%s: expected [%d,%d], got %s" % (name, o, i, w.shape))
        parts += [np.ascontiguousarray(w, np.float32).ravel(), np.ascontiguousarray(b, np.float32).ravel()]
    return np.concatenate(parts)


def aux_channel_blob(aux_sd, channel: int) -> np.ndarray:
    """One output channel of a PositionMLP as an IBLNeRF-schema blob for iblnerf_upload_aux_weights: its
    positions_linears.*, row `channel` of out_linears in the place of sigma_linear, zeros elsewhere."""
    if "positions_linears.0.weight" in aux_sd and "out_linears.weight" in aux_sd:
        aux_sd = embed_position_mlp(aux_sd)        # (a smaller network: the member of the built shape that computes the same function)
    want = [n + sfx for n, _, _ in TRUNK_SCHEMA for sfx in (".weight", ".bias")] + ["out_linears.weight", "out_linears.bias"]
    if list(aux_sd.keys()) != want:
        raise KeyError("not a PositionMLP state dict (D=8, W=256, skips=[4]): %s" % list(aux_sd.keys())[:4])
    ow, ob = _to_numpy(aux_sd["out_linears.weight"]), _to_numpy(aux_sd["out_linears.bias"])
    if ow.ndim != 2 or ow.shape[1] != 256 or not 0 <= channel < ow.shape[0] or ob.shape != (ow.shape[0],):
        raise ValueError("out_linears %s / channel %d" % (ow.shape, channel))
    full = OrderedDict()
    for name, o, i in SCHEMA:
        if name.startswith("positions_linears."):
            w, b = _to_numpy(aux_sd[name + ".weight"]), _to_numpy(aux_sd[name + ".bias"])
            if w.shape != (o, i) or b.shape != (o,):
                raise ValueError("%s: expected [%d,%d], got %s" % (name, o, i, w.shape))
        elif name == "sigma_linear":
            w, b = ow[channel:channel + 1], ob[channel:channel + 1]
        else:
            w, b = np.zeros((o, i), np.float32), np.zeros((o,), np.float32)
        full[name + ".weight"], full[name + ".bias"] = w, b
    return state_dict_to_blob(full)


def _to_numpy(v) -> np.ndarray:
    if isinstance(v, np.ndarray):
        return v
    return v.detach().cpu().numpy()  # torch.Tensor / nn.Parameter


def check_schema(sd):
    """Names, order and shapes of a state dict against SCHEMA (what the blob layout assumes) without touching data."""
    want = [(n + sfx, (o, i) if sfx == ".weight" else (o,)) for n, o, i in SCHEMA for sfx in (".weight", ".bias")]
    got = [(k, tuple(v.shape)) for k, v in sd.items()]
    if got != want:
        bad = next((g for g, w in zip(got, want) if g != w), got[len(want):] or want[len(got):])
        raise ValueError("state dict does not follow the IBLNeRF schema (first mismatch: %r)" % (bad,))


def state_dict_to_blob(sd) -> np.ndarray:
    """Flatten a reference-schema state dict (numpy arrays or torch tensors) into the fp32 blob."""
    parts = []
    for name, out_f, in_f in SCHEMA:
        w = np.ascontiguousarray(_to_numpy(sd[name + ".weight"]), dtype=np.float32)
        b = np.ascontiguousarray(_to_numpy(sd[name + ".bias"]), dtype=np.float32)
        if w.shape != (out_f, in_f) or b.shape != (out_f,):
            raise ValueError("state dict entry %s has shape %s/%s, expected (%d,%d)/(%d,)"
                             % (name, w.shape, b.shape, out_f, in_f, out_f))
        parts.append(w.ravel())
        parts.append(b.ravel())
    blob = np.concatenate(parts)
    assert blob.size == N_PARAMS
    return blob


def blob_to_state_dict(blob: np.ndarray) -> "OrderedDict[str, np.ndarray]":
    blob = np.asarray(blob, dtype=np.float32)
    if blob.size != N_PARAMS:
        raise ValueError("blob has %d floats, expected %d" % (blob.size, N_PARAMS))
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    off = 0
    for name, out_f, in_f in SCHEMA:
        sd[name + ".weight"] = blob[off:off + out_f * in_f].reshape(out_f, in_f).copy()
        off += out_f * in_f
        sd[name + ".bias"] = blob[off:off + out_f].copy()
        off += out_f
    return sd


def blob_checksum(blob: np.ndarray) -> str:
    """Short stable fingerprint of a blob (stored next to golden vectors instead of 3.2 MB of weights)."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(blob, dtype="<f4").tobytes()).hexdigest()[:16]


def find_checkpoint(basedir: str, expname: str, ft_path=None, target_load_N_iter: int = -1):
    """Checkpoint discovery rule of `ibl_nerf.py:345-350`: explicit path, else a named
    iteration, else the lexicographically last file whose name contains 'tar'."""
    if ft_path is not None and ft_path != "None":
        return ft_path
    if target_load_N_iter > 0:
        return os.path.join(basedir, expname, "{:06d}.tar".format(target_load_N_iter))
    d = os.path.join(basedir, expname)
    ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "tar" in f]
    return ckpts[-1] if ckpts else None


def read_checkpoint(path: str):
    """The raw dict of a reference `.tar` (torch.save, `train.py:180-191`), tensors on the CPU."""
    import torch
    return torch.load(path, map_location="cpu", weights_only=False)


def load_checkpoint(path: str):
    """Read a reference `.tar` (torch.save dict, `train.py:180-191`).

    Returns (global_step, coarse_state_dict, fine_state_dict_or_None)."""
    import torch
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    fine = ckpt.get("network_fine_state_dict")
    return ckpt.get("global_step", 0), ckpt["network_fn_state_dict"], fine


def load_checkpoint_aux(path: str):
    """The auxiliary networks a reference `.tar` may hold (ibl_nerf.py:369-374): {name: state dict} for the names present."""
    import torch
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    return {k: ckpt[k] for k in tuple(AUX_OUT_CH) + ("depth_mlp",) if k in ckpt}


def save_checkpoint(path: str, global_step: int, coarse_sd, fine_sd=None, elapsed_time: float = 0.0, aux=None):
    """Write the same dict schema `train.py:180-191` writes (optimizer state left empty).  aux: {name: state dict} of
    auxiliary networks, stored under the reference's keys ('albedo_mlp', ...)."""
    import torch
    to_t = lambda sd: OrderedDict((k, torch.from_numpy(np.array(_to_numpy(v)))) for k, v in sd.items())
    d = {"global_step": global_step, "network_fn_state_dict": to_t(coarse_sd),
         "optimizer_state_dict": {}, "elapsed_time": elapsed_time}
    if fine_sd is not None:
        d["network_fine_state_dict"] = to_t(fine_sd)
    for k, sd in (aux or {}).items():
        d[k] = to_t(sd)
    torch.save(d, path)
