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


# Auxiliary PositionMLPs (src/networks/MLP.py:6-30, ibl_nerf.py:312-326): the main network's trunk shape + out_linears
AUX_OUT_CH = {"albedo_mlp": 3, "roughness_mlp": 1, "irradiance_mlp": 1, "normal_mlp": 3}
TRUNK_SCHEMA = tuple(e for e in SCHEMA if e[0].startswith("positions_linears."))


def synthetic_position_mlp(seed: int, out_ch: int, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """A seeded PositionMLP state dict (D=8, W=256, input_ch=63, skips=[4]) in nn.Module registration order."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for name, o, i in TRUNK_SCHEMA + (("out_linears", out_ch, 256),):
        sd[name + ".weight"] = (rs.randn(o, i) * gain * np.sqrt(2.0 / i)).astype(np.float32)
        sd[name + ".bias"] = (rs.randn(o) * 0.1).astype(np.float32)
    return sd


def posdir_schema(out_ch: int = 1):
    """(name, out, in) of the nn.Linear layers of a PositionDirectionMLP (src/networks/MLP.py:32-49; D=8, W=256, input_ch=63,
    input_ch_views=27, skips=[4]) in registration order — the depth_mlp of infer_depth (ibl_nerf.py:293-297)."""
    return TRUNK_SCHEMA + (("feature_linear", 256, 256), ("views_linears.0", 128, 283), ("views_linears.1", 128, 128),
                           ("views_linears.2", 128, 128), ("views_linears.3", 128, 128), ("final_linear", out_ch, 128))


def synthetic_position_direction_mlp(seed: int, out_ch: int = 1, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """A seeded PositionDirectionMLP state dict in nn.Module registration order."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for name, o, i in posdir_schema(out_ch):
        sd[name + ".weight"] = (rs.randn(o, i) * gain * np.sqrt(2.0 / i)).astype(np.float32)
        sd[name + ".bias"] = (rs.randn(o) * 0.1).astype(np.float32)
    return sd


def posdir_blob(sd) -> np.ndarray:
    """A PositionDirectionMLP state dict -> the flat fp32 blob of iblnerf_upload_posdir_mlp (weight [out,in] row-major, then bias,
    layer after layer).  Validates names, order and shapes."""
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
