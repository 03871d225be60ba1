"""Host-side mirror of the reference's model factory (src/nerf_models/ibl_nerf.py).

`IBLNeRF` here is a weight container with the reference module's state-dict surface
(`state_dict()`, `load_state_dict()`, `coarse_radiance_number`); the forward pass lives in
csrc/mlp_kernel.hip and is reached through `network_query_fn` / `render_decomp`.
`create_IBLNeRF(args)` follows ibl_nerf.py:255-428: builds both networks, finds and loads the
`.tar` checkpoint by the reference's discovery rule, and returns the same render_kwargs dicts,
so `test.py`'s call sequence works unchanged.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from . import checkpoint as ck


class IBLNeRF:
    """Weight container with the schema of the reference nn.Module (ibl_nerf.py:14-75)."""

    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,), coarse_radiance_number=3,
                 is_color_independent_to_direction=False, **_ignored):
        # The kernels are built for D=8, W=256, multires 10 / 4; a SMALLER network is evaluated as the member of that architecture computing the same function
        # (checkpoint.embed_architecture, applied at upload): D <= 8 (not 5: the reference's own forward fails there), W <= 256, multires <= 10, multires_views <= 4.
        # Round 6: anything LARGER (D <= 32, even W <= 4096, multires / multires_views <= 24) runs on csrc/generic_mlp.hip — layer by layer in exact fp32 on the matrix
        # cores, every sample of every query, no backward: correct and slow (checkpoint.is_generic_arch; iblnerf_upload_weights_arch).
        shape_ok = (input_ch - 3) % 6 == 0 and (input_ch_views - 3) % 6 == 0 and input_ch >= 3 and input_ch_views >= 3
        arch = (int(D), int(W), (int(input_ch) - 3) // 6, (int(input_ch_views) - 3) // 6) if shape_ok else None
        ok = tuple(skips) == (4,) and coarse_radiance_number == 3 and shape_ok and (ck.is_member_of_built(arch) or (ck.is_generic_arch(arch) and not is_color_independent_to_direction))
        if not ok:
            raise NotImplementedError("the HIP path is built for D=8, W=256, multires=10, multires_views=4, skips=[4], coarse_radiance_number=3, evaluates smaller "
                                      "networks (D <= 8 and != 5, W <= 256, multires <= 10, multires_views <= 4) inside it and larger ones (D <= 32 and != 5, even W <= 4096, "
                                      "multires, multires_views <= 24; not colour-independent) layer by layer; got D=%d, W=%d, input_ch=%d, "
                                      "input_ch_views=%d, skips=%s, coarse_radiance_number=%d" % (D, W, input_ch, input_ch_views, list(skips), coarse_radiance_number))
        self.arch = (int(D), int(W), (int(input_ch) - 3) // 6, (int(input_ch_views) - 3) // 6)
        self.is_color_independent_to_direction = bool(is_color_independent_to_direction)   # ibl_nerf.py:75, :192
        self.coarse_radiance_number = coarse_radiance_number
        self._sd = ck.synthetic_state_dict(seed=0) if self.arch == ck.SHIPPED_ARCH else ck.synthetic_arch_state_dict(0, self.arch)       # placeholder values until load_state_dict
        self._version = 0

    def state_dict(self):
        return OrderedDict(self._sd)

    def load_state_dict(self, sd):
        if self.arch == ck.SHIPPED_ARCH:
            self._sd = ck.blob_to_state_dict(ck.state_dict_to_blob(sd))   # validates names and shapes
        else:
            sd = OrderedDict((k, np.array(ck._to_numpy(v), dtype=np.float32)) for k, v in sd.items())
            if ck.arch_of(sd) != self.arch:
                raise ValueError("state dict of IBLNeRF%s loaded into IBLNeRF%s" % (ck.arch_of(sd), self.arch))
            if ck.is_member_of_built(self.arch):
                ck.embed_architecture(sd)                              # validates names and shapes
            else:
                ck.arch_blob(sd, self.arch)                            # likewise
            self._sd = OrderedDict((n + t, sd[n + t]) for n, _, _ in ck.arch_schema(*self.arch) for t in (".weight", ".bias"))
        self._version += 1
        for v in self._sd.values():
            v.setflags(write=False)
        # a fresh first array => renderer_for() sees a new data pointer and re-uploads
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())          # a weight container: nothing here is trainable


class PositionMLP:
    """Weight container with the schema of src/networks/MLP.py:6-30 (albedo_mlp / roughness_mlp / irradiance_mlp)."""

    def __init__(self, D=8, W=256, input_ch=63, out_ch=3, skips=(4,)):
        ok = tuple(skips) == (4,) and out_ch in (1, 3) and 1 <= D <= 8 and D != 5 and 2 <= W <= 256 and 3 <= input_ch <= 63 and (input_ch - 3) % 6 == 0
        if not ok:
            raise NotImplementedError("auxiliary networks are built for D=8, W=256, multires=10, skips=[4], out_ch 1 or 3, and evaluate smaller ones (D <= 8 and != 5, "
                                      "W <= 256, multires <= 10) inside that shape")
        self.out_ch = out_ch
        self.arch = (int(D), int(W), (int(input_ch) - 3) // 6)
        self._sd = ck.synthetic_position_mlp(0, out_ch, arch=None if self.arch == (8, 256, 10) else self.arch)
        self._version = 0

    def state_dict(self):
        return OrderedDict(self._sd)

    def load_state_dict(self, sd):
        sd = OrderedDict((k, np.array(ck._to_numpy(v), dtype=np.float32)) for k, v in sd.items())
        ck.aux_channel_blob(sd, 0)                      # validates names and shapes
        if ck._trunk_arch(sd) != self.arch:
            raise ValueError("state dict of PositionMLP%s loaded into PositionMLP%s" % (ck._trunk_arch(sd), self.arch))
        if sd["out_linears.weight"].shape[0] != self.out_ch:
            raise ValueError("out_linears has %d rows, this network %d" % (sd["out_linears.weight"].shape[0], self.out_ch))
        self._sd = sd
        self._version += 1
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())


class PositionDirectionMLP:
    """Weight container with the schema of src/networks/MLP.py:32-74 (the depth_mlp of infer_depth, ibl_nerf.py:293-297)."""

    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, out_ch=1, skips=(4,)):
        ok = (tuple(skips) == (4,) and 2 <= D <= 8 and D != 5 and 2 <= W <= 256 and 3 <= input_ch <= 63 and (input_ch - 3) % 6 == 0
              and 3 <= input_ch_views <= 27 and (input_ch_views - 3) % 6 == 0)
        if not ok:
            raise NotImplementedError("PositionDirectionMLP is built for D=8, W=256, multires=10, multires_views=4, skips=[4], and evaluates smaller ones inside that shape")
        self.out_ch = out_ch
        self.arch = (int(D), int(W), (int(input_ch) - 3) // 6, (int(input_ch_views) - 3) // 6)
        self._sd = ck.synthetic_position_direction_mlp(0, out_ch, arch=None if self.arch == ck.SHIPPED_ARCH else self.arch)
        self._version = 0

    def state_dict(self):
        return OrderedDict(self._sd)

    def load_state_dict(self, sd):
        sd = OrderedDict((k, np.array(ck._to_numpy(v), dtype=np.float32)) for k, v in sd.items())
        ck.posdir_blob(sd)                              # validates names and shapes
        if sd["final_linear.weight"].shape[0] != self.out_ch:
            raise ValueError("final_linear has %d rows, this network %d" % (sd["final_linear.weight"].shape[0], self.out_ch))
        self._sd = sd
        self._version += 1
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())


_query_ctx = {}     # id(network) -> {"ref": weakref, "r": Renderer, "w": weights key}


def _ci(network_fn):
    return bool(getattr(network_fn, "is_color_independent_to_direction", False))


def _query_renderer(network_fn):
    """A small cached context holding `network_fn`'s weights; they are re-uploaded whenever a parameter
    changed (load_state_dict, or an in-place optimizer step: torch bumps the tensors' `_version`)."""
    import weakref
    from . import renderer as R
    ent = _query_ctx.get(id(network_fn))
    if ent is None or ent["ref"]() is not network_fn:
        ent = _query_ctx[id(network_fn)] = {"ref": weakref.ref(network_fn), "r": R.Renderer(64, 0, max_rays_per_launch=1, color_independent_to_direction=_ci(network_fn), range_check="lazy"), "w": None}
        for k in [k for k, e in _query_ctx.items() if e["ref"]() is None]:
            del _query_ctx[k]
    key = R._weights_key(network_fn)
    if ent["w"] != key:
        ent["r"].load_weights(0, network_fn.state_dict())
        ent["w"] = key
    return ent["r"]


def network_query_fn(inputs, viewdirs, network_fn, renderer=None, which=0):
    """ibl_nerf.py:327-329 on the fused HIP kernel (forward only: the result carries no autograd graph).
    With `renderer`, `which` selects its coarse (0) / fine (1) weights; without, `network_fn`'s own
    weights are used through a cached context."""
    if renderer is not None:
        return renderer.network_query(inputs, viewdirs, which)
    return _query_renderer(network_fn).network_query(inputs, viewdirs, 0)


TRUNK_PARAMS = tuple("positions_linears.%d.%s" % (i, k) for i in range(8) for k in ("weight", "bias")) + ("sigma_linear.weight", "sigma_linear.bias")


def fused_trunk_query(inputs, network_fn):
    """`network_query_fn(inputs, None, network_fn)` (the trunk-only query, ibl_nerf.py:175-176) WITH autograd, both directions on the
    fused kernels: forward = the trunk query, backward = iblnerf_trunk_backward (the forward is recomputed there with the pass bits
    kept, as under activation checkpointing), which hands dL/dinputs and the gradients of positions_linears.0-7 / sigma_linear back
    to torch.  `network_fn` is the torch module whose parameters carry those names (the reference's IBLNeRF).  First derivatives only."""
    import torch
    named = dict(network_fn.named_parameters())
    params = [named[k] for k in TRUNK_PARAMS]

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, pts, *ps):
            ctx.save_for_backward(pts)
            return _query_renderer(network_fn).network_query(pts, None, 0)

        @staticmethod
        def backward(ctx, grad_out):
            (pts,) = ctx.saved_tensors
            r = _query_renderer(network_fn)
            _, dpts, grads = r.trunk_backward(pts, grad_out.contiguous()[..., 0], 0)
            return (dpts if ctx.needs_input_grad[0] else None,) + tuple(
                grads[k].reshape(p.shape) if ctx.needs_input_grad[1 + i] else None for i, (k, p) in enumerate(zip(TRUNK_PARAMS, params)))

    return _Fn.apply(inputs, *params)


def fused_trunk_features(inputs, network_fn):
    """positions_linears.0-7 of IBLNeRF.forward (ibl_nerf.py:160-170) as a torch.autograd.Function on the fused kernels: [..., 3] points ->
    [..., 256] post-ReLU trunk features; backward = iblnerf_trunk_features_backward (dL/dinputs and the 16 trunk parameter gradients)."""
    import torch
    named = dict(network_fn.named_parameters())
    names = TRUNK_PARAMS[:16]
    params = [named[k] for k in names]

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, pts, *ps):
            ctx.save_for_backward(pts)
            return _query_renderer(network_fn).trunk_features(pts, 0)

        @staticmethod
        def backward(ctx, grad_out):
            (pts,) = ctx.saved_tensors
            _, dpts, grads = _query_renderer(network_fn).trunk_backward(pts, grad_out.contiguous(), 0, features=True)
            return (dpts if ctx.needs_input_grad[0] else None,) + tuple(
                grads[k].reshape(p.shape) if ctx.needs_input_grad[1 + i] else None for i, (k, p) in enumerate(zip(names, params)))

    return _Fn.apply(inputs, *params)


FEAT2_PARAMS = TRUNK_PARAMS[:16] + ("feature_linear.weight", "feature_linear.bias", "views_linears.0.weight", "views_linears.0.bias")


def fused_trunk_features2(inputs, viewdirs, network_fn):
    """positions_linears.0-7, feature_linear and views_linears.0 (ibl_nerf.py:160-170, 193-197) as one torch.autograd.Function on the fused
    kernels: -> (h7, h2), the operands of every remaining head; backward = iblnerf_trunk_features2_backward."""
    import torch
    named = dict(network_fn.named_parameters())
    params = [named[k] for k in FEAT2_PARAMS]

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, pts, vd, *ps):
            ctx.save_for_backward(pts, vd)
            return _query_renderer(network_fn).trunk_features2(pts, vd, 0)

        @staticmethod
        def backward(ctx, g7, g2):
            pts, vd = ctx.saved_tensors
            dpts, grads = _query_renderer(network_fn).trunk_features2_backward(pts, vd, g7.contiguous(), g2.contiguous(), 0)
            return (dpts if ctx.needs_input_grad[0] else None, None) + tuple(
                grads[k].reshape(p.shape) if ctx.needs_input_grad[2 + i] else None for i, (k, p) in enumerate(zip(FEAT2_PARAMS, params)))

    return _Fn.apply(inputs, viewdirs, *params)


ALL_PARAMS = tuple(n + s for n, _, _ in ck.SCHEMA for s in (".weight", ".bias"))


def fused_network_query(inputs, viewdirs, network_fn):
    """`network_query_fn(inputs, viewdirs, network_fn)` WITH autograd, the whole network on the fused kernels in both directions: forward =
    the precise network query, backward = iblnerf_network_backward (dL/d raw rows in; dL/dinputs and all 46 parameter gradients out)."""
    import torch
    named = dict(network_fn.named_parameters())
    params = [named[k] for k in ALL_PARAMS]

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, pts, vd, *ps):
            ctx.save_for_backward(pts, vd)
            return _query_renderer(network_fn).network_query(pts, vd, 0)

        @staticmethod
        def backward(ctx, graw):
            pts, vd = ctx.saved_tensors
            dpts, grads = _query_renderer(network_fn).network_backward(pts, vd, graw.contiguous(), 0)
            return (dpts if ctx.needs_input_grad[0] else None, None) + tuple(
                grads[k].reshape(p.shape) if ctx.needs_input_grad[2 + i] else None for i, (k, p) in enumerate(zip(ALL_PARAMS, params)))

    return _Fn.apply(inputs, viewdirs, *params)


def fused_query(inputs, viewdirs, network_fn):
    """`network_query_fn(inputs, viewdirs, network_fn)` WITH autograd for a training step's gradient-carrying main query: the trunk
    and the two 256-wide layers behind it (feature_linear, views_linears.0: together 79 % of the network's FLOPs) forward and backward
    on the fused kernels (`fused_trunk_features2`), the remaining head layers — ibl_nerf.py:171-191, 199-208 — in torch on the module's own parameters."""
    import torch
    import torch.nn.functional as F
    n = network_fn
    if viewdirs is not None and not _ci(n) and len(n.views_linears) == 1 and inputs.dim() == 3 and set(ALL_PARAMS) == set(dict(n.named_parameters())):
        return fused_network_query(inputs, viewdirs, n)                                # every layer, both directions
    if viewdirs is not None and not _ci(n) and len(n.views_linears) == 1 and inputs.dim() == 3:
        # feature_linear and views_linears.0 too (79 % of the FLOPs): the heads below read h7 and h2 = relu(views_linears.0([feature, dir27]))
        h, h2 = fused_trunk_features2(inputs, viewdirs, n)
    else:
        h, h2 = fused_trunk_features(inputs, n), None
    sigma = n.sigma_linear(h)
    if viewdirs is None:
        return sigma                                                                   # :175-176
    albedo = n.albedo_linear(F.relu(n.albedo_feature_linear(h)))
    rough = n.roughness_linear(h)
    irr = n.irradiance_linear(F.relu(n.irradiance_feature_linear(h)))
    if h2 is not None:
        h = h2
    elif not _ci(n):
        d = viewdirs[:, None].expand(inputs.shape)                                     # run_network expands the directions over the samples (:244-247)
        e = torch.cat([d] + [f(d * 2.0 ** k) for k in range(4) for f in (torch.sin, torch.cos)], -1)
        h = torch.cat([n.feature_linear(h), e], -1)
        for l in n.views_linears:
            h = F.relu(l(h))
    ret = [sigma, albedo, rough, irr, n.radiance_linear(h)]
    for fl, ol in zip(n.additional_radiance_feature_linear, n.additional_radiance_linear):
        ret.append(ol(F.relu(fl(h))))
    return torch.cat(ret, -1)


def fused_composite(raw, z_vals, rays_d, renderer=None):
    """The compositing of raw2outputs (ibl_nerf_renderer.py:203-206, 241-259, 281-318) as a torch.autograd.Function on the per-ray kernels:
    raw [n, S, 18] (gradient-carrying), z_vals [n, S], rays_d [n, 3] -> (maps dict {depth_map, acc_map, albedo_map, roughness_map, irradiance_map,
    radiance_map, radiance_map_1..3}, weights [n, S]).  Gradients flow to `raw` (z_vals and rays_d are treated as constants, as the sample
    positions are in the reference's training step: sample_pdf's output is detached, nerf_renderer_helper.py:704)."""
    import torch
    from . import renderer as R
    r = renderer if renderer is not None else _composite_renderer(raw.device)

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, raw_, z_, d_):
            ctx.save_for_backward(raw_, z_, d_)
            maps, w = r.composite_direct(raw_, z_, d_)
            return maps, w

        @staticmethod
        def backward(ctx, gmaps, gw):
            raw_, z_, d_ = ctx.saved_tensors
            gmaps = torch.zeros((raw_.shape[0], 19), device=raw_.device) if gmaps is None else gmaps.contiguous()
            return r.composite_direct_backward(raw_, z_, d_, gmaps, None if gw is None else gw.contiguous()), None, None

    maps, w = _Fn.apply(raw, z_vals, rays_d)
    return {k: (maps[:, o] if n == 1 else maps[:, o:o + n]) for k, o, n in R.Renderer.MAP_SLOTS}, w


_composite_ctx = {}


def _composite_renderer(device):
    from . import renderer as R
    import torch
    index = device.index if isinstance(device, torch.device) and device.index is not None else torch.cuda.current_device()
    if index not in _composite_ctx:
        _composite_ctx[index] = R.Renderer(64, 0, max_rays_per_launch=1, device=index)
    return _composite_ctx[index]


def training_network_query_fn(grad_query_fn, fused_trunk_backward=False):
    """`network_query_fn` for `render_kwargs_train` (train.py:286-297).  In the shipped training
    configuration the eps-normal queries (ibl_nerf_renderer.py:358-361) and the reflected-ray query
    (:442-448) run under `torch.no_grad()`: 1024 trunk + 128 full evaluations per ray, 59 % of the forward
    FLOPs of a training step.  Those go to the fused kernel; a query that must carry gradients (the main
    query of each pass) is handed to `grad_query_fn`, the reference's own autograd path
    (`lambda inputs, viewdirs, network_fn: run_network(...)`, ibl_nerf.py:327-329), unchanged — except with
    `fused_trunk_backward`: then the trunk of every gradient-carrying query runs forward and backward on the fused kernels — the
    trunk-only query (viewdirs None: what the depth-gradient normal modes issue, normal_from_depth.py:36, :121) entirely
    (`fused_trunk_query`), the main query with its head layers in torch (`fused_query`)."""
    import torch

    def fn(inputs, viewdirs, network_fn):
        needs_grad = torch.is_grad_enabled() and (
            any(p.requires_grad for p in network_fn.parameters()) or getattr(inputs, "requires_grad", False)
            or getattr(viewdirs, "requires_grad", False))
        if needs_grad and fused_trunk_backward:
            return fused_trunk_query(inputs, network_fn) if viewdirs is None else fused_query(inputs, viewdirs, network_fn)
        if needs_grad:
            return grad_query_fn(inputs, viewdirs, network_fn)
        return network_query_fn(inputs, viewdirs, network_fn)
    return fn


def create_IBLNeRF(args):
    """ibl_nerf.py:255-428.  Returns (render_kwargs_train, render_kwargs_test, start, elapsed_time,
    grad_vars, optimizer) with grad_vars/optimizer = None (forward-only build)."""
    if not (0 <= args.multires <= 24 and 0 <= args.multires_views <= 24) or args.i_embed != 0:
        raise NotImplementedError("the positional encoding is built for multires, multires_views <= 24 (i_embed = 0): <= 10 / <= 4 inside the fused kernels, more on the "
                                  "layer-by-layer path (csrc/generic_mlp.hip)")
    # ibl_nerf.py:292-304: both are PositionDirectionMLPs; render_rays only ever evaluates the depth_mlp (:722-726)
    in_ch, in_chv = 3 + 6 * args.multires, 3 + 6 * args.multires_views
    depth_mlp = PositionDirectionMLP(D=args.netdepth, W=args.netwidth, input_ch=in_ch, input_ch_views=in_chv, out_ch=1) if getattr(args, "infer_depth", False) else None
    visibility_mlp = PositionDirectionMLP(D=args.netdepth, W=args.netwidth, input_ch=in_ch, input_ch_views=in_chv, out_ch=1) if getattr(args, "infer_visibility", False) else None
    aux = {name: (PositionMLP(D=args.netdepth, W=args.netwidth, input_ch=in_ch, out_ch=out_ch) if getattr(args, flag, False) else None)
           for name, flag, out_ch in (("albedo_mlp", "infer_albedo_separate", 3), ("roughness_mlp", "infer_roughness_separate", 1),
                                      ("irradiance_mlp", "infer_irradiance_separate", 1), ("normal_mlp", "infer_normal", 3))}   # ibl_nerf.py:307-326
    mk = lambda: IBLNeRF(D=args.netdepth, W=args.netwidth, input_ch=3 + 6 * args.multires, input_ch_views=3 + 6 * args.multires_views,
                         coarse_radiance_number=args.coarse_radiance_number, is_color_independent_to_direction=args.color_independent_to_direction)
    model = mk()
    model_fine = mk() if args.N_importance > 0 else None
    start, elapsed = 0, 0
    path = ck.find_checkpoint(args.basedir, args.expname, getattr(args, "ft_path", None),
                              getattr(args, "target_load_N_iter", -1))
    if path is not None and not getattr(args, "no_reload", False):
        # key accesses as ibl_nerf.py:355-376: a checkpoint without the fine network (N_importance > 0) or without the
        # normal_mlp (infer_normal) is a KeyError there, never a render with placeholder weights
        ckpt = ck.read_checkpoint(path)
        start = ckpt["global_step"]
        elapsed = ckpt.get("elapsed_time", 0)
        model.load_state_dict(ckpt["network_fn_state_dict"])
        if depth_mlp is not None:
            depth_mlp.load_state_dict(ckpt["depth_mlp"])                           # :365-366
        if aux["normal_mlp"] is not None:
            aux["normal_mlp"].load_state_dict(ckpt["normal_mlp"])
        for name in ("albedo_mlp", "roughness_mlp", "irradiance_mlp"):             # lenient `in ckpt` only for these (:369-374)
            if aux[name] is not None and name in ckpt:
                aux[name].load_state_dict(ckpt[name])
        if model_fine is not None:
            model_fine.load_state_dict(ckpt["network_fine_state_dict"])
    train = {
        "network_query_fn": network_query_fn, "perturb": args.perturb, "N_importance": args.N_importance,
        "network_fine": model_fine, "N_samples": args.N_samples, "network_fn": model,
        "use_viewdirs": args.use_viewdirs, "white_bkgd": args.white_bkgd, "raw_noise_std": args.raw_noise_std,
        "ndc": False, "lindisp": args.lindisp,
        "depth_mlp": depth_mlp, "visibility_mlp": visibility_mlp, "normal_mlp": aux["normal_mlp"], "albedo_mlp": aux["albedo_mlp"],
        "roughness_mlp": aux["roughness_mlp"], "irradiance_mlp": aux["irradiance_mlp"], "infer_depth": bool(getattr(args, "infer_depth", False)),
        "infer_visibility": bool(getattr(args, "infer_visibility", False)), "infer_normal": bool(getattr(args, "infer_normal", False)),
        "infer_normal_at_surface": getattr(args, "infer_normal_at_surface", False),
        "coarse_radiance_number": args.coarse_radiance_number,
        "use_monte_carlo_integration": getattr(args, "use_monte_carlo_integration", False),
        "use_gradient_for_incident_radiance": getattr(args, "use_gradient_for_incident_radiance", False),
        "use_radiance_linear": args.use_radiance_linear, "gamma_correct": args.gamma_correct,
        "monte_carlo_integration_method": getattr(args, "monte_carlo_integration_method", "surface"),
        "use_environment_map": False, "env_map": None, "lut_coefficient": args.lut_coefficient,
        "depth_map_from_ground_truth": args.depth_map_from_ground_truth,
        "target_normal_map_for_radiance_calculation": args.calculating_normal_type,
        "calculate_albedo_from_gt": args.calculate_albedo_from_gt,
        "calculate_roughness_from_gt": args.calculate_roughness_from_gt,
        "calculate_irradiance_from_gt": args.calculate_irradiance_from_gt,
        "epsilon": args.epsilon_for_numerical_normal,
        "epsilon_direction": getattr(args, "epsilon_direction_for_numerical_normal", 0.005),
        "N_hemisphere_sample_sqrt": getattr(args, "N_hemisphere_sample_sqrt", 16),
        "roughness_exp_coefficient": getattr(args, "roughness_exp_coefficient", 1.0),
        "albedo_multiplier": getattr(args, "albedo_multiplier", 1.0),
        "correct_depth_for_prefiltered_radiance_infer": args.correct_depth_for_prefiltered_radiance_infer,
    }
    test = dict(train)
    test["perturb"] = False                                                         # ibl_nerf.py:425-426
    test["raw_noise_std"] = 0
    return train, test, start, elapsed, None, None


def default_args(**over):
    """Effective flag values of configs/IBL-NeRF/<scene>/IBL-NeRF.txt."""
    from types import SimpleNamespace
    d = dict(multires=10, multires_views=4, i_embed=0, netdepth=8, netwidth=256, N_samples=64, N_importance=128,
             netchunk=65536, chunk=1024, coarse_radiance_number=3, color_independent_to_direction=False,
             use_illumination_feature_layer=False, use_instance_feature_layer=False, basedir=".", expname="exp",
             ft_path=None, target_load_N_iter=-1, no_reload=True, perturb=1.0, use_viewdirs=True, white_bkgd=False,
             raw_noise_std=0.0, lindisp=False, use_radiance_linear=False, gamma_correct=True, lut_coefficient="F",
             depth_map_from_ground_truth=False, calculating_normal_type="normal_map_from_depth_gradient_epsilon",
             calculate_albedo_from_gt=False, calculate_roughness_from_gt=False, calculate_irradiance_from_gt=False,
             epsilon_for_numerical_normal=0.01, correct_depth_for_prefiltered_radiance_infer=True)
    d.update(over)
    return SimpleNamespace(**d)
