#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

Run from the repo root (only where /root/reference exists — never on the GPU box):

    python tests/golden/make_golden.py

What it does (SURVEY.md §8 c): puts /root/reference/src on sys.path, registers empty stub
modules for the reference's non-numeric imports (imageio, cv2, torchvision, configargparse),
imports `nerf_models.ibl_nerf_renderer` / `nerf_models.ibl_nerf`, builds the two `IBLNeRF`
modules through the reference's own `create_IBLNeRF`, loads OUR synthetic checkpoint
(`checkpoint.synthetic_state_dict`, regenerated from a seed — only its checksum is stored),
and calls the reference's `render_decomp`, `get_rays`, `sample_pdf` on seeded inputs.  Stage
boundaries are captured by wrapping the reference's own callables at run time (no reference
source is copied): network_query_fn, sample_pdf, get_normal_from_depth_gradient_epsilon,
raw2outputs_simple, F.grid_sample.

Outputs are data only: .npz files of inputs + expected outputs, plus a copy of the reference's
BRDF LUT asset (data/ibl_brdf_lut.png, MIT-licensed input data, md5 recorded in SURVEY.md §2 #9).
"""
import os
import shutil
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, REPO)
import _pkg  # noqa: E402

pkg = _pkg.load()
ck = pkg.checkpoint


def import_reference():
    sys.path.insert(0, os.path.join(REF, "src"))
    for m in ["imageio", "cv2", "torchvision", "torchvision.transforms", "configargparse"]:
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    import torch
    import nerf_models.ibl_nerf_renderer as R
    import nerf_models.ibl_nerf as M
    import nerf_models.nerf_renderer_helper as Hh
    torch.autograd.set_detect_anomaly(False)  # import switched it on (nerf_renderer_helper.py:2)
    return torch, R, M, Hh


def reference_args(tmp, n_importance, n_samples=64, color_independent=False, aux=False, infer_normal=False, infer_depth=False, arch=(8, 256, 10, 4)):
    """Effective flag values of configs/IBL-NeRF/kitchen/IBL-NeRF.txt.  arch = (netdepth, netwidth, multires, multires_views)."""
    os.makedirs(os.path.join(tmp, "exp"), exist_ok=True)
    return SimpleNamespace(
        multires=arch[2], multires_views=arch[3], i_embed=0, netdepth=arch[0], netwidth=arch[1], N_samples=n_samples,
        N_importance=n_importance, netchunk=65536, coarse_radiance_number=3,
        color_independent_to_direction=color_independent, use_illumination_feature_layer=False,
        use_instance_feature_layer=False, device="cpu", infer_depth=infer_depth, infer_visibility=False,
        infer_normal=infer_normal, infer_normal_at_surface=False, infer_albedo_separate=aux,
        infer_roughness_separate=aux, infer_irradiance_separate=aux, use_environment_map=False,
        N_envmap_size=16, lrate=5e-4, lrate_env_map=5e-4, basedir=tmp, expname="exp", ft_path=None,
        target_load_N_iter=-1, no_reload=True, perturb=1.0, use_viewdirs=True, white_bkgd=False,
        raw_noise_std=0.0, lindisp=False, use_monte_carlo_integration=False,
        monte_carlo_integration_method="surface", use_gradient_for_incident_radiance=False,
        use_radiance_linear=False, gamma_correct=True, lut_coefficient="F",
        depth_map_from_ground_truth=False,
        calculating_normal_type="normal_map_from_depth_gradient_epsilon",
        calculate_albedo_from_gt=False, calculate_roughness_from_gt=False,
        calculate_irradiance_from_gt=False, epsilon_for_numerical_normal=0.01,
        epsilon_direction_for_numerical_normal=0.005, N_hemisphere_sample_sqrt=16,
        roughness_exp_coefficient=1.0, albedo_multiplier=1.0,
        correct_depth_for_prefiltered_radiance_infer=True)


def load_lut(torch):
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(REF, "data", "ibl_brdf_lut.png")).convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(img).permute(2, 0, 1).contiguous()  # == test.py:79-87 (cv2 drops alpha)


def camera_rays(rng, n, H=800, W=800, fov_deg=60.0):
    """n seeded pixels of the synthetic 800x800 pinhole view (SURVEY.md §8 d), identity pose,
    reference ray convention (nerf_renderer_helper.py:36-45).  Returns pixel ids too."""
    f = np.float32(0.5 * W / np.tan(0.5 * np.deg2rad(fov_deg)))
    pix = rng.permutation(H * W)[:n]
    pix[:4] = [0, W - 1, (H - 1) * W, H * W - 1]  # frame corners: largest |d|
    i = (pix % W).astype(np.float32)
    j = (pix // W).astype(np.float32)
    d = np.stack([(i - np.float32(W / 2)) / f, -(j - np.float32(H / 2)) / f, -np.ones_like(i)], -1).astype(np.float32)
    o = np.zeros_like(d)
    return o, d, pix, f


EDIT_KEYS_OFF = dict(
    edit_intrinsic=False, editing_img_idx=0, num_edit_objects=0, edit_roughness=False, edit_albedo=False,
    edit_normal=False, edit_depth=False, edit_albedo_by_img=False, edit_normal_by_img=False,
    edit_roughness_by_img=False, edit_irradiance_by_img=False, editing_target_roughness_list=[],
    editing_target_albedo_list=[], editing_target_irradiance_list=[],
    insert_object=False, inserting_img_idx=0, num_insert_objects=0, inserting_target_roughness_list=[],
    inserting_target_irradiance_list=[], inserting_target_albedo_list=[])


class Recorder:
    """Wraps reference callables to record stage boundaries of ONE render_rays chunk."""

    def __init__(self, torch, R, kw, n_keep):
        self.torch, self.R, self.kw, self.n_keep = torch, R, kw, n_keep
        self.q, self.pdf, self.nrm, self.simple, self.lut = [], [], [], [], []
        self.depth = None

    def __enter__(self):
        R, kw = self.R, self.kw
        self._q0, self._pdf0 = kw["network_query_fn"], R.sample_pdf
        self._n0, self._s0, self._g0 = R.get_normal_from_depth_gradient_epsilon, R.raw2outputs_simple, R.F.grid_sample
        self._nd0 = R.get_normal_from_depth_gradient_direction_epsilon
        k = self.n_keep

        def q(inputs, viewdirs, fn):
            out = self._q0(inputs, viewdirs, fn)
            if fn is kw.get("depth_mlp"):        # one query per chunk at the ray origins: kept whole (teacher-forced stage test)
                self.depth = dict(pts=inputs.numpy().copy(), dirs=viewdirs.numpy().copy(), raw=out.numpy().copy())
                return out
            if any(fn is kw.get(a) for a in ("albedo_mlp", "roughness_mlp", "irradiance_mlp", "normal_mlp")):
                return out                       # auxiliary-network queries are not stage boundaries of the main network
            n = inputs.shape[0]
            if viewdirs is None:  # eps-normal query: 4 stacked copies of the ray set
                nr = n // 4
                sel = np.concatenate([np.arange(k) + s * nr for s in range(4)])
            else:
                sel = np.arange(k)
            self.q.append(dict(pts=inputs[sel].numpy().copy(),
                               dirs=None if viewdirs is None else viewdirs[sel].numpy().copy(),
                               raw=out[sel].numpy().copy()))
            return out

        def pdf(bins, weights, N, det=False, pytest=False):
            out = self._pdf0(bins, weights, N, det=det, pytest=pytest)
            self.pdf.append(dict(bins=bins.numpy().copy(), weights=weights.numpy().copy(), samples=out.numpy().copy()))
            return out

        def nrm(*a, **k_):
            out = self._n0(*a, **k_)
            self.nrm.append(out.numpy().copy())
            return out

        def nrm_dir(*a, **k_):
            out = self._nd0(*a, **k_)
            self.nrm.append(out.numpy().copy())
            return out

        def simple(*a, **k_):
            rad, coarse = self._s0(*a, **k_)
            self.simple.append(np.stack([rad.numpy()] + [c.numpy() for c in coarse], 1).copy())
            return rad, coarse

        def grid(inp, grid_, **k_):
            out = self._g0(inp, grid_, **k_)
            self.lut.append(dict(uv=grid_.numpy().copy().reshape(-1, 2), val=out.numpy().copy().reshape(3, -1).T.copy()))
            return out

        kw["network_query_fn"] = q
        R.sample_pdf, R.get_normal_from_depth_gradient_epsilon = pdf, nrm
        R.get_normal_from_depth_gradient_direction_epsilon = nrm_dir
        R.raw2outputs_simple, R.F.grid_sample = simple, grid
        return self

    def __exit__(self, *exc):
        R, kw = self.R, self.kw
        kw["network_query_fn"] = self._q0
        R.sample_pdf, R.get_normal_from_depth_gradient_epsilon = self._pdf0, self._n0
        R.get_normal_from_depth_gradient_direction_epsilon = self._nd0
        R.raw2outputs_simple, R.F.grid_sample = self._s0, self._g0


class GradRecorder:
    """The two autograd normal modes (normal_from_depth.py:16-52, :102-137) run with gradients enabled: records what they return."""

    def __init__(self, R):
        self.R, self.nrm = R, []

    def __enter__(self):
        R = self.R
        self._p0, self._d0 = R.get_normal_from_depth_gradient, R.get_normal_from_depth_gradient_direction

        def wrap(f):
            def g(*a, **k):
                out = f(*a, **k)
                self.nrm.append(out.detach().numpy().copy())
                return out
            return g
        R.get_normal_from_depth_gradient, R.get_normal_from_depth_gradient_direction = wrap(self._p0), wrap(self._d0)
        return self

    def __exit__(self, *exc):
        self.R.get_normal_from_depth_gradient, self.R.get_normal_from_depth_gradient_direction = self._p0, self._d0


def density_gradient_samples(torch, kw, o, d, near, far, n_rays, n_samples):
    """d raw[..., 0] / d pts by autograd through the reference's own query path (run_network -> embed -> IBLNeRF.forward), both networks."""
    z = torch.linspace(0., 1., n_samples) * (far - near) + near
    out = {}
    pts0 = torch.from_numpy(o[:n_rays, None, :] + d[:n_rays, None, :] * z.numpy()[None, :, None]).float()
    out["dg_pts"] = pts0.numpy().reshape(-1, 3).copy()
    for tag, net in (("c", kw["network_fn"]), ("f", kw["network_fine"])):
        if net is None:
            continue
        pts = pts0.clone().requires_grad_(True)
        with torch.enable_grad():
            raw = kw["network_query_fn"](pts, None, net)
            raw[..., 0].sum().backward()
        out["dg_sigma_" + tag] = raw[..., 0].detach().numpy().reshape(-1).copy()
        out["dg_grad_" + tag] = pts.grad.numpy().reshape(-1, 3).copy()
    return out


def trunk_backward_fixture(torch, M):
    """loss.backward() through the reference's own trunk-only query path (run_network -> embed -> IBLNeRF.forward, viewdirs=None) for
    L = sum_p c_p sigma_p, c seeded: the parameter gradients of positions_linears.0-7 and sigma_linear and dL/dpts — the known answer of
    iblnerf_trunk_backward.  Random-init network (seed 60) and the fitted coarse network, 384 points each."""
    out = {}
    for tag, sd in (("g10", ck.synthetic_state_dict(seed=60, gain=1.0)), ("fit", fitted_state_dicts()[0])):
        tmp = tempfile.mkdtemp()
        try:
            _, kw, *_ = M.create_IBLNeRF(reference_args(tmp, 0))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        net = kw["network_fn"]
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        rng = np.random.RandomState(61)
        pts = rng.uniform(-1.5, 1.5, (3, 128, 3)).astype(np.float32)
        c = rng.uniform(-1, 1, (3, 128)).astype(np.float32)
        p = torch.from_numpy(pts).requires_grad_(True)
        net.zero_grad()
        with torch.enable_grad():
            raw = kw["network_query_fn"](p, None, net)
            (raw[..., 0] * torch.from_numpy(c)).sum().backward()
        out[tag + "__pts"], out[tag + "__dsigma"] = pts.reshape(-1, 3), c.reshape(-1)
        out[tag + "__sigma"] = raw[..., 0].detach().numpy().reshape(-1).copy()
        out[tag + "__dpts"] = p.grad.numpy().reshape(-1, 3).copy()
        for name, prm in net.named_parameters():
            if name.startswith(("positions_linears.", "sigma_linear.")):
                out[tag + "__grad__" + name] = prm.grad.numpy().copy()
            else:
                assert prm.grad is None or float(prm.grad.abs().max()) == 0.0      # the trunk-only query touches nothing else
        out[tag + "__ck"] = np.array(ck.blob_checksum(ck.state_dict_to_blob(sd)))
    path = os.path.join(OUT, "trunk_backward.npz")
    np.savez_compressed(path, **out)
    print("%-28s parameter gradients of the trunk by the reference's autograd  %.2f MB" % ("trunk_backward", os.path.getsize(path) / 1e6))


def network_backward_fixture(torch, M):
    """loss.backward() through the reference's own FULL query path (run_network with view directions -> IBLNeRF.forward) for
    L = sum (draw * raw), draw seeded: every parameter gradient and dL/dpts — the known answer of iblnerf_network_backward.  Random-init
    network (seed 62), 4 rays x 48 points (a set without a ReLU pass-bit flip on the device, scratch/netbwd_seeds.py)."""
    out = {}
    sd = ck.synthetic_state_dict(seed=62, gain=1.0)
    tmp = tempfile.mkdtemp()
    try:
        _, kw, *_ = M.create_IBLNeRF(reference_args(tmp, 0))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    net = kw["network_fn"]
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    rng = np.random.RandomState(63)
    pts = rng.uniform(-1.5, 1.5, (4, 48, 3)).astype(np.float32)
    dirs = rng.uniform(-1, 1, (4, 3)).astype(np.float32)
    draw = rng.uniform(-1, 1, (4, 48, 18)).astype(np.float32)
    p = torch.from_numpy(pts).requires_grad_(True)
    net.zero_grad()
    with torch.enable_grad():
        raw = kw["network_query_fn"](p, torch.from_numpy(dirs), net)
        (raw * torch.from_numpy(draw)).sum().backward()
    out.update(pts=pts, dirs=dirs, draw=draw, raw=raw.detach().numpy().copy(), dpts=p.grad.numpy().copy(),
               ck=np.array(ck.blob_checksum(ck.state_dict_to_blob(sd))))
    for name, prm in net.named_parameters():
        out["grad__" + name] = prm.grad.numpy().copy()
    path = os.path.join(OUT, "network_backward.npz")
    np.savez_compressed(path, **out)
    print("%-28s every parameter gradient of the network by the reference's autograd  %.2f MB" % ("network_backward", os.path.getsize(path) / 1e6))


def fitted_state_dicts(which="fitted"):
    """The checkpoints fit_checkpoint.py produced with the reference's modules: "fitted" (tests/golden/fitted_ckpt.npz, scene 1: every fitted_*
    fixture) | "fitted2" (fitted2_ckpt.npz, scene 2: an independent second checkpoint, fixtures fitted2_*)."""
    f = np.load(os.path.join(OUT, which + "_ckpt.npz"))
    return ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])


def _run_fixture(name, torch, R, M, lut, *, n_rays, n_importance, gain, seed, mode="plain", n_keep=6, flags=None,
                n_samples=64, near=0.5, far=8.0, posed=False, color_independent=False, aux=False, infer_normal=False,
                fitted=False, record_floor=False, infer_depth=False, perturb=False, raw_noise_std=0.0, autograd=False, arch=None):
    tmp = tempfile.mkdtemp()
    try:
        _, kw, *_ = M.create_IBLNeRF(reference_args(tmp, n_importance, n_samples, color_independent, aux, infer_normal, infer_depth, arch=arch or ck.SHIPPED_ARCH))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    sd_c = ck.synthetic_state_dict(seed=2 * seed, gain=gain)
    sd_f = ck.synthetic_state_dict(seed=2 * seed + 1, gain=gain)
    if arch is not None:      # a smaller IBLNeRF (netdepth, netwidth, multires, multires_views): the reference builds and runs it as such
        sd_c, sd_f = ck.synthetic_arch_state_dict(2 * seed, arch, gain), ck.synthetic_arch_state_dict(2 * seed + 1, arch, gain)
    if fitted:                                   # the surface-bearing checkpoint instead of a random-init one
        sd_c, sd_f = fitted_state_dicts()
    n_keep = min(n_keep, n_rays)
    kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
    if kw["network_fine"] is not None:
        kw["network_fine"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    aux_seeds = {}
    if aux:  # infer_{albedo,roughness,irradiance}_separate: seeded PositionMLPs (src/networks/MLP.py) beside the main networks
        for j, aux_name in enumerate(("albedo_mlp", "roughness_mlp", "irradiance_mlp")):
            aux_seeds[aux_name] = 100 * seed + j
            sd_a = ck.synthetic_position_mlp(aux_seeds[aux_name], ck.AUX_OUT_CH[aux_name], gain, arch)
            kw[aux_name].load_state_dict({k: torch.from_numpy(v) for k, v in sd_a.items()})
    if infer_normal:  # normal_mlp: a PositionMLP with three outputs (ibl_nerf.py:307-310)
        aux_seeds["normal_mlp"] = 100 * seed + 3
        sd_a = ck.synthetic_position_mlp(aux_seeds["normal_mlp"], 3, gain, arch)
        kw["normal_mlp"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_a.items()})
        assert kw["infer_normal"] is True
    if infer_depth:   # depth_mlp: a PositionDirectionMLP evaluated once per ray at the origin (ibl_nerf_renderer.py:722-726)
        aux_seeds["depth_mlp"] = 100 * seed + 1   # (a seed whose outputs straddle 0: the relu of :724 is exercised)
        sd_a = ck.synthetic_position_direction_mlp(aux_seeds["depth_mlp"], 1, gain, arch)
        kw["depth_mlp"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_a.items()})
        assert kw["infer_depth"] is True
    kw.update(near=near, far=far)
    kw["brdf_lut"] = lut
    if perturb:   # training-time sampling through the reference's own deterministic test hook (pytest=True: numpy seed 0 per chunk)
        kw.update(perturb=1.0, pytest=True, raw_noise_std=raw_noise_std)
    kw.update(flags or {})            # flag variants outside the shipped configs

    rng = np.random.RandomState(1000 + seed)
    o, d, pix, focal = camera_rays(rng, n_rays)
    if posed:  # a rotated + translated camera: rays_o != 0 and rays_d = R dirs, as get_rays builds them (:41-44)
        q, _ = np.linalg.qr(np.eye(3) + 0.3 * rng.randn(3, 3))
        q = (q * np.sign(np.linalg.det(q))).astype(np.float32)
        d = np.sum(d[:, None, :] * q, -1).astype(np.float32)
        o = np.broadcast_to(np.array([0.3, -0.2, 0.4], np.float32), d.shape).copy()
    edit = dict(EDIT_KEYS_OFF)
    gt = {}
    if mode == "edit":  # configs/IBL-NeRF/kitchen/edit_intrinsic.txt:8-16 (+ by-list albedo on a 2nd object)
        edit.update(edit_intrinsic=True, num_edit_objects=2, edit_roughness=True, edit_normal=True,
                    edit_normal_by_img=True, editing_target_roughness_list=[0.0, 0.35],
                    edit_albedo=True, editing_target_albedo_list=[0.8, 0.1, 0.2, 0.3, 0.6, 0.9])
        level = rng.choice([0, 10, 20], size=n_rays, p=[0.5, 0.3, 0.2]).astype(np.float32) / np.float32(255)
        gt["edit_intrinsic_mask"] = np.repeat(level[:, None], 3, 1).astype(np.float32)
        gt["edit_normal"] = rng.uniform(0, 1, (n_rays, 3)).astype(np.float32)
    elif mode == "edit2":  # the image-driven edits: depth and albedo from images, one object, no normal edit
        edit.update(edit_intrinsic=True, num_edit_objects=1, edit_depth=True, edit_albedo=True, edit_albedo_by_img=True,
                    edit_roughness=True, editing_target_roughness_list=[0.6],
                    editing_target_albedo_list=[0.5, 0.5, 0.5])  # must be non-empty (:383 assert) though the image wins
        level = rng.choice([0, 10], size=n_rays, p=[0.5, 0.5]).astype(np.float32) / np.float32(255)
        gt["edit_intrinsic_mask"] = np.repeat(level[:, None], 3, 1).astype(np.float32)
        gt["edit_depth"] = rng.uniform(1, 3, (n_rays, 1)).astype(np.float32)
        gt["edit_albedo"] = rng.uniform(0, 1, (n_rays, 3)).astype(np.float32)
    elif mode == "edit3":  # edit_roughness_by_img (ibl_nerf_renderer.py:394-395): EVERY masked pixel takes the FIRST masked row of gt_values["edit_roughness"]
        edit.update(edit_intrinsic=True, num_edit_objects=2, edit_roughness=True, edit_roughness_by_img=True,
                    editing_target_roughness_list=[0.9, 0.9])   # must be non-empty (:392 assert) though the image wins
        level = rng.choice([0, 10, 20], size=n_rays, p=[0.5, 0.3, 0.2]).astype(np.float32) / np.float32(255)
        level[:3] = 0                                             # (the first masked row is not row 0)
        gt["edit_intrinsic_mask"] = np.repeat(level[:, None], 3, 1).astype(np.float32)
        gt["edit_roughness"] = rng.uniform(0.05, 0.95, (n_rays, 1)).astype(np.float32)
    elif mode == "gtnormal":  # target_normal_map_for_radiance_calculation = "ground_truth" (the parser's default), one edit on top
        edit.update(edit_intrinsic=True, num_edit_objects=1, edit_roughness=True, editing_target_roughness_list=[0.25])
        level = rng.choice([0, 10], size=n_rays, p=[0.6, 0.4]).astype(np.float32) / np.float32(255)
        gt["edit_intrinsic_mask"] = np.repeat(level[:, None], 3, 1).astype(np.float32)
        gt["normal"] = rng.uniform(0, 1, (n_rays, 3)).astype(np.float32)
    elif mode == "fromgt":  # the four *_from_gt flags with an intrinsic edit on top: the target maps no longer alias the network's
        edit.update(edit_intrinsic=True, num_edit_objects=2, edit_depth=True, edit_roughness=True,
                    editing_target_roughness_list=[0.15, 0.7], edit_albedo=True,
                    editing_target_albedo_list=[0.8, 0.1, 0.2, 0.3, 0.6, 0.9])
        level = rng.choice([0, 10, 20], size=n_rays, p=[0.5, 0.25, 0.25]).astype(np.float32) / np.float32(255)
        gt["edit_intrinsic_mask"] = np.repeat(level[:, None], 3, 1).astype(np.float32)
        gt["edit_depth"] = rng.uniform(1, 3, (n_rays, 1)).astype(np.float32)
    elif mode == "fromgt_insert":  # gt irradiance (RGB) under an object insertion: the scalar irradiance fills all channels
        edit.update(insert_object=True, num_insert_objects=2, inserting_target_roughness_list=[0.9, 0.4],
                    inserting_target_albedo_list=[0.870588, 0.3215686, 0.443137254, .05, .05, .05],
                    inserting_target_irradiance_list=[0.5, 0.0])
        level = rng.choice([0, 10, 20], size=n_rays, p=[0.4, 0.3, 0.3]).astype(np.float32) / np.float32(255)
        gt["object_insert_mask"] = np.repeat(level[:, None], 3, 1).astype(np.float32)
        gt["object_insert_depth"] = rng.uniform(1, 2, (n_rays, 1)).astype(np.float32)
        gt["object_insert_normal"] = rng.uniform(0, 1, (n_rays, 3)).astype(np.float32)
    elif mode == "insert":  # configs/IBL-NeRF/living-room-2/object_insert.txt:8-14
        edit.update(insert_object=True, num_insert_objects=4, inserting_target_roughness_list=[1, 1, 1, 1],
                    inserting_target_albedo_list=[0.870588, 0.3215686, 0.443137254, .05, .05, .05, .2, .2, .2, .05, .05, .05],
                    inserting_target_irradiance_list=[0.5, 0.1, 0.2, 0.0])
        level = rng.choice([0, 10, 20, 30, 40], size=n_rays, p=[0.4, 0.15, 0.15, 0.15, 0.15]).astype(np.float32) / np.float32(255)
        gt["object_insert_mask"] = np.repeat(level[:, None], 3, 1).astype(np.float32)
        gt["object_insert_depth"] = rng.uniform(1, 2, (n_rays, 1)).astype(np.float32)
        gt["object_insert_normal"] = rng.uniform(0, 1, (n_rays, 3)).astype(np.float32)

    fl = flags or {}   # rows the *_from_gt flags read (3-channel images as load_mitsuba.py hands them over)
    if fl.get("calculate_albedo_from_gt"):
        gt["albedo"] = rng.uniform(0, 1, (n_rays, 3)).astype(np.float32)
    if fl.get("calculate_roughness_from_gt"):
        gt["roughness"] = np.repeat(rng.uniform(0, 1, (n_rays, 1)), 3, 1).astype(np.float32)
    if fl.get("calculate_irradiance_from_gt"):
        gt["irradiance"] = rng.uniform(0, 2, (n_rays, 3)).astype(np.float32)
    if fl.get("depth_map_from_ground_truth"):
        gt["depth"] = np.repeat(rng.uniform(1, 4, (n_rays, 1)), 3, 1).astype(np.float32)

    rays = torch.from_numpy(np.stack([o, d], 0))
    gt_t = {k: torch.from_numpy(v.copy()) for k, v in gt.items()}
    K = np.array([[focal, 0, 400], [0, focal, 400], [0, 0, 1]], dtype=np.float32)
    if autograd:   # the autograd normal modes only run with gradients enabled (train.py:286-297; test.py's no_grad makes .backward() raise)
        with torch.enable_grad(), GradRecorder(R) as rec:
            ret = R.render_decomp(800, 800, K, chunk=n_rays, rays=rays, gt_values=gt_t,
                                  approximate_radiance=True, **kw, **edit)
        ret = {k: v.detach() for k, v in ret.items()}
    else:
        with torch.no_grad(), Recorder(torch, R, kw, n_keep) as rec:
            ret = R.render_decomp(800, 800, K, chunk=n_rays, rays=rays, gt_values=gt_t,
                                  approximate_radiance=True, **kw, **edit)

    floor = {}
    if record_floor:
        # the reference's own round-off sensitivity: the same render in float64 (modules, rays, gt rows, LUT) against the float32
        # run, relative L-inf per map — the yardstick for channels whose conditioning depends on the checkpoint
        nets = [n for n in (kw["network_fn"], kw["network_fine"]) if n is not None]
        for n_ in nets:
            n_.double()
        kw["brdf_lut"] = lut.double()
        q32 = kw["network_query_fn"]              # a few of the reference's intermediates are created as float32 whatever the inputs are
        kw["network_query_fn"] = lambda inputs, viewdirs, fn: q32(inputs.double(), None if viewdirs is None else viewdirs.double(), fn)
        with (torch.enable_grad() if autograd else torch.no_grad()):
            ret64 = R.render_decomp(800, 800, K, chunk=n_rays, rays=rays.double(), gt_values={k: v.double() for k, v in gt_t.items()},
                                    approximate_radiance=True, **kw, **edit)
        ret64 = {k: v.detach() for k, v in ret64.items()}
        for n_ in nets:
            n_.float()
        kw["brdf_lut"] = lut
        kw["network_query_fn"] = q32
        for k, v in ret.items():
            a, b = v.double().numpy(), ret64[k].double().numpy()
            floor[k] = float(np.nanmax(np.abs(a - b)) / max(float(np.nanmax(np.abs(b))), 1e-30))

    out = dict(rays_o=o, rays_d=d, pix=pix.astype(np.int64), near=np.float32(near), far=np.float32(far),
               gain=np.float64(gain), seed_coarse=np.int64(2 * seed), seed_fine=np.int64(2 * seed + 1),
               n_importance=np.int64(n_importance), n_samples=np.int64(n_samples),
               ck_coarse=np.array(ck.weights_checksum(sd_c)),
               ck_fine=np.array(ck.weights_checksum(sd_f)),
               mode=np.array(mode))
    if arch is not None:
        out["arch"] = np.asarray(arch, dtype=np.int64)
    for k, v in (flags or {}).items():
        out["flag__" + k] = np.asarray(v)
    if fitted:
        out["ckpt"] = np.array("fitted")
    if infer_normal:
        out["flag__infer_normal"] = np.asarray(True)
    if infer_depth:
        out["flag__infer_depth"] = np.asarray(True)
    if perturb:
        out["perturb"] = np.float32(1.0)
        out["chunk"] = np.int64(n_rays)
        out["raw_noise_std"] = np.float32(raw_noise_std)
    if color_independent:
        out["model__color_independent_to_direction"] = np.asarray(True)
    for aux_name, sd_seed in aux_seeds.items():
        out["aux__" + aux_name] = np.int64(sd_seed)
    for k, v in gt.items():
        out["gt__" + k] = v
    for k, v in edit.items():
        if isinstance(v, list):
            out["edit__" + k] = np.asarray(v, dtype=np.float32)
        else:
            out["edit__" + k] = np.asarray(v)
    for k, v in ret.items():
        out["out__" + k] = v.numpy().astype(np.float32) if v.dtype.is_floating_point else v.numpy()
    for k, v in floor.items():
        out["floor__" + k] = np.float64(v)
    passes = ["c", "f"] if n_importance > 0 else ["c"]
    if autograd:
        for pi, p_ in enumerate(passes):
            out["normal_raw_%s" % p_] = rec.nrm[pi]
        out.update(density_gradient_samples(torch, kw, o, d, near, far, n_keep, n_samples))
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-28s %4d rays  %s (autograd normal)  %.2f MB" % (name, n_rays, mode, os.path.getsize(path) / 1e6))
        return
    # stage boundaries; query order inside one raw2outputs: main, eps-normal(4x), reflected
    gt_normals = (flags or {}).get("target_normal_map_for_radiance_calculation") in ("ground_truth", "inferred_normal_map")
    nq = 2 if gt_normals else 3                          # no eps-normal query in the ground-truth normal mode
    for pi, p in enumerate(passes):
        qs = rec.q[nq * pi:nq * pi + nq]
        main, refl = qs[0], qs[-1]
        out["q_%s_main_pts" % p], out["q_%s_main_dirs" % p], out["q_%s_main_raw" % p] = main["pts"], main["dirs"], main["raw"]
        out["q_%s_refl_pts" % p], out["q_%s_refl_dirs" % p], out["q_%s_refl_raw" % p] = refl["pts"], refl["dirs"], refl["raw"]
        if not gt_normals:
            eps = qs[1]
            out["q_%s_eps_pts" % p], out["q_%s_eps_sigma" % p] = eps["pts"], eps["raw"]
            out["normal_raw_%s" % p] = rec.nrm[pi]      # before edit/insert overrides
        out["prefiltered_env_%s" % p] = rec.simple[pi]  # [N,4,3] linear (pre-gamma)
        out["lut_uv_%s" % p], out["lut_val_%s" % p] = rec.lut[pi]["uv"], rec.lut[pi]["val"]
    if rec.depth is not None:
        out["q_depth_pts"], out["q_depth_dirs"], out["q_depth_raw"] = rec.depth["pts"], rec.depth["dirs"], rec.depth["raw"]
    if n_importance > 0:
        out["pdf_bins"], out["pdf_weights"], out["pdf_samples"] = rec.pdf[0]["bins"], rec.pdf[0]["weights"], rec.pdf[0]["samples"]
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %4d rays  %s  color range [%.3f, %.3f]  %.2f MB" % (
        name, n_rays, mode, float(ret["color_map"].min()), float(ret["color_map"].max()), os.path.getsize(path) / 1e6))


def seam_fixture(name, torch, R, M, lut, kind, seed):
    """Two arguments of render_decomp itself (ibl_nerf_renderer.py:759-813) on random-init networks, 64 + 128 samples:
    "staticcam"  c2w + c2w_staticcam (:791-794): the rays come from the static camera, `viewdirs` from the other pose — they reach nothing but the
                 depth_mlp query of infer_depth (:722-726; every network query takes rays_d), so the fixture runs with infer_depth;
    "nearfar"    per-ray near / far planes as [n, 1] tensors (:802-805): a z grid per ray (:668-674) and a per-ray depth_0 in the mip level (:456)."""
    tmp = tempfile.mkdtemp()
    try:
        _, kw, *_ = M.create_IBLNeRF(reference_args(tmp, 128, 64, infer_depth=(kind == "staticcam")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    sd_c, sd_f = ck.synthetic_state_dict(seed=2 * seed, gain=1.0), ck.synthetic_state_dict(seed=2 * seed + 1, gain=1.0)
    kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
    kw["network_fine"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    kw["brdf_lut"] = lut
    rng = np.random.RandomState(2000 + seed)
    out = dict(gain=np.float64(1.0), seed_coarse=np.int64(2 * seed), seed_fine=np.int64(2 * seed + 1), n_importance=np.int64(128), n_samples=np.int64(64),
               ck_coarse=np.array(ck.blob_checksum(ck.state_dict_to_blob(sd_c))), ck_fine=np.array(ck.blob_checksum(ck.state_dict_to_blob(sd_f))),
               mode=np.array(kind))

    def pose(shift):
        q, _ = np.linalg.qr(np.eye(3) + 0.3 * rng.randn(3, 3))
        q = (q * np.sign(np.linalg.det(q))).astype(np.float32)
        return np.concatenate([q, np.asarray(shift, np.float32).reshape(3, 1)], 1).astype(np.float32)

    if kind == "staticcam":
        aux_seed = 100 * seed + 1
        kw["depth_mlp"].load_state_dict({k: torch.from_numpy(v) for k, v in ck.synthetic_position_direction_mlp(aux_seed, 1, 1.0).items()})
        H, W = 10, 12
        f = np.float32(0.5 * W / np.tan(0.5 * np.deg2rad(60.0)))
        K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=np.float32)
        c2w, c2w_static = pose([0.3, -0.2, 0.4]), pose([-0.1, 0.25, 0.2])
        kw.update(near=0.5, far=8.0)
        with torch.no_grad():
            ret = R.render_decomp(H, W, K, chunk=H * W, c2w=torch.from_numpy(c2w), c2w_staticcam=torch.from_numpy(c2w_static), gt_values={},
                                  approximate_radiance=True, **kw, **EDIT_KEYS_OFF)
            plain = R.render_decomp(H, W, K, chunk=H * W, c2w=torch.from_numpy(c2w_static), gt_values={}, approximate_radiance=True, **kw, **EDIT_KEYS_OFF)
        # (what the argument changes, recorded as a fact: only inferred_depth_map differs from a plain render from the static pose)
        assert all(torch.equal(ret[k], plain[k]) for k in ret if k != "inferred_depth_map") and not torch.equal(ret["inferred_depth_map"], plain["inferred_depth_map"])
        out.update(H=np.int64(H), W=np.int64(W), K=K, c2w=c2w, c2w_staticcam=c2w_static, near=np.float32(0.5), far=np.float32(8.0))
        out["aux__depth_mlp"] = np.int64(aux_seed)
        out["flag__infer_depth"] = np.asarray(True)
        # f-3 leftover (round 5): the same argument through the two render types only a training run takes — is_depth_only (:197-198) and approximate_radiance=False
        with torch.no_grad():
            extra = {"depthonly": R.render_decomp(H, W, K, chunk=H * W, c2w=torch.from_numpy(c2w), c2w_staticcam=torch.from_numpy(c2w_static), gt_values={},
                                                  approximate_radiance=False, is_depth_only=True, **kw, **EDIT_KEYS_OFF),
                     "direct": R.render_decomp(H, W, K, chunk=H * W, c2w=torch.from_numpy(c2w), c2w_staticcam=torch.from_numpy(c2w_static), gt_values={},
                                               approximate_radiance=False, **kw, **EDIT_KEYS_OFF)}
    else:
        n = 96
        o, d, pix, focal = camera_rays(rng, n)
        near = (0.4 + 0.6 * rng.uniform(0, 1, (n, 1))).astype(np.float32)
        far = (6.0 + 3.0 * rng.uniform(0, 1, (n, 1))).astype(np.float32)
        K = np.array([[focal, 0, 400], [0, focal, 400], [0, 0, 1]], dtype=np.float32)
        with torch.no_grad():
            ret = R.render_decomp(800, 800, K, chunk=n, rays=torch.from_numpy(np.stack([o, d], 0)), near=torch.from_numpy(near), far=torch.from_numpy(far),
                                  gt_values={}, approximate_radiance=True, **kw, **EDIT_KEYS_OFF)
        out.update(rays_o=o, rays_d=d, pix=pix.astype(np.int64), near=near, far=far)
        with torch.no_grad():
            extra = {"depthonly": R.render_decomp(800, 800, K, chunk=n, rays=torch.from_numpy(np.stack([o, d], 0)), near=torch.from_numpy(near), far=torch.from_numpy(far),
                                                  gt_values={}, approximate_radiance=False, is_depth_only=True, **kw, **EDIT_KEYS_OFF),
                     "direct": R.render_decomp(800, 800, K, chunk=n, rays=torch.from_numpy(np.stack([o, d], 0)), near=torch.from_numpy(near), far=torch.from_numpy(far),
                                               gt_values={}, approximate_radiance=False, **kw, **EDIT_KEYS_OFF)}
    for k, v in ret.items():
        out["out__" + k] = v.numpy().astype(np.float32) if v.dtype.is_floating_point else v.numpy()
    for tag, res_ in extra.items():
        for k, v in res_.items():
            out["%s__out__%s" % (tag, k)] = v.numpy().astype(np.float32) if v.dtype.is_floating_point else v.numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %s  %d maps  %.2f MB" % (name, kind, len(ret), os.path.getsize(path) / 1e6))


def train_step_fixture(torch, R, M, lut, n_rays=64, fixture="train_step", phases=("warmup", "full", "frozen", "depth"), raw_noise_std=0.0, from_gt=(), override=None,
                       color_independent=False, planes=False, aux=False, stable_rays=False, incident_gradient=False, arch=None, sparse_grads=False):
    """loss.backward() of a training step through the reference's own render_decomp (train.py:285-297, :326-441, :479-481) on the
    fitted checkpoint: render_kwargs_train (perturb = 1) with its pytest hook for deterministic draws, gradients enabled, and the losses of
    train.py that need no dataset: radiance (fine + coarse pass, :332), the three coarse radiances (:336-341), approximated radiance
    (color_map, :330, from iteration N_iter_ignore_approximated_radiance on), albedo prior (:396, beta_prior_albedo), irradiance prior and
    regulariser (:399, :404: MSE against targets) and the roughness initialisation (:411-412, before that iteration); the targets are seeded
    arrays (a loss is a loss).  Records the loss and dL/d(every parameter of network_fn and network_fine), for four phases:
        warmup    approximate_radiance=False  (the first N_iter_ignore_approximated_radiance iterations, :295)
        full      approximate_radiance=True
        frozen    approximate_radiance=True with freeze_radiance = freeze_roughness = True on both networks (:279-283: forward_freezed)
        depth     is_depth_only=True, approximate_radiance=False (:366-374): forward only (its depth_map is detached in the loss)
    override: None | "edit" | "insert" (f-3 leftover, round 5): the step with the edit / insert overrides of raw2outputs on (:218-256, :378-410) — tests/frame_overrides.py's
    analytic images at the fixture's pixels; "edit" = config 4's kwargs plus an albedo list and a depth image, so that all four masked assignments are exercised.
    color_independent: both networks with is_color_independent_to_direction=True (ibl_nerf.py:75, :192: the radiance heads read the trunk's output, feature_linear and
    views_linears are unused and get no gradient) — the fitted checkpoint's weights taken as such a network's."""
    tmp = tempfile.mkdtemp()
    try:
        kw, _, *_ = M.create_IBLNeRF(reference_args(tmp, 128, aux=aux, infer_normal=aux, arch=arch or ck.SHIPPED_ARCH))      # [0] = render_kwargs_train
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert kw["perturb"] == 1.0
    aux_seeds = {}
    if aux:      # f-3 leftover (round 5): the auxiliary networks (ibl_nerf.py:293-323: all registered with the optimizer) with seeded weights, trainable
        for j, aux_name in enumerate(("albedo_mlp", "roughness_mlp", "irradiance_mlp", "normal_mlp")):
            aux_seeds[aux_name] = 7100 + j
            kw[aux_name].load_state_dict({k: torch.from_numpy(v) for k, v in ck.synthetic_position_mlp(aux_seeds[aux_name], ck.AUX_OUT_CH.get(aux_name, 3), 1.0).items()})
        # (no depth_mlp: with infer_depth the reference's own loss.backward() raises — render_rays squeezes the ReLU's output in place, ibl_nerf_renderer.py:724-725:
        # "one of the variables needed for gradient computation has been modified by an inplace operation ... output 0 of ReluBackward0")
        assert kw["infer_normal"] is True
    sd_c, sd_f = fitted_state_dicts()
    if arch is not None:      # a smaller IBLNeRF (netdepth, netwidth, multires, multires_views) with seeded weights: there is no fitted checkpoint of that shape
        sd_c, sd_f = ck.synthetic_arch_state_dict(8100, arch, 1.5), ck.synthetic_arch_state_dict(8101, arch, 1.5)
    kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
    kw["network_fine"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    kw.update(near=0.5, far=8.0, pytest=True)
    if planes:      # per-ray near / far planes as [n, 1] tensors (:802-805): a z grid per ray (:668-674) and a per-ray depth_0 in the mip level (:455-457)
        prng = np.random.RandomState(4300)
        plane_near = (0.4 + 0.3 * prng.uniform(0, 1, (n_rays, 1))).astype(np.float32)
        plane_far = (7.0 + 2.0 * prng.uniform(0, 1, (n_rays, 1))).astype(np.float32)
        kw.update(near=torch.from_numpy(plane_near), far=torch.from_numpy(plane_far))
    if raw_noise_std > 0:       # train.py's raw_noise_std (:208-216, :242): density noise on the main query of each pass; the pytest hook draws it uniform
        kw["raw_noise_std"] = raw_noise_std
    # from_gt: names of the ground-truth substitutions switched on (render kwargs, ibl_nerf.py:411-416; raw2outputs :251-252, :320-330) with seeded gt_values
    # (train.py:294 passes target_info): the target maps become constants, so no gradient reaches the network's own map through the shading
    gt_rng = np.random.RandomState(4200)
    gt_values = {}
    for flag, key, ch, lo, hi in (("calculate_albedo_from_gt", "albedo", 3, 0.15, 0.85), ("calculate_roughness_from_gt", "roughness", 1, 0.1, 0.9),
                                  ("calculate_irradiance_from_gt", "irradiance", 3, 0.2, 1.2), ("depth_map_from_ground_truth", "depth", 1, 1.5, 5.0)):
        if flag in from_gt:
            kw[flag] = True
            gt_values[key] = gt_rng.uniform(lo, hi, (n_rays, ch)).astype(np.float32)
    kw["brdf_lut"] = lut
    if incident_gradient:      # use_gradient_for_incident_radiance (:442-453): the reflected-ray query runs with gradients
        kw["use_gradient_for_incident_radiance"] = True
    out_extra = {"incident_gradient": np.asarray(bool(incident_gradient))}
    if color_independent:
        for k_ in ("network_fn", "network_fine"):
            kw[k_].is_color_independent_to_direction = True
    rng = np.random.RandomState(4100)
    o, d, pix, focal = camera_rays(rng, n_rays)
    if stable_rays:
        # The fine network's (and the auxiliary networks') gradients depend on which BIN of the inverse CDF each stochastic fine sample falls into; a draw u within a few
        # ulps of a cdf entry lands in another bin under any other rounding of the coarse weights (tests/test_gpu_training.py: z_std of such a ray moves by 4e-3).  For a
        # fixture that pins a gradient PATH rather than that threshold, rays whose closest |u - cdf| is below 3e-6 (50 ulps of the cdf) in the reference's own run are replaced (the draw
        # of row i belongs to row i — numpy's seed-0 stream — so the replacement keeps its row) until none is left.
        pdf0, cap = R.sample_pdf, {}

        def spy(bins, weights, N, det=False, pytest=False):
            w_ = weights + 1e-5
            pdf_ = w_ / torch.sum(w_, -1, keepdim=True)
            cdf_ = torch.cat([torch.zeros_like(pdf_[..., :1]), torch.cumsum(pdf_, -1)], -1)
            np.random.seed(0)
            u_ = np.random.rand(*(list(cdf_.shape[:-1]) + [N]))
            pn = pdf_.detach().numpy().astype(np.float64)
            big = np.concatenate([pn[:, :1], np.maximum(pn[:, 1:], pn[:, :-1]), pn[:, -1:]], -1) > 1e-3      # cdf entries beside a bin that holds mass (a flip between two empty bins moves a weightless sample)
            dist = np.abs(u_[:, :, None] - cdf_.detach().numpy().astype(np.float64)[:, None, :])
            cap["margin"] = np.where(big[:, None, :], dist, np.inf).min((-1, -2))
            return pdf0(bins, weights, N, det=det, pytest=pytest)

        for _ in range(12):
            R.sample_pdf = spy
            try:
                with torch.no_grad():
                    R.render_decomp(800, 800, np.array([[focal, 0, 400], [0, focal, 400], [0, 0, 1]], dtype=np.float32), chunk=n_rays, rays=torch.from_numpy(np.stack([o, d], 0)),
                                    gt_values={}, approximate_radiance=False, **kw, **EDIT_KEYS_OFF)
            finally:
                R.sample_pdf = pdf0
            crit = np.flatnonzero(cap["margin"] < 3e-6)
            if not len(crit):
                break
            o2, d2, pix2, _ = camera_rays(rng, len(crit) + 4)      # (camera_rays pins its first four rays to the frame's corners)
            o[crit], d[crit], pix[crit] = o2[4:], d2[4:], pix2[4:]
        else:
            raise RuntimeError("stable_rays: still threshold-critical rays after 12 rounds")
        stable_margin = float(cap["margin"].min())
    edit_kw = dict(EDIT_KEYS_OFF)
    if override is not None:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import frame_overrides as FO
        if override == "edit":
            edit_kw.update(FO.EDIT_CFG4, edit_albedo=True, editing_target_albedo_list=[0.8, 0.2, 0.3], edit_depth=True)
            gt_values.update(FO.edit_rows(pix))
            gt_values["edit_depth"] = (2.0 + 0.5 * (pix % 800).astype(np.float32) / np.float32(800))[:, None].astype(np.float32)
        else:
            edit_kw.update(FO.INSERT_CFG5)
            gt_values.update(FO.insert_rows(pix))
    K = np.array([[focal, 0, 400], [0, focal, 400], [0, 0, 1]], dtype=np.float32)
    rays = torch.from_numpy(np.stack([o, d], 0))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import train_loss as TL
    tg, beta = TL.targets(rng, n_rays), TL.BETA
    if aux:
        tg.update(TL.aux_targets(rng, n_rays))
    out = dict(rays_o=o, rays_d=d, pix=pix.astype(np.int64), near=plane_near if planes else np.float32(0.5), far=plane_far if planes else np.float32(8.0), chunk=np.int64(n_rays),
               ck_coarse=np.array(ck.blob_checksum(ck.state_dict_to_blob(ck.embed_architecture(sd_c)))),
               ck_fine=np.array(ck.blob_checksum(ck.state_dict_to_blob(ck.embed_architecture(sd_f)))), ckpt=np.array("fitted" if arch is None else "arch"))
    if arch is not None:
        out["arch"] = np.asarray(arch, dtype=np.int64)
    for k, v in tg.items():
        out["target__" + k] = v
    for k, v in beta.items():
        out["beta__" + k] = np.float64(v)
    nets = (("c", kw["network_fn"]), ("f", kw["network_fine"])) + tuple((nm, kw[nm]) for nm in aux_seeds)
    for nm, sd_seed in aux_seeds.items():
        out["aux__" + nm] = np.int64(sd_seed)

    out["raw_noise_std"] = np.float32(raw_noise_std)
    for k, v in gt_values.items():
        out["gt__" + k] = v
    out["from_gt"] = np.array(sorted(from_gt))
    if stable_rays:
        out["stable_rays_min_margin"] = np.float64(stable_margin)
    out.update(out_extra)
    out["override"] = np.array(override or "")
    out["color_independent"] = np.asarray(bool(color_independent))
    import json as _json
    out["edit_kwargs"] = np.array(_json.dumps({k: v for k, v in edit_kw.items() if EDIT_KEYS_OFF.get(k, None) != v}))
    for phase in phases:
        for _, net in nets:
            net.zero_grad()
            net.freeze_radiance = net.freeze_roughness = phase == "frozen"
        approx = phase in ("full", "frozen")
        with torch.enable_grad():
            res = R.render_decomp(800, 800, K, chunk=n_rays, rays=rays, gt_values={k: torch.from_numpy(v.copy()) for k, v in gt_values.items()},
                                  approximate_radiance=approx, is_depth_only=phase == "depth", **kw, **edit_kw)
            for k, v in res.items():
                out["%s__out__%s" % (phase, k)] = v.detach().numpy().copy()
            if phase == "depth":
                continue
            loss = TL.total_loss(torch, res, tg, approx)
            loss.backward()
        out[phase + "__loss"] = np.float64(loss.item())
        for tag, net in nets:
            for name, prm in net.named_parameters():
                if tag in aux_seeds and name.endswith(".weight") and name.split(".")[1] in ("1", "2", "3", "4", "6"):
                    continue      # (an auxiliary network's fixture keeps every bias — each is the sum of its layer's dZ, so the whole dgrad chain is pinned — and the weights of layers 0, 5, 7 and out_linears)
                if sparse_grads and tag in ("c", "f") and name.endswith(".weight") and prm.numel() > 40000 and name != "positions_linears.7.weight":
                    continue      # (a variant's fixture: every bias — the sum of its layer's dZ: the whole dgrad chain — every head and 128-wide feature layer, the trunk's
                    #               first and last weight; the weight-gradient GEMMs of the wide layers are pinned by train_step.npz, which keeps all 92 tensors)
                out["%s__grad_%s__%s" % (phase, tag, name)] = (prm.grad.numpy().copy() if prm.grad is not None else np.zeros(tuple(prm.shape), np.float32))
    for _, net in nets:
        net.freeze_radiance = net.freeze_roughness = False
    path = os.path.join(OUT, fixture + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %4d rays, %d phases: losses %s  %.2f MB" % (fixture, n_rays, len(phases), {p_: round(float(out[p_ + "__loss"]), 5) for p_ in phases if p_ != "depth"},
                                                           os.path.getsize(path) / 1e6))


COMPACT_KEYS = ("depth_map", "albedo_map", "roughness_map", "irradiance_map", "radiance_map", "target_normal_map", "n_dot_v_map", "prefiltered_reflected_map",
                "specular_map", "diffuse_map", "color_map", "depth_map0", "target_normal_map0")


def launch_scale_fixture(name, torch, R, M, lut, *, n_rays, seed, mode="plain", chunk=2048, weights_every=8, n_nudge=3, posed=False, compact=False, augment=False, ckpt="fitted"):
    """The fitted checkpoint (ckpt: "fitted" | "fitted2") at launch scale: `n_rays` seeded pixels of the 800x800 bench view through the
    reference's render_decomp in float32 and, as the yardstick, in float64 (torch's default tensor type switched for that run, so that
    every tensor the reference creates itself — torch.ones, torch.Tensor(list) of the edit / insert lists — is float64 too and its
    masked assignments run).  Kept: every map of both passes; `weights` / `weights0` for every `weights_every`-th ray;
    `floor__<map>` = relative L-inf of the two runs; `floorray__<map>` [n_rays] = the same difference PER RAY (max over the map's
    channels, over the map's global max) — the distribution the GPU tests bound the reflected-ray channels with.
    A second yardstick, `nudgeray__<map>` [n_rays]: the float64 run cannot see the reference's fp32 THRESHOLDS — sample_pdf replaces
    denominators below 1e-5 by 1 (nerf_renderer_helper.py:128-129) and an empty bin's denominator is 1e-5 / sum = 167 or 168 ulps of the
    cdf, so one ulp on a coarse weight decides whether that bin's fine samples collapse onto its edge.  So the float32 render is repeated
    `n_nudge` times with the reference's own sample_pdf called on weights multiplied by (1 + s 2^-23), s in {-1, 0, 1} drawn per entry
    (the wrapper passes them on; nothing else changes), and the largest per-ray change of each map against the un-nudged run is recorded:
    what one ulp on the coarse pass's weights does to the reference's own output.
    mode: "plain" | "edit_cfg4" | "insert_cfg5" (tests/frame_overrides.py: the kwargs of the two shipped configs, analytic images).
    A third, deterministic one, `branchray__<map>`: the nudges reach a threshold-critical ray only by luck (pixel 390 016 moved by 3.1e-3 under four
    nudges in one fixture and not at all under two in another), so the float32 render is also repeated with the reference's sample_pdf output
    post-processed: every fine sample whose denominator lies within 4e-7 of the 1e-5 threshold (an empty bin: 167 or 168 ulps of the cdf) is placed
    once as the reference's `denom -> 1` branch places it (collapsed onto the bin's edge) and once as its other branch does (interpolated with
    the computed denominator) — both are outcomes the reference reaches under last-bit changes of its own sums; the larger per-ray change of each
    map against the reference's actual float32 output is recorded.
    augment=True: the fixture exists; only the branch yardstick is (re)computed and added (the float32 outputs in the file are the base).
    compact (a whole 65 536-ray launch in < 30 MB): only COMPACT_KEYS are kept, the two per-ray yardsticks as float16, and the rays are not
    stored (the test rebuilds them from `pix` with get_rays)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import frame_overrides as FO
    tmp = tempfile.mkdtemp()
    try:
        _, kw, *_ = M.create_IBLNeRF(reference_args(tmp, 128))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    sd_c, sd_f = fitted_state_dicts(ckpt)
    kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
    kw["network_fine"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    kw.update(near=0.5, far=8.0)
    kw["brdf_lut"] = lut
    rng = np.random.RandomState(1000 + seed)
    o, d, pix, focal = camera_rays(rng, n_rays)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    if posed:   # a rotated + translated camera (BASELINE configs 3 / 5: "any fixed look-at"): rays_o != 0, rays_d = R dirs as get_rays builds them (:41-44)
        q, _ = np.linalg.qr(np.eye(3) + 0.25 * rng.randn(3, 3))
        q = (q * np.sign(np.linalg.det(q))).astype(np.float32)
        c2w = np.concatenate([q, np.array([[0.35], [-0.25], [0.5]], np.float32)], 1).astype(np.float32)
        d = np.sum(d[:, None, :] * c2w[:3, :3], -1).astype(np.float32)
        o = np.broadcast_to(c2w[:3, 3], d.shape).copy()
    edit = dict(EDIT_KEYS_OFF)
    gt = {}
    if mode == "edit_cfg4":
        edit.update(FO.EDIT_CFG4)
        gt = FO.edit_rows(pix)
    elif mode == "insert_cfg5":
        edit.update(FO.INSERT_CFG5)
        gt = FO.insert_rows(pix)
    K = np.array([[focal, 0, 400], [0, focal, 400], [0, 0, 1]], dtype=np.float32)

    def render(dtype):
        rays = torch.from_numpy(np.stack([o, d], 0)).to(dtype)
        gt_t = {k: torch.from_numpy(v.copy()).to(dtype) for k, v in gt.items()}
        with torch.no_grad():
            if dtype == torch.float32:
                ret = R.render_decomp(800, 800, K, chunk=chunk, rays=rays, gt_values=gt_t, approximate_radiance=True, **kw, **edit)
            else:
                # render_decomp casts the rays to float32 (:795-802); the float64 run enters one call below it, at the reference's own
                # batchify_rays, with the ray rows [o, d, near, far, viewdirs] of :804-805 built in float64
                ro, rd = rays[0], rays[1]
                k2 = dict(kw)
                near_, far_ = k2.pop("near"), k2.pop("far")
                rows = torch.cat([ro, rd, near_ * torch.ones_like(rd[:, :1]), far_ * torch.ones_like(rd[:, :1]),
                                  rd / torch.norm(rd, dim=-1, keepdim=True)], -1)
                ret = R.batchify_rays(rows, chunk, gt_values=gt_t, approximate_radiance=True, **k2, **edit)
        return {k: v.detach().numpy() for k, v in ret.items()}

    import time
    pdf0 = R.sample_pdf

    def pdf_branch(collapse):
        """the reference's sample_pdf (det=True) with its threshold-critical samples re-placed by one of its two branches"""
        def f(bins, weights, N, det=False, pytest=False):
            out_ = pdf0(bins, weights, N, det=det, pytest=pytest)
            assert det and not pytest
            w_ = weights + 1e-5                                             # the reference's own operations (nerf_renderer_helper.py:93-96, :100-101, :116-129)
            pdf_ = w_ / torch.sum(w_, -1, keepdim=True)
            cdf_ = torch.cumsum(pdf_, -1)
            cdf_ = torch.cat([torch.zeros_like(cdf_[..., :1]), cdf_], -1)
            u_ = torch.linspace(0., 1., steps=N).expand(list(cdf_.shape[:-1]) + [N]).contiguous()
            inds = torch.searchsorted(cdf_, u_, right=True)
            below = torch.max(torch.zeros_like(inds - 1), inds - 1)
            above = torch.min((cdf_.shape[-1] - 1) * torch.ones_like(inds), inds)
            c0, c1 = torch.gather(cdf_, 1, below), torch.gather(cdf_, 1, above)
            b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
            den = c1 - c0
            crit = (den - 1e-5).abs() <= 4e-7
            t_ = (u_ - c0) / (torch.ones_like(den) if collapse else den)
            return torch.where(crit, b0 + t_ * (b1 - b0), out_)
        return f

    def branch_runs():
        outs, tt = [], time.time()
        for collapse in (True, False):
            R.sample_pdf = pdf_branch(collapse)
            try:
                outs.append(render(torch.float32))
            finally:
                R.sample_pdf = pdf0
        return outs, time.time() - tt

    path = os.path.join(OUT, name + ".npz")
    if augment == "param":
        # A fourth yardstick, `paramray__<map>`: the reference's float32 render with its OWN checkpoint rounded to 22-bit mantissas — every weight
        # and bias replaced by f16(w) + f16(w - f16(w)), a change of about one float32 ulp per parameter (1e-5 of a tensor's largest entry at
        # most), which is exactly how the three-product MFMA kernels hold the weights.  The float64 run and the one-ulp nudges probe
        # perturbations of the ARITHMETIC; a fitted network's density is a cancelling sum that amplifies a parameter perturbation ~300x
        # (2e-5 on the coarse pass's weights), and through sample placement that moves some rays' fine-pass maps by 1e-3 and more while
        # every float32 implementation with exact parameters agrees on them to 1e-6 (scratch/fp32_chaos_probe.py).
        old_ = dict(np.load(path))
        assert np.array_equal(old_["pix"], pix)

        def r22(v):
            hi = v.astype(np.float16).astype(np.float32)
            return (hi + (v - hi).astype(np.float16).astype(np.float32)).astype(np.float32)
        tt = time.time()
        kw["network_fn"].load_state_dict({k: torch.from_numpy(r22(v)) for k, v in sd_c.items()})
        kw["network_fine"].load_state_dict({k: torch.from_numpy(r22(v)) for k, v in sd_f.items()})
        try:
            rp = render(torch.float32)
        finally:
            kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
            kw["network_fine"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
        fs = float(old_["floor_scale"]) if "floor_scale" in old_ else 1.0
        we_ = int(old_["weights_every"])
        for k in [k_[5:] for k_ in old_ if k_.startswith("out__")]:
            if k == "z_std":
                continue
            base = old_["out__" + k].astype(np.float64)
            scale = max(float(np.nanmax(np.abs(base))), 1e-30)
            sub = (lambda a: a[::we_]) if k.startswith("weights") else (lambda a: a)
            nd = np.nanmax(np.abs(sub(rp[k]).astype(np.float64) - base).reshape(len(base), -1), -1) / scale
            if k.startswith("weights"):
                full = np.zeros(n_rays)
                full[::we_] = nd
                nd = full
            old_["paramray__" + k] = np.minimum(np.nan_to_num(nd) * fs, 6e4).astype(old_["floorray__" + k].dtype)
        np.savez_compressed(path, **old_)
        print("%-28s %5d rays  + parameter-rounding yardstick (%.0f s)  %.2f MB" % (name, n_rays, time.time() - tt, os.path.getsize(path) / 1e6))
        return
    if augment:
        old_ = dict(np.load(path))
        assert np.array_equal(old_["pix"], pix)
        br, t_br = branch_runs()
        fs = float(old_["floor_scale"]) if "floor_scale" in old_ else 1.0
        we_ = int(old_["weights_every"])
        for k in [k_[5:] for k_ in old_ if k_.startswith("out__")]:
            if k.endswith("0") or k == "z_std":
                continue
            base = old_["out__" + k].astype(np.float64)
            scale = max(float(np.nanmax(np.abs(base))), 1e-30)
            sub = (lambda a: a[::we_]) if k.startswith("weights") else (lambda a: a)
            nd = np.maximum.reduce([np.nanmax(np.abs(sub(r_[k]).astype(np.float64) - base).reshape(len(base), -1), -1) for r_ in br]) / scale
            if k.startswith("weights"):      # per-ray arrays are full length; the weights themselves are kept for every `weights_every`-th ray
                full = np.zeros(n_rays)
                full[::we_] = nd
                nd = full
            old_["branchray__" + k] = np.minimum(nd * fs, 6e4).astype(old_["floorray__" + k].dtype)
        np.savez_compressed(path, **old_)
        print("%-28s %5d rays  + threshold-branch yardstick (%.0f s)  %.2f MB" % (name, n_rays, t_br, os.path.getsize(path) / 1e6))
        return
    t0 = time.time()
    ret = render(torch.float32)
    t32 = time.time() - t0
    nudged, t_nudge = [], 0.0
    br, t_br = branch_runs()
    for m in range(n_nudge):
        nrng = np.random.RandomState(9000 + 10 * seed + m)

        def pdf_nudged(bins, weights, N, det=False, pytest=False):
            s_ = torch.from_numpy(nrng.randint(-1, 2, size=tuple(weights.shape)).astype(np.float32))
            return pdf0(bins, weights * (1.0 + s_ * 2.0 ** -23), N, det=det, pytest=pytest)
        R.sample_pdf = pdf_nudged
        try:
            t0 = time.time()
            nudged.append(render(torch.float32))
            t_nudge += time.time() - t0
        finally:
            R.sample_pdf = pdf0
    nets = [kw["network_fn"], kw["network_fine"]]
    torch.set_default_tensor_type(torch.DoubleTensor)     # (not set_default_dtype: torch.Tensor(list) must become float64 as well)
    try:
        for n_ in nets:
            n_.double()
        kw["brdf_lut"] = lut.double()
        t0 = time.time()
        ret64 = render(torch.float64)
        t64 = time.time() - t0
    finally:
        torch.set_default_tensor_type(torch.FloatTensor)
        for n_ in nets:
            n_.float()
        kw["brdf_lut"] = lut
    assert all(v.dtype == np.float64 for v in ret64.values() if v.dtype.kind == "f")

    out = dict(rays_o=o, rays_d=d, pix=pix.astype(np.int64), near=np.float32(0.5), far=np.float32(8.0), gain=np.float64(1.0),
               seed_coarse=np.int64(2 * seed), seed_fine=np.int64(2 * seed + 1), n_importance=np.int64(128), n_samples=np.int64(64),
               ck_coarse=np.array(ck.blob_checksum(ck.state_dict_to_blob(sd_c))),
               ck_fine=np.array(ck.blob_checksum(ck.state_dict_to_blob(sd_f))), mode=np.array(mode), ckpt=np.array(ckpt),
               weights_every=np.int64(weights_every), chunk=np.int64(chunk), n_nudge=np.int64(n_nudge), c2w=c2w)
    for k, v in gt.items():
        out["gt__" + k] = v
    for k, v in edit.items():
        out["edit__" + k] = np.asarray(v, dtype=np.float32) if isinstance(v, list) else np.asarray(v)
    if compact:
        del out["rays_o"], out["rays_d"]
        out["compact"] = np.asarray(True)
        out["floor_scale"] = np.float64(16384.0)
    for k, v in ret.items():
        if compact and k not in COMPACT_KEYS:
            continue
        a, b = v.astype(np.float64), ret64[k]
        scale = max(float(np.nanmax(np.abs(b))), 1e-30)
        diff = np.abs(a - b).reshape(n_rays, -1)
        out["floor__" + k] = np.float64(np.nanmax(diff) / scale)
        fdt, fs = (np.float16, 16384.0) if compact else (np.float32, 1.0)      # compact: float16 of 2^14 x the value (resolution down to 4e-9)
        out["floorray__" + k] = np.minimum(np.nanmax(diff, -1) / scale * fs, 6e4).astype(fdt)
        if nudged and not k.endswith("0"):         # (the coarse pass comes before sample_pdf: its maps cannot move)
            nd = np.maximum.reduce([np.nanmax(np.abs(r_[k].astype(np.float64) - a).reshape(n_rays, -1), -1) for r_ in nudged]) / scale
            out["nudgeray__" + k] = np.minimum(nd * fs, 6e4).astype(fdt)
        if not k.endswith("0") and k != "z_std":
            bd = np.maximum.reduce([np.nanmax(np.abs(r_[k].astype(np.float64) - a).reshape(n_rays, -1), -1) for r_ in br]) / scale
            out["branchray__" + k] = np.minimum(bd * fs, 6e4).astype(fdt)
        out["out__" + k] = v.astype(np.float32)[::weights_every] if k.startswith("weights") else v.astype(np.float32)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %5d rays  %s  reference: %.0f s float32 (%.0f rays/s), %.0f s float64, %.0f s for %d nudged runs  %.2f MB" % (
        name, n_rays, mode, t32, n_rays / t32, t64, t_nudge, n_nudge, os.path.getsize(path) / 1e6))


def small_vectors(torch, R, Hh):
    """get_rays / sample_pdf / embedder known-answer vectors on tiny seeded inputs."""
    rng = np.random.RandomState(7)
    out = {}
    # get_rays with a non-trivial pose, non-square image
    H, W, f = 5, 7, np.float32(6.3)
    K = np.array([[f, 0, 3.25], [0, f * 1.1, 2.5], [0, 0, 1]], dtype=np.float32)
    A = rng.normal(size=(3, 3))
    Q, _ = np.linalg.qr(A)
    c2w = np.concatenate([Q, rng.normal(size=(3, 1))], 1).astype(np.float32)
    ro, rd = Hh.get_rays(H, W, K, torch.from_numpy(c2w))
    out.update(gr_H=np.int64(H), gr_W=np.int64(W), gr_K=K, gr_c2w=c2w, gr_o=ro.numpy().copy(), gr_d=rd.numpy().copy())
    # sample_pdf(det=True): generic, spiky, flat/zero weights, and non-uniform bins
    bins = np.sort(rng.uniform(0.5, 8, (6, 63)).astype(np.float32), -1)
    w = rng.uniform(0, 1, (6, 62)).astype(np.float32)
    w[1] = 0; w[1, 30] = 1.0          # one spike
    w[2] = 0                          # all zero -> uniform pdf from the 1e-5 floor
    w[3, :31] = 0                     # half empty
    w[4] = 1e-7                       # tiny
    s = Hh.sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), 128, det=True)
    out.update(sp_bins=bins, sp_weights=w, sp_samples=s.numpy().copy())
    s16 = Hh.sample_pdf(torch.from_numpy(bins[:, :9].copy()), torch.from_numpy(w[:, :8].copy()), 16, det=True)
    out.update(sp16_samples=s16.numpy().copy())
    # embedders (positional_embedder.py:4-52)
    import nerf_models.positional_embedder as PE
    e10, d10 = PE.get_embedder(10, 0)
    e4, d4 = PE.get_embedder(4, 0)
    x = np.concatenate([rng.uniform(-9, 9, (24, 3)), np.array([[0, 0, 0], [8.0, -8.0, 1e-4], [3.1415927, -1.5707964, 6.2831855]])], 0).astype(np.float32)
    out.update(pe_x=x, pe_e10=e10(torch.from_numpy(x)).numpy().copy(), pe_e4=e4(torch.from_numpy(x)).numpy().copy())
    assert d10 == 63 and d4 == 27
    np.savez_compressed(os.path.join(OUT, "small_vectors.npz"), **out)
    print("small_vectors.npz written")


def sample_pdf_spiky(torch, Hh):
    """sample_pdf(det=True) of the reference on rows from the regime a checkpoint with surfaces produces: one to four samples carry the
    whole weight, the other bins are empty (exactly 0, or ~1e-8) — where the `denom < 1e-5` replacement (nerf_renderer_helper.py:128-129)
    flips with the last bit of torch.sum.  The known answer of the summation order both the oracle and the HIP kernel follow."""
    rng = np.random.RandomState(77)
    N = 2048
    z = (np.linspace(0., 1., 64, dtype=np.float64) * 7.5 + 0.5).astype(np.float32)
    mids = np.broadcast_to((np.float32(0.5) * (z[1:] + z[:-1])).astype(np.float32), (N, 63)).copy()
    w = np.zeros((N, 62), np.float32)
    for r in range(N):
        k = rng.randint(1, 5)
        w[r, rng.choice(62, k, replace=False)] = (rng.dirichlet(np.ones(k)) * rng.uniform(0.7, 1.0)).astype(np.float32)
        if r % 3 == 0:
            w[r] += ((rng.rand(62) < 0.2) * rng.rand(62) * 1e-7).astype(np.float32)
    s = Hh.sample_pdf(torch.from_numpy(mids), torch.from_numpy(w), 128, det=True).numpy()
    tot = torch.sum(torch.from_numpy(w) + 1e-5, -1, keepdim=True).numpy()
    np.savez_compressed(os.path.join(OUT, "sample_pdf_spiky.npz"), bins=mids[0], weights=w, samples=s, row_sum=tot)
    print("sample_pdf_spiky.npz: %d rows  %.2f MB" % (N, os.path.getsize(os.path.join(OUT, "sample_pdf_spiky.npz")) / 1e6))


def export_fixture(torch, R, M, lut):
    """render_decomp_path (ibl_nerf_renderer.py:819-910) on a tiny 2-view synthetic 'dataset':
    records the float maps it returns and every 8-bit image it hands to imageio.imwrite."""
    tmp = tempfile.mkdtemp()
    try:
        _, kw, *_ = M.create_IBLNeRF(reference_args(tmp, 32))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    sd_c, sd_f = ck.synthetic_state_dict(seed=8, gain=1.0), ck.synthetic_state_dict(seed=9, gain=1.0)
    kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
    kw["network_fine"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    kw.update(near=0.5, far=8.0)
    kw["brdf_lut"] = lut
    Hh_, Ww_, focal = 6, 8, 7.5
    rng = np.random.RandomState(42)
    poses = []
    for _ in range(2):
        Q, _r = np.linalg.qr(rng.normal(size=(3, 3)))
        poses.append(np.concatenate([np.concatenate([Q, rng.normal(size=(3, 1)) * 0.3], 1), [[0, 0, 0, 1]]], 0))
    poses = np.stack(poses).astype(np.float32)

    class FakeDataset:
        far = 8.0

        def __init__(self):
            self.poses = torch.from_numpy(poses)

        def get_resized_normal_albedo(self, render_factor, i):
            return {}

    written = {}
    R.imageio.imwrite = lambda fn, img: written.__setitem__(os.path.basename(fn), np.array(img))
    with torch.no_grad():
        res = R.render_decomp_path(FakeDataset(), (Hh_, Ww_, focal), None, 1024, kw, savedir="/nonexistent",
                                   render_factor=1, approximate_radiance=True, **EDIT_KEYS_OFF)
    out = dict(H=np.int64(Hh_), W=np.int64(Ww_), focal=np.float64(focal), poses=poses, far=np.float64(8.0),
               near=np.float64(0.5), seed_coarse=np.int64(8), seed_fine=np.int64(9), n_importance=np.int64(32))
    for k, v in res.items():
        out["res__" + k] = np.asarray(v, dtype=np.float32)
    for k, v in written.items():
        out["png__" + k] = v
    np.savez_compressed(os.path.join(OUT, "export_path.npz"), **out)
    print("export_path.npz: %d result maps, %d images (%s ...)" % (len(res), len(written), sorted(written)[:3]))


def main(only=None):
    """`python tests/golden/make_golden.py [name ...]` regenerates all fixtures, or only the named ones."""
    torch, R, M, Hh = import_reference()
    torch.manual_seed(0)
    lut = load_lut(torch)
    shutil.copyfile(os.path.join(REF, "data", "ibl_brdf_lut.png"), os.path.join(OUT, "ibl_brdf_lut.png"))
    if not only or "small_vectors" in only:
        small_vectors(torch, R, Hh)
    if not only or "train_step" in only:
        train_step_fixture(torch, R, M, lut)
    if not only or "train_step_noise" in only:      # the same step with raw_noise_std = 1 (f-3: density noise inside a gradient-carrying render)
        train_step_fixture(torch, R, M, lut, fixture="train_step_noise", phases=("warmup", "full"), raw_noise_std=1.0, sparse_grads=True)
    if not only or "train_step_from_gt" in only:    # ... with the four ground-truth substitutions on (f-3: constants of the backward)
        train_step_fixture(torch, R, M, lut, fixture="train_step_from_gt", phases=("full",), sparse_grads=True,
                           from_gt=("calculate_albedo_from_gt", "calculate_roughness_from_gt", "calculate_irradiance_from_gt", "depth_map_from_ground_truth"))
    if not only or "train_step_from_gt_warmup" in only:   # f-3 leftover (round 5): the substitutions during the warm-up iterations (approximate_radiance=False)
        train_step_fixture(torch, R, M, lut, fixture="train_step_from_gt_warmup", phases=("warmup",), sparse_grads=True,
                           from_gt=("calculate_albedo_from_gt", "calculate_roughness_from_gt", "calculate_irradiance_from_gt", "depth_map_from_ground_truth"))
    if not only or "train_step_edit" in only:       # f-3 leftover (round 5): edit / insert overrides inside a gradient-carrying render
        train_step_fixture(torch, R, M, lut, fixture="train_step_edit", phases=("warmup", "full"), override="edit", sparse_grads=True)
    if not only or "train_step_insert" in only:
        train_step_fixture(torch, R, M, lut, fixture="train_step_insert", phases=("warmup", "full"), override="insert", sparse_grads=True)
    if not only or "train_step_planes" in only:     # f-3 leftover (round 5): per-ray near / far planes in a gradient-carrying render
        train_step_fixture(torch, R, M, lut, fixture="train_step_planes", phases=("warmup", "full", "depth"), planes=True, sparse_grads=True)
    if not only or "train_step_aux" in only:        # f-3 leftover (round 5): auxiliary networks (albedo / roughness / irradiance / normal) trained by the step
        train_step_fixture(torch, R, M, lut, fixture="train_step_aux", phases=("warmup", "full"), aux=True, stable_rays=True, sparse_grads=True)
    if not only or "train_step_incident" in only:   # f-3 leftover (round 5): use_gradient_for_incident_radiance
        train_step_fixture(torch, R, M, lut, fixture="train_step_incident", phases=("full", "frozen"), incident_gradient=True, sparse_grads=True)
    if not only or "train_step_arch" in only:       # round 5: a training step of a smaller architecture (evaluated inside the built one; gradients = sub-blocks)
        train_step_fixture(torch, R, M, lut, fixture="train_step_arch", phases=("warmup", "full"), arch=(6, 128, 6, 2), stable_rays=True, sparse_grads=True)
    if not only or "train_step_ci" in only:         # f-3 leftover (round 5): colour-independent networks in the backward
        train_step_fixture(torch, R, M, lut, fixture="train_step_ci", phases=("warmup", "full", "frozen"), color_independent=True, sparse_grads=True)
    if not only or "train_step_from_gt2" in only:   # ... and with two of them: albedo and irradiance from the networks, roughness and depth from the ground truth
        train_step_fixture(torch, R, M, lut, fixture="train_step_from_gt2", phases=("full",), sparse_grads=True,
                           from_gt=("calculate_roughness_from_gt", "depth_map_from_ground_truth"))
    if not only or "sample_pdf_spiky" in only:
        sample_pdf_spiky(torch, Hh)
    if not only or "export_path" in only:
        export_fixture(torch, R, M, lut)
    if not only or "trunk_backward" in only:
        trunk_backward_fixture(torch, M)
    if not only or "network_backward" in only:
        network_backward_fixture(torch, M)

    def run_fixture(name, *a, **k):
        if not only or name in only:
            _run_fixture(name, *a, **k)

    # config 1 (BASELINE.json configs[0]): coarse only
    run_fixture("cfg1_coarse_g10", torch, R, M, lut, n_rays=128, n_importance=0, gain=1.0, seed=0)
    # configs 2/3 kernel mix: 64+128, well-conditioned and wide-range checkpoints
    run_fixture("plain_g10", torch, R, M, lut, n_rays=256, n_importance=128, gain=1.0, seed=0)
    run_fixture("plain_g16", torch, R, M, lut, n_rays=128, n_importance=128, gain=1.6, seed=1, n_keep=128, record_floor=True)
    # config 4 / config 5 override paths
    run_fixture("edit_g10", torch, R, M, lut, n_rays=128, n_importance=128, gain=1.0, seed=2, mode="edit")
    run_fixture("insert_g10", torch, R, M, lut, n_rays=128, n_importance=128, gain=1.0, seed=3, mode="insert")
    # f-4 flag variants: HDR radiance (ReLU + Reinhard), inverse-depth sampling, F0 LUT coefficient
    run_fixture("variant_lin_g10", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=4,
                flags=dict(use_radiance_linear=True, lindisp=True, lut_coefficient="F0"))
    # image-driven edits (edit_depth, edit_albedo_by_img) — flags of config_parser.py:246-256 the shipped edit config leaves off
    run_fixture("edit2_g10", torch, R, M, lut, n_rays=96, n_importance=128, gain=1.0, seed=5, mode="edit2")
    # edit_roughness_by_img (:394-395; first-masked-row semantics), and two arguments of render_decomp itself: c2w_staticcam, per-ray near / far
    run_fixture("edit3_g10", torch, R, M, lut, n_rays=96, n_importance=128, gain=1.0, seed=40, mode="edit3")
    for nm, kind, sd_ in (("staticcam_g10", "staticcam", 41), ("nearfar_g10", "nearfar", 42)):
        if not only or nm in only:
            seam_fixture(nm, torch, R, M, lut, kind, sd_)
    # other sample counts / epsilon / planes / no gamma / roughness-only mip level, posed camera
    run_fixture("variant_small_g10", torch, R, M, lut, n_rays=96, n_importance=48, gain=1.0, seed=6, n_samples=32,
                near=1.0, far=5.0, posed=True,
                flags=dict(epsilon=0.02, gamma_correct=False, correct_depth_for_prefiltered_radiance_infer=False))
    # smaller architectures (round 5; ibl_nerf.py:14-60 takes any netdepth / netwidth, config_parser.py any multires / multires_views): the reference builds and runs them
    # as such; this library evaluates them inside the built 8 x 256 / 10 / 4 architecture (checkpoint.embed_architecture).
    run_fixture("arch_6x128_g10", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=21, arch=(6, 128, 6, 2))       # own skip layer, fewer frequencies
    run_fixture("arch_4x64_g10", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=22, arch=(4, 64, 10, 4))        # no skip layer: four identity layers
    run_fixture("arch_7x200_g10", torch, R, M, lut, n_rays=48, n_importance=64, gain=1.0, seed=23, arch=(7, 200, 8, 3), mode="insert")     # odd width, one identity layer, insert overrides
    # ... with every auxiliary network in the same small shape (PositionMLP x 4, the depth_mlp a PositionDirectionMLP with D // 2 = 3 view layers of W // 2)
    run_fixture("arch_aux_6x128_g10", torch, R, M, lut, n_rays=48, n_importance=128, gain=1.0, seed=24, arch=(6, 128, 6, 2), aux=True, infer_normal=True, infer_depth=True)
    # LARGER architectures (round 6: csrc/generic_mlp.hip evaluates them layer by layer in exact fp32): deeper, wider, more frequencies than the built 8 x 256 / 10 / 4
    run_fixture("arch_10x384_g10", torch, R, M, lut, n_rays=48, n_importance=64, gain=1.0, seed=25, arch=(10, 384, 12, 5))     # two layers behind the skip layer more, 1.5 x the width, 12 / 5 frequencies
    run_fixture("arch_8x512_g10", torch, R, M, lut, n_rays=40, n_importance=64, gain=1.0, seed=26, arch=(8, 512, 10, 4), mode="edit")      # the built depth at twice the width, edit overrides
    # is_color_independent_to_direction (ibl_nerf.py:192): radiance heads on the trunk output, no feature / view layers
    run_fixture("colorindep_g10", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=8, color_independent=True)
    # ground-truth normals instead of the eps-normal (no offset queries)
    run_fixture("gtnormal_g10", torch, R, M, lut, n_rays=96, n_importance=128, gain=1.0, seed=7, mode="gtnormal",
                flags=dict(target_normal_map_for_radiance_calculation="ground_truth"))
    # the other finite-difference normal: four rays with tilted directions (normal_from_depth.py:55-100), posed camera
    run_fixture("dirnormal_g10", torch, R, M, lut, n_rays=96, n_importance=128, gain=1.0, seed=11, posed=True,
                flags=dict(target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient_direction_epsilon",
                           epsilon_direction=0.005))
    # auxiliary PositionMLPs for albedo / roughness / irradiance (infer_*_separate), HDR radiance so that the irradiance_mlp's
    # sigmoid differs from radiance_f
    run_fixture("auxmlp_g10", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=12, aux=True)
    run_fixture("auxmlp_lin_g10", torch, R, M, lut, n_rays=48, n_importance=128, gain=1.0, seed=13, aux=True,
                flags=dict(use_radiance_linear=True))
    # infer_normal: the normal_mlp's composited output as an extra map beside the eps-normal, and as the target normal
    run_fixture("infernormal_g10", torch, R, M, lut, n_rays=48, n_importance=128, gain=1.0, seed=14, infer_normal=True)
    run_fixture("infernormal_target_g10", torch, R, M, lut, n_rays=48, n_importance=128, gain=1.0, seed=15, infer_normal=True,
                flags=dict(target_normal_map_for_radiance_calculation="inferred_normal_map"))
    # ... evaluated once per ray at the surface point, under an edited depth (the surface point follows the edit)
    run_fixture("infernormal_surface_g10", torch, R, M, lut, n_rays=48, n_importance=128, gain=1.0, seed=16, infer_normal=True,
                mode="edit2", flags=dict(target_normal_map_for_radiance_calculation="inferred_normal_map", infer_normal_at_surface=True))
    # perturb = 1 (stratified jitter + stochastic fine samples) through the reference's pytest seed path
    run_fixture("perturb_g10", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=18, perturb=True)
    # ... with density noise on top (raw_noise_std = 1; the pytest hook draws it uniform)
    run_fixture("perturb_noise_g10", torch, R, M, lut, n_rays=48, n_importance=128, gain=1.0, seed=19, perturb=True, raw_noise_std=1.0)
    # infer_depth: the depth_mlp (PositionDirectionMLP) once per ray at the origin with the normalised direction; posed camera
    run_fixture("inferdepth_g10", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=17, infer_depth=True, posed=True)
    # the fitted (surface-bearing) checkpoint of fit_checkpoint.py: plain, material edit, object insertion; raw recorded for all rays
    run_fixture("fitted_plain", torch, R, M, lut, n_rays=96, n_importance=128, gain=1.0, seed=20, fitted=True, n_keep=96, record_floor=True)
    run_fixture("fitted_edit", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=21, mode="edit", fitted=True, n_keep=64)
    run_fixture("fitted_insert", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=22, mode="insert", fitted=True, n_keep=64)   # (the reference's masked assignments do not run in float64: the floor of fitted_plain stands for all three)
    # the same checkpoint on 1 024 rays (maps only): how the worst ray grows with the sample, and the reference's own fp64-vs-fp32 run on it
    run_fixture("fitted_wide", torch, R, M, lut, n_rays=1024, n_importance=128, gain=1.0, seed=23, fitted=True, n_keep=2, record_floor=True)
    # launch scale (NOT part of the default run: minutes of reference CPU time each; name them on the command line): 16 384 pixels of
    # the bench view, and BASELINE configs 4 / 5 (the shipped edit / insert kwargs on analytic mask / normal / depth images) at 4 096
    for nm, kws in (("fitted_launch16k", dict(n_rays=16384, seed=30)),
                    ("fitted_edit_cfg4", dict(n_rays=4096, seed=31, mode="edit_cfg4", weights_every=4, n_nudge=4)),
                    ("fitted_insert_cfg5", dict(n_rays=4096, seed=32, mode="insert_cfg5", weights_every=4, n_nudge=4)),
                    ("fitted_posed4k", dict(n_rays=4096, seed=34, weights_every=4, n_nudge=4, posed=True)),
                    ("fitted2_launch4k", dict(n_rays=4096, seed=36, weights_every=4, n_nudge=4, ckpt="fitted2")),   # the second, independent checkpoint (fit_checkpoint.py --scene 2)
                    ("fitted2_posed4k", dict(n_rays=4096, seed=37, weights_every=4, n_nudge=4, posed=True, ckpt="fitted2")),
                    # the HOLD-OUT checkpoint (round 5, fit_checkpoint.py --scene 3: thin discs, a grazing floor, empty space at -3.5, the sharpest steps), fitted and
                    # rendered after every threshold of the estimate / list route, the calibration and the launch-scale rules was frozen
                    ("fitted3_launch4k", dict(n_rays=4096, seed=38, weights_every=4, n_nudge=4, ckpt="fitted3")),
                    ("fitted3_posed4k", dict(n_rays=4096, seed=39, weights_every=4, n_nudge=4, posed=True, ckpt="fitted3")),
                    ("fitted_launch64k", dict(n_rays=65536, seed=35, n_nudge=2, compact=True)),        # one whole launch of bench.py's frame: ~40 minutes of reference CPU time
                    ("_launch_probe", dict(n_rays=64, seed=33, mode="insert_cfg5", weights_every=1))):
        if only and nm in only:
            launch_scale_fixture(nm, torch, R, M, lut, **kws)
        elif only and nm + "+branch" in only:       # add the threshold-branch yardstick to an existing fixture (two float32 renders)
            launch_scale_fixture(nm, torch, R, M, lut, augment=True, **kws)
        elif only and nm + "+param" in only:        # add the parameter-rounding yardstick to an existing fixture (one float32 render)
            launch_scale_fixture(nm, torch, R, M, lut, augment="param", **kws)
    # the autograd normal modes (normal_from_depth.py:16-52 direction, :102-137 position), run with gradients enabled as in training;
    # posed cameras; one on the fitted checkpoint
    run_fixture("gradnormal_g10", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=24, posed=True, autograd=True, n_keep=4, record_floor=True,
                flags=dict(target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient"))
    run_fixture("graddir_g10", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=25, posed=True, autograd=True, n_keep=4, record_floor=True,
                flags=dict(target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient_direction"))
    run_fixture("fitted_gradnormal", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=26, fitted=True, autograd=True, n_keep=8, record_floor=True,
                flags=dict(target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient"))
    # *_from_gt: shade with ground-truth intrinsics (config_parser.py's calculate_*_from_gt, depth_map_from_ground_truth)
    run_fixture("fromgt_g10", torch, R, M, lut, n_rays=96, n_importance=128, gain=1.0, seed=9, mode="fromgt",
                flags=dict(calculate_albedo_from_gt=True, calculate_roughness_from_gt=True,
                           calculate_irradiance_from_gt=True, depth_map_from_ground_truth=True))
    run_fixture("fromgt_insert_g10", torch, R, M, lut, n_rays=64, n_importance=128, gain=1.0, seed=10, mode="fromgt_insert",
                flags=dict(calculate_irradiance_from_gt=True, calculate_roughness_from_gt=True))


if __name__ == "__main__":
    main(set(sys.argv[1:]))
