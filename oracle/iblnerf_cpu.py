"""CPU ORACLE (test infrastructure, NOT the product) — numpy-facing wrapper of oracle/_build/libiblnerf_cpu.so, the C restatement of
the reference's forward path (oracle/iblnerf_cpu.h, oracle/csrc/*.c; built by oracle/build_cpu.py).

Only `tests/`, `__graft_entry__` and `bench.py`'s `cpu_baseline` leg may import this module.  The functions take and return what the
numpy oracle's functions of the same name do (oracle/iblnerf_oracle.py), so a test can run either checker on a fixture.  The ctypes
mirrors of the ABI structs are the shipped binding's own classes (ibl-nerf_amd/binding.py: plain ctypes, no torch, no library load) —
the C side includes include/iblnerf.h itself, so the two cannot drift apart unnoticed (tests/test_host_logic.py checks the mirror)."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "_build", "libiblnerf_cpu.so")
F32 = np.float32

MAPS3 = ["color_map", "radiance_map", "reflected_radiance_map", "prefiltered_reflected_map", "albedo_map", "specular_map", "diffuse_map",
         "target_normal_map"]
MAPS1 = ["roughness_map", "n_dot_v_map", "disp_map", "acc_map", "depth_map", "target_depth_map"]
# raw2outputs' key order (ibl_nerf_renderer.py:494-525), as the numpy oracle returns it
ORDER = (["color_map", "radiance_map"] + ["radiance_map_%d" % k for k in (1, 2, 3)] + ["reflected_coarse_radiance_map_%d" % k for k in (1, 2, 3)]
         + ["irradiance_map", "reflected_radiance_map", "prefiltered_reflected_map", "albedo_map", "roughness_map", "specular_map", "diffuse_map",
            "n_dot_v_map", "target_normal_map", "disp_map", "acc_map", "depth_map", "target_depth_map", "weights"])

_lib = None
_B = None


def _binding():
    global _B
    if _B is None:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import _pkg
        _B = _pkg.load().binding
    return _B


def build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("iblnerf_build_cpu", os.path.join(HERE, "build_cpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.build()


def lib():
    """The loaded library; built on first use where gcc is present (the GPU box receives the prebuilt file)."""
    global _lib
    if _lib is None:
        try:
            build()          # (a stamp of sources + flags makes this a no-op when nothing changed; a stale library would silently check against old code)
        except Exception:
            if not os.path.exists(LIB_PATH):     # no gcc here and no prebuilt file: nothing to load
                raise
        L = C.CDLL(LIB_PATH)
        L.iblnerf_cpu_last_error.restype = C.c_char_p
        L.iblnerf_cpu_isa.restype = C.c_char_p
        fp, vp = C.POINTER(C.c_float), C.c_void_p
        L.iblnerf_render_cpu.argtypes = [vp, fp, fp, C.c_size_t, fp, fp, fp, C.c_int64, C.c_float, C.c_float, vp, vp, C.c_int]
        L.iblnerf_network_query_cpu.argtypes = [fp, C.c_size_t, C.c_int, fp, C.c_int64, C.c_int, fp, fp, C.c_int]
        L.iblnerf_sample_pdf_cpu.argtypes = [fp, fp, C.c_int64, C.c_int, C.c_int, fp]
        L.iblnerf_get_rays_cpu.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp]
        _lib = L
    return _lib


def usable_cpus():
    """Hardware threads this process may actually keep busy: the affinity mask, cut down to the cgroup's CPU quota (the GPU boxes of this pool
    show 256 logical CPUs under a quota of 16: more threads than that only time-slice — 999 rays/s on 256 threads against 2 042 on 32)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                       # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                      # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(np.ceil(quota / period))))
        except (OSError, ValueError):
            pass
    return n


def isa():
    return lib().iblnerf_cpu_isa().decode()


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=F32)


def _check(rc):
    if rc != 0:
        raise RuntimeError("libiblnerf_cpu: %s (status %d)" % (lib().iblnerf_cpu_last_error().decode(), rc))


def _blob(sd):
    if isinstance(sd, np.ndarray):
        return _f32(sd)
    return _f32(np.concatenate([np.asarray(v, dtype=F32).ravel() for v in sd.values()]))


def get_rays(H, W, K, c2w):
    ro, rd = np.empty((H, W, 3), F32), np.empty((H, W, 3), F32)
    _check(lib().iblnerf_get_rays_cpu(H, W, _fp(_f32(np.asarray(K)[:3, :3])), _fp(_f32(np.asarray(c2w)[:3, :4])), _fp(ro), _fp(rd)))
    return ro, rd


def network_query(sd, pts, viewdirs, color_independent=False, n_threads=None):
    n_threads = usable_cpus() if n_threads is None else n_threads
    pts = _f32(pts)
    N, S, _ = pts.shape
    d = _f32(viewdirs) if viewdirs is not None else None
    out = np.empty((N, S, 18 if d is not None else 1), F32)
    b = _blob(sd)
    _check(lib().iblnerf_network_query_cpu(_fp(b), b.size, int(color_independent), _fp(pts), N, S, _fp(d), _fp(out), n_threads))
    return out


def sample_pdf(bins, weights, n_samples):
    bins, weights = _f32(bins), _f32(weights)
    out = np.empty((bins.shape[0], n_samples), F32)
    _check(lib().iblnerf_sample_pdf_cpu(_fp(bins), _fp(weights), bins.shape[0], bins.shape[1], n_samples, _fp(out)))
    return out


def _maps(B, n, S, keep):
    m, t = B.Maps(), {}

    def new(key, *shape):
        a = np.full(shape, np.nan, F32)
        keep.append(a)
        t[key] = a
        return a.ctypes.data

    for k in MAPS3:
        setattr(m, k, new(k, n, 3))
    for k in MAPS1:
        setattr(m, k, new(k, n))
    m.irradiance_map = new("irradiance_map", n, 1)
    m.weights = new("weights", n, S)
    for i in range(3):
        m.radiance_map_k[i] = new("radiance_map_%d" % (i + 1), n, 3)
        m.reflected_coarse_radiance_map_k[i] = new("reflected_coarse_radiance_map_%d" % (i + 1), n, 3)
    return m, t


def render_rays(sd_coarse, sd_fine, rays_o, rays_d, near, far, lut, n_samples=64, n_importance=128, gt=None, edit=None, flags=None,
                color_independent=False, coarse_outputs=True, n_threads=None):
    """The numpy oracle's render_rays (same arguments, same result dict) on the C restatement.  flags: the subset the C path restates —
    use_radiance_linear, lut_coefficient, gamma_correct, epsilon, correct_depth_for_prefiltered_radiance_infer, lindisp,
    target_normal_map_for_radiance_calculation in {normal_map_from_depth_gradient_epsilon, ground_truth}; anything else raises."""
    B = _binding()
    flags, gt, edit = dict(flags or {}), gt or {}, edit or {}
    if n_threads is None:
        n_threads = usable_cpus()          # (0 = omp_get_max_threads(), which ignores a cgroup quota)
    o = B.Options()
    o.n_samples, o.n_importance = int(n_samples), int(n_importance)
    o.epsilon = float(flags.pop("epsilon", 0.01))
    o.gamma_correct = int(bool(flags.pop("gamma_correct", True)))
    lutc = flags.pop("lut_coefficient", "F")
    if lutc not in ("F", "F0"):
        raise ValueError(lutc)
    o.lut_coefficient_f0 = int(lutc == "F0")
    o.correct_depth_for_prefiltered_radiance = int(bool(flags.pop("correct_depth_for_prefiltered_radiance_infer", True)))
    o.coarse_outputs = int(bool(coarse_outputs))
    o.lindisp = int(bool(flags.pop("lindisp", False)))
    o.use_radiance_linear = int(bool(flags.pop("use_radiance_linear", False)))
    nmode = flags.pop("target_normal_map_for_radiance_calculation", "normal_map_from_depth_gradient_epsilon")
    if nmode not in ("normal_map_from_depth_gradient_epsilon", "ground_truth"):
        raise NotImplementedError("the C restatement has the epsilon normal and ground-truth normals; %r is in the numpy oracle" % nmode)
    o.normal_mode = 1 if nmode == "ground_truth" else 0
    o.color_independent_to_direction = int(bool(color_independent))
    left = {k: v for k, v in flags.items() if v}
    if left:
        raise NotImplementedError("flags not restated in C (the numpy oracle has them): %s" % sorted(left))
    assert not (edit.get("edit_intrinsic") and edit.get("insert_object")), "edit_intrinsic and insert_object cannot be True at the same time"   # :218
    ro, rd = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    n = ro.shape[0]
    keep = []
    ov = None

    def rows(key, width):
        a = _f32(np.asarray(gt[key], dtype=F32).reshape(n, -1)[:, :width])
        keep.append(a)
        return a.ctypes.data

    if o.normal_mode == 1 or edit.get("edit_intrinsic") or edit.get("insert_object"):
        ov = B.Overrides()
        if o.normal_mode == 1:
            ov.d_gt_normal = rows("normal", 3)
        if edit.get("edit_intrinsic"):
            ov.mode, ov.num_objects = 1, int(edit["num_edit_objects"])
            ov.d_mask = rows("edit_intrinsic_mask", 3)
            for f in ("edit_depth", "edit_normal", "edit_albedo", "edit_albedo_by_img", "edit_roughness"):
                setattr(ov, f, int(bool(edit.get(f, False))))
            if ov.edit_depth:
                ov.d_depth = rows("edit_depth", 1)
            if ov.edit_normal:
                ov.d_normal = rows("edit_normal", 3)
            if ov.edit_albedo and ov.edit_albedo_by_img:
                ov.d_albedo = rows("edit_albedo", 3)
            rgh = list(edit.get("editing_target_roughness_list") or [])
            alb = list(edit.get("editing_target_albedo_list") or [])
            ov.n_roughness_list = len(rgh)
            if ov.edit_roughness and edit.get("edit_roughness_by_img"):     # :394-395: the first masked row (one chunk = this call), handed over per ray
                m = np.asarray(gt["edit_intrinsic_mask"], dtype=F32).reshape(n, -1)[:, 0] > 0
                img = np.asarray(gt["edit_roughness"], dtype=F32).reshape(n, -1)
                per_ray = np.full((n,), img[m][0][0] if m.any() else 0.0, dtype=F32)
                keep.append(per_ray)
                ov.edit_roughness_by_img, ov.d_roughness = 1, per_ray.ctypes.data
        elif edit.get("insert_object"):
            ov.mode, ov.num_objects = 2, int(edit["num_insert_objects"])
            ov.d_mask, ov.d_depth, ov.d_normal = rows("object_insert_mask", 3), rows("object_insert_depth", 1), rows("object_insert_normal", 3)
            rgh = list(edit["inserting_target_roughness_list"])
            alb = list(edit["inserting_target_albedo_list"])
            ov.n_roughness_list = len(rgh)
            for i, v in enumerate(edit["inserting_target_irradiance_list"]):
                ov.irradiance_list[i] = float(v)
        if ov.mode:
            for i, v in enumerate(rgh):
                ov.roughness_list[i] = float(v)
            for i, v in enumerate(alb):
                ov.albedo_list[i] = float(v)
    outs = B.Outputs()
    fine = n_importance > 0
    outs.fine, t_fine = _maps(B, n, n_samples + n_importance, keep)
    t_coarse = {}
    if fine and coarse_outputs:
        outs.coarse, t_coarse = _maps(B, n, n_samples, keep)
    z_std = None
    if fine:
        z_std = np.full((n,), np.nan, F32)
        outs.z_std = z_std.ctypes.data
    bc = _blob(sd_coarse)
    bf = _blob(sd_fine) if sd_fine is not None else None
    lut = _f32(lut)
    _check(lib().iblnerf_render_cpu(C.addressof(o), _fp(bc), _fp(bf), bc.size, _fp(lut), _fp(ro), _fp(rd), n, float(near), float(far),
                                    C.addressof(ov) if ov is not None else None, C.addressof(outs), int(n_threads)))
    res = {k: t_fine[k] for k in ORDER}
    for k in ORDER:
        if k in t_coarse:
            res[k + "0"] = t_coarse[k]
    if z_std is not None:
        res["z_std"] = z_std
    return res
