"""ctypes binding of the C-ABI in include/iblnerf.h (libiblnerf_hip.so, built by build.py).

There is no CPU fallback: if the library is missing, or no HIP device is present, the compute
entry points raise.  Nothing in this package imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libiblnerf_hip.so")   # (an ablation build is loaded by passing its path to load_library first)

EXPORTS = [
    "iblnerf_default_options", "iblnerf_create", "iblnerf_destroy", "iblnerf_last_error", "iblnerf_blob_floats",
    "iblnerf_upload_weights", "iblnerf_upload_lut", "iblnerf_pack_weights_host", "iblnerf_stream_bytes",
    "iblnerf_table_floats", "iblnerf_encode_host", "iblnerf_get_rays", "iblnerf_network_query",
    "iblnerf_sample_pdf", "iblnerf_render_rays", "iblnerf_set_profiling", "iblnerf_last_mlp_time",
    "iblnerf_range_status", "iblnerf_pack_weights_host_mx", "iblnerf_stream_bytes_mx", "iblnerf_upload_weights_device",
    "iblnerf_upload_aux_weights", "iblnerf_clear_aux", "iblnerf_composite_pass", "iblnerf_range_peek", "iblnerf_pack_weights_host_f16x3",
    "iblnerf_posdir_floats", "iblnerf_upload_posdir_mlp", "iblnerf_clear_posdir_mlp", "iblnerf_posdir_query", "iblnerf_render_rays_sampled", "iblnerf_sample_pdf_u",
    "iblnerf_density_gradient", "iblnerf_trunk_backward", "iblnerf_trunk_features", "iblnerf_trunk_features_backward",
    "iblnerf_trunk_features2", "iblnerf_trunk_features2_backward", "iblnerf_network_backward",
    "iblnerf_composite_direct", "iblnerf_composite_direct_backward", "iblnerf_trim", "iblnerf_composite_direct_backward_full",
    "iblnerf_coarse_z", "iblnerf_sample_points", "iblnerf_fine_z", "iblnerf_composite_sigma", "iblnerf_render_rays_tapped",
    "iblnerf_ray_outputs_backward", "iblnerf_range_flags_async", "iblnerf_set_query_routing", "iblnerf_layer_ranges", "iblnerf_last_selection",
    "iblnerf_last_executed_flops", "iblnerf_estimate_policy", "iblnerf_ray_outputs_backward_gt", "iblnerf_ray_outputs_backward_rays", "iblnerf_coarse_z_rays", "iblnerf_aux_query", "iblnerf_aux_backward", "iblnerf_ray_outputs_backward_env", "iblnerf_set_select_tmin", "iblnerf_set_chunk_cuts",
    "iblnerf_decide_route", "iblnerf_set_route", "iblnerf_get_route", "iblnerf_copy_route", "iblnerf_describe_route", "iblnerf_last_slot_units", "iblnerf_trunk_density_fp32", "iblnerf_set_offset_tier_threshold",
    "iblnerf_escalate_route", "iblnerf_set_lists", "iblnerf_set_tapped_lists", "iblnerf_get_rays_strided", "iblnerf_get_rays_pixels", "iblnerf_set_tier_thresholds", "iblnerf_decide_route_outputs", "iblnerf_upload_weights_arch",
]


class IblNerfError(RuntimeError):
    """A negative iblnerf_status came back; message from iblnerf_last_error."""


class Options(C.Structure):
    _fields_ = [("n_samples", C.c_int32), ("n_importance", C.c_int32), ("epsilon", C.c_float),
                ("gamma_correct", C.c_int32), ("lut_coefficient_f0", C.c_int32),
                ("correct_depth_for_prefiltered_radiance", C.c_int32), ("coarse_outputs", C.c_int32),
                ("max_rays_per_launch", C.c_int32), ("device", C.c_int32), ("lindisp", C.c_int32),
                ("use_radiance_linear", C.c_int32), ("normal_mode", C.c_int32), ("color_independent_to_direction", C.c_int32), ("mlp_precision", C.c_int32),
                ("epsilon_direction", C.c_float), ("infer_normal_at_surface", C.c_int32),
                ("query_routing", C.c_int32), ("persistent_workgroups", C.c_int32)]


MLP_BF16X3, MLP_F16_MXFP6, MLP_F16_MIXED, MLP_F16X3, MLP_F16X3_MXFP6, MLP_F16X3_MAIN, MLP_F16X3_MXFP6X = 0, 1, 2, 3, 4, 5, 6
MLP_PRECISIONS = {"bf16x3": MLP_BF16X3, "f16_mxfp6": MLP_F16_MXFP6, "f16_mixed": MLP_F16_MIXED, "f16x3": MLP_F16X3,
                  "f16x3_mxfp6": MLP_F16X3_MXFP6, "f16x3_main": MLP_F16X3_MAIN, "f16x3_mxfp6x": MLP_F16X3_MXFP6X}
ROUTE_COARSE_OFFSETS_MIXED, ROUTE_USER_TRUNK_MIXED, ROUTE_FINE_MAIN_PRECISE, ROUTE_POINT_BATCH, ROUTE_COARSE_MAIN_22BIT, ROUTE_USER_TRUNK_P, ROUTE_FINE_OFFSETS_PRECISE, ROUTE_COARSE_DENSITY_ALL_POINTS = 1, 2, 4, 8, 16, 32, 64, 128   # iblnerf_options.query_routing bits
ROUTE_ESTIMATES_6SLOT, ROUTE_ESTIMATES_WHOLE, ROUTE_OFFSETS_ESTIMATE_ALL, ROUTE_COARSE_DENSITY_15SLOT, ROUTE_NO_OFFSET_TIERS, ROUTE_FINE_TIERS = 256, 512, 1024, 4096, 8192, 16384
AUX_KINDS = {"albedo_mlp": (0, 3), "roughness_mlp": (1, 1), "irradiance_mlp": (2, 1), "normal_mlp": (3, 3)}   # render kwarg -> (IBLNERF_AUX_*, out_ch)


FP = C.c_void_p  # device float*


class Overrides(C.Structure):
    _fields_ = [("mode", C.c_int32), ("num_objects", C.c_int32), ("edit_depth", C.c_int32),
                ("edit_normal", C.c_int32), ("edit_albedo", C.c_int32), ("edit_albedo_by_img", C.c_int32),
                ("edit_roughness", C.c_int32), ("n_roughness_list", C.c_int32),
                ("d_mask", FP), ("d_depth", FP), ("d_normal", FP), ("d_albedo", FP),
                ("roughness_list", C.c_float * 8), ("albedo_list", C.c_float * 24),
                ("irradiance_list", C.c_float * 8), ("d_gt_normal", FP),
                ("d_gt_albedo", FP), ("d_gt_roughness", FP), ("d_gt_irradiance", FP), ("d_gt_depth", FP),
                ("edit_roughness_by_img", C.c_int32), ("d_roughness", FP)]


class Maps(C.Structure):
    _fields_ = [("color_map", FP), ("radiance_map", FP), ("radiance_map_k", FP * 3),
                ("reflected_coarse_radiance_map_k", FP * 3), ("irradiance_map", FP),
                ("reflected_radiance_map", FP), ("prefiltered_reflected_map", FP), ("albedo_map", FP),
                ("roughness_map", FP), ("specular_map", FP), ("diffuse_map", FP), ("n_dot_v_map", FP),
                ("target_normal_map", FP), ("disp_map", FP), ("acc_map", FP), ("depth_map", FP),
                ("target_depth_map", FP), ("weights", FP), ("inferred_normal_map", FP)]


class StageInputs(C.Structure):
    _fields_ = [("n_samples", C.c_int32), ("d_z", FP), ("d_raw", FP), ("d_sigma_offsets", FP), ("d_refl_raw", FP),
                ("d_normal_raw", FP), ("d_stage", FP), ("d_refl_o", FP), ("d_refl_d", FP)]


class Sampling(C.Structure):
    _fields_ = [("d_t_rand", FP), ("d_u", FP), ("d_noise_coarse", FP), ("d_noise_fine", FP), ("d_near", FP), ("d_far", FP)]


class Taps(C.Structure):
    _fields_ = [("d_z_coarse", FP), ("d_z_fine", FP), ("d_raw_coarse", FP), ("d_raw_fine", FP), ("d_env_coarse", FP), ("d_env_fine", FP)]


class Route(C.Structure):
    """iblnerf_route: the checkpoint's measured route (which queries run as estimate + list, on which estimates)."""
    _fields_ = [("decided", C.c_int32), ("estimates_plain_f16", C.c_int32 * 2), ("tripped", C.c_int32),
                ("coarse_share", C.c_double), ("fine_main_share", C.c_double), ("fine_offsets_share", C.c_double),
                ("select_margin", C.c_float * 2), ("estimate_error", C.c_float * 2)]

    def as_dict(self):
        return {"decided": bool(self.decided), "estimates_plain_f16": [bool(self.estimates_plain_f16[0]), bool(self.estimates_plain_f16[1])],
                "tripped": int(self.tripped), "coarse_share": float(self.coarse_share), "fine_main_share": float(self.fine_main_share),
                "fine_offsets_share": float(self.fine_offsets_share), "select_margin": [float(self.select_margin[0]), float(self.select_margin[1])],
                "estimate_error": [float(self.estimate_error[0]), float(self.estimate_error[1])]}


class Outputs(C.Structure):
    _fields_ = [("fine", Maps), ("coarse", Maps), ("z_std", FP), ("inferred_depth_map", FP), ("trip_rays", FP)]


_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library and declare prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise IblNerfError("%s not found: build it with `python ibl-nerf_amd/build.py` "
                           "(there is no CPU fallback for the render path)" % path)
    try:
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so); this library is linked against /opt/rocm's.  Whichever is loaded FIRST
        # serves both (same soname) — and it has to be torch's: with /opt/rocm's runtime loaded first, torch initialises a second runtime and
        # this library's hipGetDeviceCount then reports "no ROCm-capable device" (seen when build() dlopen'ed the library before smoke()
        # imported torch)
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.iblnerf_default_options.argtypes = [C.POINTER(Options)]
    lib.iblnerf_default_options.restype = None
    lib.iblnerf_create.argtypes = [C.POINTER(Options), C.POINTER(C.c_void_p)]
    lib.iblnerf_destroy.argtypes = [C.c_void_p]
    lib.iblnerf_destroy.restype = None
    lib.iblnerf_last_error.argtypes = [C.c_void_p]
    lib.iblnerf_last_error.restype = C.c_char_p
    lib.iblnerf_blob_floats.restype = C.c_size_t
    lib.iblnerf_stream_bytes.restype = C.c_size_t
    lib.iblnerf_table_floats.restype = C.c_size_t
    lib.iblnerf_stream_bytes_mx.restype = C.c_size_t
    lib.iblnerf_pack_weights_host_mx.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.iblnerf_range_status.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.iblnerf_range_peek.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.iblnerf_range_peek.restype = C.c_int
    lib.iblnerf_layer_ranges.argtypes = [C.c_void_p, C.c_void_p, FP, C.c_size_t, FP, FP, C.c_int64, FP]
    lib.iblnerf_layer_ranges.restype = C.c_int
    lib.iblnerf_last_selection.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.iblnerf_last_selection.restype = C.c_int
    lib.iblnerf_last_executed_flops.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.iblnerf_last_executed_flops.restype = C.c_int
    lib.iblnerf_estimate_policy.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.iblnerf_estimate_policy.restype = C.c_int
    lib.iblnerf_decide_route.argtypes = [C.c_void_p, C.c_void_p, FP, FP, C.c_int64, C.c_float, C.c_float, C.POINTER(Route)]
    lib.iblnerf_decide_route.restype = C.c_int
    lib.iblnerf_decide_route_outputs.argtypes = [C.c_void_p, C.c_void_p, FP, FP, C.c_int64, C.c_float, C.c_float, C.POINTER(Overrides), C.POINTER(Outputs), C.POINTER(Route)]
    lib.iblnerf_decide_route_outputs.restype = C.c_int
    lib.iblnerf_set_route.argtypes = [C.c_void_p, C.POINTER(Route)]
    lib.iblnerf_set_route.restype = C.c_int
    lib.iblnerf_get_route.argtypes = [C.c_void_p, C.POINTER(Route)]
    lib.iblnerf_get_route.restype = C.c_int
    lib.iblnerf_copy_route.argtypes = [C.c_void_p, C.c_void_p]
    lib.iblnerf_copy_route.restype = C.c_int
    lib.iblnerf_describe_route.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.iblnerf_describe_route.restype = C.c_int
    lib.iblnerf_set_tier_thresholds.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.iblnerf_set_tier_thresholds.restype = C.c_int
    lib.iblnerf_set_offset_tier_threshold.argtypes = [C.c_void_p, C.c_float]
    lib.iblnerf_set_offset_tier_threshold.restype = C.c_int
    lib.iblnerf_last_slot_units.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.iblnerf_last_slot_units.restype = C.c_int
    lib.iblnerf_set_query_routing.argtypes = [C.c_void_p, C.c_int]
    lib.iblnerf_set_query_routing.restype = C.c_int
    lib.iblnerf_range_flags_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.iblnerf_range_flags_async.restype = C.c_int
    lib.iblnerf_upload_weights.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.iblnerf_upload_weights_arch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.iblnerf_upload_weights_arch.restype = C.c_int
    lib.iblnerf_upload_weights_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.iblnerf_upload_weights_device.restype = C.c_int
    lib.iblnerf_upload_aux_weights.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.iblnerf_upload_aux_weights.restype = C.c_int
    lib.iblnerf_clear_aux.argtypes = [C.c_void_p, C.c_int]
    lib.iblnerf_clear_aux.restype = C.c_int
    lib.iblnerf_upload_lut.argtypes = [C.c_void_p, C.c_void_p]
    lib.iblnerf_pack_weights_host.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.iblnerf_pack_weights_host_f16x3.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.iblnerf_pack_weights_host_f16x3.restype = C.c_int
    lib.iblnerf_encode_host.argtypes = [C.c_float, C.c_int, C.c_void_p]
    lib.iblnerf_encode_host.restype = None
    lib.iblnerf_get_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_int, FP, FP]
    lib.iblnerf_get_rays_strided.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, FP, FP]
    lib.iblnerf_get_rays_strided.restype = C.c_int
    lib.iblnerf_get_rays_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, FP, C.c_int64, FP, FP]
    lib.iblnerf_get_rays_pixels.restype = C.c_int
    lib.iblnerf_escalate_route.argtypes = [C.c_void_p, C.c_int]
    lib.iblnerf_escalate_route.restype = C.c_int
    lib.iblnerf_set_lists.argtypes = [C.c_void_p, C.c_int]
    lib.iblnerf_set_lists.restype = C.c_int
    lib.iblnerf_set_tapped_lists.argtypes = [C.c_void_p, C.c_int]
    lib.iblnerf_set_tapped_lists.restype = C.c_int
    lib.iblnerf_network_query.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FP, C.c_int64, C.c_int, FP, FP]
    lib.iblnerf_density_gradient.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FP, C.c_int64, FP]
    lib.iblnerf_density_gradient.restype = C.c_int
    lib.iblnerf_trunk_density_fp32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FP, C.c_int64, FP]
    lib.iblnerf_trunk_density_fp32.restype = C.c_int
    lib.iblnerf_trunk_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FP, C.c_int64, FP, C.c_float, FP, FP]
    lib.iblnerf_trunk_backward.restype = C.c_int
    lib.iblnerf_trunk_features.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FP, C.c_int64, FP]
    lib.iblnerf_trunk_features.restype = C.c_int
    lib.iblnerf_trunk_features_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FP, C.c_int64, FP, C.c_float, FP, FP]
    lib.iblnerf_trunk_features_backward.restype = C.c_int
    lib.iblnerf_trunk_features2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FP, C.c_int64, C.c_int, FP, FP, FP]
    lib.iblnerf_trunk_features2.restype = C.c_int
    lib.iblnerf_trunk_features2_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FP, C.c_int64, C.c_int, FP, FP, FP, C.c_float, FP, FP]
    lib.iblnerf_trunk_features2_backward.restype = C.c_int
    lib.iblnerf_network_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FP, C.c_int64, C.c_int, FP, FP, C.c_float, FP, FP]
    lib.iblnerf_network_backward.restype = C.c_int
    lib.iblnerf_composite_direct.argtypes = [C.c_void_p, C.c_void_p, FP, FP, FP, C.c_int64, C.c_int, FP, FP]
    lib.iblnerf_composite_direct.restype = C.c_int
    lib.iblnerf_composite_direct_backward.argtypes = [C.c_void_p, C.c_void_p, FP, FP, FP, C.c_int64, C.c_int, FP, FP, FP]
    lib.iblnerf_composite_direct_backward.restype = C.c_int
    lib.iblnerf_composite_direct_backward_full.argtypes = [C.c_void_p, C.c_void_p, FP, FP, FP, C.c_int64, C.c_int, FP, FP, FP]
    lib.iblnerf_composite_direct_backward_full.restype = C.c_int
    lib.iblnerf_ray_outputs_backward.argtypes = [C.c_void_p, C.c_void_p, FP, FP, FP, C.c_float, C.POINTER(Maps), C.c_int64, FP]
    lib.iblnerf_ray_outputs_backward.restype = C.c_int
    lib.iblnerf_ray_outputs_backward_gt.argtypes = [C.c_void_p, C.c_void_p, FP, FP, FP, C.c_float, C.POINTER(Maps), C.POINTER(Overrides), C.c_int64, FP]
    lib.iblnerf_ray_outputs_backward_gt.restype = C.c_int
    lib.iblnerf_coarse_z.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, FP, C.c_int64, FP]
    lib.iblnerf_coarse_z.restype = C.c_int
    lib.iblnerf_ray_outputs_backward_env.argtypes = [C.c_void_p, C.c_void_p, FP, FP, FP, C.c_float, FP, C.POINTER(Maps), C.POINTER(Overrides), C.c_int64, FP, FP]
    lib.iblnerf_ray_outputs_backward_env.restype = C.c_int
    lib.iblnerf_set_chunk_cuts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.iblnerf_set_chunk_cuts.restype = C.c_int
    lib.iblnerf_set_select_tmin.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    lib.iblnerf_set_select_tmin.restype = C.c_int
    lib.iblnerf_aux_query.argtypes = [C.c_void_p, C.c_void_p, C.c_int, FP, C.c_int64, FP]
    lib.iblnerf_aux_query.restype = C.c_int
    lib.iblnerf_aux_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, FP, C.c_int64, FP, C.c_float, FP, FP]
    lib.iblnerf_aux_backward.restype = C.c_int
    lib.iblnerf_coarse_z_rays.argtypes = [C.c_void_p, C.c_void_p, FP, FP, FP, C.c_int64, FP]
    lib.iblnerf_coarse_z_rays.restype = C.c_int
    lib.iblnerf_ray_outputs_backward_rays.argtypes = [C.c_void_p, C.c_void_p, FP, FP, FP, FP, C.POINTER(Maps), C.POINTER(Overrides), C.c_int64, FP]
    lib.iblnerf_ray_outputs_backward_rays.restype = C.c_int
    lib.iblnerf_sample_points.argtypes = [C.c_void_p, C.c_void_p, FP, FP, FP, C.c_int64, C.c_int, FP]
    lib.iblnerf_sample_points.restype = C.c_int
    lib.iblnerf_fine_z.argtypes = [C.c_void_p, C.c_void_p, FP, FP, C.c_int64, FP, FP, FP]
    lib.iblnerf_fine_z.restype = C.c_int
    lib.iblnerf_composite_sigma.argtypes = [C.c_void_p, C.c_void_p, FP, FP, FP, C.c_int64, C.c_int, FP, FP, FP]
    lib.iblnerf_composite_sigma.restype = C.c_int
    lib.iblnerf_trim.argtypes = [C.c_void_p]
    lib.iblnerf_trim.restype = C.c_int
    lib.iblnerf_sample_pdf.argtypes = [C.c_void_p, C.c_void_p, FP, FP, C.c_int64, C.c_int, C.c_int, FP]
    lib.iblnerf_render_rays.argtypes = [C.c_void_p, C.c_void_p, FP, FP, C.c_int64, C.c_float, C.c_float,
                                        C.POINTER(Overrides), C.POINTER(Outputs)]
    lib.iblnerf_composite_pass.argtypes = [C.c_void_p, C.c_void_p, FP, FP, C.c_int64, C.c_float, C.c_float,
                                           C.POINTER(Overrides), C.POINTER(StageInputs), C.POINTER(Maps)]
    lib.iblnerf_composite_pass.restype = C.c_int
    lib.iblnerf_posdir_floats.argtypes = [C.c_int]
    lib.iblnerf_posdir_floats.restype = C.c_size_t
    lib.iblnerf_upload_posdir_mlp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.iblnerf_upload_posdir_mlp.restype = C.c_int
    lib.iblnerf_clear_posdir_mlp.argtypes = [C.c_void_p]
    lib.iblnerf_clear_posdir_mlp.restype = C.c_int
    lib.iblnerf_posdir_query.argtypes = [C.c_void_p, C.c_void_p, FP, FP, C.c_int64, FP]
    lib.iblnerf_posdir_query.restype = C.c_int
    lib.iblnerf_render_rays_sampled.argtypes = [C.c_void_p, C.c_void_p, FP, FP, C.c_int64, C.c_float, C.c_float,
                                                C.POINTER(Overrides), C.POINTER(Sampling), C.POINTER(Outputs)]
    lib.iblnerf_render_rays_sampled.restype = C.c_int
    lib.iblnerf_render_rays_tapped.argtypes = [C.c_void_p, C.c_void_p, FP, FP, C.c_int64, C.c_float, C.c_float,
                                               C.POINTER(Overrides), C.POINTER(Sampling), C.POINTER(Outputs), C.POINTER(Taps)]
    lib.iblnerf_render_rays_tapped.restype = C.c_int
    lib.iblnerf_sample_pdf_u.argtypes = [C.c_void_p, C.c_void_p, FP, FP, C.c_int64, C.c_int, C.c_int, FP, FP]
    lib.iblnerf_sample_pdf_u.restype = C.c_int
    lib.iblnerf_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.iblnerf_last_mlp_time.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    for n in ("iblnerf_create", "iblnerf_upload_weights", "iblnerf_upload_lut", "iblnerf_pack_weights_host",
              "iblnerf_get_rays", "iblnerf_network_query", "iblnerf_sample_pdf", "iblnerf_render_rays",
              "iblnerf_set_profiling", "iblnerf_last_mlp_time", "iblnerf_range_status", "iblnerf_pack_weights_host_mx"):
        getattr(lib, n).restype = C.c_int
    _lib = lib
    return lib


def default_options() -> Options:
    o = Options()
    load_library().iblnerf_default_options(C.byref(o))
    return o


def check(ctx, rc: int):
    if rc != 0:
        msg = load_library().iblnerf_last_error(ctx)
        raise IblNerfError("iblnerf status %d: %s" % (rc, (msg or b"").decode()))
