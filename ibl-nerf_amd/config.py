"""Reader for the reference's config files (`configs/**.txt`): `key = value` lines, bare `flag`
lines for store_true options, `#` comments, and the recursive `include = <relative path>` chain
that `recursive_config_parser` (src/config_parser.py:6-26) resolves base-first, so that a leaf
file overrides what it includes.  Only flags that reach the forward/inference path carry typed
defaults here (config_parser.py:35-271); every other key is kept verbatim as a string so nothing
is lost when a full training config is read."""
from __future__ import annotations

import ast
import os
from types import SimpleNamespace

# flag -> default, with the reference parser's types (config_parser.py line numbers in SURVEY.md §5)
DEFAULTS = dict(
    expname=None, basedir="./logs/", export_basedir=None, datadir="./data/llff/fern", dataset_type="mitsuba",
    netdepth=8, netwidth=256, multires=10, multires_views=4, i_embed=0, N_samples=64, N_importance=0,
    chunk=1024 * 16, netchunk=1024 * 64, perturb=1.0, raw_noise_std=0.0, render_factor=1, testskip=8,
    image_scale=1.0, near_plane=1.0, far_plane=20.0,
    coarse_radiance_number=0, target_load_N_iter=-1, ft_path=None, lut_coefficient="F",
    calculating_normal_type="ground_truth", epsilon_for_numerical_normal=0.01,
    epsilon_direction_for_numerical_normal=0.005,
    # store_true flags
    no_reload=False, use_viewdirs=False, white_bkgd=False, lindisp=False, gamma_correct=False,
    color_independent_to_direction=False, correct_depth_for_prefiltered_radiance_infer=False,
    use_radiance_linear=False, use_environment_map=False, use_gradient_for_incident_radiance=False,
    use_illumination_feature_layer=False, use_instance_feature_layer=False, load_depth_range_from_file=False,
    infer_normal=False, infer_normal_at_surface=False, infer_depth=False, infer_visibility=False,
    infer_albedo_separate=False, infer_roughness_separate=False, infer_irradiance_separate=False,
    calculate_irradiance_from_gt=False, calculate_roughness_from_gt=False, calculate_albedo_from_gt=False,
    depth_map_from_ground_truth=False,
    edit_intrinsic=False, editing_img_idx=0, edit_roughness=False, edit_albedo=False, edit_normal=False,
    edit_depth=False, num_edit_objects=1, edit_albedo_by_img=False, edit_normal_by_img=False,
    edit_roughness_by_img=False, edit_irradiance_by_img=False, editing_target_roughness_list=None,
    editing_target_albedo_list=None, editing_target_irradiance_list=None,
    insert_object=False, inserting_img_idx=0, num_insert_objects=1, inserting_target_roughness_list=None,
    inserting_target_albedo_list=None, inserting_target_irradiance_list=None,
)
_LIST_FLAGS = {k for k in DEFAULTS if k.endswith("_list")}


def _parse_file(path):
    """-> (include or None, [(key, raw value or None)...]) in file order."""
    include, items = None, []
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            if "=" in line:
                k, v = (s.strip() for s in line.split("=", 1))
            elif ":" in line and " " not in line.split(":", 1)[0]:
                k, v = (s.strip() for s in line.split(":", 1))
            else:
                k, v = line, None
            if k == "include":
                include = v
            else:
                items.append((k, v))
    return include, items


def include_chain(path):
    """Files in application order: deepest include first, the given file last
    (config_parser.py:6-26: includes become default_config_files in reversed discovery order)."""
    chain, seen = [], set()
    while path is not None:
        path = os.path.normpath(path)
        if path in seen:
            raise ValueError("config include cycle at %s" % path)
        seen.add(path)
        chain.append(path)
        inc, _ = _parse_file(path)
        path = os.path.join(os.path.dirname(path), inc) if inc else None
    return list(reversed(chain))


def _convert(key, raw):
    if key in _LIST_FLAGS:
        v = ast.literal_eval(raw) if raw is not None else []
        return [float(x) for x in (v if isinstance(v, (list, tuple)) else [v])]
    if key not in DEFAULTS:
        return True if raw is None else raw
    d = DEFAULTS[key]
    if isinstance(d, bool):
        if raw is None:
            return True
        if raw.lower() in ("true", "1", "yes"):
            return True
        if raw.lower() in ("false", "0", "no"):
            return False
        raise ValueError("boolean flag %s has value %r" % (key, raw))
    if raw is None:
        raise ValueError("flag %s needs a value" % key)
    if isinstance(d, int) and not isinstance(d, bool):
        return int(float(raw)) if raw.replace(".", "", 1).lstrip("-").isdigit() else int(raw)
    if isinstance(d, float):
        return float(raw)
    return None if raw == "None" else raw


def load_config(path, **overrides):
    """Effective flag namespace of a config file (the `args` the reference's CLIs build)."""
    values = dict(DEFAULTS)
    for f in include_chain(path):
        for k, raw in _parse_file(f)[1]:
            values[k] = _convert(k, raw)
    values.update(overrides)
    for k in _LIST_FLAGS:
        if values.get(k) is None:
            values[k] = []
    if values.get("expname") is None:                      # test.py:160-163
        values["expname"] = os.path.basename(path).split(".")[0]
    values["config"] = path
    return SimpleNamespace(**values)


def edit_params(args):
    """The 20 edit/insert kwargs test.py:115-139 forwards to render_decomp_path."""
    keys = ["edit_intrinsic", "editing_img_idx", "num_edit_objects", "edit_roughness", "edit_albedo", "edit_normal",
            "edit_depth", "edit_albedo_by_img", "edit_normal_by_img", "edit_roughness_by_img", "edit_irradiance_by_img",
            "editing_target_roughness_list", "editing_target_albedo_list", "editing_target_irradiance_list",
            "insert_object", "inserting_img_idx", "num_insert_objects", "inserting_target_roughness_list",
            "inserting_target_irradiance_list", "inserting_target_albedo_list"]
    return {k: getattr(args, k) for k in keys}
