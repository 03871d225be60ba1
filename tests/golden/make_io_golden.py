#!/usr/bin/env python3
"""Pins the on-disk format readers by RUNNING THE REFERENCE'S OWN reader and parser in this container.

    python tests/golden/make_io_golden.py                 # writes io_dataset_expected.npz, io_config_expected.json
    python tests/golden/make_io_golden.py --dump-config F # prints the reference parser's values for config file F as JSON

The reference imports four packages this image lacks.  They are replaced by MINIMAL WORKING stand-ins defined in this file (never
shipped, never imported by the product), so a fixture produced here pins the reference's semantics only as far as a stand-in is
faithful to the package it stands for:
  cv2.imread / cvtColor / COLOR_BGR2RGB   PIL decode to 8-bit RGB, reversed to BGR and back (cv2.resize is not provided: fixtures use
                                          image_scale 1; the 0.5 path stays unpinned)
  imageio.imread(path, pilmode='RGB')     PIL
  torchvision.transforms.Resize           torch.nn.functional.interpolate(bilinear, antialias) — what torchvision calls for tensors
  configargparse.ArgumentParser           an argparse subclass: `key = value` / `key: value` / bare `flag` lines, [a, b] lists fed to
                                          action="append", store_true flags taking true / false, default_config_files applied in
                                          order before the --config file, the command line last (configargparse's documented rules)
Inputs are the synthetic scene of tests/test_dataset.py (write_scene) and the config texts of tests/test_config.py (FILES): test data
written for this repository, not reference files.
"""
import argparse
import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))   # tests/test_dataset.py imports the oracle at module level


# ---------------------------------------------------------------------------------------------- stand-ins
def _install_standins():
    import torch
    from PIL import Image

    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2RGB = 4

    def imread(path):
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1])

    def cvtColor(img, code):
        assert code == cv2.COLOR_BGR2RGB
        return np.ascontiguousarray(img[..., ::-1])

    def resize(*a, **k):
        raise NotImplementedError("cv2.resize has no stand-in: fixtures are generated at image_scale 1")

    cv2.imread, cv2.cvtColor, cv2.resize = imread, cvtColor, resize
    imageio = types.ModuleType("imageio")
    imageio.imread = lambda path, pilmode="RGB": np.asarray(Image.open(path).convert(pilmode))
    imageio.imwrite = lambda *a, **k: None
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Resize:
        def __init__(self, size, antialias=True):
            self.size, self.antialias = tuple(size), antialias

        def __call__(self, x):
            return torch.nn.functional.interpolate(x[None], size=self.size, mode="bilinear", align_corners=False, antialias=self.antialias)[0]

    tvt.Resize = Resize
    tv.transforms = tvt

    cap = types.ModuleType("configargparse")

    class ArgumentParser(argparse.ArgumentParser):
        def __init__(self, *a, default_config_files=None, **k):
            super().__init__(*a, **k)
            self._default_config_files = list(default_config_files or [])
            self._config_file_actions = []

        def add_argument(self, *a, is_config_file=False, **k):
            act = super().add_argument(*a, **k)
            if is_config_file:
                self._config_file_actions.append(act)
            return act

        @staticmethod
        def _items(path):
            with open(path) as f:
                for line in f:
                    line = line.strip()
                    if not line or line[0] in "#;[" or line.startswith("---"):
                        continue
                    for sep in ("=", ":", " "):
                        if sep in line:
                            k, v = (s.strip() for s in line.split(sep, 1))
                            break
                    else:
                        k, v = line, "true"
                    yield k, v.split(" #")[0].strip() if sep != " " or True else v

        def parse_args(self, args=None, namespace=None):
            args = sys.argv[1:] if args is None else (args.split() if isinstance(args, str) else list(args))
            files = list(self._default_config_files)
            for act in self._config_file_actions:
                for opt in act.option_strings:
                    if opt in args:
                        files.append(args[args.index(opt) + 1])
            by_opt = {o: a for a in self._actions for o in a.option_strings}
            cfg_args = []
            for path in files:
                for key, value in self._items(path):
                    act = by_opt.get("--" + key)
                    if act is None:
                        self.error("unrecognized config key: %s" % key)
                    if any(o in args for o in act.option_strings):
                        continue                               # the command line wins
                    if isinstance(act, argparse._StoreTrueAction):
                        if value.lower() in ("true", "yes", "1"):
                            cfg_args.append("--" + key)
                        elif value.lower() not in ("false", "no", "0"):
                            self.error("%s takes true / false" % key)
                    elif value.startswith("[") and value.endswith("]"):
                        for elt in (e.strip() for e in value[1:-1].split(",")):
                            if elt:
                                cfg_args += ["--" + key, elt]
                    else:
                        cfg_args += ["--" + key, value]
            return super().parse_args(cfg_args + args, namespace)

    cap.ArgumentParser = ArgumentParser
    for name, mod in (("cv2", cv2), ("imageio", imageio), ("torchvision", tv), ("torchvision.transforms", tvt), ("configargparse", cap)):
        sys.modules[name] = mod
    sys.path.insert(0, os.path.join(REF, "src"))


def _jsonable(v):
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    return v


def reference_config_values(path, keys=None):
    """vars(recursive_config_parser().parse_args()) of the reference for `--config path` (config_parser.py:6-26)."""
    import config_parser as CP
    argv, cwd = sys.argv, os.getcwd()
    try:
        sys.argv = ["x", "--config", path]
        args = CP.recursive_config_parser().parse_args()
    finally:
        sys.argv = argv
        os.chdir(cwd)
    d = vars(args)
    return {k: _jsonable(d[k]) for k in (keys or d) if k in d}


# test.py:57-76: what the reference's test driver asks its reader for
def load_params(**over):
    d = dict(image_scale=1, coarse_radiance_number=0,   # (3 in test.py: the prefiltered training targets need images of 64+ pixels)
              near_plane=1.0, far_plane=20.0, load_depth_range_from_file=True, gamma_correct=True,
             load_priors=False, load_edit_intrinsic_mask=False, load_edit_albedo=False, load_edit_normal=False, load_edit_irradiance=False,
             load_edit_depth=False, object_insert=False, editing_idx=None)
    d.update(over)
    return d


MODES = {"plain": dict(skip=2), "all": dict(skip=1),
         "edit": dict(skip=1, load_edit_intrinsic_mask=True, load_edit_albedo=True, load_edit_normal=True, load_edit_depth=True, editing_idx=2),
         "insert": dict(skip=1, object_insert=True, editing_idx=3),
         "train": dict(split="train", load_priors=True, coarse_radiance_number=1)}     # train.py:100-130's reader: priors + one prefiltered target level


def read_with(load_dataset, root, mode, as_numpy):
    ds = load_dataset("mitsuba", str(root), **dict(dict(split="test"), **load_params(**MODES[mode])))
    ds.load_all_data(num_of_workers=0)
    ds.to_tensor("cpu")
    out = {"hwf": np.array([ds.height, ds.width, ds.focal], np.float64), "near_far": np.array([ds.near, ds.far], np.float64),
           "len": np.int64(len(ds)), "K": np.asarray(ds.get_focal_matrix()), "poses": as_numpy(ds.poses), "images": as_numpy(ds.images)}
    for i in range(len(ds)):
        for k, v in ds.get_resized_normal_albedo(1, i).items():
            out["gt%d__%s" % (i, k)] = as_numpy(v)
    if mode == "train":
        out["prior_irradiance_mean"] = np.float64(ds.prior_irradiance_mean)
        out["prefiltered_1"] = as_numpy(ds.prefiltered_images[0])
        for k, v in ds.get_info(1, np.array([0, 3, 7]), np.array([5, 2, 0])).items():
            out["info__" + k] = as_numpy(v)
    return out


def main():
    _install_standins()
    if len(sys.argv) > 2 and sys.argv[1] == "--dump-config":
        print(json.dumps(reference_config_values(os.path.abspath(sys.argv[2]))))
        return
    import _pkg
    _pkg.load()
    import test_config as TC
    import test_dataset as TD
    from ibl_nerf_amd import config as Cfg
    # ---- configs
    expected = {}
    with tempfile.TemporaryDirectory() as d:
        for rel, text in TC.FILES.items():
            p = Path(d) / rel
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_text(text)
        for rel in TC.FILES:
            expected[rel] = reference_config_values(str(Path(d) / rel), list(Cfg.DEFAULTS))
    json.dump(expected, open(os.path.join(HERE, "io_config_expected.json"), "w"), indent=1, sort_keys=True)
    print("io_config_expected.json: %d config files x %d keys" % (len(expected), len(next(iter(expected.values())))))
    # ---- dataset
    from dataset.dataset_interface import load_dataset
    out = {}
    with tempfile.TemporaryDirectory() as d:
        root = Path(d) / "data" / "tiny"
        os.makedirs(root)
        TD.write_scene(root)
        for mode in MODES:
            for k, v in read_with(load_dataset, root, mode, lambda t: t.numpy() if hasattr(t, "numpy") else np.asarray(t)).items():
                out["%s__%s" % (mode, k)] = v
    np.savez_compressed(os.path.join(HERE, "io_dataset_expected.npz"), **out)
    print("io_dataset_expected.npz: %d arrays (%s ...)" % (len(out), sorted(out)[:4]))


if __name__ == "__main__":
    main()
