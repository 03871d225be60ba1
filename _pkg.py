"""Loader for the package directory `ibl-nerf_amd/` (the hyphen makes it un-importable by name).

    import _pkg; pkg = _pkg.load()      # registers it as `ibl_nerf_amd` in sys.modules
    from ibl_nerf_amd import renderer   # works afterwards
"""
import importlib.util
import os
import sys

_NAME = "ibl_nerf_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ibl-nerf_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
