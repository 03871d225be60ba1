"""Config include-chain reader (ibl-nerf_amd/config.py) — effective values of the shipped
Kitchen / Living-room-2 configs as listed in SURVEY.md §5 and Appendix D.  The config text below is
test data written for this test (same keys/values as the reference's files, which are exercised
directly when /root/reference is present)."""
import os

import pytest

from ibl_nerf_amd import config as Cfg

FILES = {
    "common.txt": """basedir = ../logs
lindisp = False
correct_depth_for_prefiltered_radiance_infer = True
use_viewdirs = True
N_samples = 64
N_importance = 128
# if calculate normal chunk should be small
chunk = 1024
coarse_radiance_number = 3
calculating_normal_type = normal_map_from_depth_gradient_epsilon
""",
    "IBL-NeRF/common.txt": """include = ../common.txt
basedir = ../logs/IBL-NeRF
dataset_type = mitsuba
load_depth_range_from_file
testskip = 32
""",
    "IBL-NeRF/kitchen/common.txt": """include = ../common.txt
datadir = ../data/IBL-NeRF/kitchen
basedir = ../logs/IBL-NeRF/kitchen
""",
    "IBL-NeRF/kitchen/IBL-NeRF.txt": "include = common.txt\ngamma_correct=True\nrender_factor = 1\n",
    "IBL-NeRF/kitchen/edit_intrinsic.txt": """include = common.txt
expname = IBL-NeRF
gamma_correct=True
edit_intrinsic
editing_img_idx = 14
num_edit_objects = 1
edit_roughness
edit_normal
editing_target_roughness_list = [0]
edit_normal_by_img
""",
    "IBL-NeRF/kitchen/object_insert.txt": """include = common.txt
expname = IBL-NeRF
insert_object
num_insert_objects = 4
inserting_target_roughness_list = [1, 1, 1, 1]
inserting_target_albedo_list = [0.870588, 0.3215686, 0.443137254, 0.05, 0.05, 0.05, 0.2, 0.2, 0.2, 0.05, 0.05, 0.05]
inserting_target_irradiance_list = [0.5, 0.1, 0.2, 0.2]
""",
}


@pytest.fixture()
def cfgdir(tmp_path):
    for rel, text in FILES.items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    return tmp_path


def check_effective(a):
    assert (a.N_samples, a.N_importance, a.chunk, a.netchunk) == (64, 128, 1024, 65536)
    assert (a.multires, a.multires_views, a.netdepth, a.netwidth, a.coarse_radiance_number) == (10, 4, 8, 256, 3)
    assert a.use_viewdirs is True and a.lindisp is False and a.gamma_correct is True
    assert a.lut_coefficient == "F" and a.calculating_normal_type == "normal_map_from_depth_gradient_epsilon"
    assert a.epsilon_for_numerical_normal == 0.01 and a.correct_depth_for_prefiltered_radiance_infer is True
    assert a.load_depth_range_from_file is True and a.dataset_type == "mitsuba" and a.render_factor == 1


def test_include_chain_and_effective_values(cfgdir):
    leaf = str(cfgdir / "IBL-NeRF/kitchen/IBL-NeRF.txt")
    chain = Cfg.include_chain(leaf)
    assert [os.path.relpath(c, cfgdir) for c in chain] == ["common.txt", "IBL-NeRF/common.txt",
                                                            "IBL-NeRF/kitchen/common.txt", "IBL-NeRF/kitchen/IBL-NeRF.txt"]
    a = Cfg.load_config(leaf)
    check_effective(a)
    assert a.basedir == "../logs/IBL-NeRF/kitchen" and a.expname == "IBL-NeRF" and a.testskip == 32
    assert a.edit_intrinsic is False and a.editing_target_roughness_list == []
    e = Cfg.load_config(str(cfgdir / "IBL-NeRF/kitchen/edit_intrinsic.txt"))
    assert e.edit_intrinsic and e.edit_roughness and e.edit_normal and e.edit_normal_by_img and not e.edit_albedo
    assert e.editing_target_roughness_list == [0.0] and e.editing_img_idx == 14 and e.num_edit_objects == 1
    o = Cfg.load_config(str(cfgdir / "IBL-NeRF/kitchen/object_insert.txt"))
    assert o.insert_object and o.num_insert_objects == 4 and o.inserting_target_roughness_list == [1.0] * 4
    assert len(o.inserting_target_albedo_list) == 12 and o.inserting_target_irradiance_list == [0.5, 0.1, 0.2, 0.2]
    assert set(Cfg.edit_params(o)) == set(Cfg.edit_params(e)) and len(Cfg.edit_params(o)) == 20
    # the factory accepts the namespace directly
    from ibl_nerf_amd import model as M
    os.makedirs(cfgdir / "logs" / "x")
    a2 = Cfg.load_config(leaf, basedir=str(cfgdir / "logs"), expname="x")
    _, kw, *_ = M.create_IBLNeRF(a2)
    assert kw["N_importance"] == 128 and kw["gamma_correct"] is True and kw["perturb"] is False


def test_include_cycle_and_bad_values(tmp_path):
    (tmp_path / "a.txt").write_text("include = b.txt\n")
    (tmp_path / "b.txt").write_text("include = a.txt\n")
    with pytest.raises(ValueError):
        Cfg.include_chain(str(tmp_path / "a.txt"))
    (tmp_path / "c.txt").write_text("gamma_correct = maybe\n")
    with pytest.raises(ValueError):
        Cfg.load_config(str(tmp_path / "c.txt"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference configs only exist in the build container")
def test_reference_config_files():
    for scene in ("kitchen", "living-room-2"):
        check_effective(Cfg.load_config("/root/reference/configs/IBL-NeRF/%s/IBL-NeRF.txt" % scene))
    e = Cfg.load_config("/root/reference/configs/IBL-NeRF/kitchen/edit_intrinsic.txt")
    assert e.edit_intrinsic and e.editing_target_roughness_list == [0.0]
    o = Cfg.load_config("/root/reference/configs/IBL-NeRF/living-room-2/object_insert.txt")
    assert o.insert_object and o.num_insert_objects == 4 and len(o.inserting_target_albedo_list) == 12


def _same(ours, ref, key):
    if key.endswith("_list"):
        return [float(x) for x in (ours or [])] == [float(x) for x in (ref or [])]     # the reference keeps an unset list as None
    if isinstance(ref, (int, float)) and not isinstance(ref, bool):
        return float(ours) == float(ref)
    return ours == ref


def test_values_match_the_reference_parser(cfgdir):
    """The effective values the REFERENCE's recursive_config_parser gives for the config texts above (tests/golden/
    io_config_expected.json, produced by tests/golden/make_io_golden.py with a configargparse stand-in: pinned as far as that
    stand-in is faithful), key by key for every flag this reader types."""
    import json
    from conftest import GOLDEN
    expected = json.load(open(os.path.join(GOLDEN, "io_config_expected.json")))
    assert sorted(expected) == sorted(FILES)
    for rel, ref in expected.items():
        ours = vars(Cfg.load_config(str(cfgdir / rel)))
        for key, want in ref.items():
            if key == "expname" and want is None:
                assert ours[key] == os.path.basename(rel).split(".")[0]              # test.py:160-163 fills it from the file name
                continue
            assert _same(ours[key], want, key), (rel, key, ours[key], want)


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="needs the reference checkout (build container only)")
def test_shipped_configs_against_the_reference_parser_live():
    """Every config file the reference ships for the IBL-NeRF scenes, read by the reference's own parser (fresh process, stand-in
    for configargparse) and by this reader."""
    import glob
    import json
    import subprocess
    import sys
    from conftest import GOLDEN
    files = sorted(glob.glob("/root/reference/configs/IBL-NeRF/*/*.txt"))
    assert len(files) >= 10
    checked = 0
    for path in files:
        out = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_io_golden.py"), "--dump-config", path], capture_output=True, text=True,
                             timeout=120, cwd=os.path.dirname(path))
        assert out.returncode == 0, (path, out.stderr[-800:])
        ref = json.loads(out.stdout.strip().splitlines()[-1])
        ours = vars(Cfg.load_config(path))
        for key in Cfg.DEFAULTS:
            if key not in ref or (key == "expname" and ref[key] is None):
                continue
            assert _same(ours[key], ref[key], key), (path, key, ours[key], ref[key])
            checked += 1
    assert checked > 700
