"""render_decomp_path mirror (ibl-nerf_amd/export.py) against what the REFERENCE's own
render_decomp_path returned and handed to imageio.imwrite (tests/golden/export_path.npz).
CPU part: the export mapping with the oracle as renderer.  GPU part: with the HIP renderer."""
import os

import numpy as np
import pytest

import iblnerf_oracle as O
from conftest import GOLDEN, rel_linf
from ibl_nerf_amd import checkpoint as ck
from ibl_nerf_amd import export as E


class FakeDataset:
    def __init__(self, g):
        self.poses = g["poses"]
        self.far = float(g["far"])

    def get_resized_normal_albedo(self, render_factor, i):
        return {}


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "export_path.npz"))


def check_against_golden(res, savedir, g, float_tol, max_lsb_off_frac):
    names = sorted(k[5:] for k in g.files if k.startswith("res__"))
    assert sorted(res.keys()) == names
    for k in names:
        assert res[k].shape == g["res__" + k].shape, k
        tol = 5e-3 if ("normal" in k or k in ("rgb", "specular", "n_dot_v", "prefiltered_reflected", "reflected_radiance")
                       or k.startswith("reflected_coarse")) else float_tol
        assert rel_linf(res[k], g["res__" + k]) <= tol, (k, rel_linf(res[k], g["res__" + k]))
    from PIL import Image
    pngs = sorted(k[5:] for k in g.files if k.startswith("png__"))
    assert sorted(os.listdir(savedir)) == pngs                         # same file names <out_name>_{i:03d}.png
    for name in pngs:
        got = np.asarray(Image.open(os.path.join(savedir, name)))
        ref = g["png__" + name]
        ref = ref[..., 0] if ref.ndim == 3 and ref.shape[-1] == 1 else ref
        assert got.shape == ref.shape and got.dtype == np.uint8, name
        diff = np.abs(got.astype(int) - ref.astype(int))
        assert diff.max() <= (3 if "normal" in name else 1), name      # to8b truncates: 1e-6 float differences flip an LSB
        assert (diff > 0).mean() <= max_lsb_off_frac, (name, (diff > 0).mean())


def test_export_mapping_with_oracle_renderer(gold, lut, tmp_path):
    g = gold
    sdc, sdf = ck.synthetic_state_dict(int(g["seed_coarse"])), ck.synthetic_state_dict(int(g["seed_fine"]))

    def render_fn(H, W, K, chunk, c2w, gt_values, **kw):
        return O.render_decomp(H, W, K, sdc, sdf, lut, float(g["near"]), float(g["far"]), c2w=np.asarray(c2w),
                               n_importance=int(g["n_importance"]))

    res = E.render_decomp_path(FakeDataset(g), (int(g["H"]), int(g["W"]), float(g["focal"])), None, 1024,
                               {"coarse_radiance_number": 3}, savedir=str(tmp_path), render_factor=1, render_fn=render_fn)
    check_against_golden(res, str(tmp_path), g, float_tol=2e-5, max_lsb_off_frac=0.35)


def test_depth_to_normal_image_space_teacher_forced(gold):
    g = gold
    H, W, f = int(g["H"]), int(g["W"]), float(g["focal"])
    K = np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]], dtype=np.float32)
    for i in range(2):
        depth = (np.float32(0.1 * float(g["far"])) / g["res__depth"][i]).astype(np.float32)     # undo the export mapping
        n = E.depth_to_normal_image_space(depth, g["poses"][i][:3, :4], K)
        ref = g["res__normal_from_depth"][i] * 2 - 1
        assert np.abs(n - ref).max() <= 2e-3       # a cross product of small central differences of float32 positions
    assert E.to8b(np.array([0.0, 0.999, 1.0, 1.7, -0.2, 0.5])).tolist() == [0, 254, 255, 255, 0, 127]
    assert abs(E.psnr(np.full(4, 0.5), np.full(4, 0.6)) - 20.0) < 1e-9


@pytest.mark.gpu
def test_export_path_with_hip_renderer(gold, lut, tmp_path):
    torch = pytest.importorskip("torch")
    from ibl_nerf_amd import model as M
    g = gold
    os.makedirs(tmp_path / "exp")
    _, kw, *_ = M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), N_importance=int(g["n_importance"])))
    kw["network_fn"].load_state_dict(ck.synthetic_state_dict(int(g["seed_coarse"])))
    kw["network_fine"].load_state_dict(ck.synthetic_state_dict(int(g["seed_fine"])))
    kw.update(near=float(g["near"]), far=float(g["far"]), brdf_lut=torch.from_numpy(lut))
    ds = FakeDataset(g)
    ds.poses = torch.from_numpy(g["poses"])
    out = tmp_path / "png"
    res = E.render_decomp_path(ds, (int(g["H"]), int(g["W"]), float(g["focal"])), None, 1024, kw, savedir=str(out),
                               render_factor=1, approximate_radiance=True)
    check_against_golden(res, str(out), g, float_tol=2e-4, max_lsb_off_frac=0.35)
