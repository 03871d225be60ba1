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


def test_ssim_and_metrics(tmp_path):
    """piq is absent: SSIM is checked against an independent torch.conv2d evaluation of the same published
    definition and against the properties the definition implies."""
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F
    rs = np.random.RandomState(0)
    a = rs.uniform(0, 1, (40, 52, 3))
    b = np.clip(a + 0.1 * rs.randn(*a.shape), 0, 1)
    assert abs(E.ssim(a, a) - 1.0) < 1e-12 and abs(E.ssim(a, b) - E.ssim(b, a)) < 1e-12
    assert E.ssim(a, b) > E.ssim(a, np.clip(a + 0.3 * rs.randn(*a.shape), 0, 1))

    def ssim_torch(x, y):                                        # grouped 'valid' convolution, as piq builds it
        k = torch.from_numpy(E._gaussian_kernel()).repeat(3, 1, 1, 1)
        x, y = (torch.from_numpy(t).permute(2, 0, 1)[None] for t in (x, y))
        f = max(1, round(min(x.shape[-2:]) / 256))
        if f > 1:
            x, y = F.avg_pool2d(x, f), F.avg_pool2d(y, f)
        mx, my = F.conv2d(x, k, groups=3), F.conv2d(y, k, groups=3)
        sxx, syy, sxy = F.conv2d(x * x, k, groups=3) - mx * mx, F.conv2d(y * y, k, groups=3) - my * my, F.conv2d(x * y, k, groups=3) - mx * my
        cs = (2 * sxy + 0.03 ** 2) / (sxx + syy + 0.03 ** 2)
        return float(((2 * mx * my + 0.01 ** 2) / (mx * mx + my * my + 0.01 ** 2) * cs).mean((-1, -2)).mean())

    assert abs(E.ssim(a, b) - ssim_torch(a, b)) < 1e-10
    big_a = rs.uniform(0, 1, (700, 650, 3))                      # f = round(650/256) = 3: pooled to 233 x 216
    big_b = np.clip(big_a + 0.05 * rs.randn(*big_a.shape), 0, 1)
    assert abs(E.ssim(big_a, big_b) - ssim_torch(big_a, big_b)) < 1e-10
    with pytest.raises(ValueError):
        E.ssim(a[:8], b[:8])
    from PIL import Image
    os.makedirs(tmp_path / "gt" / "test")
    os.makedirs(tmp_path / "pred")
    for i in range(2):
        Image.fromarray(E.to8b(a)).save(tmp_path / "gt" / "test" / ("%d.png" % (i + 1)))
        Image.fromarray(E.to8b(b if i else a)).save(tmp_path / "pred" / ("rgb_%03d.png" % i))
    m = E.calculate_metrics(str(tmp_path / "gt"), str(tmp_path / "pred"), "rgb", n_views=2)
    assert m["mse"][0] == 0.0 and m["psnr"][0] == float("inf") and abs(m["ssim"][0] - 1) < 1e-12
    assert 0 < m["ssim"][1] < 1 and abs(m["psnr"][1] - 10 * np.log10(1 / m["mse"][1])) < 1e-9
