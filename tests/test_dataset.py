"""Mitsuba scene reader (ibl-nerf_amd/dataset.py) and the test.py-shaped driver (render_views.py) on a tiny
synthetic scene written in the reference's on-disk layout (dataset_mitsuba.py:8-137).  The
reference's own reader needs cv2 / imageio / torchvision, none of which is in this image, so its
behaviour is pinned here by the semantics stated in its source (cited per assertion), not by a run.
CPU: reader + driver with the oracle as renderer.  GPU: the same chain with the HIP renderer."""
import json
import os

import numpy as np
import pytest

import iblnerf_oracle as O
from ibl_nerf_amd import checkpoint as ck
from ibl_nerf_amd import config as C
from ibl_nerf_amd import dataset as DS
from ibl_nerf_amd import render_views as RT

H, W, FOV = 6, 8, 50.0
N_TEST = 3


def look(i):
    """A Mitsuba-convention camera-to-world (forward +Z): small seeded rotation + translation."""
    rs = np.random.RandomState(100 + i)
    q, _ = np.linalg.qr(np.eye(3) + 0.1 * rs.randn(3, 3))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    m = np.eye(4)
    m[:3, :3] = q
    m[:3, 3] = 0.2 * rs.randn(3)
    return m


def write_scene(root):
    from PIL import Image
    rs = np.random.RandomState(7)
    os.makedirs(root / "train")
    os.makedirs(root / "test")
    img = lambda c=3: rs.randint(0, 256, (H, W, c)).astype(np.uint8)
    Image.fromarray(img()).save(root / "train" / "1.png")
    truth = {}
    for n in range(1, N_TEST + 1):
        truth[n] = {}
        for suffix in ("", "_edit_normal", "_insert_normal"):
            a = img()
            truth[n][suffix] = a
            Image.fromarray(a).save(root / "test" / ("%d%s.png" % (n, suffix)))
        rgba = np.concatenate([img(), np.full((H, W, 1), 128, np.uint8)], -1)       # alpha must be dropped
        truth[n]["_edit_albedo"] = rgba[..., :3]
        Image.fromarray(rgba).save(root / "test" / ("%d_edit_albedo.png" % n))
        mask = np.zeros((H, W, 3), np.uint8)
        mask[1:4, 2:6] = 10                                                          # object 1 (decode: round(255 m / 10))
        mask[4:, :2] = 20                                                            # object 2
        truth[n]["mask"] = mask
        Image.fromarray(mask).save(root / "test" / ("%d_edit_intrinsic_mask.png" % n))
        Image.fromarray(mask).save(root / "test" / ("%d_insert_mask.png" % n))
        d = rs.uniform(1.0, 2.0, (H, W))
        truth[n]["insert_depth"] = d
        np.save(root / "test" / ("%d_insert_depth.npy" % n), d)                      # float64 on disk -> float32 in memory
        np.save(root / "test" / ("%d_edit_depth.npy" % n), d.astype(np.float32) + 1)
    for n in range(1, N_TEST + 1):       # the train split: images (1.png exists), prior albedo / irradiance of prior_type "bell" (dataset_mitsuba.py:66-67)
        if n > 1:
            Image.fromarray(img()).save(root / "train" / ("%d.png" % n))
        Image.fromarray(img()).save(root / "train" / ("%d_bell_r.png" % n))
        Image.fromarray(img()).save(root / "train" / ("%d_bell_s.png" % n))
    json.dump({"mean_bell": 0.4375, "mean_other": 0.9}, open(root / "avg_irradiance.json", "w"))
    frames = [{"fov_degree": FOV, "transform": look(i).tolist()} for i in range(N_TEST)]
    for split in ("train", "test"):
        json.dump({"frames": frames}, open(root / ("transforms_%s.json" % split), "w"))
    json.dump({"min_depth": 1.0, "max_depth": 5.0}, open(root / "min_max_depth.json", "w"))
    return truth


@pytest.fixture()
def scene(tmp_path):
    root = tmp_path / "data" / "tiny"
    os.makedirs(root)
    return root, write_scene(root)


def test_reader_plain(scene):
    root, truth = scene
    ds = DS.load_dataset("mitsuba", str(root), split="test", skip=1, near_plane=1.0, far_plane=20.0)
    assert (ds.height, ds.width, len(ds)) == (H, W, N_TEST) and ds.scene_name == "tiny"
    assert ds.get_near_far_plane() == {"near": 1.0, "far": 20.0}                      # dataset_interface.py:64-65
    f = 0.5 * W / np.tan(0.5 * FOV / 180.0 * np.pi)                                    # dataset_mitsuba.py:34-44
    assert abs(ds.focal - f) < 1e-12
    K = ds.get_focal_matrix()
    assert K.dtype == np.float32 and np.array_equal(K, np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32))
    ds.load_all_data(num_of_workers=1)
    assert len(ds.poses) == N_TEST and len(ds.images) == N_TEST
    for i in range(N_TEST):
        m = look(i).astype(np.float32)
        m[:3, 0] *= -1                                                                 # :128-130
        m[:3, 2] *= -1
        assert np.array_equal(ds.poses[i], m)
        assert np.array_equal(ds.images[i], truth[i + 1][""].astype(np.float32) / np.float32(255))   # file <skip*i+1>.png
    assert ds.get_resized_normal_albedo(1, 0) == {}                                   # nothing but images was requested
    ds2 = DS.load_dataset("mitsuba", str(root), split="test", skip=2, load_depth_range_from_file=True)
    assert len(ds2) == 2 and (ds2.near, ds2.far) == (0.9, 5.5)                        # :12-16
    assert np.array_equal(ds2[1]["image"], truth[3][""].astype(np.float32) / np.float32(255))        # n = skip*index + 1
    assert np.array_equal(ds2[1]["pose"][:3, 3], look(2)[:3, 3].astype(np.float32))
    assert DS.load_dataset("mitsuba", str(root), split="train", skip=5).skip == 1     # :30-32
    with pytest.raises(ValueError):
        DS.load_dataset("blender", str(root))
    quarter = DS.load_dataset("mitsuba", str(root), split="test", image_scale=0.25)    # any other scale: OpenCV's general bilinear path
    assert quarter[0]["image"].shape == (int(np.rint(H * 0.25)), int(np.rint(W * 0.25)), 3) and abs(quarter.focal - f / 4) < 1e-12
    half = DS.load_dataset("mitsuba", str(root), split="test", image_scale=0.5)        # configs/real use 0.5
    assert (half.height, half.width) == (H // 2, W // 2) and abs(half.focal - f / 2) < 1e-12
    a = truth[1][""].astype(np.uint32)                                                 # 2x2 cell mean, rounded half up
    want = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.float32) / np.float32(255)
    assert np.array_equal(half[0]["image"], want)


def test_reader_colmap(tmp_path):
    """dataset_colmap.py: one transforms.json for all splits, every 8th frame is the test view, the other seven of each
    group of eight train views; poses are taken as stored (no axis flips); size and focal from h / w / camera_angle_x."""
    from PIL import Image
    root = tmp_path / "real" / "desk"
    os.makedirs(root / "images")
    n, hh, ww, ang = 19, 4, 6, 0.9
    rs = np.random.RandomState(3)
    imgs = [rs.randint(0, 256, (hh, ww, 3)).astype(np.uint8) for _ in range(n)]
    frames = []
    for i in range(n):
        Image.fromarray(imgs[i]).save(root / "images" / ("frame_%03d.png" % i))
        frames.append({"file_path": "./some/dir/frame_%03d.png" % i, "transform_matrix": look(i).tolist()})
    json.dump({"camera_angle_x": ang, "h": hh, "w": ww, "frames": frames}, open(root / "transforms.json", "w"))
    test = DS.load_dataset("colmap", str(root), split="test", skip=1, near_plane=0.5, far_plane=20)
    train = DS.load_dataset("colmap", str(root), split="train", skip=4)
    assert test.index_list == [0, 8, 16] and train.skip == 1                          # :25-26, :36-41
    assert train.index_list == [i for i in range(n) if i % 8]
    assert (test.height, test.width, test.scene_name) == (hh, ww, "desk")
    assert abs(test.focal - 0.5 * ww / np.tan(0.5 * ang)) < 1e-12 and test.get_near_far_plane() == {"near": 0.5, "far": 20}
    test.load_all_data()
    assert len(test.poses) == 3 and np.array_equal(test.poses[1], look(8).astype(np.float32))
    assert np.array_equal(test.images[2], imgs[16].astype(np.float32) / np.float32(255))   # basename under images/ (:47)
    assert test.get_resized_normal_albedo(1, 0) == {}
    half = DS.load_dataset("colmap", str(root), split="val", image_scale=0.5, load_image=False)
    assert (half.height, half.width) == (2, 3) and "image" not in half[0] and len(half) == 3
    pri = DS.load_dataset("colmap", str(root), load_priors=True)                    # prior files sit beside the images (:49-60); none written here
    assert pri.prior_irradiance_mean == 0.7 and pri.prior_type == "bell"           # dataset_interface.py:42-44
    with pytest.raises(FileNotFoundError):
        pri[0]


def test_reader_mitsuba_eval(tmp_path):
    from PIL import Image
    root = tmp_path / "eval" / "monte_carlo_run"
    os.makedirs(root)
    rs = np.random.RandomState(4)
    truth = {}
    for i in range(2):
        for stem in ("rgb", "diffuse", "specular", "irradiance", "roughness", "albedo"):
            truth[stem, i] = rs.randint(0, 256, (3, 5, 3)).astype(np.uint8)
            Image.fromarray(truth[stem, i]).save(root / ("%s_%03d.png" % (stem, i)))
    ds = DS.load_dataset("mitsuba_eval", str(root))
    assert len(ds) == 2 and sorted(ds[1]) == ["albedo", "diffuse", "image", "irradiance", "roughness", "specular"]
    assert np.array_equal(ds[1]["image"], truth["rgb", 1].astype(np.float32) / np.float32(255))
    alb = truth["albedo", 0].astype(np.float32) / np.float32(255)
    assert np.array_equal(ds[0]["albedo"], np.power(alb, 1 / 2.2))                    # "monte_carlo" in the path (:52-53)


def test_reader_edit_and_insert(scene):
    torch = pytest.importorskip("torch")
    root, truth = scene
    ds = DS.load_dataset("mitsuba", str(root), split="test", editing_idx=2, load_edit_intrinsic_mask=True,
                         load_edit_normal=True, load_edit_albedo=True, load_edit_depth=True)
    assert len(ds) == 1                                                                # :46-47
    s = ds[0]
    assert np.array_equal(s["pose"][:3, 3], look(1)[:3, 3].astype(np.float32))         # frames[editing_idx - 1] (:57-58)
    assert np.array_equal(s["image"], truth[2][""].astype(np.float32) / np.float32(255))             # file <editing_idx>.png
    assert np.array_equal(s["edit_albedo"], truth[2]["_edit_albedo"].astype(np.float32) / np.float32(255))
    assert s["edit_depth"].shape == (H, W, 1) and s["edit_depth"].dtype == np.float32
    ds.load_all_data()
    ds.to_tensor("cpu")
    assert ds.poses.shape == (1, 4, 4) and torch.is_tensor(ds.poses)
    gt = ds.get_resized_normal_albedo(1, 0)
    assert sorted(gt) == ["edit_albedo", "edit_depth", "edit_intrinsic_mask", "edit_normal"]
    assert gt["edit_intrinsic_mask"].shape == (H, W, 3)
    labels, _ = O.decode_masks(gt["edit_intrinsic_mask"].reshape(-1, 3).numpy(), 2)
    assert [int(m.sum()) for m in labels] == [12, 4]
    ins = DS.load_dataset("mitsuba", str(root), split="test", editing_idx=3, object_insert=True)
    ins.load_all_data()
    g2 = ins.get_resized_normal_albedo(0, 0)
    assert sorted(g2) == ["object_insert_depth", "object_insert_mask", "object_insert_normal"]
    assert np.array_equal(g2["object_insert_depth"][..., 0], truth[3]["insert_depth"].astype(np.float32))
    half = ins.get_resized_normal_albedo(2, 0)                                          # antialiased bilinear, H//2 x W//2
    assert tuple(half["object_insert_mask"].shape) == (H // 2, W // 2, 3)


def write_experiment(tmp_path, root, extra_lines):
    """configs/<...>.txt with an include chain + a checkpoint under <basedir>/<expname>/."""
    cfg = tmp_path / "configs"
    os.makedirs(cfg, exist_ok=True)
    (cfg / "common.txt").write_text("\n".join([
        "basedir = %s" % (tmp_path / "logs"), "datadir = %s" % root, "use_viewdirs = True", "N_samples = 64",
        "N_importance = 128", "chunk = 1024", "coarse_radiance_number = 3", "image_scale = 1", "lindisp = False",
        "correct_depth_for_prefiltered_radiance_infer = True",
        "calculating_normal_type = normal_map_from_depth_gradient_epsilon", "load_depth_range_from_file"]))
    (cfg / "tiny.txt").write_text("\n".join(["include = common.txt", "expname = tiny", "gamma_correct=True", "render_factor = 1"]
                                            + extra_lines))
    exp = tmp_path / "logs" / "tiny"
    os.makedirs(exp, exist_ok=True)
    sdc, sdf = ck.synthetic_state_dict(0), ck.synthetic_state_dict(1)
    ck.save_checkpoint(str(exp / "002000.tar"), 2000, sdc, sdf)
    return str(cfg / "tiny.txt"), sdc, sdf


EDIT_LINES = ["edit_intrinsic", "editing_img_idx = 2", "num_edit_objects = 2", "edit_roughness", "edit_normal",
              "editing_target_roughness_list = [0, 0.7]", "edit_normal_by_img"]


def oracle_render_fn(sdc, sdf, lut):
    def fn(Hh, Ww, K, chunk, c2w, gt_values, **kw):
        gt = {k: np.asarray(v) for k, v in gt_values.items()}
        edit = {k: v for k, v in kw.items() if k.startswith(("edit", "insert", "num_edit", "num_insert"))}
        return O.render_decomp(Hh, Ww, K, sdc, sdf, lut, float(kw["near"]), float(kw["far"]), c2w=np.asarray(c2w),
                               n_importance=int(kw["N_importance"]), gt_values=gt, **edit)
    return fn


@pytest.mark.parametrize("extra", [[], EDIT_LINES], ids=["plain", "edit"])
def test_driver_with_oracle_renderer(scene, tmp_path, lut, extra):
    """config -> dataset -> checkpoint discovery -> per-view loop -> PNG names, with the oracle rendering."""
    pytest.importorskip("torch")
    root, _ = scene
    path, sdc, sdf = write_experiment(tmp_path, root, extra)
    args = C.load_config(path, device="cpu")
    res, out = RT.test(args, brdf_lut_path=os.path.join(os.path.dirname(__file__), "golden", "ibl_brdf_lut.png"),
                       render_fn=oracle_render_fn(sdc, sdf, lut))
    n_views = 1 if extra else N_TEST
    assert out == str(tmp_path / "logs_eval" / "tiny" / "testset_002000")             # test.py:141, :165-166
    assert res["rgb"].shape == (n_views, H, W, 3) and res["roughness"].shape == (n_views, H, W)
    assert len(os.listdir(out)) == 21 * n_views and os.path.exists(os.path.join(out, "normal_from_depth_%03d.png" % (n_views - 1)))
    if extra:
        m = np.zeros((H, W), bool)
        m[1:4, 2:6] = True
        assert np.all(res["roughness"][0][m] == 0.0) and np.all(res["roughness"][0][4:, :2] == np.float32(0.7))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], EDIT_LINES], ids=["plain", "edit"])
def test_render_test_cli_on_gpu(scene, tmp_path, lut, extra):
    """`render_test.py --config ...` end to end on the HIP path == the same driver with the oracle."""
    pytest.importorskip("torch")
    from PIL import Image
    root, _ = scene
    path, sdc, sdf = write_experiment(tmp_path, root, extra)
    lut_path = os.path.join(os.path.dirname(__file__), "golden", "ibl_brdf_lut.png")
    ref, ref_dir = RT.test(C.load_config(path, device="cpu", export_basedir=str(tmp_path / "ref")), brdf_lut_path=lut_path,
                           render_fn=oracle_render_fn(sdc, sdf, lut))
    RT.main(["--config", path, "--brdf_lut", lut_path, "--export_basedir", str(tmp_path / "hip")])
    hip_dir = str(tmp_path / "hip" / "tiny" / "testset_002000")
    assert sorted(os.listdir(hip_dir)) == sorted(os.listdir(ref_dir))
    for name in sorted(os.listdir(ref_dir)):
        a = np.asarray(Image.open(os.path.join(hip_dir, name))).astype(int)
        b = np.asarray(Image.open(os.path.join(ref_dir, name))).astype(int)
        assert a.shape == b.shape, name
        assert np.abs(a - b).max() <= (3 if "normal" in name else 1), name             # to8b truncation flips LSBs at 1e-6


def test_reader_matches_the_reference_reader(scene):
    """The same scene through the REFERENCE's MitsubaDataset / NerfDataset (tests/golden/io_dataset_expected.npz, produced by
    tests/golden/make_io_golden.py with PIL-backed stand-ins for cv2 / imageio and torch's antialiased bilinear filter for
    torchvision's Resize: pinned as far as those stand-ins are faithful; image_scale 1 only) with test.py's four load-parameter
    sets — every array bit for bit."""
    from conftest import GOLDEN
    root, _ = scene
    g = np.load(os.path.join(GOLDEN, "io_dataset_expected.npz"))
    base = dict(image_scale=1, coarse_radiance_number=0, near_plane=1.0, far_plane=20.0, load_depth_range_from_file=True, gamma_correct=True,
                load_priors=False)
    modes = {"plain": dict(skip=2), "all": dict(skip=1),
             "edit": dict(skip=1, load_edit_intrinsic_mask=True, load_edit_albedo=True, load_edit_normal=True, load_edit_depth=True, editing_idx=2),
             "insert": dict(skip=1, object_insert=True, editing_idx=3),
             "train": dict(split="train", load_priors=True, coarse_radiance_number=1)}     # train.py's reader: priors + one prefiltered target level
    seen = set()
    for mode, kw in modes.items():
        ds = DS.load_dataset("mitsuba", str(root), **dict(dict(base, split="test"), **kw))
        ds.load_all_data(num_of_workers=1)
        ds.to_tensor("cpu")
        got = {"hwf": np.array([ds.height, ds.width, ds.focal], np.float64), "near_far": np.array([ds.near, ds.far], np.float64),
               "len": np.int64(len(ds)), "K": ds.get_focal_matrix(), "poses": ds.poses.numpy(), "images": ds.images.numpy()}
        for i in range(len(ds)):
            for k, v in ds.get_resized_normal_albedo(1, i).items():
                got["gt%d__%s" % (i, k)] = v.numpy() if hasattr(v, "numpy") else np.asarray(v)
        if mode == "train":
            got["prior_irradiance_mean"] = np.float64(ds.prior_irradiance_mean)
            got["prefiltered_1"] = ds.prefiltered_images[0].numpy()
            for k, v in ds.get_info(1, np.array([0, 3, 7]), np.array([5, 2, 0])).items():
                got["info__" + k] = np.asarray(v)
        for k, v in got.items():
            want = g["%s__%s" % (mode, k)]
            assert np.asarray(v).shape == want.shape and np.array_equal(np.asarray(v), want), (mode, k)
            seen.add("%s__%s" % (mode, k))
    assert seen == set(g.files)


def test_resize_linear_known_answers():
    """dataset.resize_linear = cv2.resize(..., INTER_LINEAR) for scales other than 1 and an exact 1/2, restated from OpenCV's algorithm
    (no cv2 in this image: UNPINNED by a run; these are the algorithm's own invariants and two hand-computed cases)."""
    from ibl_nerf_amd import dataset as DS
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (12, 16, 3)).astype(np.uint8)
    const = np.full((9, 7, 3), 173, np.uint8)
    for scale in (0.25, 0.75, 1.5, 2.0, 3.0):
        out = DS.resize_linear(img, scale)
        assert out.dtype == np.uint8 and out.shape == (int(np.rint(12 * scale)), int(np.rint(16 * scale)), 3)
        assert np.array_equal(DS.resize_linear(const, scale), np.full((int(np.rint(9 * scale)), int(np.rint(7 * scale)), 3), 173, np.uint8))
        f32 = DS.resize_linear(img[..., 0].astype(np.float32), scale)                  # depth maps: float arithmetic
        assert f32.dtype == np.float32 and f32.shape == out.shape[:2]
        assert np.abs(f32 - out[..., 0]).max() <= 1.0                                  # fixed point against float: within one level
        assert f32.min() >= img[..., 0].min() and f32.max() <= img[..., 0].max()       # a convex combination of the source
    # 2x upscale of one row [0, 255]: destination centres 0.25 / 0.75 pixel apart from the source centres; the first and last tap leave
    # the image (fraction zeroed): [0, 0.25*255, 0.75*255, 255] -> fixed point: (0*1536 + 255*512)/2048 = 63.75 -> 64, 191.25 -> 191
    row = np.array([[0, 255]], np.uint8)
    assert DS.resize_linear(row, 2.0).tolist() == [[0, 64, 191, 255], [0, 64, 191, 255]]
    assert np.allclose(DS.resize_linear(row.astype(np.float32), 2.0), [[0, 63.75, 191.25, 255]] * 2)
    # 4x reduction samples between source pixels 1 and 2 of every four (centre (d + 0.5) * 4 - 0.5 = 4d + 1.5): their mean
    ramp = np.arange(16, dtype=np.float32)[None, :].repeat(4, 0)
    assert np.allclose(DS.resize_linear(ramp, 0.25), [[1.5, 5.5, 9.5, 13.5]])
    assert DS.resize_linear(ramp.astype(np.uint8), 0.25).tolist() == [[2, 6, 10, 14]]     # x.5 -> (.. + 2) >> 2 rounds half up
    with pytest.raises(ValueError):
        DS.resize_linear(row, 0.1)
