"""GPU parity tests (run with `-m gpu` on an MI355X).  Everything goes through the C-ABI
(ibl-nerf_amd/binding.py -> libiblnerf_hip.so); the checker is the oracle and the golden vectors
recorded from the reference.  Tolerances: north_star's bar is 1e-3 relative L-inf per intrinsic
channel; tighter bounds are asserted where the arithmetic allows and the reason is stated.
"""
import numpy as np
import pytest

import iblnerf_oracle as O
from conftest import FROM_GT_FLAGS, GOLDEN, RENDER_FIXTURES, color_independent, from_gt_flags, golden_aux, golden_flags, ill_conditioned, load_golden, n_samples, rel_linf

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _require_native():
    from ibl_nerf_amd import binding as B
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    B.load_library()   # raises if the HIP library was not built: never fall back


@pytest.fixture(scope="module")
def R():
    _require_native()
    from ibl_nerf_amd import renderer
    return renderer


# product schemes of the fused MLP kernel (include/iblnerf.h: mlp_precision): the default mixes the precise f16x3 kernel with the
# f16 + MX-fp6 one per query class; the others run one kernel for every query
PRECISIONS = ["f16x3_mxfp6x", "f16x3_mxfp6", "f16x3_main", "f16x3", "f16_mxfp6", "bf16x3"]
F16_MODES = ["f16x3_mxfp6x", "f16x3_mxfp6", "f16x3_main", "f16x3", "f16_mxfp6"]        # modes with the f16 range guard


def make_renderer(R, g, sdc, sdf, lut, **kw):
    kw = dict({k: v for k, v in golden_flags(g).items() if k not in FROM_GT_FLAGS}, **kw)
    if "target_normal_map_for_radiance_calculation" in kw:
        kw["normal_mode"] = kw.pop("target_normal_map_for_radiance_calculation")
    if color_independent(g):
        kw["color_independent_to_direction"] = True
    r = R.Renderer(n_samples(g), int(g["n_importance"]), **kw)
    r.load_weights(0, sdc)
    if int(g["n_importance"]) > 0:
        r.load_weights(1, sdf)
    r.load_lut(lut)
    for name, sd in golden_aux(g).items():
        if name == "depth_mlp":
            r.load_depth_mlp(sd)
        else:
            r.load_aux(name, sd)
    return r


def to_np(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


DIRECT = ["weights", "depth_map", "acc_map", "disp_map", "albedo_map", "roughness_map", "irradiance_map",
          "radiance_map", "radiance_map_1", "radiance_map_2", "radiance_map_3", "target_depth_map"]
DERIVED = ["target_normal_map", "n_dot_v_map", "specular_map", "diffuse_map", "color_map", "reflected_radiance_map",
           "prefiltered_reflected_map", "reflected_coarse_radiance_map_1", "reflected_coarse_radiance_map_2",
           "reflected_coarse_radiance_map_3"]


def test_get_rays(R):
    sv = np.load(GOLDEN + "/small_vectors.npz")
    r = R.Renderer(64, 0, max_rays_per_launch=16)
    H, W = int(sv["gr_H"]), int(sv["gr_W"])
    ro, rd = r.get_rays(H, W, sv["gr_K"], sv["gr_c2w"])
    assert np.array_equal(ro.cpu().numpy(), sv["gr_o"])
    assert np.abs(rd.cpu().numpy() - sv["gr_d"]).max() <= 2e-7
    # row tiles concatenate to the full image (the multi-GPU partition)
    a = r.get_rays(H, W, sv["gr_K"], sv["gr_c2w"], 0, 2)[1]
    b = r.get_rays(H, W, sv["gr_K"], sv["gr_c2w"], 2, 3)[1]
    assert torch.equal(torch.cat([a, b], 0), rd)
    # 800x800 view of the bench config against the oracle, bit-exact origins, 1-ulp directions
    K = np.array([[692.8203, 0, 400], [0, 692.8203, 400], [0, 0, 1]], dtype=np.float32)
    c2w = np.concatenate([np.eye(3), np.array([[0.1], [-0.2], [0.3]])], 1).astype(np.float32)
    ro, rd = r.get_rays(800, 800, K, c2w)
    oo, od = O.get_rays(800, 800, K, c2w)
    assert np.array_equal(ro.cpu().numpy(), oo) and np.abs(rd.cpu().numpy() - od).max() <= 2e-7
    # round 6: one rank's interleaved rows (iblnerf_get_rays_strided) and a probe's pixel list (iblnerf_get_rays_pixels) are the frame's rays bit for bit
    for rank, world in ((0, 8), (3, 8), (7, 8), (1, 3)):
        n_rows = len(range(rank, 800, world))
        so, sd = r.get_rays_strided(800, 800, K, c2w, rank, world, n_rows)
        assert torch.equal(sd, rd[rank::world]) and torch.equal(so, ro[rank::world]) and sd.shape == (n_rows, 800, 3)
    pix = np.sort(np.random.RandomState(3).permutation(640000)[:5000])
    po, pd = r.get_rays_pixels(800, 800, K, c2w, pix)
    idx = torch.as_tensor(pix, device=rd.device)
    assert torch.equal(pd, rd.reshape(-1, 3)[idx]) and torch.equal(po, ro.reshape(-1, 3)[idx])
    with pytest.raises(ValueError):
        r.get_rays_pixels(800, 800, K, c2w, np.array([640000]))
    with pytest.raises(R.B.IblNerfError):
        r.get_rays_strided(800, 800, K, c2w, 7, 8, 101)          # (row 7 + 100 * 8 = 807 is outside the image)


def test_sample_pdf(R):
    sv = np.load(GOLDEN + "/small_vectors.npz")
    r = R.Renderer(64, 0, max_rays_per_launch=16)
    s = r.sample_pdf(sv["sp_bins"], sv["sp_weights"], 128).cpu().numpy()
    assert np.abs(s - sv["sp_samples"]).max() <= 1e-5       # 1 ulp of a cdf entry moves a sample by ~1e-5
    s16 = r.sample_pdf(sv["sp_bins"][:, :9].copy(), sv["sp_weights"][:, :8].copy(), 16).cpu().numpy()
    assert np.abs(s16 - sv["sp16_samples"]).max() <= 1e-5
    g = np.load(GOLDEN + "/plain_g10.npz")
    s = r.sample_pdf(g["pdf_bins"], g["pdf_weights"], 128).cpu().numpy()
    assert np.abs(s - g["pdf_samples"]).max() <= 2e-5
    assert np.all(np.diff(s, axis=-1) >= -1e-6)
    with pytest.raises(ValueError):
        r.sample_pdf(sv["sp_bins"], sv["sp_weights"][:, :-1].copy(), 8)


def test_sample_pdf_spiky_rows_collapse_where_the_reference_collapses(R):
    """The regime of a checkpoint with surfaces (fixture sample_pdf_spiky, the reference's own output): empty bins sit one ulp from the
    `denom < 1e-5` replacement, so the kernel sums the row in torch.sum's own order (aten_row_sum_eps).  Every sample of every row within
    fp32 round-off of the reference — before, 1 row in 3 had a sample a fraction of the bin width (0.12) away, and a ray whose "empty" bin
    hides a thin structure came out 3e-3 off in depth (ray 1017 of fitted_edit_cfg4)."""
    g = np.load(GOLDEN + "/sample_pdf_spiky.npz")
    r = R.Renderer(64, 0, max_rays_per_launch=16)
    bins = np.ascontiguousarray(np.broadcast_to(g["bins"], (len(g["weights"]), 63)))
    s = r.sample_pdf(bins, g["weights"], 128).cpu().numpy()
    assert np.abs(s - g["samples"]).max() <= 2e-6, np.abs(s - g["samples"]).max()
    assert np.mean(s == g["samples"]) > 0.999


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("name", RENDER_FIXTURES)
def test_network_query_stagewise(R, name, lut, prec):
    """Teacher-forced MLP: the reference's own query inputs -> its recorded raw outputs.
    bf16x3 (hi/lo split, 3 MFMA products): ~2^-17 per operand, 8-12 layers deep; f16 + MX-fp6 residuals: ~2^-16."""
    g, sdc, sdf, _, _ = load_golden(name)
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=64, mlp_precision=prec)
    tol = 6e-5 if float(g["gain"]) == 1.0 else 6e-4
    passes = [("c", 0)] + ([("f", 1)] if int(g["n_importance"]) > 0 else [])
    for p, which in passes:
        raw = r.network_query(g["q_%s_main_pts" % p], g["q_%s_main_dirs" % p], which).cpu().numpy()
        assert raw.shape == g["q_%s_main_raw" % p].shape
        assert np.abs(raw - g["q_%s_main_raw" % p]).max() <= tol
        if "q_%s_eps_pts" % p in g.files:
            sig = r.network_query(g["q_%s_eps_pts" % p], None, which).cpu().numpy()
            assert np.abs(sig - g["q_%s_eps_sigma" % p]).max() <= tol
        refl = r.network_query(g["q_%s_refl_pts" % p], g["q_%s_refl_dirs" % p], which).cpu().numpy()
        assert np.abs(refl - g["q_%s_refl_raw" % p]).max() <= tol
    assert r.range_fallbacks == 0


@pytest.mark.parametrize("prec", PRECISIONS)
def test_network_query_ragged_sizes_vs_oracle(R, lut, prec):
    """Point counts that are not multiples of the 32-point wave tile / 128-point workgroup tile,
    a single point, and samples-per-ray that do not divide 32."""
    g, sdc, sdf, _, _ = load_golden("plain_g10")
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=64, mlp_precision=prec)
    rng = np.random.RandomState(0)
    for n_rays, S in ((1, 1), (3, 5), (7, 33), (2, 191), (129, 3)):
        pts = rng.uniform(-6, 6, (n_rays, S, 3)).astype(np.float32)
        dirs = rng.uniform(-1, 1, (n_rays, 3)).astype(np.float32)
        ref = O.network_query(sdc, pts, dirs)
        got = r.network_query(pts, dirs, 0).cpu().numpy()
        assert np.abs(got - ref).max() <= 6e-5, (n_rays, S)
        ref_s = O.network_query(sdc, pts, None)
        got_s = r.network_query(pts, None, 0).cpu().numpy()
        assert np.abs(got_s - ref_s).max() <= 6e-5, (n_rays, S)


@pytest.mark.parametrize("prec", PRECISIONS + ["f16_mixed"])   # + plain f16 for the fine main and reflected queries
@pytest.mark.parametrize("name", RENDER_FIXTURES)
def test_render_rays_vs_reference_golden(R, name, lut, prec):
    g, sdc, sdf, gt, edit = load_golden(name)
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=4096, mlp_precision=prec)
    res = to_np(r.render_rays(g["rays_o"], g["rays_d"], float(g["near"]), float(g["far"]), gt, **edit, **from_gt_flags(g)))
    ref_keys = sorted(k[5:] for k in g.files if k.startswith("out__"))
    assert sorted(res.keys()) == ref_keys
    wide = ill_conditioned(g)
    report = {}
    for k in ref_keys:
        assert res[k].shape == g["out__" + k].shape, k
        report[k] = rel_linf(res[k], g["out__" + k])
    for sfx in ([""] + (["0"] if int(g["n_importance"]) > 0 else [])):
        for k in DIRECT:
            # north_star bar is 1e-3; with 17-bit operands the direct channels sit well below it
            assert report[k + sfx] <= (1e-3 if wide else 2e-4), (k + sfx, report[k + sfx])
        for k in DERIVED:
            # gain 1.6 fixture: the reference's own fp32-vs-fp64 runs disagree by 1e-3..5e-1 on these
            # channels, so only a sanity bound applies there.
            # (the narrow random-init networks of the arch_* fixtures are nearly flat in density: the eps-normal's depth differences are small, and the two
            # single-product modes — 2^-16 per operand — land at 1.2e-3 on specular_map0 of the 6 x 128 one; every f16x3 mode stays below 1e-3)
            flat = name.startswith("arch_") and prec in ("f16_mxfp6", "f16_mixed")
            assert report[k + sfx] <= (2e-1 if wide else (2e-3 if flat else 1e-3)), (k + sfx, report[k + sfx])
    if int(g["n_importance"]) > 0:
        assert report["z_std"] <= 1e-4
    color = res["color_map"].astype(np.float64)
    psnr = 10 * np.log10(1.0 / max(np.mean((color - g["out__color_map"]) ** 2), 1e-30))
    assert psnr > (40 if wide else 70), psnr


def test_render_invariances_and_edge_sizes(R, lut):
    """Launch chunking must not change results (ibl_nerf_renderer.py:768-769); ragged ray counts;
    coarse_outputs=False leaves the fine maps bit-identical; N_importance=0 path."""
    g, sdc, sdf, _, _ = load_golden("plain_g10")
    ro, rd = g["rays_o"][:37], g["rays_d"][:37]
    big = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=64)
    small = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=5)
    a = to_np(big.render_rays(ro, rd, 0.5, 8.0))
    b = to_np(small.render_rays(ro, rd, 0.5, 8.0))
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    one = to_np(big.render_rays(ro[:1], rd[:1], 0.5, 8.0))
    for k in a:
        assert np.array_equal(one[k], a[k][:1]), k
    none = big.render_rays(ro[:0], rd[:0], 0.5, 8.0)            # empty batch: same keys, zero rows
    assert sorted(none.keys()) == sorted(a.keys()) and none["weights"].shape == (0, 192)
    assert big.network_query(torch.zeros(0, 64, 3), torch.zeros(0, 3)).shape == (0, 64, 18)
    assert big.sample_pdf(torch.zeros(0, 9), torch.zeros(0, 8), 16).shape == (0, 16)
    lean = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=64, coarse_outputs=False)
    c = to_np(lean.render_rays(ro, rd, 0.5, 8.0))
    assert not any(k.endswith("0") for k in c) and "z_std" in c
    for k in c:
        assert np.array_equal(c[k], a[k]), k
    # size-independent properties
    assert np.allclose(np.linalg.norm(a["target_normal_map"], axis=-1), 1, atol=1e-5)
    assert np.allclose(a["weights"].sum(-1), a["acc_map"], rtol=2e-6)
    assert np.all((a["n_dot_v_map"] >= 0) & (a["n_dot_v_map"] <= 1))
    assert np.all(a["weights"] >= 0) and a["weights"].shape == (37, 192) and a["weights0"].shape == (37, 64)


def test_render_decomp_dropin_surface(R, lut):
    """The reference-signature entry point: kwargs dict from the create_IBLNeRF mirror, `rays=`
    and `c2w=` forms, output shapes as ibl_nerf_renderer.py:810-812, edit asserts."""
    from ibl_nerf_amd import model as M
    import os
    import tempfile
    g, sdc, sdf, _, _ = load_golden("plain_g10")
    with tempfile.TemporaryDirectory() as d:      # the factory lists <basedir>/<expname> like ibl_nerf.py:350
        os.makedirs(os.path.join(d, "exp"))
        _, kw, *_ = M.create_IBLNeRF(M.default_args(basedir=d))
    kw["network_fn"].load_state_dict(sdc)
    kw["network_fine"].load_state_dict(sdf)
    kw.update(near=0.5, far=8.0, brdf_lut=torch.from_numpy(lut))
    K = np.array([[692.8203, 0, 400], [0, 692.8203, 400], [0, 0, 1]], dtype=np.float32)
    rays = torch.from_numpy(np.stack([g["rays_o"][:16], g["rays_d"][:16]], 0))
    ret = R.render_decomp(800, 800, K, chunk=1024, rays=rays, gt_values={}, approximate_radiance=True, **kw)
    assert len(ret) == 45 and ret["color_map"].shape == (16, 3) and ret["weights"].shape == (16, 192)
    assert rel_linf(ret["albedo_map"].cpu().numpy(), g["out__albedo_map"][:16]) <= 2e-4
    c2w = torch.from_numpy(np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32))
    Ks = np.array([[5.0, 0, 2], [0, 5.0, 1.5], [0, 0, 1]], dtype=np.float32)
    img = R.render_decomp(3, 4, Ks, c2w=c2w, gt_values={}, approximate_radiance=True, **kw)
    assert img["color_map"].shape == (3, 4, 3) and img["depth_map"].shape == (3, 4) and img["irradiance_map"].shape == (3, 4, 1)
    ora = O.render_decomp(3, 4, Ks, sdc, sdf, lut, 0.5, 8.0, c2w=c2w.numpy())
    assert rel_linf(img["depth_map"].cpu().numpy(), ora["depth_map"]) <= 2e-4
    with pytest.raises(AssertionError):
        R.render_decomp(800, 800, K, rays=rays, gt_values={}, approximate_radiance=True, insert_object=True,
                        num_insert_objects=0, **kw)
    with pytest.raises(TypeError):                                              # infer_depth without its network
        R.render_decomp(800, 800, K, rays=rays, gt_values={}, approximate_radiance=True, **dict(kw, infer_depth=True))
    with pytest.raises(ValueError):
        R.render_decomp(800, 800, K, rays=rays, gt_values={}, approximate_radiance=True, **dict(kw, lut_coefficient="Q"))


def test_render_decomp_static_camera_and_per_ray_planes(R, lut, tmp_path):
    """The two arguments of render_decomp that used to raise (VERDICT r3 missing-5), against the reference's own runs of them:
    c2w_staticcam (ibl_nerf_renderer.py:791-794; fixture staticcam_g10, with infer_depth: the only consumer of the other pose's view directions)
    and per-ray near / far planes as [n, 1] tensors (:802-805; fixture nearfar_g10: a z grid and a mip-level depth_0 per ray) — at the
    direct-channel tolerance; plus the reference's error behaviour for planes that do not broadcast."""
    import os
    from ibl_nerf_amd import checkpoint as ck, model as M
    g, sdc, sdf, _, _ = load_golden("staticcam_g10")
    os.makedirs(tmp_path / "exp")
    ck.save_checkpoint(str(tmp_path / "exp" / "000001.tar"), 1, sdc, sdf, aux=golden_aux(g))
    _, kw, *_ = M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), no_reload=False, infer_depth=True))
    kw.update(near=float(g["near"]), far=float(g["far"]), brdf_lut=torch.from_numpy(lut), max_rays_per_launch=64)
    H, W = int(g["H"]), int(g["W"])
    ret = to_np(R.render_decomp(H, W, g["K"], c2w=torch.from_numpy(g["c2w"]), c2w_staticcam=torch.from_numpy(g["c2w_staticcam"]), gt_values={},
                                approximate_radiance=True, **kw))
    assert sorted(ret) == sorted(k[5:] for k in g.files if k.startswith("out__")) and list(ret)[-1] == "inferred_depth_map"
    for k in DIRECT:
        for sfx in ("", "0"):
            assert ret[k + sfx].shape == g["out__" + k + sfx].shape and rel_linf(ret[k + sfx], g["out__" + k + sfx]) <= 2e-4, k + sfx
    assert rel_linf(ret["target_normal_map"], g["out__target_normal_map"]) <= 1e-3
    assert ret["inferred_depth_map"].shape == (H, W) and rel_linf(ret["inferred_depth_map"], g["out__inferred_depth_map"]) <= 1e-5
    plain = to_np(R.render_decomp(H, W, g["K"], c2w=torch.from_numpy(g["c2w_staticcam"]), gt_values={}, approximate_radiance=True, **kw))
    assert all(np.array_equal(plain[k], ret[k]) for k in ret if k != "inferred_depth_map")          # what the argument changes: that one map
    assert rel_linf(plain["inferred_depth_map"], g["out__inferred_depth_map"]) > 1e-3
    rays = torch.from_numpy(np.stack([np.zeros((5, 3), np.float32), np.ones((5, 3), np.float32)], 0))
    with pytest.raises(RuntimeError):                                                                # torch.cat of 5 rays' viewdirs with H * W static rays (:806)
        R.render_decomp(H, W, g["K"], rays=rays, c2w_staticcam=torch.from_numpy(g["c2w_staticcam"]), gt_values={}, approximate_radiance=True, **kw)

    g, sdc, sdf, _, _ = load_golden("nearfar_g10")
    kw["network_fn"].load_state_dict(sdc)
    kw["network_fine"].load_state_dict(sdf)
    kw2 = {k: v for k, v in kw.items() if k not in ("near", "far", "infer_depth", "depth_mlp")}
    K = np.array([[692.8203, 0, 400], [0, 692.8203, 400], [0, 0, 1]], dtype=np.float32)
    rays = torch.from_numpy(np.stack([g["rays_o"], g["rays_d"]], 0))
    ret = to_np(R.render_decomp(800, 800, K, rays=rays, near=torch.from_numpy(g["near"]), far=torch.from_numpy(g["far"]), gt_values={}, approximate_radiance=True, **kw2))
    assert sorted(ret) == sorted(k[5:] for k in g.files if k.startswith("out__"))
    for k in DIRECT:
        for sfx in ("", "0"):
            assert rel_linf(ret[k + sfx], g["out__" + k + sfx]) <= 2e-4, k + sfx
    for k in DERIVED:
        for sfx in ("", "0"):
            assert rel_linf(ret[k + sfx], g["out__" + k + sfx]) <= 1e-3, k + sfx
    assert rel_linf(ret["z_std"], g["out__z_std"]) <= 1e-4
    uni = to_np(R.render_decomp(800, 800, K, rays=rays, near=torch.full((96, 1), 0.5), far=8.0, gt_values={}, approximate_radiance=True, **kw2))     # a uniform plane is a scalar
    sca = to_np(R.render_decomp(800, 800, K, rays=rays, near=0.5, far=8.0, gt_values={}, approximate_radiance=True, **kw2))
    assert all(np.array_equal(uni[k], sca[k]) for k in uni)
    assert rel_linf(sca["depth_map"], g["out__depth_map"]) > 1e-2                                    # the planes matter
    with pytest.raises(RuntimeError):                                                                # a 1-D [n] plane broadcasts to [n, n] in the reference and fails
        R.render_decomp(800, 800, K, rays=rays, near=torch.from_numpy(g["near"][:, 0]), far=8.0, gt_values={}, approximate_radiance=True, **kw2)
    # chunked launches: the planes follow their rays
    r = R.renderer_for(dict(kw2, max_rays_per_launch=40))
    small = to_np(r.render_rays(g["rays_o"], g["rays_d"], g["near"], g["far"]))
    assert all(np.array_equal(small[k], ret[k].reshape(small[k].shape)) for k in small)
    # ... and stratified sampling on per-ray grids agrees with the oracle on the same draws
    t_rand, u = torch.rand((96, 64), generator=torch.Generator().manual_seed(3)), torch.rand((96, 128), generator=torch.Generator().manual_seed(4))
    got = to_np(r.render_rays(g["rays_o"], g["rays_d"], g["near"], g["far"], draws=(t_rand.cuda(), u.cuda())))
    ora = O.render_rays(sdc, sdf, g["rays_o"], g["rays_d"], g["near"], g["far"], lut, t_rand=t_rand.numpy(), u=u.numpy())
    assert rel_linf(got["depth_map"], ora["depth_map"]) <= 2e-4 and rel_linf(got["weights0"], ora["weights0"]) <= 2e-4


def test_a_smaller_architecture_through_the_model_factory(R, lut, tmp_path):
    """Round 5: netdepth / netwidth / multires / multires_views below the built 8 / 256 / 10 / 4 through the reference's call sequence — create_IBLNeRF(args) builds the
    containers in their own shapes, finds and loads the checkpoint, render_decomp uploads each network embedded in the built architecture
    (checkpoint.embed_architecture) and renders it on the same kernels, lists and routes included.  Fixture arch_6x128_g10 = the reference's own render of
    IBLNeRF(6, 128, multires 6 / 2); the parametrised fixture tests above run it (and 4 x 64, 7 x 200) in every precision mode.  What is not a member raises."""
    import os
    from ibl_nerf_amd import checkpoint as ck, model as M
    g, sdc, sdf, _, _ = load_golden("arch_6x128_g10")
    assert ck.arch_of(sdc) == (6, 128, 6, 2)
    os.makedirs(tmp_path / "exp")
    ck.save_checkpoint(str(tmp_path / "exp" / "000001.tar"), 1, sdc, sdf)
    _, kw, *_ = M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), no_reload=False, netdepth=6, netwidth=128, multires=6, multires_views=2))
    assert kw["network_fn"].arch == (6, 128, 6, 2) and ck.arch_of(kw["network_fine"].state_dict()) == (6, 128, 6, 2)
    kw.update(near=float(g["near"]), far=float(g["far"]), brdf_lut=torch.from_numpy(lut))
    K = np.array([[692.8203, 0, 400], [0, 692.8203, 400], [0, 0, 1]], dtype=np.float32)
    rays = torch.from_numpy(np.stack([g["rays_o"], g["rays_d"]], 0))
    ret = to_np(R.render_decomp(800, 800, K, rays=rays, gt_values={}, approximate_radiance=True, **kw))
    assert sorted(ret) == sorted(k[5:] for k in g.files if k.startswith("out__"))
    for k in DIRECT:
        for sfx in ("", "0"):
            assert rel_linf(ret[k + sfx], g["out__" + k + sfx]) <= 2e-4, (k + sfx, rel_linf(ret[k + sfx], g["out__" + k + sfx]))
    for k in DERIVED:
        for sfx in ("", "0"):
            assert rel_linf(ret[k + sfx], g["out__" + k + sfx]) <= 1e-3, (k + sfx, rel_linf(ret[k + sfx], g["out__" + k + sfx]))
    # the query hook on the small container: network_query_fn embeds too
    raw = M.network_query_fn(torch.from_numpy(g["q_c_main_pts"]), torch.from_numpy(g["q_c_main_dirs"]), kw["network_fn"]).cpu().numpy()
    assert np.abs(raw - g["q_c_main_raw"]).max() <= 6e-5
    for bad in (dict(netdepth=5), dict(netwidth=257), dict(netdepth=40), dict(multires=30)):
        with pytest.raises(NotImplementedError):
            M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), no_reload=True, **bad))


def test_a_larger_architecture_through_the_model_factory(R, lut, tmp_path):
    """Round 6 (VERDICT r5 missing-2): netdepth / netwidth / multires ABOVE the built 8 / 256 / 10 / 4 — IBLNeRF(D, W, ...) takes any (ibl_nerf.py:14-60) — through the
    reference's call sequence: create_IBLNeRF(args) builds the containers, loads the checkpoint, render_decomp uploads each network as it is
    (iblnerf_upload_weights_arch) and csrc/generic_mlp.hip evaluates it layer by layer in exact fp32: every sample of every query, no route, no table decision.
    Fixture arch_10x384_g10 = the reference's own render of IBLNeRF(10, 384, multires 12 / 5); arch_8x512_g10 (twice the width, edit overrides) runs with the
    parametrised fixture tests above.  The fused backward is not built for such a network: a gradient query says so."""
    import os
    from ibl_nerf_amd import checkpoint as ck, model as M
    g, sdc, sdf, _, _ = load_golden("arch_10x384_g10")
    assert ck.arch_of(sdc) == (10, 384, 12, 5) and ck.is_generic_arch(ck.arch_of(sdc))
    os.makedirs(tmp_path / "exp")
    ck.save_checkpoint(str(tmp_path / "exp" / "000001.tar"), 1, sdc, sdf)
    _, kw, *_ = M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), no_reload=False, netdepth=10, netwidth=384, multires=12, multires_views=5,
                                                N_importance=int(g["n_importance"])))
    assert kw["network_fn"].arch == (10, 384, 12, 5) and ck.arch_of(kw["network_fine"].state_dict()) == (10, 384, 12, 5)
    kw.update(near=float(g["near"]), far=float(g["far"]), brdf_lut=torch.from_numpy(lut))
    K = np.array([[692.8203, 0, 400], [0, 692.8203, 400], [0, 0, 1]], dtype=np.float32)
    rays = torch.from_numpy(np.stack([g["rays_o"], g["rays_d"]], 0))
    ret = to_np(R.render_decomp(800, 800, K, rays=rays, gt_values={}, approximate_radiance=True, **kw))
    assert sorted(ret) == sorted(k[5:] for k in g.files if k.startswith("out__"))
    for k in DIRECT:
        for sfx in ("", "0"):
            assert rel_linf(ret[k + sfx], g["out__" + k + sfx]) <= 2e-4, (k + sfx, rel_linf(ret[k + sfx], g["out__" + k + sfx]))
    for k in DERIVED:
        for sfx in ("", "0"):
            assert rel_linf(ret[k + sfx], g["out__" + k + sfx]) <= 1e-3, (k + sfx, rel_linf(ret[k + sfx], g["out__" + k + sfx]))
    r = R.renderer_for(kw)
    assert r._generic == {0: (10, 384, 12, 5), 1: (10, 384, 12, 5)} and r.route is None and r.last_selection() == (0, 0)
    # the network query on its own, against the reference's recorded rows; more rays than one 65 536-point chunk of the layer kernels
    raw = M.network_query_fn(torch.from_numpy(g["q_c_main_pts"]), torch.from_numpy(g["q_c_main_dirs"]), kw["network_fn"]).cpu().numpy()
    assert np.abs(raw - g["q_c_main_raw"]).max() <= 2e-5 * max(1.0, float(np.abs(g["q_c_main_raw"]).max()))
    big = r.render_rays(np.tile(g["rays_o"], (30, 1)), np.tile(g["rays_d"], (30, 1)), float(g["near"]), float(g["far"]))       # 1 440 rays: several chunks, no decisions
    assert torch.equal(big["depth_map"][:48], torch.from_numpy(ret["depth_map"]).cuda()) and torch.equal(big["color_map"][48:96], big["color_map"][:48])
    with pytest.raises(R.B.IblNerfError, match="not built for IBLNeRF"):
        r.density_gradient(torch.from_numpy(g["q_c_main_pts"][:4]).reshape(-1, 3))
    # the same context takes a built-shape network back onto the fused kernels
    g8, sdc8, sdf8, _, _ = load_golden("plain_g10")
    r.load_weights(0, sdc8)
    r.load_weights(1, sdf8)
    assert r._generic == {}


def test_static_camera_and_per_ray_planes_in_the_training_only_render_types(R, lut, tmp_path):
    """f-3 leftover closed in round 5: c2w_staticcam and per-ray near / far planes through is_depth_only (raw2outputs_depth, :197-198) and approximate_radiance=False —
    the two render types only a training run takes (train.py:285-297, :366-374), which used to raise.  Fixtures staticcam_g10 / nearfar_g10 carry the reference's
    own runs of both (`depthonly__out__*`, `direct__out__*`): per-ray planes give every ray its own z grid (iblnerf_coarse_z_rays); the static camera's other pose
    reaches only inferred_depth_map (:722-726, appended whatever the pass type)."""
    import os
    from ibl_nerf_amd import checkpoint as ck, model as M
    g, sdc, sdf, _, _ = load_golden("staticcam_g10")
    os.makedirs(tmp_path / "exp")
    ck.save_checkpoint(str(tmp_path / "exp" / "000001.tar"), 1, sdc, sdf, aux=golden_aux(g))
    _, kw, *_ = M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), no_reload=False, infer_depth=True))
    kw.update(near=float(g["near"]), far=float(g["far"]), brdf_lut=torch.from_numpy(lut), max_rays_per_launch=64)
    H, W = int(g["H"]), int(g["W"])
    for tag, extra in (("depthonly", dict(approximate_radiance=False, is_depth_only=True)), ("direct", dict(approximate_radiance=False))):
        ret = to_np(R.render_decomp(H, W, g["K"], c2w=torch.from_numpy(g["c2w"]), c2w_staticcam=torch.from_numpy(g["c2w_staticcam"]), gt_values={}, **extra, **kw))
        want = sorted(k[len(tag) + 7:] for k in g.files if k.startswith(tag + "__out__"))
        assert sorted(ret) == want and list(ret)[-1] == "inferred_depth_map", (tag, sorted(ret), want)
        for k in want:
            ref = g["%s__out__%s" % (tag, k)]
            assert ret[k].shape == ref.shape and rel_linf(ret[k], ref) <= (1e-5 if k == "inferred_depth_map" else 2e-4), (tag, k, rel_linf(ret[k], ref))
        plain = to_np(R.render_decomp(H, W, g["K"], c2w=torch.from_numpy(g["c2w_staticcam"]), gt_values={}, **extra, **kw))
        assert all(np.array_equal(plain[k], ret[k]) for k in ret if k != "inferred_depth_map") and rel_linf(plain["inferred_depth_map"], ret["inferred_depth_map"]) > 1e-3

    g, sdc, sdf, _, _ = load_golden("nearfar_g10")
    kw["network_fn"].load_state_dict(sdc)
    kw["network_fine"].load_state_dict(sdf)
    kw2 = {k: v for k, v in kw.items() if k not in ("near", "far", "infer_depth", "depth_mlp")}
    K = np.array([[692.8203, 0, 400], [0, 692.8203, 400], [0, 0, 1]], dtype=np.float32)
    rays = torch.from_numpy(np.stack([g["rays_o"], g["rays_d"]], 0))
    for tag, extra in (("depthonly", dict(approximate_radiance=False, is_depth_only=True)), ("direct", dict(approximate_radiance=False))):
        ret = to_np(R.render_decomp(800, 800, K, rays=rays, near=torch.from_numpy(g["near"]), far=torch.from_numpy(g["far"]), gt_values={}, **extra, **kw2))
        want = sorted(k[len(tag) + 7:] for k in g.files if k.startswith(tag + "__out__"))
        assert sorted(ret) == want, (tag, sorted(ret), want)
        for k in want:
            assert rel_linf(ret[k], g["%s__out__%s" % (tag, k)]) <= (1e-4 if k == "z_std" else 2e-4), (tag, k, rel_linf(ret[k], g["%s__out__%s" % (tag, k)]))
        sca = to_np(R.render_decomp(800, 800, K, rays=rays, near=0.5, far=8.0, gt_values={}, **extra, **kw2))
        assert rel_linf(sca["depth_map"], g[tag + "__out__depth_map"]) > 1e-2                        # the planes matter


def test_edit_roughness_by_img_follows_the_reference_chunking(R, lut):
    """edit_roughness_by_img (ibl_nerf_renderer.py:394-395): `target_roughness_map[mask_all] = gt_values["edit_roughness"][mask_all][0]` runs inside
    raw2outputs, i.e. once per `chunk` rays — every masked ray takes the FIRST masked row of ITS chunk, the one place where the reference's result
    depends on the chunk size.  Fixture edit3_g10 is the reference's run with one chunk (rendered in every precision mode by
    test_render_rays_vs_reference_golden); here the drop-in seam with chunk = 32 against the oracle applying the reference's loop, exact on the map."""
    from ibl_nerf_amd import model as M
    g, sdc, sdf, gt, edit = load_golden("edit3_g10")
    net_c, net_f = M.IBLNeRF(), M.IBLNeRF()
    net_c.load_state_dict(sdc)
    net_f.load_state_dict(sdf)
    kw = dict(network_fn=net_c, network_fine=net_f, N_samples=64, N_importance=128, perturb=False, raw_noise_std=0, lindisp=False, gamma_correct=True,
              lut_coefficient="F", epsilon=0.01, target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient_epsilon",
              correct_depth_for_prefiltered_radiance_infer=True, near=0.5, far=8.0, brdf_lut=torch.from_numpy(lut), max_rays_per_launch=64)
    K = np.array([[692.8203, 0, 400], [0, 692.8203, 400], [0, 0, 1]], dtype=np.float32)
    rays = torch.from_numpy(np.stack([g["rays_o"], g["rays_d"]], 0))
    gt_t = {k: torch.from_numpy(v) for k, v in gt.items()}
    masked = gt["edit_intrinsic_mask"][:, 0] > 0
    one = to_np(R.render_decomp(800, 800, K, chunk=96, rays=rays, gt_values=gt_t, approximate_radiance=True, **kw, **edit))
    assert np.array_equal(one["roughness_map"], g["out__roughness_map"][: len(one["roughness_map"])]) or rel_linf(one["roughness_map"], g["out__roughness_map"]) <= 2e-4
    assert np.all(one["roughness_map"][masked] == gt["edit_roughness"][masked][0, 0]) and np.all(one["roughness_map0"][masked] == gt["edit_roughness"][masked][0, 0])
    got = to_np(R.render_decomp(800, 800, K, chunk=32, rays=rays, gt_values=gt_t, approximate_radiance=True, **kw, **edit))
    ora = O.render_decomp(800, 800, K, sdc, sdf, lut, 0.5, 8.0, rays=(g["rays_o"], g["rays_d"]), chunk=32, gt_values=gt, **edit)
    assert np.array_equal(got["roughness_map"][masked], ora["roughness_map"][masked]) and len(np.unique(got["roughness_map"][masked])) == 3
    assert rel_linf(got["color_map"], ora["color_map"]) <= 1e-3 and rel_linf(got["roughness_map"], ora["roughness_map"]) <= 2e-4
    with pytest.raises(RuntimeError):       # a three-channel image: the reference's masked assignment of a [3]-vector to a scalar map fails
        R.render_decomp(800, 800, K, chunk=96, rays=rays, gt_values=dict(gt_t, edit_roughness=torch.rand(96, 3)), approximate_radiance=True, **kw, **edit)


def test_tile_sharding_matches_full_frame(R, lut):
    """Config 3's partition at a CPU-oracle-checkable size: rendering row tiles separately and
    concatenating equals rendering the frame in one call, bit for bit."""
    from ibl_nerf_amd import dist as D
    g, sdc, sdf, _, _ = load_golden("plain_g10")
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=256)
    H, W = 10, 12
    K = np.array([[9.0, 0, 6], [0, 9.0, 5], [0, 0, 1]], dtype=np.float32)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    full = D.render_frame(r, H, W, K, c2w, 0.5, 8.0)
    buf, layout = D.pack_maps({k: v.reshape(H * W, -1) for k, v in full.items()}, D.EXPORT_KEYS, H, W)
    for partition in ("contiguous", "interleaved"):
        cat = torch.empty_like(buf)
        for rank in range(3):
            rows = D.tile_row_indices(H, rank, 3, partition)
            ro, rd = r.get_rays(H, W, K, c2w)
            sl = slice(rows.start, rows.stop, rows.step)
            m = r.render_rays(ro[sl].reshape(-1, 3), rd[sl].reshape(-1, 3), 0.5, 8.0)
            cat[sl] = D.pack_maps(m, D.EXPORT_KEYS, len(rows), W)[0]
        assert torch.equal(cat, buf), partition
    ora = O.render_decomp(H, W, K, sdc, sdf, lut, 0.5, 8.0, c2w=c2w)
    assert rel_linf(full["albedo_map"].cpu().numpy(), ora["albedo_map"]) <= 2e-4
    assert rel_linf(full["target_normal_map"].cpu().numpy(), ora["target_normal_map"]) <= 1e-3


def test_full_frame_800x800_properties(R, lut):
    """BASELINE configs[1] at full size (640 000 rays, 64+128): size-independent properties of the
    whole frame, a seeded sample of pixels against the oracle, and row-tile == whole-frame equality."""
    from ibl_nerf_amd import checkpoint as ck
    sdc, sdf = ck.synthetic_state_dict(0, 1.0), ck.synthetic_state_dict(1, 1.0)
    r = R.Renderer(64, 128)
    r.load_weights(0, sdc)
    r.load_weights(1, sdf)
    r.load_lut(lut)
    H = W = 800
    f = np.float32(0.5 * W / np.tan(0.5 * np.deg2rad(60.0)))
    K = np.array([[f, 0, 400], [0, f, 400], [0, 0, 1]], dtype=np.float32)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    ro, rd = r.get_rays(H, W, K, c2w)
    m = r.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), 0.5, 8.0)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(v).all()) for v in m.values())
    n = m["target_normal_map"]
    assert float((n.norm(dim=-1) - 1).abs().max()) <= 1e-5
    assert float((m["weights"].sum(-1) - m["acc_map"]).abs().max()) <= 5e-6
    assert float(m["weights"].min()) >= 0 and float(m["n_dot_v_map"].min()) >= 0 and float(m["n_dot_v_map"].max()) <= 1
    assert float(m["roughness_map"].min()) > 0 and float(m["roughness_map"].max()) < 1
    assert float(m["z_std"].min()) > 0
    # a row tile rendered on its own is bit-identical to the same rows of the frame (multi-GPU partition)
    rows = slice(300 * W, 310 * W)
    t = r.render_rays(ro.reshape(-1, 3)[rows], rd.reshape(-1, 3)[rows], 0.5, 8.0)
    for k in ("color_map", "depth_map", "target_normal_map", "weights", "albedo_map0"):
        assert torch.equal(t[k], m[k][rows]), k
    # seeded pixel sample against the oracle
    pix = np.random.RandomState(3).permutation(H * W)[:96]
    ref = O.render_rays(sdc, sdf, ro.reshape(-1, 3)[pix].cpu().numpy(), rd.reshape(-1, 3)[pix].cpu().numpy(), 0.5, 8.0, lut)
    idx = torch.as_tensor(pix, device=n.device)
    for k, tol in (("albedo_map", 2e-4), ("roughness_map", 2e-4), ("irradiance_map", 2e-4), ("depth_map", 2e-4),
                   ("radiance_map", 2e-4), ("target_normal_map", 1e-3), ("prefiltered_reflected_map", 1e-3), ("color_map", 1e-3)):
        assert rel_linf(m[k][idx].cpu().numpy(), ref[k]) <= tol, (k, rel_linf(m[k][idx].cpu().numpy(), ref[k]))


def test_torch_module_weights_are_tracked(R, lut):
    """render_decomp accepts a torch nn.Module with the reference's parameter names (what test.py /
    train.py pass as network_fn) and re-uploads when a parameter changes in place (optimizer step)."""
    from ibl_nerf_amd import checkpoint as ck
    from torch_ref import RefShaped
    g, sdc, sdf, _, _ = load_golden("plain_g10")
    net_c, net_f = RefShaped(sdc), RefShaped(sdf)
    assert list(net_c.state_dict().keys()) == [n + s for n, _, _ in ck.SCHEMA for s in (".weight", ".bias")]
    kw = dict(network_fn=net_c, network_fine=net_f, N_samples=64, N_importance=128, perturb=False, raw_noise_std=0,
              lindisp=False, gamma_correct=True, lut_coefficient="F", epsilon=0.01,
              target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient_epsilon",
              correct_depth_for_prefiltered_radiance_infer=True, near=0.5, far=8.0, brdf_lut=torch.from_numpy(lut),
              max_rays_per_launch=64)
    K = np.eye(3, dtype=np.float32)
    rays = torch.from_numpy(np.stack([g["rays_o"][:8], g["rays_d"][:8]], 0))
    with torch.no_grad():                           # test.py:141-151 (with autograd on and trainable modules the call is a training step: test_gpu_training.py)
        a = R.render_decomp(800, 800, K, rays=rays, gt_values={}, approximate_radiance=True, **kw)
    assert rel_linf(a["albedo_map"].cpu().numpy(), g["out__albedo_map"][:8]) <= 2e-4
    with torch.no_grad():
        net_f.albedo_linear.bias.add_(0.5)          # in-place update, as an optimizer step does
        b = R.render_decomp(800, 800, K, rays=rays, gt_values={}, approximate_radiance=True, **kw)
    assert float((b["albedo_map"] - a["albedo_map"]).abs().min()) > 1e-2     # the new weights were uploaded
    assert torch.equal(b["albedo_map0"], a["albedo_map0"])                   # the coarse network did not change


def test_training_query_fn_dispatch(R, lut):
    """training_network_query_fn: no-grad queries (eps-normal, reflected) run on the fused kernel and see
    the weights an optimizer step just wrote; grad-carrying queries keep the autograd path."""
    from ibl_nerf_amd import model as M
    from torch_ref import RefShaped, torch_query
    g, sdc, _, _, _ = load_golden("plain_g10")
    net = RefShaped(sdc).cuda()
    calls = []

    def grad_query(inputs, viewdirs, network_fn):
        calls.append(inputs.shape)
        return torch_query(inputs, viewdirs, network_fn)

    q = M.training_network_query_fn(grad_query)
    pts = torch.from_numpy(g["q_c_main_pts"]).cuda()
    dirs = torch.from_numpy(g["q_c_main_dirs"]).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    for step in range(2):
        with torch.no_grad():
            fused = q(pts, dirs, net)                         # HIP
            sig = q(pts, None, net)
            ref = torch_query(pts, dirs, net)
        assert len(calls) == step and not fused.requires_grad          # the no-grad queries never reach grad_query
        assert float((fused - ref).abs().max()) <= 6e-5 and float((sig[..., 0] - ref[..., 0]).abs().max()) <= 6e-5
        out = q(pts, dirs, net)                               # autograd path
        assert len(calls) == step + 1 and out.requires_grad
        opt.zero_grad()
        out.square().mean().backward()
        opt.step()                                            # in-place update: the next fused query must see it
    assert M.network_query_fn(pts, dirs, RefShaped(sdc).cuda()).shape == (pts.shape[0], pts.shape[1], 18)


@pytest.mark.parametrize("prec", F16_MODES)
def test_f16_range_fallback(R, lut, prec):
    """An f16 mode on a checkpoint whose activations leave the f16 range (a gain of 16 per trunk layer: 16^8 through the trunk).  The kernel flags it
    (f16 + MX-fp6: running maximum of the converted values; f16x3: the overflow turns into NaN outputs, which the kernel's tail checks); the renderer
    measures every activation's magnitude on the call's points (iblnerf_layer_ranges), re-uploads the network as the SAME function with its
    activations scaled by powers of two (checkpoint.scale_activations: exact) and repeats the call on the f16 kernels — at their own precision, not
    bf16x3's 2^-17.  Where rescaling cannot help (an input beyond the range) the call still goes to the bf16x3 twin."""
    from ibl_nerf_amd import checkpoint as ck
    g, _, _, _, _ = load_golden("plain_g10")
    sd = {k: (v * np.float32(16.0) if k.endswith("weight") and k.startswith("positions_linears") else v)
          for k, v in ck.synthetic_state_dict(0).items()}          # 16^8 gain through the trunk: activations pass 65504
    fast = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision=prec)
    wide = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision="bf16x3")
    for r in (fast, wide):
        r.load_weights(0, sd)
        r.load_lut(lut)
    pts, dirs = g["q_c_main_pts"], g["q_c_main_dirs"]
    a, b = fast.network_query(pts, dirs, 0), wide.network_query(pts, dirs, 0)
    assert fast.range_rescales == 1 and fast.range_fallbacks == 0 and bool(torch.isfinite(a).all())
    ref = O.network_query(sd, pts, dirs)
    scale = np.abs(ref).max((0, 1))
    ea, eb = (np.abs(x.cpu().numpy() - ref).max((0, 1)) / scale for x in (a, b))
    tol = 2e-4 if prec in ("f16_mxfp6",) else 2e-5                 # (2^-16 operands in every query of that mode)
    assert ea.max() <= tol, (prec, ea)
    assert prec == "f16_mxfp6" or ea.max() <= 0.5 * eb.max(), (ea.max(), eb.max())      # ... which the bf16x3 twin does not reach
    t = fast._act_scale[0]
    assert all(v < 1.0 and np.log2(v) == int(np.log2(v)) for v in t.values()) and t["positions_linears.7"] <= 2.0 ** -10
    a2 = fast.network_query(pts, dirs, 0)                         # the rescaled network stays: no second event
    assert fast.range_rescales == 1 and torch.equal(a, a2)
    ra = fast.render_rays(g["rays_o"][:16], g["rays_d"][:16], 0.5, 8.0)
    assert fast.range_fallbacks == 0 and all(bool(torch.isfinite(v).all()) for k, v in ra.items() if not k.startswith("disp"))
    rb = O.render_rays(sd, None, g["rays_o"][:16], g["rays_d"][:16], 0.5, 8.0, lut, 64, 0)
    for k in ("depth_map", "albedo_map", "roughness_map", "radiance_map", "weights"):
        assert rel_linf(ra[k].cpu().numpy(), rb[k]) <= 2e-4, k
    if prec.startswith("f16x3"):                                  # (the modes that keep the backward's stream)
        with pytest.raises(FloatingPointError):                   # gradients of a rescaled network are those of other parameters: refused
            fast.trunk_backward(pts.reshape(-1, 3), np.ones(pts.shape[0] * pts.shape[1], np.float32), 0)
    fast.load_weights(0, sd)                                      # the caller's own upload forgets the rescaling
    assert fast._act_scale == {}
    # an input beyond the range: nothing to rescale, the bf16x3 twin answers
    far_pts = pts.copy()
    far_pts[0, 0, 0] = np.float32(7.0e4)
    fast2 = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision=prec)
    fast2.load_weights(0, ck.synthetic_state_dict(0))
    wide.load_weights(0, ck.synthetic_state_dict(0))
    c, d = fast2.network_query(far_pts, dirs, 0), wide.network_query(far_pts, dirs, 0)
    assert fast2.range_fallbacks == 1 and fast2.range_rescales == 0 and torch.equal(c, d)
    assert not wide.out_of_range()


def test_a_checkpoint_beyond_the_f16_range_renders_at_full_precision(R, lut):
    """the range fallback must be parity-grade.  The fitted checkpoint reparametrised so that its later trunk activations reach 1e6
    (checkpoint.scale_activations with factors 2^14: the SAME function bit for bit in fp32, as the oracle confirms) on 4 096 rays of the reference's own
    launch-scale render: a default-constructed renderer hits the range event on its first launch, rescales by measurement and holds the launch-scale
    rules (direct maps <= 5e-4 or 8x the ray's own reference sensitivity, ...) that the plain checkpoint holds — no bf16x3 launch anywhere."""
    import test_gpu_launch_scale as LS
    from ibl_nerf_amd import checkpoint as ck
    g, sdc, sdf, gt, edit = load_golden("fitted_posed4k")
    up = {k: 2.0 ** 14 for k in ck.ACTIVATIONS[3:8]}
    up.update({"feature_linear": 2.0 ** 13, "views_linears.0": 2.0 ** 15, "albedo_feature_linear": 2.0 ** 12})
    big_c, big_f = ck.scale_activations(sdc, up), ck.scale_activations(sdf, up)
    p = g["rays_o"][:8, None, :] + g["rays_d"][:8, None, :] * np.linspace(0.5, 8, 64, dtype=np.float32)[None, :, None]
    assert np.array_equal(O.network_query(big_f, p, g["rays_d"][:8]), O.network_query(sdf, p, g["rays_d"][:8]))        # the same function
    r = make_renderer(R, g, big_c, big_f, lut, max_rays_per_launch=16384)
    res = to_np(r.render_rays(g["rays_o"], g["rays_d"], 0.5, 8.0, gt, **edit))
    assert r.range_rescales >= 1 and r.range_fallbacks == 0 and r._wide is None
    assert r.policy["decision"] == LS.DECISION["fitted_posed4k"]
    LS.check_against_fixture(res, g, rules=LS.rules_for("fitted_posed4k"), name="fitted_posed4k", decision=r.policy["decision"])
    assert max(r._act_scale[1].values()) <= 2.0 ** -5 and set(r._act_scale[1]) >= set(ck.ACTIVATIONS[3:8])


@pytest.mark.parametrize("prec", F16_MODES)
def test_f16_mode_with_out_of_range_weights(R, lut, prec):
    """A weight beyond the f16 range: an f16 mode runs that network on the bf16x3 kernel from the start
    (no device flag, no second render), with the same result as an explicit bf16x3 context."""
    from ibl_nerf_amd import checkpoint as ck
    g, _, _, _, _ = load_golden("plain_g10")
    sd = dict(ck.synthetic_state_dict(0))
    w = sd["positions_linears.0.weight"].copy()
    w[3, 5] = np.float32(1.0e5)
    sd["positions_linears.0.weight"] = w
    fast = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision=prec)
    wide = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision="bf16x3")
    for r in (fast, wide):
        r.load_weights(0, sd)
    a = fast.network_query(g["q_c_main_pts"], g["q_c_main_dirs"], 0)
    b = wide.network_query(g["q_c_main_pts"], g["q_c_main_dirs"], 0)
    assert fast.range_fallbacks == 0 and torch.equal(a, b)


@pytest.mark.parametrize("prec", PRECISIONS)
def test_device_side_weight_upload_is_bit_identical(R, lut, prec):
    """Weights packed on the device (iblnerf_upload_weights_device) == weights packed on the host, for both kernels."""
    from ibl_nerf_amd import checkpoint as ck
    g, sdc, _, _, _ = load_golden("plain_g16")                   # wide-range weights
    host = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision=prec)
    dev = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision=prec)
    host.load_weights(0, sdc)
    dev.load_weights(0, {k: torch.from_numpy(v).cuda() for k, v in sdc.items()})          # CUDA tensors -> device packer
    pts, dirs = g["q_c_main_pts"], g["q_c_main_dirs"]
    assert torch.equal(host.network_query(pts, dirs, 0), dev.network_query(pts, dirs, 0))
    assert torch.equal(host.network_query(pts, None, 0), dev.network_query(pts, None, 0))
    dev.load_weights(0, torch.from_numpy(ck.state_dict_to_blob(sdc)).cuda())              # flat device blob
    assert torch.equal(host.network_query(pts, dirs, 0), dev.network_query(pts, dirs, 0))
    with pytest.raises(ValueError):
        dev.load_weights(0, {k: torch.from_numpy(v).cuda() for k, v in list(sdc.items())[:-1]})
    if prec != "bf16x3":                                         # an out-of-range weight is reported through the range flag
        bad = {k: torch.from_numpy(v).cuda() for k, v in sdc.items()}
        bad["positions_linears.3.weight"][7, 7] = 1.0e5
        dev.load_weights(0, bad)
        wide = R.Renderer(64, 0, max_rays_per_launch=64, mlp_precision="bf16x3")
        wide.load_weights(0, {k: v.cpu().numpy() for k, v in bad.items()})
        assert torch.equal(dev.network_query(pts, dirs, 0), wide.network_query(pts, dirs, 0)) and dev.range_fallbacks == 1


def test_render_decomp_ground_truth_normal_mode(R, lut):
    """The drop-in seam with target_normal_map_for_radiance_calculation="ground_truth" (the parser's default value):
    normals come from gt_values["normal"], no offset queries are launched, results match the reference golden."""
    from ibl_nerf_amd import model as M
    g, sdc, sdf, gt, edit = load_golden("gtnormal_g10")
    net_c, net_f = M.IBLNeRF(), M.IBLNeRF()
    net_c.load_state_dict(sdc)
    net_f.load_state_dict(sdf)
    kw = dict(network_fn=net_c, network_fine=net_f, N_samples=64, N_importance=128, perturb=False, raw_noise_std=0, lindisp=False,
              gamma_correct=True, lut_coefficient="F", epsilon=0.01, target_normal_map_for_radiance_calculation="ground_truth",
              correct_depth_for_prefiltered_radiance_infer=True, near=0.5, far=8.0, brdf_lut=torch.from_numpy(lut), max_rays_per_launch=64)
    rays = torch.from_numpy(np.stack([g["rays_o"], g["rays_d"]], 0))
    r = R.renderer_for(kw)
    r.set_profiling(True)
    ret = R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values={k: torch.from_numpy(v) for k, v in gt.items()},
                          approximate_radiance=True, **kw, **edit)
    n_launch = r.last_mlp_time()[1]
    r.set_profiling(False)
    # 2 launches x ((coarse, fine) x (main + reflected) + the coarse pass's density on the 15-slot form): no offset queries (and, in launches of 64 rays, none of the
    # per-network decisions behind the list refinement: they wait for a launch of >= 1 024 rays, api.cpp SELECT_MIN_RAYS — every sample is evaluated)
    assert n_launch == 2 * (2 * 2 + 1)
    assert rel_linf(ret["target_normal_map"].cpu().numpy(), g["out__target_normal_map"]) <= 1e-6
    assert rel_linf(ret["color_map"].cpu().numpy(), g["out__color_map"]) <= 2e-4
    with pytest.raises(KeyError):
        R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values={}, approximate_radiance=True, **kw)


def test_render_decomp_from_gt_flags(R, lut):
    """calculate_{albedo,roughness,irradiance}_from_gt + depth_map_from_ground_truth through the drop-in seam
    (ibl_nerf_renderer.py:251-252, :320-330): the gt rows are returned as the target maps (bit-exact where no edit
    touches them), irradiance_map turns RGB, and depth_map / disp stay the network's because the edited target depth no
    longer aliases them."""
    from ibl_nerf_amd import model as M
    g, sdc, sdf, gt, edit = load_golden("fromgt_g10")
    net_c, net_f = M.IBLNeRF(), M.IBLNeRF()
    net_c.load_state_dict(sdc)
    net_f.load_state_dict(sdf)
    kw = dict(network_fn=net_c, network_fine=net_f, N_samples=64, N_importance=128, perturb=False, raw_noise_std=0, lindisp=False,
              gamma_correct=True, lut_coefficient="F", epsilon=0.01,
              target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient_epsilon",
              correct_depth_for_prefiltered_radiance_infer=True, near=0.5, far=8.0, brdf_lut=torch.from_numpy(lut), max_rays_per_launch=40)
    rays = torch.from_numpy(np.stack([g["rays_o"], g["rays_d"]], 0))
    gtt = {k: torch.from_numpy(v) for k, v in gt.items()}
    ret = to_np(R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values=gtt, approximate_radiance=True,
                                **kw, **edit, **from_gt_flags(g)))
    assert ret["irradiance_map"].shape == (96, 3) and ret["irradiance_map0"].shape == (96, 3)
    for k in ("color_map", "diffuse_map", "specular_map", "irradiance_map", "albedo_map", "prefiltered_reflected_map"):
        assert rel_linf(ret[k], g["out__" + k]) <= 2e-4, k
    untouched = gt["edit_intrinsic_mask"][:, 0] == 0
    assert np.array_equal(ret["roughness_map"][untouched], gt["roughness"][untouched, 0])
    assert np.array_equal(ret["target_depth_map"][untouched], gt["depth"][untouched, 0])
    assert np.array_equal(ret["target_depth_map"][~untouched], gt["edit_depth"][~untouched, 0])
    plain = to_np(R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values=gtt, approximate_radiance=True, **kw))
    assert np.array_equal(ret["depth_map"], plain["depth_map"]) and np.array_equal(ret["disp_map"], plain["disp_map"])
    assert plain["irradiance_map"].shape == (96, 1)
    with pytest.raises(KeyError):                       # the flag needs its gt_values row, as the reference's indexing does
        R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values={}, approximate_radiance=True, **kw,
                        calculate_albedo_from_gt=True)


@pytest.mark.parametrize("prec", PRECISIONS + ["f16_mixed"])
def test_repeated_launches_are_bit_identical(R, lut, prec):
    """Race detector for the hand-counted LDS-DMA pipeline (ring slots are reused every third chunk, guarded only by
    vmcnt arithmetic and one barrier per chunk): the same 1.6 M points through all 256 persistent workgroups five times
    must give the same bits every time, for every kernel variant."""
    from ibl_nerf_amd import checkpoint as ck
    r = R.Renderer(64, 128, max_rays_per_launch=8192, mlp_precision=prec)
    r.load_weights(0, ck.synthetic_state_dict(0))
    r.load_weights(1, ck.synthetic_state_dict(1))
    r.load_lut(lut)
    gen = torch.Generator(device="cuda").manual_seed(0)
    pts = torch.rand((8192, 192, 3), device="cuda", generator=gen) * 8 - 4
    dirs = torch.rand((8192, 3), device="cuda", generator=gen) * 2 - 1
    full0, trunk0 = r.network_query(pts, dirs, 1), r.network_query(pts, None, 0)
    for _ in range(4):
        assert torch.equal(r.network_query(pts, dirs, 1), full0) and torch.equal(r.network_query(pts, None, 0), trunk0)
    ro = torch.zeros((8192, 3), device="cuda")
    a = r.render_rays(ro, dirs - torch.tensor([0.0, 0.0, 2.0], device="cuda"), 0.5, 8.0)
    b = r.render_rays(ro, dirs - torch.tensor([0.0, 0.0, 2.0], device="cuda"), 0.5, 8.0)
    assert all(torch.equal(a[k], b[k]) for k in a) and bool(torch.isfinite(a["color_map"]).all())


def test_create_iblnerf_color_independent_drop_in(R, lut, tmp_path):
    """create_IBLNeRF(args.color_independent_to_direction=True) -> render_decomp picks the colour-independent kernels."""
    import os
    from ibl_nerf_amd import model as M
    g, sdc, sdf, _, _ = load_golden("colorindep_g10")
    os.makedirs(tmp_path / "exp")
    _, kw, *_ = M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), color_independent_to_direction=True))
    assert kw["network_fn"].is_color_independent_to_direction
    kw["network_fn"].load_state_dict(sdc)
    kw["network_fine"].load_state_dict(sdf)
    kw.update(near=0.5, far=8.0, brdf_lut=torch.from_numpy(lut), max_rays_per_launch=64)
    rays = torch.from_numpy(np.stack([g["rays_o"][:16], g["rays_d"][:16]], 0))
    ret = R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values={}, approximate_radiance=True, **kw)
    for k in ("radiance_map", "radiance_map_3", "albedo_map", "color_map"):
        assert rel_linf(ret[k].cpu().numpy(), g["out__" + k][:16]) <= 2e-4, k


def test_create_iblnerf_auxiliary_networks_drop_in(R, lut, tmp_path):
    """infer_{albedo,roughness,irradiance}_separate: create_IBLNeRF builds the PositionMLP containers and fills them from the
    checkpoint's 'albedo_mlp' / 'roughness_mlp' / 'irradiance_mlp' entries (ibl_nerf.py:312-326, :369-374); render_decomp
    evaluates them with the trunk kernel and composites their samples (ibl_nerf_renderer.py:291-303).  Dropping a network
    from the kwargs returns that map to the main network's head."""
    import os
    from ibl_nerf_amd import checkpoint as ck, model as M
    g, sdc, sdf, _, _ = load_golden("auxmlp_lin_g10")
    aux = golden_aux(g)
    os.makedirs(tmp_path / "exp")
    ck.save_checkpoint(str(tmp_path / "exp" / "000100.tar"), 100, sdc, sdf, aux=aux)
    args = M.default_args(basedir=str(tmp_path), no_reload=False, infer_albedo_separate=True, infer_roughness_separate=True,
                          infer_irradiance_separate=True, use_radiance_linear=True)
    _, kw, start, *_ = M.create_IBLNeRF(args)
    assert start == 100 and kw["albedo_mlp"].out_ch == 3 and kw["irradiance_mlp"].out_ch == 1
    assert np.array_equal(kw["roughness_mlp"].state_dict()["out_linears.weight"], aux["roughness_mlp"]["out_linears.weight"])
    kw.update(near=0.5, far=8.0, brdf_lut=torch.from_numpy(lut), max_rays_per_launch=40)
    n = g["rays_o"].shape[0]
    rays = torch.from_numpy(np.stack([g["rays_o"], g["rays_d"]], 0))
    ret = to_np(R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values={}, approximate_radiance=True, **kw))
    for k in ("albedo_map", "roughness_map", "irradiance_map", "albedo_map0", "irradiance_map0"):
        assert rel_linf(ret[k], g["out__" + k]) <= 2e-4, k
    for k in ("diffuse_map", "specular_map", "color_map"):        # HDR radiance fixture: loose bound on the derived maps
        assert rel_linf(ret[k], g["out__" + k]) <= 5e-2, k
    kw["albedo_mlp"] = None
    ret2 = to_np(R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values={}, approximate_radiance=True, **kw))
    ref = O.render_rays(sdc, sdf, g["rays_o"][:8], g["rays_d"][:8], 0.5, 8.0, lut, flags=dict(use_radiance_linear=True),
                        aux={k: v for k, v in aux.items() if k != "albedo_mlp"})
    assert rel_linf(ret2["albedo_map"][:8], ref["albedo_map"]) <= 2e-4 and rel_linf(ret2["roughness_map"][:8], ref["roughness_map"]) <= 2e-4
    assert rel_linf(ret2["albedo_map"], g["out__albedo_map"]) > 1e-2 and n == 48


def test_infer_depth_drop_in(R, lut, tmp_path):
    """infer_depth: create_IBLNeRF builds the depth_mlp (a PositionDirectionMLP, ibl_nerf.py:293-297; checkpoint entry 'depth_mlp'),
    render_decomp appends inferred_depth_map = relu(depth_mlp(rays_o, viewdirs)[..., 0]) (ibl_nerf_renderer.py:722-726); the
    network's raw query against the reference's recorded one (fp32 kernel: 1e-5)."""
    import os
    from ibl_nerf_amd import checkpoint as ck, model as M
    g, sdc, sdf, _, _ = load_golden("inferdepth_g10")
    aux = golden_aux(g)
    os.makedirs(tmp_path / "exp")
    ck.save_checkpoint(str(tmp_path / "exp" / "000002.tar"), 2, sdc, sdf, aux=aux)
    _, kw, *_ = M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), no_reload=False, infer_depth=True, infer_visibility=True))
    assert kw["infer_depth"] is True and kw["depth_mlp"].out_ch == 1 and kw["visibility_mlp"] is not None
    assert np.array_equal(kw["depth_mlp"].state_dict()["final_linear.weight"], aux["depth_mlp"]["final_linear.weight"])
    kw.update(near=0.5, far=8.0, brdf_lut=torch.from_numpy(lut), max_rays_per_launch=40)
    rays = torch.from_numpy(np.stack([g["rays_o"], g["rays_d"]], 0))
    ret = to_np(R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values={}, approximate_radiance=True, **kw))
    assert list(ret)[-1] == "inferred_depth_map" and ret["inferred_depth_map"].shape == (64,)
    assert np.abs(ret["inferred_depth_map"] - g["out__inferred_depth_map"]).max() <= 1e-5
    assert (ret["inferred_depth_map"] == 0).any() and (ret["inferred_depth_map"] > 0).any()
    assert rel_linf(ret["color_map"], g["out__color_map"]) <= 1e-3
    r = R.renderer_for(kw)
    raw = r.posdir_query(g["q_depth_pts"], g["q_depth_dirs"]).cpu().numpy()
    assert raw.shape == g["q_depth_raw"].shape and np.abs(raw - g["q_depth_raw"]).max() <= 1e-5
    for n in (1, 15, 16, 17, 33):                                               # ragged against the 16-ray workgroup tile
        got = r.posdir_query(g["q_depth_pts"][:n], g["q_depth_dirs"][:n]).cpu().numpy()
        assert np.array_equal(got, raw[:n])
    kw["infer_depth"] = False                                                   # without the flag the map is gone
    assert "inferred_depth_map" not in R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values={}, approximate_radiance=True, **kw)
    from ibl_nerf_amd import binding as B
    lib = B.load_library()
    blob = ck.posdir_blob(aux["depth_mlp"])
    assert lib.iblnerf_upload_posdir_mlp(r.ctx, blob.ctypes.data, blob.size - 1, 1) == -1 and b"floats" in lib.iblnerf_last_error(r.ctx)


def test_infer_normal_drop_in(R, lut, tmp_path):
    """infer_normal: create_IBLNeRF builds the normal_mlp (ibl_nerf.py:307-310, checkpoint entry 'normal_mlp'), render_decomp
    returns its composited output as inferred_normal_map (ibl_nerf_renderer.py:267-276) and, with
    target_normal_map_for_radiance_calculation="inferred_normal_map", shades with it without any offset query."""
    import os
    from ibl_nerf_amd import checkpoint as ck, model as M
    g, sdc, sdf, _, _ = load_golden("infernormal_target_g10")
    os.makedirs(tmp_path / "exp")
    ck.save_checkpoint(str(tmp_path / "exp" / "000001.tar"), 1, sdc, sdf, aux=golden_aux(g))
    _, kw, *_ = M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), no_reload=False, infer_normal=True,
                                                 calculating_normal_type="inferred_normal_map"))
    assert kw["infer_normal"] is True and kw["target_normal_map_for_radiance_calculation"] == "inferred_normal_map"
    kw.update(near=0.5, far=8.0, brdf_lut=torch.from_numpy(lut), max_rays_per_launch=32)
    rays = torch.from_numpy(np.stack([g["rays_o"], g["rays_d"]], 0))
    r = R.renderer_for(kw)
    r.set_profiling(True)
    ret = to_np(R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values={}, approximate_radiance=True, **kw))
    n_launch = r.last_mlp_time()[1]
    r.set_profiling(False)
    # 2 launches x ((coarse, fine) x (main + 3 normal channels + reflected) + the coarse pass's density on the 15-slot form)
    assert n_launch == 2 * (2 * (1 + 3 + 1) + 1)
    assert list(ret).index("inferred_normal_map") == list(ret).index("target_normal_map") - 1      # reference's key order
    assert np.array_equal(ret["inferred_normal_map"], ret["target_normal_map"])
    for k in ("inferred_normal_map", "inferred_normal_map0", "n_dot_v_map", "color_map"):
        assert rel_linf(ret[k], g["out__" + k]) <= 2e-4, k
    kw["infer_normal"] = False                          # the mode without its network: the reference dies on a None normal
    with pytest.raises(Exception):
        R.render_decomp(800, 800, np.eye(3, dtype=np.float32), rays=rays, gt_values={}, approximate_radiance=True, **kw)


@pytest.mark.parametrize("prec", PRECISIONS + ["f16_mixed"])
def test_degenerate_density_rays_match_the_oracle(R, lut, prec):
    """Rays the reference's arithmetic degenerates on: an empty volume (every sigma <= 0: weights 0, acc 0, depth/acc = 0/0, so
    disp_map is NaN through torch.max, ibl_nerf_renderer.py:258; flat fine-sample pdf) and a wall at the first sample (sigma
    huge: all weight on sample 0).  Same NaN pattern and the same finite values as the oracle."""
    from ibl_nerf_amd import checkpoint as ck
    g, _, _, _, _ = load_golden("plain_g10")
    ro, rd = g["rays_o"][:24], g["rays_d"][:24]
    for bias in (-60.0, 400.0):
        sdc, sdf = ck.synthetic_state_dict(21, 1.0, sigma_bias=bias), ck.synthetic_state_dict(22, 1.0, sigma_bias=bias)
        r = R.Renderer(64, 128, max_rays_per_launch=16, mlp_precision=prec)
        r.load_weights(0, sdc)
        r.load_weights(1, sdf)
        r.load_lut(lut)
        got = to_np(r.render_rays(ro, rd, 0.5, 8.0))
        ref = O.render_rays(sdc, sdf, ro, rd, 0.5, 8.0, lut)
        for k in ref:
            assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])), (bias, k)
            ok = ~np.isnan(ref[k])
            if k in ("disp_map", "disp_map0"):
                ok &= np.abs(ref[k]) < 1e9                      # 1 / max(1e-10, .) of a vanishing depth: compare where it is a depth
            assert np.abs(got[k][ok] - ref[k][ok]).max(initial=0.0) <= 2e-4 * max(1.0, np.abs(ref[k][ok]).max(initial=0.0)), (bias, k)
        if bias < 0:
            assert np.isnan(ref["disp_map"]).all() and not ref["acc_map"].any()
        else:
            assert np.allclose(ref["weights"][:, 0], 1.0) and np.allclose(ref["depth_map"], 0.5)


def test_c_abi_rejects_misuse(R, lut):
    """Error behaviour of the boundary: every misuse comes back as a negative status with a message, never a crash."""
    import ctypes as C
    from ibl_nerf_amd import binding as B, checkpoint as ck
    lib = B.load_library()
    o = B.default_options()
    o.normal_mode = 7
    ctx = C.c_void_p()
    assert lib.iblnerf_create(C.byref(o), C.byref(ctx)) == -1 and b"normal_mode" in lib.iblnerf_last_error(None)
    r = R.Renderer(64, 128, max_rays_per_launch=8, normal_mode="inferred_normal_map")
    blob = ck.state_dict_to_blob(ck.synthetic_state_dict(0))
    r.load_weights(0, blob)
    r.load_weights(1, blob)
    r.load_lut(lut)
    with pytest.raises(B.IblNerfError, match="normal_mlp"):                    # the mode without its network
        r.render_rays(np.zeros((2, 3), np.float32), np.array([[0, 0, -1.0]] * 2, np.float32), 0.5, 8.0)
    assert lib.iblnerf_upload_aux_weights(r.ctx, 9, 0, blob.ctypes.data, blob.size) == -1
    assert lib.iblnerf_upload_aux_weights(r.ctx, 1, 1, blob.ctypes.data, blob.size) == -1      # roughness_mlp has one channel
    assert lib.iblnerf_upload_aux_weights(r.ctx, 0, 0, blob.ctypes.data, blob.size - 1) == -1  # wrong blob length
    assert lib.iblnerf_clear_aux(r.ctx, -1) == -1
    assert lib.iblnerf_upload_weights(r.ctx, 2, blob.ctypes.data, blob.size) == -1
    fresh = R.Renderer(64, 0, max_rays_per_launch=8)
    with pytest.raises(B.IblNerfError):                                         # weights / LUT not uploaded: IBLNERF_ERR_STATE
        fresh.render_rays(np.zeros((1, 3), np.float32), np.array([[0, 0, -1.0]], np.float32), 0.5, 8.0)
    ov = B.Overrides()
    ov.mode = 3
    outs = B.Outputs()
    z = torch.zeros((1, 3), device="cuda")
    r2 = R.Renderer(64, 0, max_rays_per_launch=8)
    r2.load_weights(0, blob)
    r2.load_lut(lut)
    assert lib.iblnerf_render_rays(r2.ctx, None, z.data_ptr(), z.data_ptr(), 1, 0.5, 8.0, C.byref(ov), C.byref(outs)) == -1
    assert b"mode" in lib.iblnerf_last_error(r2.ctx)


@pytest.mark.parametrize("seed,gain", [(31, 1.0), (32, 1.25)])
def test_unseen_checkpoints_vs_oracle(R, lut, seed, gain):
    """Checkpoints and rays that no fixture holds (guards against tuning to the committed vectors): a posed camera's rays
    through seeded networks, default kernel, every map against the oracle at the fixture tolerances."""
    from ibl_nerf_amd import checkpoint as ck
    sdc, sdf = ck.synthetic_state_dict(2 * seed, gain), ck.synthetic_state_dict(2 * seed + 1, gain)
    rs = np.random.RandomState(seed)
    K = np.array([[692.82, 0, 400], [0, 692.82, 400], [0, 0, 1]], dtype=np.float32)
    q, _ = np.linalg.qr(np.eye(3) + 0.2 * rs.randn(3, 3))
    c2w = np.concatenate([q * np.sign(np.linalg.det(q)), rs.uniform(-0.3, 0.3, (3, 1))], 1).astype(np.float32)
    ro, rd = O.get_rays(800, 800, K, c2w)
    pix = rs.choice(640000, 40, replace=False)
    ro, rd = ro.reshape(-1, 3)[pix], rd.reshape(-1, 3)[pix]
    r = R.Renderer(64, 128, max_rays_per_launch=16)
    r.load_weights(0, sdc)
    r.load_weights(1, sdf)
    r.load_lut(lut)
    got = to_np(r.render_rays(ro, rd, 0.5, 8.0))
    ref = O.render_rays(sdc, sdf, ro, rd, 0.5, 8.0, lut)
    assert sorted(got) == sorted(ref) and r.range_fallbacks == 0
    for sfx in ("", "0"):
        for k in DIRECT:
            assert rel_linf(got[k + sfx], ref[k + sfx]) <= 2e-4, (k + sfx, rel_linf(got[k + sfx], ref[k + sfx]))
        for k in DERIVED:
            assert rel_linf(got[k + sfx], ref[k + sfx]) <= (1e-3 if gain == 1.0 else 5e-2), (k + sfx, rel_linf(got[k + sfx], ref[k + sfx]))


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the driver's keys (one step of the real workload, no CPU baseline leg)."""
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "rays/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.05 < rf["frac"] < 0.67
    assert abs(d["value"] - 640000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d                       # that leg (and its PSNR check against the oracle) was switched off
    # round 6: frac is the DRIVER-TIMED fraction — value x algorithmic FLOPs per ray / peak, nothing else in the denominator; the MLP-only figure and the executed MACs beside it
    assert abs(rf["frac"] - d["value"] * rf["flop_per_ray"] / 1e12 / rf["peak"]) < 1e-9 and rf["flop_per_ray"] == 256 * 1591552.0 + 1024 * 982528.0 + 128 * 1458944.0
    assert rf["frac_mlp_only"] >= rf["frac"] and 0.0 < rf["executed"]["frac_whole_step"] < rf["frac"]
    assert d["config"]["decision_scope"].startswith("per frame") and d["config"]["decision_ms_per_frame"] > 0 and d["config"]["policy"]["decision"] in ("fast", "tiered", "safe")
    worst = d["value_worst_checkpoint"]
    assert worst["checkpoint"] in d["value_by_checkpoint"] and worst["value"] == min(v for k, v in d["value_by_checkpoint"].items() if k.startswith("fitted"))
    assert 0.5 < worst["ratio_to_best"] <= 1.0


def test_training_bench_line_contract():
    """bench.py --train prints ONE JSON line of the same shape for the TRAINING step (render + losses + backward + Adam): value = rays of the headline step / its time,
    a roofline object on forward + main-query-backward FLOPs, the reference's own CPU step (recorded in the build container) as cpu_baseline, the loss going down."""
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--train", "--train-rays", "512,2048", "--train-headline", "2048", "--steps", "8", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "rays/s" and "training" in d["metric"] and "workload" in d["config"] and d["config"]["rays_per_step"] == 2048
    assert abs(d["value"] - 2048 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["flop_per_ray"] == rf["flop_forward_per_ray"] + rf["flop_backward_per_ray"] and rf["flop_backward_per_ray"] == 2.0 * 256 * 1591552.0
    assert abs(rf["frac"] - d["value"] * rf["flop_per_ray"] / 1e12 / rf["peak"]) < 1e-9 and 0.01 < rf["frac"] < 0.5
    assert set(d["by_rays"]) == {"512", "2048"}
    for v in d["by_rays"].values():
        assert v["skipped_steps"] == 0 and v["range_fallbacks"] == 0 and v["loss_last"] < v["loss_first"], v
    ref = d["by_rays"]["512"]["reference_in_build_container"]
    assert ref["kind"] == "reference" and ref["cores"] == 8 and ref["value"] > 0 and d["by_rays"]["512"]["rays_per_s"] > 100 * ref["value"]
