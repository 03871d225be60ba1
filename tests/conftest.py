import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import _pkg  # noqa: E402

_pkg.load()  # registers `ibl_nerf_amd`

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_lut_rgb():
    """tests/golden/ibl_brdf_lut.png -> float32 [3,512,512] exactly as test.py:79-87 builds it."""
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(GOLDEN, "ibl_brdf_lut.png")).convert("RGB"), dtype=np.float32)
    return np.ascontiguousarray((img / np.float32(255.0)).transpose(2, 0, 1))


@pytest.fixture(scope="session")
def lut():
    return load_lut_rgb()


def load_golden(name):
    """Returns (npz, coarse_sd, fine_sd, gt dict, edit dict) for a render fixture."""
    from ibl_nerf_amd import checkpoint as ck
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if "ckpt" in g.files and str(g["ckpt"]) in ("fitted", "fitted2", "fitted3"):       # the surface-bearing checkpoints (tests/golden/fit_checkpoint.py, scene 1 | 2)
        f = np.load(os.path.join(GOLDEN, str(g["ckpt"]) + "_ckpt.npz"))
        sdc, sdf = ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])
    elif "arch" in g.files:      # a smaller IBLNeRF (netdepth, netwidth, multires, multires_views): its own shapes, as the reference built it
        arch = tuple(int(v) for v in g["arch"])
        sdc = ck.synthetic_arch_state_dict(int(g["seed_coarse"]), arch, float(g["gain"]))
        sdf = ck.synthetic_arch_state_dict(int(g["seed_fine"]), arch, float(g["gain"]))
    else:
        sdc = ck.synthetic_state_dict(int(g["seed_coarse"]), float(g["gain"]))
        sdf = ck.synthetic_state_dict(int(g["seed_fine"]), float(g["gain"]))
    assert ck.weights_checksum(sdc) == str(g["ck_coarse"])
    assert ck.weights_checksum(sdf) == str(g["ck_fine"])
    edit = {k[6:]: g[k].tolist() for k in g.files if k.startswith("edit__")}
    gt = {k[4:]: g[k] for k in g.files if k.startswith("gt__")}
    return g, sdc, sdf, gt, edit


def golden_aux(g):
    """{'albedo_mlp' | 'roughness_mlp' | 'irradiance_mlp': PositionMLP state dict} of a fixture rendered with auxiliary networks."""
    from ibl_nerf_amd import checkpoint as ck
    arch = tuple(int(v) for v in g["arch"]) if "arch" in g.files else None       # (a smaller architecture: every network of the fixture in its own shapes)
    return {k[5:]: (ck.synthetic_position_direction_mlp(int(g[k]), 1, float(g["gain"]), arch) if k == "aux__depth_mlp"
                    else ck.synthetic_position_mlp(int(g[k]), ck.AUX_OUT_CH[k[5:]], float(g["gain"]), arch)) for k in g.files if k.startswith("aux__")}


def rel_linf(x, ref):
    """SURVEY.md §8 d: max|x - ref| / max|ref| (the per-channel parity metric of north_star)."""
    x, ref = np.asarray(x, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.nanmax(np.abs(x.reshape(ref.shape) - ref)) / max(float(np.nanmax(np.abs(ref))), 1e-30))


RENDER_FIXTURES = ["cfg1_coarse_g10", "plain_g10", "plain_g16", "edit_g10", "insert_g10", "variant_lin_g10",
                   "edit2_g10", "variant_small_g10", "gtnormal_g10", "colorindep_g10", "fromgt_g10", "fromgt_insert_g10", "dirnormal_g10", "auxmlp_g10",
                   "auxmlp_lin_g10", "infernormal_g10", "infernormal_target_g10",
                   "infernormal_surface_g10", "inferdepth_g10", "edit3_g10", "arch_6x128_g10", "arch_4x64_g10", "arch_7x200_g10", "arch_aux_6x128_g10",
                   "arch_10x384_g10", "arch_8x512_g10"]      # (the last two: LARGER than the built architecture — round 6, the layer-by-layer path csrc/generic_mlp.hip)
FITTED_FIXTURES = ["fitted_plain", "fitted_edit", "fitted_insert", "fitted_wide"]   # rendered by the reference from the fitted checkpoint (fitted_wide: 1 024 rays, maps only)


def reference_floor(key, name="fitted_plain"):
    """The reference's own float64-vs-float32 relative L-inf on map `key` for the fitted checkpoint, as make_golden.py recorded it with
    fixture `name` (fixtures rendered with edits have none — the reference's masked assignments do not run in float64 — and take
    fitted_plain's)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if "floor__" + key not in g.files:
        g = np.load(os.path.join(GOLDEN, "fitted_plain.npz"))
    return float(g["floor__" + key])


def color_independent(g):
    """The fixture's networks were built with is_color_independent_to_direction (ibl_nerf.py:192)."""
    return "model__color_independent_to_direction" in g.files


@pytest.fixture(autouse=True)
def _oracle_model_flags():
    """The oracle takes the model flag from a module global; every test starts from the shipped value."""
    import iblnerf_oracle as O
    O.COLOR_INDEPENDENT = False
    yield
    O.COLOR_INDEPENDENT = False


def n_samples(g):
    """N_samples of a fixture (64 unless the fixture records another count)."""
    return int(g["n_samples"]) if "n_samples" in g.files else 64


def golden_flags(g):
    """Flag variants recorded with a fixture (use_radiance_linear / lindisp / lut_coefficient / epsilon /
    gamma_correct / correct_depth_for_prefiltered_radiance_infer / target_normal_map_for_radiance_calculation)."""
    return {k[6:]: g[k].item() for k in g.files if k.startswith("flag__")}


FROM_GT_FLAGS = ("calculate_albedo_from_gt", "calculate_roughness_from_gt", "calculate_irradiance_from_gt",
                 "depth_map_from_ground_truth", "infer_normal", "infer_depth")   # per-call flags (render kwargs), not construction options


def from_gt_flags(g):
    """The per-call raw2outputs flags of a fixture (render kwargs, not renderer construction options)."""
    return {k: v for k, v in golden_flags(g).items() if k in FROM_GT_FLAGS}


def ill_conditioned(g):
    """Fixtures whose DERIVED channels are chaotic even between two fp32 implementations: the
    wide-range checkpoint and HDR radiance (ReLU kinks + gamma of values near 0)."""
    return float(g["gain"]) > 1.0 or bool(golden_flags(g).get("use_radiance_linear", False))


def teacher_pass(g, p):
    """Recorded stage boundaries of pass p ('c' coarse | 'f' fine) of a fixture, for teacher forcing: the reference's own
    network-query results for the first k rays (k = rays whose raw rows the fixture keeps) and the pass's z_vals, rebuilt exactly
    as render_rays builds them (coarse grid; fine = sort(cat(coarse, recorded sample_pdf output)), ibl_nerf_renderer.py:701-707).
    Returns dict(k, z, zc, raw [k,S,18], sigma_offsets [4k,S] | None, refl_raw [k,Sc,18])."""
    import iblnerf_oracle as O
    raw = g["q_%s_main_raw" % p]
    k = raw.shape[0]
    zc = O.coarse_z(float(g["near"]), float(g["far"]), n_samples(g), k, bool(golden_flags(g).get("lindisp", False)))
    z = zc if p == "c" else np.sort(np.concatenate([zc, g["pdf_samples"][:k]], -1), -1)
    sig = g["q_%s_eps_sigma" % p][..., 0] if "q_%s_eps_sigma" % p in g.files else None
    assert raw.shape[1] == z.shape[1] and (sig is None or sig.shape == (4 * k, z.shape[1]))
    return dict(k=k, z=z.astype(np.float32), zc=zc, raw=raw, sigma_offsets=sig, refl_raw=g["q_%s_refl_raw" % p])


# fixtures whose recorded main-network raw rows are the whole input of raw2outputs (no auxiliary / normal network outputs, which the
# recorder does not keep)
TEACHER_FIXTURES = [n for n in RENDER_FIXTURES if not n.startswith(("auxmlp", "infernormal", "inferdepth", "arch_aux"))] + FITTED_FIXTURES
