"""Parity at launch scale, against the REFERENCE's own render.

Fixtures (tests/golden/make_golden.py launch_scale_fixture, the reference's render_decomp on the fitted checkpoint, float32 and — as the
yardstick — float64):
    fitted_launch16k     16 384 seeded pixels of the 800x800 bench view (BASELINE configs[1])
    fitted_edit_cfg4      4 096 pixels under the kwargs of configs/IBL-NeRF/kitchen/edit_intrinsic.txt:8-16      (configs[3])
    fitted_insert_cfg5    4 096 pixels under the kwargs of configs/IBL-NeRF/living-room-2/object_insert.txt:8-14 (configs[4])
    fitted_posed4k        4 096 pixels of the same view from a rotated and translated camera
    fitted2_launch4k / fitted2_posed4k   4 096 pixels each, frontal / rotated camera, of a second, independently fitted checkpoint (scene 2)
The masks / normal / depth images of the two override configs are analytic functions of the pixel (tests/frame_overrides.py), so the
whole 800x800 frames of configs 4 and 5 are rendered here too and compared at the fixtures' pixels.

Rules.  A ray's own sensitivity in the reference = the largest of three per-ray yardsticks recorded with the fixture (ray_floor): its float64-vs-float32
difference, one-ulp nudges of the coarse weights, both branches of sample_pdf's threshold.  (A fourth recorded column, the reference's float32 render with its checkpoint
rounded to 22-bit mantissas, describes round 3's kernels and is reported, not used.)
The reference's own float64-vs-float32 difference is recorded PER RAY (fixture arrays floorray__*): a ray that grazes a
surface amplifies round-off without bound in the reference itself (its two runs differ by 7e-2 on the normal of the worst of 16 384 rays, by
5e-4 on depth), so an absolute L-inf bar over a launch is not attainable by any arithmetic; what is asserted instead:
  (i)   direct channels (and `weights`, per sample): every ray <= 5e-4, or <= 8x that ray's own reference difference where that is larger — for
        all but <= 0.05 % of the rays; and EVERY ray <= 1e-3 (the normal: 2e-3) or 16x its own (a ray's own sensitivity is a one-sample estimate; of
        65 536 rays one sits at 6.0e-4 in depth with its own reference runs 1.6e-5 apart, two at 1.0e-3 and 1.4e-3 on the normal — the first also with
        all-precise offsets) — so a ray of a direct map above the north-star 1e-3 is one the reference itself flags, and their number stays below the
        number of rays so flagged (depth: 2 against 17; normal: 20 against 642);
        99.9 % of the rays <= 2e-4 (`weights`: 1e-3) or 1.5x the reference's own 99.9th percentile.  The normal and n.v (a 50x amplified depth
        difference): the same with 1e-3 in both places (from the rotated camera the reference's own 99.9th percentile is 1.4e-3, the HIP path's 1.2e-3).  The worst rays are reported in DESIGN.md section 2 with the reference's own numbers;
  (ii)  the reflected-ray channels are ill-conditioned in the reference itself (its two runs differ by 1e-1 .. 6e-1 on the worst ray): their
        per-ray error DISTRIBUTION is bounded: median / 99 % / 99.9 % of the HIP path's per-ray error against the reference's float32 run
        stay within DIST_FACTOR x the same percentiles of the reference's own float64-vs-float32 per-ray difference, with an absolute floor of
        fp32 round-off; the worst ray within 4x the reference's own worst ray;
  (iii) whole frames of configs 4 / 5: the fixture pixels by rules (i) / (ii), and the override properties (masked pixels carry their
        override rows exactly).
"""
import json
import os

import numpy as np
import pytest

import frame_overrides as FO
from conftest import load_golden, rel_linf
from test_gpu_parity import DERIVED, DIRECT, make_renderer, to_np

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

REFLECTED = ["specular_map", "color_map", "reflected_radiance_map", "prefiltered_reflected_map",
             "reflected_coarse_radiance_map_1", "reflected_coarse_radiance_map_2", "reflected_coarse_radiance_map_3"]
DIST_FACTOR = 4.0           # per-ray percentiles of the reflected-ray channels: within this factor of the reference's own float64-vs-float32 percentiles
DIST_FLOOR = {50: 2e-6, 99: 2e-5, 99.9: 1e-4}   # ... or this (fp32 round-off of a gamma-corrected sum of 64 samples), whichever is larger


@pytest.fixture(scope="module")
def R():
    from ibl_nerf_amd import binding as B, renderer
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    B.load_library()
    return renderer


def per_ray(got, ref):
    """max over a map's channels of |got - ref|, over the map's global max: [n]."""
    import warnings
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(float(np.nanmax(np.abs(ref))), 1e-30)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)      # a ray whose map is NaN in the reference and here (disp_map of an empty ray): NaN, compared as such by the caller
        return np.nanmax(np.abs(np.asarray(got, dtype=np.float64).reshape(ref.shape) - ref).reshape(len(ref), -1), -1) / scale


NORMAL_LIKE = ["target_normal_map", "n_dot_v_map"]


def ray_floor(g, key, with_param=False):
    """The reference's own per-ray sensitivity on map `key`: the largest of THREE yardsticks the reference recorded about itself — its
    float64-vs-float32 difference (smooth conditioning), what one ulp on the coarse pass's weights does to its float32 output (`nudgeray__*`: the
    `denom < 1e-5` replacement of sample_pdf sits one ulp from an empty bin's denominator, which no float64 run can see; fine-pass maps only), and
    `branchray__*`: the same threshold deterministically, every critical sample placed by either branch.
    with_param=True adds `paramray__*` — the reference's float32 render with its checkpoint rounded to 22-bit mantissas, which is how the
    three-f16-product kernels hold the weights: that column describes THIS LIBRARY's round-3 arithmetic, not the reference, so it
    is a report column (scratch/rule_report.py) and no part of any pass criterion.  Since round 4 the one query it mattered for — the coarse pass's
    density, which places the fine samples — runs on the 15-slot form (operands to ~2^-26, csrc/mlp_kernel_mx.hip VAR_TRUNK_P)."""
    f = g["floorray__" + key].astype(np.float64)
    for y in ("nudgeray__", "branchray__") + (("paramray__",) if with_param else ()):
        if y + key in g.files:
            f = np.maximum(f, g[y + key].astype(np.float64))
    return f / (float(g["floor_scale"]) if "floor_scale" in g.files else 1.0)      # (the compact 65 536-ray fixture stores float16 of 2^14 x the value)


# Rule parameters: ONE set for both checkpoints and both cameras, on a default-constructed renderer (mlp_precision="auto").  (Round 3 kept a second,
# looser set for the checkpoint fitted after the policy was fixed — per-sample `weights` at 1.6e-3, a handful of normals from the rotated camera —
# and admitted the 22-bit-parameter column into ray_floor; both are gone: the coarse pass's density runs on the 15-slot form, and the renderer
# measures at load whether a checkpoint tolerates the fast table of the fine pass, Renderer.calibrate.)
# (refl_worst: the worst ray of a reflected-ray channel within this multiple of the reference's own worst ray — a one-sample statistic, "NOT a
# parity claim": on the second checkpoint one ray of color_map0 flips its reflected direction, 0.22 against the reference's own 0.034, so only the
# distribution is asserted there; depth_p99: the bulk of the depth map — from the rotated camera most rays of scene 2 cross unfitted space.)
STRICT = dict(frac8=2000, n16=0, w_base=5e-4, w_cap=1e-3, w_p999=1e-3, p999=2e-4, depth_p99=2e-5, depth_p999=1e-4)

# THE YARDSTICK (round 5): what an actual fp32 implementation attains.  tests/golden/c_restatement_column.json (tests/golden/make_c_column.py) holds,
# per launch-scale fixture and map, the fp32 C restatement's own rays above 1e-3 / 99.9th percentile / worst ray against the same reference render on the same rays.
# Asserted against it: on the direct maps and the normal the HIP path has no more rays above the north-star 1e-3 than the C restatement has, plus an allowance that
# scales with the launch — per 65 536 rays 2 on depth / albedo / roughness / irradiance, 5 on the normal / n.v, under either table (never less than 1 / 2: counts of
# single rays) — measured at the end of this round at 0 / 2 (fast table; safe: 0 / 1) on the 65 536-ray launch against the C restatement's 0 / 0, and EQUAL to the C
# restatement's counts on the seven 4 096 / 16 384-ray fixtures (round 4: 2 / 20 on the launch).  (Round 4 bounded the count by the number of rays the
# reference's own sensitivity yardsticks flag — 17 / 642 there: true of the reference's conditioning, not an allowance anyone earned.)  The reflected-ray channels are
# chaotic in every fp32 implementation — the C restatement itself has 2 - 8 % of a launch above 1e-3 there, and its WORST ray ranges from 1.3e-3 to 5.7e-1 over the eight
# fixtures (0.47 - 0.57 on the hold-out's coarse pass, where the HIP path's is 0.02; 3.9e-3 on the second checkpoint's color_map0, where the HIP path's is 0.22): a worst-ray
# bound there says nothing (round 4's `refl_worst` rule and its per-checkpoint exception are gone); their count is held to 5 x the C restatement's + 30, their distribution
# to the reference's own as before.
_COL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c_restatement_column.json")
C_COLUMN = json.load(open(_COL)) if os.path.exists(_COL) else {}


def c_allowance(key, n, decision):
    base = key.rstrip("0")
    # (depth / albedo / roughness / irradiance: 2 per launch; the maps that multiply two of them or carry the view-dependent radiance — diffuse, radiance_k, disp, acc: 4)
    per_launch = 5 if base in NORMAL_LIKE else (2 if base in ("depth_map", "albedo_map", "roughness_map", "irradiance_map", "target_depth_map") else 4)
    return max(2 if base in NORMAL_LIKE else 1, int(np.ceil(per_launch * n / 65536.0)))
# what the calibration decides on each fixture's checkpoint and camera (asserted: a fast decision on the second checkpoint would be a parity bug, a safe
# one on the first a 17 % slower frame for nothing)
# (fitted_posed4k: "fast" until round 5 tightened the calibration's limits on the normal; fitted3_*: the hold-out checkpoint, decided by limits frozen before it existed)
# (round 6: where FAST does not hold, the TIERED table — the fast forms, three f16 products on the samples k_importance flags — is measured against the same SAFE yardstick and
# limits, and holds them on every one of these cases: "safe" is now what remains when neither does)
DECISION = {"fitted_launch16k": "fast", "fitted_edit_cfg4": "fast", "fitted_insert_cfg5": "fast", "fitted_posed4k": "tiered", "fitted_launch64k": "fast",
            "fitted2_launch4k": "tiered", "fitted2_posed4k": "tiered", "fitted3_launch4k": "tiered", "fitted3_posed4k": "tiered"}


def rules_for(name):
    """ONE rule set for every checkpoint and camera (round 4's two exceptions for the second checkpoint are gone: its depth map's bulk is inside the strict bound since
    the coarse density runs in fp32, and the reflected channels' worst-ray bound is gone for everyone — see C_COLUMN above)."""
    return STRICT


def check_against_fixture(res, g, report=None, rules=STRICT, name=None, decision="fast"):
    """Rules (i) and (ii) for the rays of fixture `g`; `res` holds the HIP maps of exactly those rays.  name / decision: the fixture's row of the C-restatement column
    and the table the renderer chose (the allowance on the normal depends on it)."""
    col = C_COLUMN.get(name or "", {})
    we = int(g["weights_every"])
    compact = "compact" in g.files          # the 65 536-ray fixture keeps a subset of the maps
    if not compact:
        assert sorted(res.keys()) == sorted(k[5:] for k in g.files if k.startswith("out__"))
    for sfx in ("", "0"):
        for k in DIRECT + ["diffuse_map"] + NORMAL_LIKE:
            key = k + sfx
            if "out__" + key not in g.files:
                continue
            got, f = res[key], ray_floor(g, key)
            if k == "weights":
                got, f = got[::we], f[::we]
            e = per_ray(got, g["out__" + key])
            fine_w = key == "weights"          # per-sample weights of the fine pass
            base = 1e-3 if k in NORMAL_LIKE else (rules["w_base"] if fine_w else 5e-4)
            bad, worse = e > np.maximum(base, 8 * f), e > np.maximum(2e-3 if k in NORMAL_LIKE else (rules["w_cap"] if fine_w else 1e-3), 16 * f)
            if report is not None:
                report[key] = (float(np.nanmax(e)), float(g["floor__" + key]), int((e > 1e-3).sum()), int((f > 1e-3 / 8).sum()))
            # a ray's own sensitivity is sampled once per yardstick (one float64 run, a few nudged runs): 8x it holds for all but <= 0.05 % of the
            # rays (measured: none of 16 384 + 2 x 4 096 in the frontal view; one of 4 096 from the rotated camera, 13x, on the mixed trunk form —
            # 1.7x with all-precise offsets), 16x for every ray
            assert bad.sum() <= max(1, len(e) // rules["frac8"]), (key, "rays beyond max(%.0e, 8x their own reference difference):" % base, np.flatnonzero(bad)[:8], e[bad][:8], f[bad][:8])
            assert worse.sum() <= rules["n16"], (key, "rays beyond max(1e-3 | 2e-3, 16x their own reference difference):", np.flatnonzero(worse)[:8], e[worse][:8], f[worse][:8])
            # ... so a ray above the north-star 1e-3 is one the reference itself flags (own difference > 1e-3 / 8), and there are fewer of them
            assert (e > 1e-3).sum() <= (f > 1e-3 / 8).sum(), (key, int((e > 1e-3).sum()), int((f > 1e-3 / 8).sum()))
            # ... and, the yardstick: no more of them than an fp32 implementation has on these very rays, plus the allowance
            if key in col and k != "weights":
                allowed = col[key]["above_1e-3"] + c_allowance(key, len(e), decision)
                if report is not None:
                    report[key] = report[key] + (col[key]["above_1e-3"], allowed)
                assert (e > 1e-3).sum() <= allowed, (key, "rays above 1e-3:", int((e > 1e-3).sum()), "C restatement:", col[key]["above_1e-3"], "allowed:", allowed)
            p999 = max(1e-3 if k in NORMAL_LIKE else (rules["w_p999"] if fine_w else (1e-3 if k == "weights" else rules["p999"])), 1.5 * float(np.nanpercentile(f, 99.9)))     # ... or the reference's own 99.9th percentile (x1.5)
            assert float(np.nanpercentile(e, 99.9)) <= p999, (key, float(np.nanpercentile(e, 99.9)), p999)
        for k in REFLECTED:
            key = k + sfx
            if "out__" + key not in g.files:
                continue
            e, f = per_ray(res[key], g["out__" + key]), ray_floor(g, key)
            for q in (50, 99, 99.9):
                bound = max(DIST_FACTOR * float(np.nanpercentile(f, q)), DIST_FLOOR[q])
                if report is not None:
                    report["%s p%s" % (key, q)] = (float(np.nanpercentile(e, q)), bound)
                assert float(np.nanpercentile(e, q)) <= bound, (key, q, float(np.nanpercentile(e, q)), bound)
            if key in col:        # (the worst ray of these channels is chaotic in every fp32 implementation: the COUNT against the C restatement's)
                assert (e > 1e-3).sum() <= 5 * col[key]["above_1e-3"] + 30, (key, int((e > 1e-3).sum()), col[key]["above_1e-3"])
    if "out__z_std" in g.files:
        assert rel_linf(res["z_std"], g["out__z_std"]) <= max(1e-4, 4 * float(g["floor__z_std"]))
    d = per_ray(res["depth_map"], g["out__depth_map"])
    assert np.median(d) <= 2e-7 and np.percentile(d, 99) <= rules["depth_p99"] and np.percentile(d, 99.9) <= rules["depth_p999"], (np.median(d), np.percentile(d, 99), np.percentile(d, 99.9))


@pytest.mark.parametrize("name", ["fitted_launch16k", "fitted_edit_cfg4", "fitted_insert_cfg5", "fitted_posed4k", "fitted2_launch4k", "fitted2_posed4k", "fitted3_launch4k",
                                  "fitted3_posed4k"])
def test_launch_scale_render_vs_reference(R, lut, name):
    """The default mode on 16 384 / 4 096 / 4 096 / 4 096 rays of the reference's own render, in ONE launch (fitted_posed4k: a rotated and
    translated camera, BASELINE configs 3 / 5's "any fixed look-at": ray origins off the axis, directions through get_rays' rotation).
    fitted2_*: the same on a SECOND checkpoint (tests/golden/fit_checkpoint.py --scene 2: other geometry, materials, light, sharper density
    steps, another seed; fitted after the precision policy and these rules were fixed) — frontal and from the rotated camera.
    fitted3_*: the HOLD-OUT (round 5, VERDICT r4 next-2 a; fit_checkpoint.py --scene 3: thin discs of one or two fine samples, a floor at grazing incidence, empty space
    at raw density -3.5 — 1.5 above the selection margin —, the sharpest density steps of the three, other materials / light / seed), fitted and rendered by the reference
    AFTER the selection margin, the lists' break-evens, the estimate guard, the calibration limits and these rules were frozen: a default-constructed renderer, no tripwire
    event, no range event, the calibration's own decision, the same STRICT rules."""
    g, sdc, sdf, gt, edit = load_golden(name)
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384)
    if "c2w" in g.files:      # the fixture's rays are what get_rays builds on the device for its pose
        f_ = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
        ro_d, rd_d = r.get_rays(800, 800, np.array([[f_, 0, 400], [0, f_, 400], [0, 0, 1]], dtype=np.float32), g["c2w"])
        idx = torch.as_tensor(g["pix"], device=rd_d.device)
        assert np.array_equal(ro_d.reshape(-1, 3)[idx].cpu().numpy(), g["rays_o"]) and np.abs(rd_d.reshape(-1, 3)[idx].cpu().numpy() - g["rays_d"]).max() <= 2e-7
    assert r.mlp_precision == "auto" and r.policy is None
    res = to_np(r.render_rays(g["rays_o"], g["rays_d"], float(g["near"]), float(g["far"]), gt, **edit))
    assert r.range_fallbacks == 0 and r.trips == 0 and r.route["decided"] and r.route["estimates_plain_f16"] == [True, True], r.route
    assert r.policy["decision"] == DECISION[name], r.policy                  # decided on 4 096 of these very rays, before they were rendered
    for k in res:                                      # rays that end in empty space (acc = 0, fitted2_posed4k): disp = 1 / max(1e-10, depth / acc) is NaN in both
        if not k.startswith("weights"):                 # (the fixtures keep every weights_every-th row of the two weights tensors)
            assert np.array_equal(np.isnan(res[k]), np.isnan(g["out__" + k])), k
    check_against_fixture(res, g, rules=rules_for(name), name=name, decision=r.policy["decision"])
    psnr = 10 * np.log10(1.0 / max(np.mean((res["color_map"].astype(np.float64) - g["out__color_map"]) ** 2), 1e-30))
    # 55 dB, or what the reference's own two runs reach on these rays where that is less (its per-ray difference taken for all three channels:
    # 63.5 / 50.2 / 69.5 dB on the three fixtures; color_map carries the reflected-ray term)
    own = -10 * np.log10(np.mean((ray_floor(g, "color_map") * np.abs(g["out__color_map"]).max()) ** 2))
    assert psnr > min(55.0, own), (psnr, own)


def test_the_fast_table_on_the_second_checkpoint_is_what_the_calibration_says(R, lut):
    """The pinned fast table (mlp_precision="f16x3_mxfp6x") on the second checkpoint: the per-sample `weights` of the fine pass — the one output that
    sees the fine main query's 2^-16 density error unaveraged — sit at 1.6e-3 (99.9 %), beyond the strict rules; query_routing = FINE_MAIN_PRECISE alone
    repairs them (1.9e-4) and the frontal view then holds STRICT, the rotated one needs the fine offsets precise too: the calibration's "safe"."""
    from ibl_nerf_amd import binding as B
    g, sdc, sdf, gt, edit = load_golden("fitted2_launch4k")
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384, mlp_precision="f16x3_mxfp6x")
    res = to_np(r.render_rays(g["rays_o"], g["rays_d"], float(g["near"]), float(g["far"]), gt, **edit))
    e = per_ray(res["weights"][::int(g["weights_every"])], g["out__weights"])
    assert 1e-3 < np.percentile(e, 99.9) < 3e-3, np.percentile(e, 99.9)
    with pytest.raises(AssertionError):
        check_against_fixture(res, g, rules=rules_for("fitted2_launch4k"), name="fitted2_launch4k", decision="fast")
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384, mlp_precision="f16x3_mxfp6x", query_routing=B.ROUTE_FINE_MAIN_PRECISE)
    res = to_np(r.render_rays(g["rays_o"], g["rays_d"], float(g["near"]), float(g["far"]), gt, **edit))
    check_against_fixture(res, g, rules=rules_for("fitted2_launch4k"))            # round 4's rules (every ray against its own sensitivity in the reference) hold ...
    e = per_ray(res["weights"][::int(g["weights_every"])], g["out__weights"])
    assert np.percentile(e, 99.9) <= 4e-4 and e.max() <= 5e-4, (np.percentile(e, 99.9), e.max())
    check_against_fixture(res, g, rules=rules_for("fitted2_launch4k"), name="fitted2_launch4k", decision="fast")      # ... and so does the C-restatement yardstick
    # the safe table = f16x3_mxfp6, bit for bit (the same kernels on every query); the auto mode's TIERED decision stays within the calibration's limits of it
    g2, sdc2, sdf2, gt2, edit2 = load_golden("fitted2_posed4k")
    ra = make_renderer(R, g2, sdc2, sdf2, lut, max_rays_per_launch=16384)
    rb = make_renderer(R, g2, sdc2, sdf2, lut, max_rays_per_launch=16384, mlp_precision="f16x3_mxfp6")
    rc = make_renderer(R, g2, sdc2, sdf2, lut, max_rays_per_launch=16384, mlp_precision="f16x3_mxfp6x", query_routing=ra.SAFE_ROUTING)
    a, b, c = (x.render_rays(g2["rays_o"], g2["rays_d"], 0.5, 8.0) for x in (ra, rb, rc))
    assert all(_same(c[k], b[k]) for k in c)
    assert ra.policy["decision"] == "tiered" and not ra.policy["triggers_tiered"] and ra.policy["triggers"], ra.policy
    for k, lim in ra.CAL_LIMITS.items():
        e = (a[k].double() - b[k].double()).abs().reshape(a[k].shape[0], -1).amax(-1) / b[k].double().abs().max()
        assert float(torch.quantile(e.cpu(), 0.999)) <= lim and float((e > 1e-3).double().mean()) <= ra.CAL_MAX_SHARE_ABOVE_1E3.get(k, 1.0), k


def _same(x, y):
    """torch.equal, NaNs matching; the per-sample weights to 1e-15 (see above)."""
    if x.dim() == 2 and x.shape[1] > 18:
        return float((x.nan_to_num(7.0) - y.nan_to_num(7.0)).abs().max()) <= 1e-15
    return torch.equal(x.nan_to_num(7.0), y.nan_to_num(7.0))


def test_calibration_measures_and_decides(R, lut):
    """Renderer.calibrate (mlp_precision="auto"): FAST against SAFE routing of one context on the same rays.  First checkpoint: every limit held with
    room on four different subsets of the rays (weights <= 2.5e-4 against 5e-4) -> "fast", from either camera; second checkpoint: the per-sample
    weights at 1.1-1.8e-3 on every subset -> "safe".  A new checkpoint resets the decision; a call too small to measure on renders safe and leaves
    it open; pinned modes never calibrate; precision_report is the same measurement against any reference mode."""
    # (what triggers: the frontal view of the first checkpoint nothing — weights <= 2.5e-4 against 5e-4, the normal <= 2e-4 against 4e-4, no ray above 1e-3; its rotated
    # view the NORMAL, since round 5 — 99.9 % at 5e-4 .. 1.4e-3 and 0.05 - 0.24 % of the rays above 1e-3; the second checkpoint the per-sample weights, 1.1-1.8e-3)
    for name, want, by in (("fitted_launch16k", "fast", None), ("fitted_posed4k", "tiered", "target_normal_map"), ("fitted2_launch4k", "tiered", "weights"), ("fitted2_posed4k", "tiered", "weights")):
        g, sdc, sdf, _, _ = load_golden(name)
        r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384)
        n = g["rays_o"].shape[0]
        ro, rd = torch.from_numpy(g["rays_o"]).cuda(), torch.from_numpy(g["rays_d"]).cuda()
        for idx in (torch.linspace(0, n - 1, 4096).long(), torch.arange(2048), torch.arange(n - 2048, n), torch.arange(0, n, 2)[:4096]):
            r.policy = None
            p = r.calibrate(ro[idx.cuda()].contiguous(), rd[idx.cuda()].contiguous(), 0.5, 8.0)
            assert p["decision"] == want and p["rays"] == len(idx), (name, p)
            w, nm = p["metrics"]["weights"]["p999"], p["metrics"]["target_normal_map"]
            if want == "fast":
                assert w <= 2.5e-4 and nm["p999"] <= 2e-4 and nm["above_1e-3"] == 0.0 and not p["triggers"], (name, p)
            else:
                assert any(t.startswith(by) for t in p["triggers"]) and not p["triggers_tiered"], (name, p)
                mt = p["metrics_tiered"]
                assert mt["weights"]["p999"] <= 1.5e-4 and mt["target_normal_map"]["p999"] <= 2.5e-4 and mt["target_normal_map"]["above_1e-3"] == 0.0, (name, mt)
                assert (w >= 1e-3) if by == "weights" else (w <= 2.5e-4 and (nm["p999"] > 4e-4 or nm["above_1e-3"] > 3e-4)), (name, p)
        assert r.route["decided"] and r.route["imposed"] and r.trips == 0  # (an explicit calibrate() imposes route and table: measured on the first probe, the later decisions ran under that route)
    # a call too small to measure on: safe, undecided; a frame-sized call decides FOR ITSELF (round 6): the next small call is the first one again; another checkpoint likewise
    g, sdc, sdf, _, _ = load_golden("fitted_launch16k")
    g2, sdc2, sdf2, _, _ = load_golden("fitted2_launch4k")
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384)
    safe = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384, mlp_precision="f16x3_mxfp6")
    fast = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=16384, mlp_precision="f16x3_mxfp6x")
    few = r.render_rays(g["rays_o"][:300], g["rays_d"][:300], 0.5, 8.0)
    assert r.policy is None and torch.equal(few["weights"], safe.render_rays(g["rays_o"][:300], g["rays_d"][:300], 0.5, 8.0)["weights"])
    many = r.render_rays(g["rays_o"][:4000], g["rays_d"][:4000], 0.5, 8.0)
    assert r.policy["decision"] == "fast" and _same(many["weights"], fast.render_rays(g["rays_o"][:4000], g["rays_d"][:4000], 0.5, 8.0)["weights"])
    few2 = r.render_rays(g["rays_o"][:300], g["rays_d"][:300], 0.5, 8.0)                 # the frame-sized call's decision was its own: no memory
    assert r.policy is None and torch.equal(few2["weights"], few["weights"])
    r.load_weights(0, sdc2)
    r.load_weights(1, sdf2)
    assert r.policy is None
    r.render_rays(g2["rays_o"], g2["rays_d"], 0.5, 8.0)
    assert r.policy["decision"] == "tiered" and r.policy["routing"] & r.TIERED_ROUTING == r.TIERED_ROUTING and not r.policy["routing"] & r.SAFE_ROUTING
    assert fast.policy["decision"] == "pinned" and fast.calibrate(g["rays_o"][:2048], g["rays_d"][:2048], 0.5, 8.0)["decision"] == "pinned"
    rep = fast.precision_report(g2["rays_o"], g2["rays_d"], 0.5, 8.0, reference="f16x3_mxfp6")      # (on checkpoint 1's weights: the twin copies them)
    # (the coarse pass's weights are the same in both tables up to the estimates behind saturation — weights below 1e-8, from the fast or the precise FULL kernel)
    assert rep["weights"]["p999"] < 3e-4 and rep["weights0"]["max"] < 1e-9, (rep["weights"], rep["weights0"])


def _frame_rays(r):
    H = W = 800
    f = np.float32(0.5 * W / np.tan(0.5 * np.deg2rad(60.0)))
    K = np.array([[f, 0, 400], [0, f, 400], [0, 0, 1]], dtype=np.float32)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    ro, rd = r.get_rays(H, W, K, c2w)
    return ro.reshape(-1, 3), rd.reshape(-1, 3)


def test_full_frame_of_config_2_at_the_fixture_pixels(R, lut):
    """BASELINE configs[1] at full size, the frame bench.py times (fitted checkpoint, default mode, 65 536-ray launches): the 16 384 pixels the
    reference rendered, picked out of the 640 000, by the launch-scale rules — the frame path (get_rays on the device, ten equal launches)
    against the reference, not only a 16 384-ray call."""
    g, sdc, sdf, _, _ = load_golden("fitted_launch16k")
    r = make_renderer(R, g, sdc, sdf, lut)
    ro, rd = _frame_rays(r)
    m = r.render_rays(ro, rd, 0.5, 8.0)
    torch.cuda.synchronize()
    assert r.range_fallbacks == 0
    assert all(bool(torch.isfinite(v).all()) for v in m.values())
    idx = torch.as_tensor(g["pix"], device=rd.device)
    assert np.abs(rd[idx].cpu().numpy() - g["rays_d"]).max() <= 2e-7
    check_against_fixture({k: v[idx].cpu().numpy() for k, v in m.items()}, g, name="fitted_launch16k", decision=r.policy["decision"])
    assert r.policy["decision"] == "fast" and r.trips == 0


def test_one_whole_launch_against_the_reference(R, lut):
    """65 536 rays — one full launch of the frame bench.py times — rendered by the reference (fixture fitted_launch64k, compact: thirteen maps,
    both per-ray yardsticks as scaled float16; rays rebuilt from the pixel ids by get_rays): the launch-scale rules on every one of them."""
    import os
    from conftest import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, "fitted_launch64k.npz")):
        pytest.skip("fitted_launch64k.npz not generated (40 minutes of reference CPU time: python tests/golden/make_golden.py fitted_launch64k)")
    g, sdc, sdf, _, _ = load_golden("fitted_launch64k")
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=65536)
    ro, rd = _frame_rays(r)
    idx = torch.as_tensor(g["pix"], device=rd.device)
    m = r.render_rays(ro[idx].contiguous(), rd[idx].contiguous(), 0.5, 8.0)       # ONE launch of 65 536 rays
    torch.cuda.synchronize()
    assert r.range_fallbacks == 0 and r.trips == 0 and r.policy["decision"] == "fast"
    rep = {}
    check_against_fixture({k: v.cpu().numpy() for k, v in m.items()}, g, rep, name="fitted_launch64k", decision="fast")
    # (VERDICT r4's bar for this launch: 0 depth / <= 5 normal rays above 1e-3, where round 4 had 2 / 20 and the fp32 C restatement has 0 / 0.  Measured: 0 / 2 under the
    # fast table — the coarse density in the reference's own fp32 summation order, the copies' own-selection samples on three f16 products —, 0 / 1 under the safe one)
    assert rep["depth_map"][2] == 0 and rep["albedo_map"][2] == 0 and rep["target_normal_map"][2] <= 5, (rep["depth_map"], rep["albedo_map"], rep["target_normal_map"])
    # ... and the estimates' z-chunks / ranges stopping at each query's own selection threshold (the round's last change) against estimating every sample of every ray
    # (IBLNERF_ROUTE_ESTIMATES_WHOLE): the same samples refined; every map of the launch within 1e-7 of its largest value (the samples behind the threshold are dropped on
    # either route — what differs is their weight: exactly zero instead of below 1e-8 — on 8 192 rays every map is bit-identical, tests/test_gpu_fitted.py)
    # (both with round 4's offsets route, IBLNERF_ROUTE_OFFSETS_ESTIMATE_ALL: the copies' estimates in z-chunks too, the same candidates on either side)
    outs = {}
    for label, routing in (("chunks", ("offsets_estimate_all",)), ("whole", ("offsets_estimate_all", "estimates_whole"))):
        w = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=65536, query_routing=routing)
        outs[label] = (w.render_rays(ro[idx].contiguous(), rd[idx].contiguous(), 0.5, 8.0), w.last_selection(), w.last_executed_flops())
    assert outs["chunks"][1] == outs["whole"][1] and outs["chunks"][2] < 0.95 * outs["whole"][2], (outs["chunks"][1:], outs["whole"][1:])
    for k in outs["whole"][0]:
        d = float((outs["chunks"][0][k] - outs["whole"][0][k]).abs().max() / outs["whole"][0][k].abs().max().clamp_min(1e-30))
        assert d <= 1e-7, (k, d)


@pytest.mark.parametrize("name,rows_fn", [("fitted_edit_cfg4", FO.edit_rows), ("fitted_insert_cfg5", FO.insert_rows)])
def test_full_frame_of_configs_4_and_5(R, lut, name, rows_fn):
    """BASELINE configs[3] / [4] at full size on the HIP path: 640 000 rays of the fitted checkpoint under the shipped edit / insert kwargs
    with image-shaped gt_values; the fixture's 4 096 pixels against the reference, and the override properties on every masked pixel."""
    g, sdc, sdf, _, edit = load_golden(name)
    r = make_renderer(R, g, sdc, sdf, lut)
    ro, rd = _frame_rays(r)
    pix = g["pix"]
    # the fixture's rays are these pixels' rays (get_rays is bit-exact on origins, 2e-7 on directions)
    assert np.abs(rd[torch.as_tensor(pix, device=rd.device)].cpu().numpy() - g["rays_d"]).max() <= 2e-7
    gt = rows_fn(np.arange(800 * 800))
    for k, v in gt.items():
        assert np.array_equal(v[pix], g["gt__" + k]), k
    m = r.render_rays(ro, rd, 0.5, 8.0, {k: torch.from_numpy(v).cuda() for k, v in gt.items()}, **edit)
    torch.cuda.synchronize()
    assert r.range_fallbacks == 0
    assert all(bool(torch.isfinite(v).all()) for k, v in m.items() if not k.startswith("disp_map"))
    idx = torch.as_tensor(pix, device=rd.device)
    check_against_fixture({k: v[idx].cpu().numpy() for k, v in m.items()}, g, name=name, decision=r.policy["decision"])
    # override properties, all 640 000 pixels (ibl_nerf_renderer.py:253-256, :378-410): both passes
    mask_key = "edit_intrinsic_mask" if name.endswith("cfg4") else "object_insert_mask"
    level = gt[mask_key][:, 0]
    any_obj = level > 0
    assert 0.05 < any_obj.mean() < 0.5
    nimg = 2.0 * gt["edit_normal" if name.endswith("cfg4") else "object_insert_normal"].astype(np.float64) - 1.0
    nimg /= np.linalg.norm(nimg, axis=-1, keepdims=True)
    for sfx in ("", "0"):
        nrm = m["target_normal_map" + sfx].cpu().numpy()
        assert np.abs(nrm[any_obj] - nimg[any_obj]).max() <= 2e-6
        rough = m["roughness_map" + sfx].cpu().numpy()
        if name.endswith("cfg4"):
            assert np.all(rough[any_obj] == np.float32(FO.EDIT_CFG4["editing_target_roughness_list"][0]))
            assert np.all(rough[~any_obj] > 0)
        else:
            alb, irr, tdep = (m[k + sfx].cpu().numpy() for k in ("albedo_map", "irradiance_map", "target_depth_map"))
            assert np.array_equal(tdep[any_obj], gt["object_insert_depth"][any_obj, 0])
            for k in range(4):
                sel = np.abs(level - np.float32(10 * (k + 1)) / np.float32(255)) < 1e-6
                assert sel.sum() > 1000
                assert np.all(rough[sel] == np.float32(FO.INSERT_CFG5["inserting_target_roughness_list"][k]))
                # albedo / irradiance leave raw2outputs through its output lambdas (gamma): every pixel of object k carries ONE value, the one
                # the reference returns at the fixture's pixels of that object
                fk = np.flatnonzero(np.abs(g["gt__object_insert_mask"][:, 0] - np.float32(10 * (k + 1)) / np.float32(255)) < 1e-6)
                assert len(fk) > 10
                for got, key in ((alb, "albedo_map"), (irr, "irradiance_map")):
                    want = g["out__" + key + sfx][fk[0]]
                    assert np.all(g["out__" + key + sfx][fk] == want)
                    assert np.abs(got[sel].reshape(sel.sum(), -1) - np.asarray(want).reshape(1, -1)).max() <= 1e-6, (key, k)
                    assert np.all(got[sel] == got[sel][0])
    # unmasked pixels are the plain render's: the overrides act per ray
    plain = r.render_rays(ro[:8000], rd[:8000], 0.5, 8.0)
    keep = torch.as_tensor(~any_obj[:8000], device=rd.device)
    for k in ("color_map", "target_normal_map", "roughness_map", "depth_map"):
        assert torch.equal(plain[k][keep], m[k][:8000][keep]), k


@pytest.mark.parametrize("config", ["plain", "insert"])
def test_a_band_of_the_frame_against_the_c_restatement(R, lut, config):
    """32 contiguous rows of the bench frame (25 600 rays no fixture holds) on the HIP path against the C restatement of the reference path
    (oracle/csrc — an independent fp32 implementation, itself pinned to the reference's fixtures and inside the launch-scale rules with room:
    tests/test_oracle_c.py) run on the host's cores: per-ray error distribution of every map.  The bounds are 2-4x what the 100 central rows
    measured (scratch/full_frame_vs_c.py; depth 99.9 % 1.4e-4, normal 99.9 % 6.1e-4, 3 / 50 of 80 000 rays above 1e-3 on depth / normal) —
    a regression guard over image area, where the fixture tests are the parity claim.  config "insert": the same band under BASELINE
    configs[4]'s kwargs with image-shaped override rows (tests/frame_overrides.py)."""
    import iblnerf_cpu as OC
    g, sdc, sdf, _, edit = load_golden("fitted_launch16k" if config == "plain" else "fitted_insert_cfg5")
    r = make_renderer(R, g, sdc, sdf, lut)
    ro, rd = _frame_rays(r)
    r0 = 384 if config == "plain" else 500                                                     # (rows 500-531 cross two of the inserted spheres)
    rows = np.arange(r0 * 800, (r0 + 32) * 800)
    sel = torch.as_tensor(rows, device=rd.device)
    gt = {k: v for k, v in FO.insert_rows(rows).items()} if config == "insert" else {}
    got = to_np(r.render_rays(ro[sel].contiguous(), rd[sel].contiguous(), 0.5, 8.0, {k: torch.from_numpy(v).cuda() for k, v in gt.items()}, **(edit if gt else {})))
    assert r.range_fallbacks == 0
    ref = OC.render_rays(sdc, sdf, ro[sel].cpu().numpy(), rd[sel].cpu().numpy(), 0.5, 8.0, lut, gt=gt, edit=edit if gt else None)
    assert sorted(got) == sorted(ref)
    n = len(rows)
    stat = {}
    for k in ref:
        e = per_ray(got[k], ref[k])
        stat[k] = (float(np.percentile(e, 50)), float(np.percentile(e, 99)), float(np.percentile(e, 99.9)), float(e.max()), int((e > 1e-3).sum()))
    for k in ("depth_map", "albedo_map", "roughness_map", "irradiance_map", "radiance_map", "diffuse_map", "disp_map", "weights"):
        p50, p99, p999, mx, over = stat[k]
        assert p50 <= 1e-5 and p99 <= 3e-4 and p999 <= 1e-3 and over <= n // 2000, (k, stat[k])
        assert stat[k + "0"][3] <= 1e-4, (k + "0", stat[k + "0"])                              # coarse pass: the all-precise kernels, no sampling step before it
    for k in ("target_normal_map", "n_dot_v_map"):
        p50, p99, p999, mx, over = stat[k]
        assert p50 <= 5e-5 and p99 <= 3e-4 and p999 <= 2.5e-3 and over <= n // 400, (k, stat[k])
        assert stat[k + "0"][2] <= 3e-4 and stat[k + "0"][3] <= 4e-3, (k + "0", stat[k + "0"])
    for k in REFLECTED:                                                                         # ill-conditioned in the reference itself: the bulk only
        assert stat[k][0] <= 2e-5 and stat[k + "0"][0] <= 2e-5, (k, stat[k], stat[k + "0"])
    psnr = 10 * np.log10(1.0 / max(np.mean((got["color_map"].astype(np.float64) - ref["color_map"]) ** 2), 1e-30))
    assert psnr > 55.0, psnr
    if gt:                                                                                      # both implementations carry the override rows themselves
        m = gt["object_insert_mask"][:, 0] > 0
        assert m.sum() > 1000 and np.array_equal(got["target_depth_map"][m], ref["target_depth_map"][m])
