"""Round 6: WHAT a render call is rendered under is decided for that call alone (route and precision table measured on a probe of the call's own rays, or on the
probe its caller hands it), and a ray the estimate tripwire marks is rendered once more by itself.  Nothing carries over from one call to the next but the weights —
what the reference guarantees by construction: every view of an export, and every chunk of a view, goes through the same arithmetic whatever came before it
(/root/reference/src/nerf_models/ibl_nerf_renderer.py:735-756, :768-769, :819-910).  Round 5 froze both decisions on the first frame-sized call of a checkpoint although
the answer is camera-dependent (VERDICT r5 weak-1 / missing-1)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import test_gpu_launch_scale as LS  # noqa: E402
from test_gpu_parity import make_renderer, to_np  # noqa: E402


@pytest.fixture(scope="module")
def R():
    from ibl_nerf_amd import binding as B, renderer
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    B.load_library()
    return renderer


def _same(a, b):
    return all(torch.equal(a[k].nan_to_num(7.0), b[k].nan_to_num(7.0)) for k in a)


def test_one_renderer_two_poses_each_view_decides_for_itself(R, lut):
    """One default-constructed Renderer renders the frontal view's fixture (fitted_launch16k: the FAST table holds) and then the rotated camera's (fitted_posed4k: the
    normal refuses FAST) — each view is measured on its own rays and passes the STRICT launch-scale rules against the reference's render; then the frontal view again,
    which is its first render bit for bit (round 5: the first view's table served every later pose).  The reverse order on a fresh context gives the same bits."""
    ga, sdc, sdf, gta, edita = load_golden("fitted_launch16k")
    gb, _, _, gtb, editb = load_golden("fitted_posed4k")
    assert LS.DECISION["fitted_launch16k"] == "fast" and LS.DECISION["fitted_posed4k"] == "tiered"
    r = make_renderer(R, ga, sdc, sdf, lut)
    assert r.mlp_precision == "auto" and r.policy is None and r.route is None
    a1 = r.render_rays(ga["rays_o"], ga["rays_d"], 0.5, 8.0, gta, **edita)
    assert r.policy["decision"] == "fast" and not r.policy.get("imposed") and r.route["decided"] and not r.route.get("imposed")
    LS.check_against_fixture(to_np(a1), ga, rules=LS.rules_for("fitted_launch16k"), name="fitted_launch16k", decision="fast")
    b1 = r.render_rays(gb["rays_o"], gb["rays_d"], float(gb["near"]), float(gb["far"]), gtb, **editb)
    assert r.policy["decision"] == "tiered", r.policy                                # measured on THIS view's rays
    LS.check_against_fixture(to_np(b1), gb, rules=LS.rules_for("fitted_posed4k"), name="fitted_posed4k", decision="tiered")
    a2 = r.render_rays(ga["rays_o"], ga["rays_d"], 0.5, 8.0, gta, **edita)
    assert r.policy["decision"] == "fast" and _same(a1, a2)                          # ... and the frontal view is FAST again: no demotion, no memory
    assert r.trips == 0 and r.alarms == 0 and r.range_fallbacks == 0
    r2 = make_renderer(R, ga, sdc, sdf, lut)
    b0 = r2.render_rays(gb["rays_o"], gb["rays_d"], float(gb["near"]), float(gb["far"]), gtb, **editb)
    a0 = r2.render_rays(ga["rays_o"], ga["rays_d"], 0.5, 8.0, gta, **edita)
    assert _same(b0, b1) and _same(a0, a1)


def test_a_call_is_rendered_as_if_it_were_the_first(R, lut):
    """Call history changes nothing: a 64-ray call (too small to measure on: every sample evaluated, SAFE table) gives the same bits before and after a frame-sized
    call has decided FAST with lists for itself; an IMPOSED route / table (decide_route, set_route, calibrate) is the explicit exception and is withdrawn by
    set_route(None) / policy = None / load_weights."""
    g, sdc, sdf, _, _ = load_golden("fitted_launch16k")
    r = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=4096)
    small = r.render_rays(g["rays_o"][:64], g["rays_d"][:64], 0.5, 8.0)
    assert r.last_selection() == (0, 0) and r.route is None and r.policy is None and not r.get_route()["decided"]
    big = r.render_rays(g["rays_o"][:8192], g["rays_d"][:8192], 0.5, 8.0)
    assert r.last_selection()[0] > 0 and r.route["decided"] and r.route["probe_rays"] == 4096 and r.policy["decision"] == "fast"
    again = r.render_rays(g["rays_o"][:64], g["rays_d"][:64], 0.5, 8.0)
    assert r.last_selection() == (0, 0) and r.route is None and r.policy is None and not r.get_route()["decided"] and _same(small, again)
    # imposing: the measured route and table now serve the small call too ...
    route = r.decide_route(g["rays_o"][:8192], g["rays_d"][:8192], 0.5, 8.0)
    assert route["imposed"] and r.calibrate(torch.from_numpy(g["rays_o"][:8192:2].copy()).cuda(), torch.from_numpy(g["rays_d"][:8192:2].copy()).cuda(), 0.5, 8.0)["imposed"]
    listed = r.render_rays(g["rays_o"][:64], g["rays_d"][:64], 0.5, 8.0)
    assert r.last_selection()[1] >= 64 * (64 * 7 + 192) and r.route["imposed"] and r.policy["decision"] == "fast"
    for k in ("depth_map", "albedo_map", "target_normal_map", "weights"):
        assert float((listed[k] - small[k]).abs().max() / small[k].abs().max()) <= 5e-4, k          # (another table, lists: the same picture)
    # ... until it is withdrawn
    r.set_route(None)
    r.policy = None
    assert _same(r.render_rays(g["rays_o"][:64], g["rays_d"][:64], 0.5, 8.0), small)
    assert _same(r.render_rays(g["rays_o"][:8192], g["rays_d"][:8192], 0.5, 8.0), big)


def _frame(r, K=None, c2w=None):
    H = W = 800
    fl = np.float32(0.5 * W / np.tan(0.5 * np.deg2rad(60.0)))
    K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32) if K is None else K
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32) if c2w is None else c2w
    ro, rd = r.get_rays(H, W, K, c2w)
    return H, W, K, c2w, ro.reshape(-1, 3), rd.reshape(-1, 3)


def test_tripped_rays_are_repeated_by_themselves_and_the_split_of_a_frame_changes_nothing(R, lut):
    """The second fitted checkpoint's 800 x 800 frame: the probe's 4 096 pixels measure a selection margin of 2; somewhere in the frame's 640 000 rays a list launch then
    refines a positive density whose estimate lay below -1 — the tripwire.  Round 5 doubled the context's margins for good and rendered the whole frame again (so an 8-rank
    frame was not the 1-rank frame: only the rank that owned the sample re-rendered, under other margins than its peers).  Round 6: k_tripwire marks the RAY, the wrapper
    renders the marked rays once more with every sample evaluated and overwrites their rows; the route is untouched.  A ray's result is then a function of (probe, ray):
    the frame in one call == the frame as two halves == the frame as 8 interleaved row tiles, bit for bit, on every map."""
    from ibl_nerf_amd import dist as D
    g, sdc, sdf, _, _ = load_golden("fitted2_launch4k")
    r = make_renderer(R, g, sdc, sdf, lut)
    H, W, K, c2w, ro, rd = _frame(r)
    probe = D.frame_probe_for_call(r, H, W, K, c2w, 0.5, 8.0)
    assert probe["rays_d"].shape == (4096, 3) and torch.equal(probe["rays_d"], rd[torch.as_tensor(D.probe_pixels(H, W), device=ro.device)])     # the frame's own rays, bit for bit
    whole = r.render_rays(ro, rd, 0.5, 8.0, probe=probe)
    assert r.trips >= 1 and r.alarms == 0 and r.route["select_margin"] == [2.0, 2.0] and r.route["tripped"] == 0 and r.policy["decision"] == "tiered", (r.trips, r.route)
    marked = set(int(i) for i in r.last_trip_rays.cpu())
    assert 1 <= len(marked) <= 64, len(marked)
    n = ro.shape[0]
    t0 = r.trips
    halves = [r.render_rays(ro[a:b].contiguous(), rd[a:b].contiguous(), 0.5, 8.0, probe=probe) for a, b in ((0, n // 3), (n // 3, n))]
    assert r.trips - t0 == len(marked)                                                # the same rays are marked whatever call they are rendered in
    for k in whole:
        assert torch.equal(torch.cat([h[k] for h in halves]).nan_to_num(7.0), whole[k].nan_to_num(7.0)), k
    tiles = D.render_frame(r, H, W, K, c2w, 0.5, 8.0)                                 # (no process group: one tile; the rays generated row-strided)
    for k in D.EXPORT_KEYS:
        assert torch.equal(tiles[k].reshape(-1).nan_to_num(7.0), whole[k].reshape(-1).nan_to_num(7.0)), k
    rows = [D.tile_row_indices(H, t, 8) for t in range(8)]
    parts = []
    for rr in rows:
        to, td = r.get_rays_strided(H, W, K, c2w, rr.start, rr.step, len(rr))
        parts.append(r.render_rays(to.reshape(-1, 3), td.reshape(-1, 3), 0.5, 8.0, probe=probe))
    for k in ("depth_map", "target_normal_map", "color_map", "albedo_map", "weights"):
        full = torch.stack([p[k].reshape((len(rows[0]), W) + tuple(p[k].shape[1:])) for p in parts], 1).reshape((H * W,) + tuple(parts[0][k].shape[1:]))
        assert torch.equal(full.nan_to_num(7.0), whole[k].nan_to_num(7.0)), k
    # the marked rays' rows are the every-sample evaluation's (iblnerf_set_lists 0; the coarse density in exact fp32 there too): rendering exactly those rays with the lists
    # off gives their rows bit for bit — whichever call marked them
    idx = torch.as_tensor(sorted(marked), device=ro.device)
    r.lib.iblnerf_set_lists(r.ctx, 0)
    try:
        alone, _, _ = r._render(ro[idx].contiguous(), rd[idx].contiguous(), 0.5, 8.0, None, {})
        one, _, _ = r._render(ro[idx[:1]].contiguous(), rd[idx[:1]].contiguous(), 0.5, 8.0, None, {})
    finally:
        r.lib.iblnerf_set_lists(r.ctx, 1)
    for k in whole:
        assert torch.equal(alone[k].nan_to_num(7.0), whole[k][idx].nan_to_num(7.0)), k
        assert torch.equal(one[k].nan_to_num(7.0), whole[k][idx[:1]].nan_to_num(7.0)), k


def test_a_route_that_does_not_fit_the_call_raises_the_alarm(R, lut):
    """An audited sample — dropped as clearly empty — that turns out NOT to be empty (bit 4 of the range flags), or marks on more rays than max(16, n / 256), say the ROUTE
    is wrong for the call (its probe did not see what the call's rays see, or it was imposed from elsewhere): the route climbs the ladder (iblnerf_escalate_route: margins
    2 -> 4 -> 6, six-slot estimates, lists off) and the whole call is rendered again.  Built here by imposing the route measured on the fitted network — plain-f16 estimates,
    margin 2 — onto a network built to break plain-f16 estimates (test_gpu_fitted.cancelling_network: errors up to 6 in raw density); the result is the every-sample
    render's, bit for bit."""
    import test_gpu_fitted as TF
    g, sdc, sdf, _, _ = load_golden("fitted_launch16k")
    sd = TF.cancelling_network(g, sdc)
    n = 8192
    ro, rd = g["rays_o"][:n], g["rays_d"][:n]
    whole = make_renderer(R, g, sd, sdf, lut, max_rays_per_launch=4096, mlp_precision="f16x3_mxfp6x", query_routing=("coarse_density_all_points",)).render_rays(ro, rd, 0.5, 8.0)
    good = make_renderer(R, g, sdc, sdf, lut, max_rays_per_launch=4096, mlp_precision="f16x3_mxfp6x")
    good.render_rays(ro, rd, 0.5, 8.0)
    assert good.trips == 0 and good.route["estimates_plain_f16"] == [True, True]
    r = make_renderer(R, g, sd, sdf, lut, max_rays_per_launch=4096, mlp_precision="f16x3_mxfp6x")
    r.set_route(good.route)
    assert r.estimate_policy(0) == (True, True) and r.route["imposed"]
    got = r.render_rays(ro, rd, 0.5, 8.0)
    assert r.alarms >= 1 and r.route["tripped"] == 2 and r.route["imposed"] and r.range_fallbacks == 0, (r.alarms, r.route)
    for k in got:
        assert torch.equal(got[k], whole[k]), k
    a = r.alarms
    r.render_rays(ro, rd, 0.5, 8.0)
    assert r.alarms == a                                                              # (an imposed route keeps what it learned)
    # left to itself the call's own probe sees the network for what it is: plain-f16 estimates refused on sight, the ladder climbed on the probe (and, for what only the
    # call's other rays show, by the alarm): the same result
    r = make_renderer(R, g, sd, sdf, lut, max_rays_per_launch=4096, mlp_precision="f16x3_mxfp6x")
    got = r.render_rays(ro, rd, 0.5, 8.0)
    assert r.route["tripped"] == 2 and r.probe_escalations >= 1 and r.last_selection() == (0, 0), (r.route, r.probe_escalations, r.alarms)
    for k in got:
        assert torch.equal(got[k], whole[k]), k


def test_call_sizes_around_the_decision_thresholds(R, lut):
    """0, 1, 1 023 rays: nothing to measure on — every sample evaluated, SAFE table.  1 024 .. 2 047: a route of the call's own, SAFE table.  2 048 .. 4 096: the call is its
    own probe and the decided table's probe render is returned.  4 097 and more: a strided probe, then the render.  Whatever the size, a ray's maps agree with the same ray's
    in a call of another size to well within the parity bar (the tables differ by at most the calibration limits), and the same call twice gives the same bits."""
    g, sdc, sdf, _, _ = load_golden("fitted_launch16k")
    r = make_renderer(R, g, sdc, sdf, lut)
    ro, rd = torch.from_numpy(g["rays_o"]).cuda(), torch.from_numpy(g["rays_d"]).cuda()
    empty = r.render_rays(ro[:0], rd[:0], 0.5, 8.0)
    assert all(v.shape[0] == 0 for v in empty.values()) and r.route is None
    ref = r.render_rays(ro[:8192], rd[:8192], 0.5, 8.0)
    assert r.policy["decision"] == "fast" and r.route["probe_rays"] == 4096
    seen = {}
    for n in (1, 1023, 1024, 2047, 2048, 4096, 4097):
        out = r.render_rays(ro[:n].contiguous(), rd[:n].contiguous(), 0.5, 8.0)
        seen[n] = (None if r.route is None else r.route["probe_rays"], None if r.policy is None else r.policy["decision"], r.last_selection()[1] > 0)
        again = r.render_rays(ro[:n].contiguous(), rd[:n].contiguous(), 0.5, 8.0)
        assert _same(out, again), n
        for k in ("depth_map", "albedo_map", "roughness_map", "irradiance_map", "target_normal_map", "radiance_map"):
            e = float((out[k] - ref[k][:n]).abs().max() / ref[k].abs().max())
            assert e <= 5e-4, (n, k, e)
    assert seen[1] == seen[1023] == (None, None, False)
    assert seen[1024] == (1024, None, True) and seen[2047] == (2047, None, True)
    assert seen[2048] == (2048, "fast", True) and seen[4096] == (4096, "fast", True) and seen[4097] == (4096, "fast", True), seen


def _big_call(g, copies=10):
    ro = torch.from_numpy(np.tile(g["rays_o"], (copies, 1))).cuda()
    rd = torch.from_numpy(np.tile(g["rays_d"], (copies, 1))).cuda()
    return ro, rd


def test_two_halves_on_two_streams_are_the_call(R, lut):
    """Round 6 (_render_pair): a frame-sized call (>= PAIR_MIN_RAYS rays) is rendered in two halves on two contexts and two HIP streams — one half's per-ray kernels
    under the other half's matrix kernels — under one decision (iblnerf_copy_route).  163 840 rays of the second checkpoint (the one whose frames trip the
    tripwire): every map, the per-call counters and the tripped rays against the same call on one context and one stream; and the insert configuration (per-ray
    override rows follow their rays into the halves)."""
    g, sdc, sdf, gt, edit = load_golden("fitted2_launch4k")
    ro, rd = _big_call(g, 40)
    n = ro.shape[0]
    assert n >= R.Renderer.PAIR_MIN_RAYS
    r = make_renderer(R, g, sdc, sdf, lut)
    r.pair_streams = False
    nf = (float(g["near"]), float(g["far"]))
    one = r.render_rays(ro, rd, *nf)
    sel_one, dec_one, trips_one = r.last_selection(), r.policy["decision"], r.trips
    assert r._pair is None
    r.pair_streams = True
    two = r.render_rays(ro, rd, *nf)
    assert r._pair is not None and r._pair_last and r.policy["decision"] == dec_one and r.trips == 2 * trips_one
    assert r.last_selection() == sel_one or trips_one                                    # (with tripped rays the last library call is their repeat)
    assert _same(one, two)
    again = r.render_rays(ro, rd, *nf)
    assert _same(two, again)
    # per-ray override rows (config 5: an inserted object) are dealt to the halves with their rays
    gi, sdc_i, sdf_i, gti, editi = load_golden("fitted_insert_cfg5")
    copies = -(-R.Renderer.PAIR_MIN_RAYS // gi["rays_o"].shape[0])
    roi, rdi = _big_call(gi, copies)
    gt_big = {k: (np.tile(v, (copies,) + (1,) * (np.ndim(v) - 1)) if hasattr(v, "shape") and np.ndim(v) >= 1 and len(v) == gi["rays_o"].shape[0] else v) for k, v in gti.items()}
    ri = make_renderer(R, gi, sdc_i, sdf_i, lut)
    ri.pair_streams = False
    a = ri.render_rays(roi, rdi, float(gi["near"]), float(gi["far"]), gt_big, **editi)
    ri.pair_streams = True
    b = ri.render_rays(roi, rdi, float(gi["near"]), float(gi["far"]), gt_big, **editi)
    assert ri._pair_last and _same(a, b)
    # new weights: the twin is rebuilt from them
    ri.load_weights(0, sdc_i)
    assert ri._pair is None


def test_a_range_event_on_the_pair_is_answered_on_one_context(R, lut):
    """An activation that leaves the f16 range in either half sends the call back to this context alone, where such events are answered (rescaling, or the bf16x3
    twin) — here the event is injected (the flag read of the pair reports bit 0 once)."""
    g, sdc, sdf, _, _ = load_golden("fitted_launch16k")
    ro, rd = _big_call(g, 8)
    r = make_renderer(R, g, sdc, sdf, lut)
    ref = r.render_rays(ro, rd, 0.5, 8.0)
    assert r._pair_last
    orig, fired = r.range_bits, []

    def once():
        bits = orig()
        if r._pair is not None and not fired and getattr(r, "_in_pair", False):
            fired.append(1)
            return bits | 1
        return bits
    r.range_bits = once
    orig_pair = r._render_pair

    def pair(*a, **k):
        r._in_pair = True
        try:
            return orig_pair(*a, **k)
        finally:
            r._in_pair = False
    r._render_pair = pair
    got = r.render_rays(ro, rd, 0.5, 8.0)
    assert fired and not r._pair_last and _same(ref, got)
