#!/usr/bin/env python3
"""One rank of the sharded-frame GPU test (tests/test_gpu_dist.py starts two of these with torch.distributed.run).

Every rank renders its row tile of a small frame with the HIP renderer under object-insertion / material-edit gt_values
(BASELINE configs 3 and 5: partition + per-tile override rows + pack + all-gather + unpack, all on device tensors), then
renders the WHOLE frame by itself and checks that the gathered frame is bit-identical to it.  Prints "DIST_OK <rank>".

Backend: "nccl" (= RCCL; one GPU per rank) or, with `--backend gloo`, gloo with all ranks sharing the one GPU of
a test box (RCCL refuses two ranks on one device).  Not a pytest file.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def fitted_frame(torch, dist, rank, world, backend, kind="fitted"):
    """a FITTED checkpoint's 800x800 frame (the list route on: estimates, predicted offset copies, exact-fp32 coarse density) rendered by
    `world` ranks — interleaved rows, route and precision table measured by every rank on the frame's seeded probe pixels — against the same frame rendered by this rank
    alone in one call: every export map bit for bit, the same route and table on every rank.  kind = "fitted2": the checkpoint whose frame TRIPS the estimate wire
    (round 5: only the rank that owned the sample doubled its margins and re-rendered its tile; round 6: the marked rays are rendered once more by whoever owns them,
    and a ray's result depends on the probe and the ray alone)."""
    import _pkg
    _pkg.load()
    from conftest import load_lut_rgb
    from ibl_nerf_amd import checkpoint as ck, dist as D, renderer as R
    H = W = 800
    fl = np.float32(0.5 * W / np.tan(0.5 * np.deg2rad(60.0)))
    K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    f = np.load(os.path.join(ROOT, "tests", "golden", kind + "_ckpt.npz"))
    r = R.Renderer(64, 128)
    r.load_weights(0, ck.blob_to_state_dict(f["coarse"]))
    r.load_weights(1, ck.blob_to_state_dict(f["fine"]))
    r.load_lut(load_lut_rgb())
    full = D.render_frame(r, H, W, K, c2w, 0.5, 8.0)                      # sharded: rows rank, rank + world, ... + one all-gather
    route, policy, tile_trips = r.get_route(), r.policy, r.trips
    assert route["decided"] and route["coarse_share"] < 0.3 and policy["decision"] in ("fast", "tiered", "safe") and r.alarms == 0 and r.range_fallbacks == 0
    assert not r.route.get("imposed") and not policy.get("imposed") and route["tripped"] == 0
    ro, rd = r.get_rays(H, W, K, c2w)
    # the same frame in one call of this rank (ten launches of 64 000 rays), on the frame's probe — what dist.render_frame does without a group
    whole = r.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), 0.5, 8.0, probe=D.frame_probe_for_call(r, H, W, K, c2w, 0.5, 8.0))
    assert r.last_selection()[0] > 0 or r.trips > tile_trips                   # (the lists are on: this is the route a frame takes)
    frame_trips = r.trips - tile_trips
    assert (frame_trips >= 1) == (kind == "fitted2"), (kind, frame_trips)      # (the second checkpoint's frame trips; the first one's does not)
    for k in D.EXPORT_KEYS:
        assert torch.equal(full[k].reshape(-1).nan_to_num(7.0), whole[k].reshape(-1).nan_to_num(7.0)), k
    # the tiles' marked rays add up to the frame's
    t = torch.tensor([float(tile_trips)], dtype=torch.float64)
    dist.all_reduce(t)
    assert int(t.item()) == frame_trips, (int(t.item()), frame_trips)
    # one route, one decision on every rank
    mine = torch.tensor([route["coarse_share"], route["fine_main_share"], route["fine_offsets_share"], float(sum(route["estimates_plain_f16"])),
                         float(("fast", "tiered", "safe").index(policy["decision"]))], dtype=torch.float64)
    ref = mine.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(ref, mine), (ref, mine)
    dist.barrier()
    print("BACKEND %s" % dist.get_backend(), flush=True)
    print("DIST_OK %d" % rank, flush=True)
    dist.destroy_process_group()


class _Poses:
    """What export.render_decomp_path needs of a dataset (ibl_nerf_renderer.py:819-910 reads .poses, .far and get_resized_normal_albedo)."""
    def __init__(self, poses, far):
        self.poses, self.far = poses, far

    def get_resized_normal_albedo(self, render_factor, i):
        return {}


def sharded_views(torch, dist, rank, world):
    """a view-sharded export (render_views.test deals the views round-robin: dist.view_indices) of two poses of the first fitted checkpoint — the
    frontal camera, whose view the calibration calls FAST, and a rotated one, which needs SAFE — against the same export rendered by ONE rank: every exported map of
    every view bit for bit, whichever rank rendered it and whatever it rendered before (round 5: each rank froze route and table on ITS first view)."""
    import _pkg
    _pkg.load()
    from conftest import load_golden, load_lut_rgb
    from ibl_nerf_amd import dist as D, export as E, model as M
    g, sdc, sdf, _, _ = load_golden("fitted_posed4k")
    H = W = 128
    focal = float(0.5 * W / np.tan(0.5 * np.deg2rad(60.0)))
    front = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    poses = [front, np.asarray(g["c2w"], dtype=np.float32)[:3, :4] if "c2w" in g.files else front, front]
    net_c, net_f = M.IBLNeRF(), M.IBLNeRF()
    net_c.load_state_dict(sdc)
    net_f.load_state_dict(sdf)
    kw = dict(network_fn=net_c, network_fine=net_f, N_samples=64, N_importance=128, perturb=False, raw_noise_std=0, lindisp=False, gamma_correct=True, lut_coefficient="F",
              epsilon=0.01, target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient_epsilon", correct_depth_for_prefiltered_radiance_infer=True,
              near=float(g["near"]), far=float(g["far"]), brdf_lut=torch.from_numpy(load_lut_rgb()).cuda(), coarse_radiance_number=3)
    ds = _Poses(poses, float(g["far"]))
    mine = list(D.view_indices(len(poses), rank, world))
    with torch.no_grad():
        part = E.render_decomp_path(ds, (H, W, focal), None, 32768, kw, savedir=None, render_factor=1, approximate_radiance=True, views=mine)
        alone = E.render_decomp_path(ds, (H, W, focal), None, 32768, kw, savedir=None, render_factor=1, approximate_radiance=True)
    r = next(iter(__import__("ibl_nerf_amd").renderer._renderers.values()))["r"]
    assert r.route is not None and r.route["decided"] and not r.route.get("imposed") and r.alarms == 0
    for k in alone:
        assert part[k].shape[0] == len(mine), (k, part[k].shape)
        for j, v in enumerate(mine):
            assert np.array_equal(part[k][j], alone[k][v], equal_nan=True), (k, v)
    dist.barrier()
    print("DIST_OK %d" % rank, flush=True)
    dist.destroy_process_group()


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = sys.argv[sys.argv.index("--backend") + 1] if "--backend" in sys.argv else "nccl"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    import _pkg
    _pkg.load()
    from conftest import load_golden, load_lut_rgb
    from ibl_nerf_amd import dist as D, renderer as R
    if "--fitted-frame" in sys.argv:
        return fitted_frame(torch, dist, rank, world, backend, sys.argv[sys.argv.index("--checkpoint") + 1] if "--checkpoint" in sys.argv else "fitted")
    if "--views" in sys.argv:
        return sharded_views(torch, dist, rank, world)
    H, W = 9, 16                                           # odd row count: tiles of 5 and 4 rows -> the padded all-gather
    K = np.array([[14.0, 0, 8], [0, 14.0, 4.5], [0, 0, 1]], dtype=np.float32)
    c2w = np.concatenate([np.eye(3), np.array([[0.05], [-0.1], [0.2]])], 1).astype(np.float32)
    lut = load_lut_rgb()
    for name in ("insert_g10", "edit_g10"):
        g, sdc, sdf, _, edit = load_golden(name)
        rng = np.random.RandomState(5)
        n = H * W
        gt = {}
        if name.startswith("insert"):
            level = rng.choice([0, 10, 20, 30, 40], size=n).astype(np.float32) / np.float32(255)
            gt["object_insert_mask"] = np.repeat(level[:, None], 3, 1)
            gt["object_insert_depth"] = rng.uniform(1, 2, (n, 1)).astype(np.float32)
            gt["object_insert_normal"] = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        else:
            level = rng.choice([0, 10, 20], size=n).astype(np.float32) / np.float32(255)
            gt["edit_intrinsic_mask"] = np.repeat(level[:, None], 3, 1)
            gt["edit_normal"] = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        gt = {k: torch.from_numpy(v).cuda() for k, v in gt.items()}
        r = R.Renderer(64, 128, max_rays_per_launch=64)
        r.load_weights(0, sdc)
        r.load_weights(1, sdf)
        r.load_lut(lut)
        full = D.render_frame(r, H, W, K, c2w, 0.5, 8.0, gt_values=gt, **edit)          # sharded: this rank's tile + all-gather
        assert full["color_map"].is_cuda and full["color_map"].shape == (H, W, 3)
        ro, rd = r.get_rays(H, W, K, c2w)
        whole = r.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), 0.5, 8.0, gt, **edit)   # the same frame on one rank
        for k in D.EXPORT_KEYS:
            assert torch.equal(full[k].reshape(-1), whole[k].reshape(-1)), (name, k)
        assert bool(torch.isfinite(full["color_map"]).all()) and r.range_fallbacks == 0
        # every rank holds the same frame
        mine = full["color_map"].contiguous() if backend == "nccl" else full["color_map"].cpu()
        ref = mine.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, mine)
    dist.barrier()
    print("BACKEND %s" % dist.get_backend(), flush=True)
    print("DIST_OK %d" % rank, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
