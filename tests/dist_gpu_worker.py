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


def fitted_frame(torch, dist, rank, world, backend):
    """the FITTED checkpoint's 800x800 frame (the list route on: estimates, predicted offset copies, exact-fp32 coarse density) rendered by `world`
    ranks — interleaved rows, the route measured by every rank on the frame's seeded probe pixels — against the same frame rendered by this rank alone in one call:
    every export map bit for bit, the same route on every rank, no tripwire event."""
    import _pkg
    _pkg.load()
    from conftest import load_lut_rgb
    from ibl_nerf_amd import checkpoint as ck, dist as D, renderer as R
    H = W = 800
    fl = np.float32(0.5 * W / np.tan(0.5 * np.deg2rad(60.0)))
    K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    f = np.load(os.path.join(ROOT, "tests", "golden", "fitted_ckpt.npz"))
    r = R.Renderer(64, 128)
    r.load_weights(0, ck.blob_to_state_dict(f["coarse"]))
    r.load_weights(1, ck.blob_to_state_dict(f["fine"]))
    r.load_lut(load_lut_rgb())
    full = D.render_frame(r, H, W, K, c2w, 0.5, 8.0)                      # sharded: rows rank, rank + world, ... + one all-gather
    route, policy = r.get_route(), r.policy
    assert route["decided"] and route["coarse_share"] < 0.3 and policy["decision"] in ("fast", "safe") and r.trips == 0 and r.range_fallbacks == 0
    ro, rd = r.get_rays(H, W, K, c2w)
    whole = r.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), 0.5, 8.0)       # the same frame in one call of this rank (ten launches of 64 000 rays)
    assert r.last_selection()[0] > 0                                           # (the lists are on: this is the route a frame takes)
    for k in D.EXPORT_KEYS:
        assert torch.equal(full[k].reshape(-1), whole[k].reshape(-1)), k
    # one route, one decision on every rank
    mine = torch.tensor([route["coarse_share"], route["fine_main_share"], route["fine_offsets_share"], float(sum(route["estimates_plain_f16"])),
                         float(policy["decision"] == "safe")], dtype=torch.float64)
    ref = mine.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(ref, mine), (ref, mine)
    dist.barrier()
    print("BACKEND %s" % dist.get_backend(), flush=True)
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
        return fitted_frame(torch, dist, rank, world, backend)
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
