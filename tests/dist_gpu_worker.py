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
