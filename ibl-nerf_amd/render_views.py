"""Test-view render driver: mirror of `test(args)` (src/test.py:30-151) on top of the HIP renderer.
Reads a reference config (include chain), the Mitsuba test split, the latest checkpoint under
<basedir>/<expname>, renders every test view (or the one edited / inserted view) and writes the
reference's PNG set to <export_basedir>/<expname>/testset_<step:06d>/.

    python render_test.py --config ../configs/IBL-NeRF/kitchen/IBL-NeRF.txt [--key value ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 render_test.py --config ...
        (one process per GPU; the test views are dealt out round-robin, each rank writes its own files)
"""
from __future__ import annotations

import os

import numpy as np

from . import config as C, dataset as DS, dist as D, export as E, model as M


def load_brdf_lut(path, device):
    """test.py:79-87: 8-bit RGB / 255 as a [3, 512, 512] tensor."""
    import torch
    from PIL import Image
    lut = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / np.float32(255.0)
    return torch.from_numpy(lut).to(device).permute(2, 0, 1)


def test(args, brdf_lut_path=None, render_fn=None):
    import torch
    device = getattr(args, "device", None) or torch.device("cuda")
    editing_idx = args.editing_img_idx if args.edit_intrinsic else (args.inserting_img_idx if args.insert_object else None)
    load_params = {                                                            # test.py:52-70
        "image_scale": args.image_scale, "coarse_radiance_number": args.coarse_radiance_number,
        "near_plane": args.near_plane, "far_plane": args.far_plane,
        "load_depth_range_from_file": args.load_depth_range_from_file, "gamma_correct": args.gamma_correct,
        "load_priors": False, "load_edit_intrinsic_mask": args.edit_intrinsic,
        "load_edit_albedo": args.edit_albedo_by_img, "load_edit_normal": args.edit_normal_by_img,
        "load_edit_irradiance": args.edit_irradiance_by_img, "load_edit_depth": args.edit_depth,
        "object_insert": args.insert_object, "editing_idx": editing_idx,
    }
    dataset = DS.load_dataset(args.dataset_type, args.datadir, split="test", skip=1, **load_params)
    dataset.load_all_data(num_of_workers=1, editing_idx=editing_idx)
    hwf = [dataset.height, dataset.width, dataset.focal]
    dataset.to_tensor(device)
    brdf_lut = load_brdf_lut(brdf_lut_path or "../data/ibl_brdf_lut.png", device)
    _, render_kwargs_test, start, _, _, _ = M.create_IBLNeRF(args)
    render_kwargs_test.update(dataset.get_near_far_plane())
    render_kwargs_test["brdf_lut"] = brdf_lut
    K = dataset.get_focal_matrix()
    if getattr(args, "export_basedir", None) is None:                          # test.py:165-166
        args.export_basedir = args.basedir.replace("logs", "logs_eval")
    testsavedir = os.path.join(args.export_basedir, args.expname, "testset_{:06d}".format(start))
    os.makedirs(testsavedir, exist_ok=True)
    views = None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        views = D.view_indices(len(dataset.poses), dist.get_rank(), dist.get_world_size())   # independent views: no collective
    with torch.no_grad():
        return E.render_decomp_path(dataset, hwf, K, args.chunk, render_kwargs_test, savedir=testsavedir, render_factor=1,
                                    approximate_radiance=True, render_fn=render_fn, views=views,
                                    **C.edit_params(args)), testsavedir


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", required=True)
    ap.add_argument("--brdf_lut", default=None, help="path of ibl_brdf_lut.png (default ../data/ibl_brdf_lut.png, as test.py)")
    ns, rest = ap.parse_known_args(argv)
    over = {}
    it = iter(rest)
    for tok in it:                                                             # --key value | --flag
        if not tok.startswith("--"):
            raise SystemExit("unexpected argument %r" % tok)
        k = tok[2:]
        d = C.DEFAULTS.get(k)
        over[k] = True if isinstance(d, bool) else C._convert(k, next(it))
    args = C.load_config(ns.config, **over)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                        # python -m torch.distributed.run --nproc-per-node N render_test.py --config ...
        import torch
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    _, out = test(args, brdf_lut_path=ns.brdf_lut)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    print("Done! ->", out)
