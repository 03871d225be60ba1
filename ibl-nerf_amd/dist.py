"""Ray-tile sharding of one frame across the GPUs of a node + reassembly by ONE all-gather.

The reference is single-process; rays are independent (no cross-ray op in
render_rays / raw2outputs), cost per ray is constant, so a static partition into contiguous
row tiles is perfectly balanced.  One process per GPU (`torch.distributed`, backend "nccl" =
RCCL over xGMI; "gloo" in the CPU tests).  Per frame and rank: generate the tile's rays from
(K, c2w, rows), render them, pack the wanted maps into one [rows, W, C] buffer, all-gather it
(the only exchange step on the path), unpack.  Weights/LUT are replicated by each rank's own
upload — no broadcast is needed because every rank reads the same checkpoint.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

# maps render_decomp_path exports as PNGs (ibl_nerf_renderer.py:870-900) — the default gather set
EXPORT_KEYS = ["color_map", "radiance_map", "radiance_map_1", "radiance_map_2", "radiance_map_3",
               "reflected_coarse_radiance_map_1", "reflected_coarse_radiance_map_2", "reflected_coarse_radiance_map_3",
               "irradiance_map", "reflected_radiance_map", "prefiltered_reflected_map", "albedo_map", "roughness_map",
               "specular_map", "diffuse_map", "n_dot_v_map", "target_normal_map", "disp_map", "depth_map"]


def tile_rows(H: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row tile of `rank`: (row0, n_rows); the first H % world ranks get one extra row."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(H, world)
    n = base + (1 if rank < extra else 0)
    row0 = rank * base + min(rank, extra)
    return row0, n


def view_indices(n_views: int, rank: int, world: int) -> range:
    """Whole-view sharding for a multi-view export (test.py renders ~100 independent views): rank r takes
    views r, r + world, ...  No exchange step: every rank writes its own views' files."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return range(rank, n_views, world)


def slice_gt_rows(gt_values: Optional[dict], W: int, row0: int, n_rows: int) -> dict:
    """gt_values arrive flattened to [H*W, C] (ibl_nerf_renderer.py:864-866); take this tile's rows."""
    if not gt_values:
        return {}
    return {k: v[row0 * W:(row0 + n_rows) * W] for k, v in gt_values.items()}


def pack_maps(maps: Dict[str, "object"], keys: Sequence[str], n_rows: int, W: int):
    """-> ([n_rows, W, C] tensor, [(key, channels)...]).  1-D maps contribute one channel."""
    import torch
    cols, layout = [], []
    for k in keys:
        v = maps[k].reshape(n_rows, W, -1)
        cols.append(v)
        layout.append((k, v.shape[-1]))
    return torch.cat(cols, -1).contiguous(), layout


def unpack_maps(buf, layout: List[Tuple[str, int]]) -> Dict[str, "object"]:
    out, c = {}, 0
    for k, n in layout:
        v = buf[..., c:c + n]
        out[k] = v[..., 0] if n == 1 and k not in ("irradiance_map",) else v
        c += n
    return out


def all_gather_frame(local, H: int, W: int, group=None):
    """local: [n_rows(rank), W, C] on every rank -> [H, W, C] on every rank (one collective)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    C = local.shape[-1]
    max_rows = tile_rows(H, 0, world)[1]
    row0, n = tile_rows(H, rank, world)
    assert local.shape[0] == n and local.shape[1] == W
    padded = local
    if n < max_rows:   # equal-size contributions let RCCL run one flat all-gather
        padded = torch.zeros((max_rows, W, C), dtype=local.dtype, device=local.device)
        padded[:n] = local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # test hook (several ranks sharing one GPU, which RCCL refuses): gloo exchanges host buffers
        staged = torch.empty((world * max_rows, W, C), dtype=local.dtype)
        dist.all_gather_into_tensor(staged, padded.contiguous().cpu(), group=group)
        gathered = staged.to(local.device)
    else:
        gathered = torch.empty((world * max_rows, W, C), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(gathered, padded.contiguous(), group=group)
    if H % world == 0:
        return gathered
    parts = [gathered[r * max_rows: r * max_rows + tile_rows(H, r, world)[1]] for r in range(world)]
    return torch.cat(parts, 0)


def render_frame_sharded(render_tile: Callable[[int, int], Dict[str, "object"]], H: int, W: int,
                         keys: Sequence[str] = EXPORT_KEYS, group=None) -> Dict[str, "object"]:
    """`render_tile(row0, n_rows)` renders this rank's rows and returns maps shaped [n_rows*W, ...] or
    [n_rows, W, ...].  Returns the full-frame maps ([H, W, ...]) on every rank."""
    import torch.distributed as dist
    grouped = dist.is_available() and dist.is_initialized()
    if grouped:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    row0, n = tile_rows(H, rank, world)
    maps = render_tile(row0, n)
    keys = list(keys) + [k for k in ("inferred_normal_map",) if k in maps and k not in keys]   # present under infer_normal only
    buf, layout = pack_maps(maps, keys, n, W)
    full = all_gather_frame(buf, H, W, group) if grouped else buf     # (a one-rank group still goes through the collective)
    return unpack_maps(full, layout)


def calibrate_on_frame(renderer, H, W, K, c2w, near, far, n=4096, seed=0):
    """mlp_precision="auto": decide the checkpoint's query routing (Renderer.calibrate) on `n` seeded pixels of the WHOLE frame — the same pixels on
    every rank, so that all tiles of a frame are rendered under one decision (the kernels are deterministic: same rays, same measurement)."""
    import numpy as np
    import torch
    pix = np.sort(np.random.RandomState(seed).permutation(H * W)[:min(n, H * W)])
    ro, rd = renderer.get_rays(H, W, K, c2w)                       # (15 MB for 800 x 800; once per checkpoint)
    idx = torch.as_tensor(pix, device=ro.device)
    if len(pix) < renderer.CAL_MIN_RAYS:       # a frame too small to measure on (the 9-row frames of the tests): rendered SAFE, question left open
        renderer._set_routing(renderer.SAFE_ROUTING)
        return None
    return renderer.calibrate(ro.reshape(-1, 3)[idx].contiguous(), rd.reshape(-1, 3)[idx].contiguous(), near, far)


def render_frame(renderer, H, W, K, c2w, near, far, keys: Sequence[str] = EXPORT_KEYS, gt_values=None, group=None,
                 **edit):
    """Full frame with the HIP renderer, sharded over the ranks of `group` (or unsharded without one)."""
    if getattr(renderer, "_auto", False) and renderer.policy is None:
        calibrate_on_frame(renderer, H, W, K, c2w, near, far)

    def tile(row0, n_rows):
        ro, rd = renderer.get_rays(H, W, K, c2w, row0, n_rows)
        return renderer.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), near, far,
                                    slice_gt_rows(gt_values, W, row0, n_rows), **edit)
    return render_frame_sharded(tile, H, W, keys, group)
