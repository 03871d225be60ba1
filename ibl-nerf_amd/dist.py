"""Ray-tile sharding of one frame across the GPUs of a node + reassembly by ONE all-gather.

The reference is single-process; rays are independent (no cross-ray op in
render_rays / raw2outputs).  Since round 4 a ray's cost depends on what it sees (estimates in
z-chunks, list lengths), so the partition is INTERLEAVED: rank r renders image rows r, r + N,
r + 2N, ... — every rank sees the same mix of sky, floor and silhouettes (measured:
profiles/r05_tiles) — instead of one contiguous band.  One process per GPU (`torch.distributed`,
backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  Per frame and rank: the tile's rays
from (K, c2w, rows), render them, pack the wanted maps into one [rows, W, C] buffer, ONE flat
all-gather (the only exchange step on the path), undo the interleave with a transposed view, unpack.
Weights/LUT are replicated by each rank's own upload — no broadcast is needed because every rank
reads the same checkpoint.  What a frame is rendered under (the route: which queries take lists,
on which estimates; the precision table of mlp_precision="auto") is measured PER FRAME on the
frame's seeded probe pixels — generated and measured by every rank for itself, same rays, same
deterministic kernels — and handed to the tile's render call (Renderer.render_rays probe=); a
ray the estimate tripwire marks is rendered once more by the rank that owns it.  No decision
outlives a frame and none depends on a rank's tile: the N-rank frame is the 1-rank frame bit
for bit, whatever the checkpoint, and so is a view rendered by any rank of a view-sharded export.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

# maps render_decomp_path exports as PNGs (ibl_nerf_renderer.py:870-900) — the default gather set
EXPORT_KEYS = ["color_map", "radiance_map", "radiance_map_1", "radiance_map_2", "radiance_map_3",
               "reflected_coarse_radiance_map_1", "reflected_coarse_radiance_map_2", "reflected_coarse_radiance_map_3",
               "irradiance_map", "reflected_radiance_map", "prefiltered_reflected_map", "albedo_map", "roughness_map",
               "specular_map", "diffuse_map", "n_dot_v_map", "target_normal_map", "disp_map", "depth_map"]


def tile_rows(H: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row tile of `rank`: (row0, n_rows); the first H % world ranks get one extra row.  (Round 4's partition; kept for the balance measurement
    and as `partition="contiguous"`.)"""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(H, world)
    n = base + (1 if rank < extra else 0)
    row0 = rank * base + min(rank, extra)
    return row0, n


def tile_row_indices(H: int, rank: int, world: int, partition: str = "interleaved") -> range:
    """The image rows of `rank`: rows rank, rank + world, ... ("interleaved", the default) or tile_rows' contiguous band."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    if partition == "interleaved":
        return range(rank, H, world)
    if partition != "contiguous":
        raise ValueError("partition must be 'interleaved' or 'contiguous'")
    row0, n = tile_rows(H, rank, world)
    return range(row0, row0 + n)


def view_indices(n_views: int, rank: int, world: int) -> range:
    """Whole-view sharding for a multi-view export (test.py renders ~100 independent views): rank r takes
    views r, r + world, ...  No exchange step: every rank writes its own views' files."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return range(rank, n_views, world)


def slice_gt_rows(gt_values: Optional[dict], W: int, row0, n_rows: Optional[int] = None) -> dict:
    """gt_values arrive flattened to [H*W, C] (ibl_nerf_renderer.py:864-866); take this tile's image rows: (row0, n_rows) contiguous, or a range of rows."""
    if not gt_values:
        return {}
    if n_rows is not None:
        return {k: v[row0 * W:(row0 + n_rows) * W] for k, v in gt_values.items()}
    rows = row0
    H = None
    out = {}
    for k, v in gt_values.items():
        H = v.shape[0] // W
        out[k] = v.reshape((H, W) + tuple(v.shape[1:]))[rows.start:rows.stop:rows.step].reshape((len(rows) * W,) + tuple(v.shape[1:]))
    return out


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


def all_gather_frame(local, H: int, W: int, group=None, partition: str = "interleaved"):
    """local: [n_rows(rank), W, C] on every rank -> [H, W, C] on every rank (one collective)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    C = local.shape[-1]
    max_rows = len(tile_row_indices(H, 0, world, partition))
    n = len(tile_row_indices(H, rank, world, partition))
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
    if partition == "interleaved":
        # block r of the gathered buffer holds rows r, r + world, ...: row j * world + r sits at [r, j] — a transposed view puts the frame back in order; the padding
        # rows of the shorter tiles (j = max_rows - 1, r >= H % world) are exactly the indices >= H
        if world == 1:
            return gathered
        return gathered.reshape(world, max_rows, W, C).transpose(0, 1).reshape(world * max_rows, W, C)[:H].contiguous()
    if H % world == 0:
        return gathered
    parts = [gathered[r * max_rows: r * max_rows + tile_rows(H, r, world)[1]] for r in range(world)]
    return torch.cat(parts, 0)


def render_frame_sharded(render_tile: Callable[..., Dict[str, "object"]], H: int, W: int,
                         keys: Sequence[str] = EXPORT_KEYS, group=None, partition: str = "interleaved") -> Dict[str, "object"]:
    """`render_tile(rows)` renders this rank's image rows (a range: rank, rank + world, ... — or the contiguous band of partition="contiguous") and returns maps
    shaped [n_rows*W, ...] or [n_rows, W, ...].  Returns the full-frame maps ([H, W, ...]) on every rank."""
    import torch.distributed as dist
    grouped = dist.is_available() and dist.is_initialized()
    if grouped:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    rows = tile_row_indices(H, rank, world, partition)
    n = len(rows)
    maps = render_tile(rows)
    keys = list(keys) + [k for k in ("inferred_normal_map",) if k in maps and k not in keys]   # present under infer_normal only
    buf, layout = pack_maps(maps, keys, n, W)
    full = all_gather_frame(buf, H, W, group, partition) if grouped else buf     # (a one-rank group still goes through the collective)
    return unpack_maps(full, layout)


def probe_pixels(H, W, n=4096, seed=0):
    """`n` seeded pixels of an H x W frame (sorted flat indices) — a function of (H, W, n, seed) alone: the same on every rank and for every view."""
    import numpy as np
    return np.sort(np.random.RandomState(seed).permutation(H * W)[:min(n, H * W)])


def frame_probe(renderer, H, W, K, c2w, n=4096, seed=0):
    """`n` seeded pixels of the WHOLE frame — the same on every rank — as rays: what a frame's route and precision table are measured on.  Only those pixels' rays are
    generated (iblnerf_get_rays_pixels: every pixel's ray is computed by itself, so they are the frame's rays bit for bit)."""
    return renderer.get_rays_pixels(H, W, K, c2w, probe_pixels(H, W, n, seed))


def frame_probe_for_call(renderer, H, W, K, c2w, near, far, gt_values=None, n=4096, seed=0):
    """The `probe` argument of Renderer.render_rays for a tile of this frame: the frame's seeded pixels as rays, with their rows of the frame's gt_values."""
    import torch
    pix = probe_pixels(H, W, n, seed)
    ro, rd = renderer.get_rays_pixels(H, W, K, c2w, pix)
    probe = {"rays_o": ro, "rays_d": rd}
    idx = torch.as_tensor(pix, device=ro.device)
    for name, v in (("near", near), ("far", far)):
        if hasattr(v, "shape") and int(torch.as_tensor(v).numel()) == H * W:
            probe[name] = torch.as_tensor(v, device=ro.device).reshape(-1)[idx]
    if gt_values:
        probe["gt_values"] = {k: (torch.as_tensor(v, device=ro.device).reshape(H * W, -1)[idx] if hasattr(v, "shape") and len(v) == H * W else v) for k, v in gt_values.items()}
    return probe


def decide_on_frame(renderer, H, W, K, c2w, near, far, n=4096, seed=0):
    """IMPOSES on the renderer the route (Renderer.decide_route: which queries run as estimate + list, on which estimates) measured on the frame's seeded probe pixels,
    until its next load_weights — an explicit, per-checkpoint decision for callers that want one (A/B measurements, tests).  render_frame does not need it: it hands the
    same probe to every tile's render call, which then decides for that frame alone.  A frame too small to measure on (the 9-row frames of the tests) leaves the route
    undecided: every query then evaluates all of its samples."""
    if renderer.route is not None or H * W < renderer.ROUTE_MIN_RAYS or renderer.mlp_precision == "bf16x3" or int(renderer.opt.max_rays_per_launch) < renderer.ROUTE_MIN_RAYS:
        return renderer.route
    ro, rd = frame_probe(renderer, H, W, K, c2w, n, seed)
    return renderer.decide_route(ro, rd, near, far)


def calibrate_on_frame(renderer, H, W, K, c2w, near, far, n=4096, seed=0):
    """mlp_precision="auto": IMPOSES the query routing (Renderer.calibrate) decided on `n` seeded pixels of the WHOLE frame, and the route measured on them
    (decide_on_frame), until the next load_weights — see decide_on_frame."""
    decide_on_frame(renderer, H, W, K, c2w, near, far, n, seed)
    if not getattr(renderer, "_auto", False):
        return renderer.policy
    if min(n, H * W) < renderer.CAL_MIN_RAYS:       # a frame too small to measure on (the 9-row frames of the tests): rendered SAFE, question left open
        renderer._set_routing(renderer.SAFE_ROUTING)
        return None
    ro, rd = frame_probe(renderer, H, W, K, c2w, n, seed)
    return renderer.calibrate(ro, rd, near, far)


def render_frame(renderer, H, W, K, c2w, near, far, keys: Sequence[str] = EXPORT_KEYS, gt_values=None, group=None, partition: str = "interleaved",
                 **edit):
    """Full frame with the HIP renderer, sharded over the ranks of `group` (or unsharded without one).  Every rank generates only its own rows' rays
    (iblnerf_get_rays_strided) and the frame's seeded probe pixels' (frame_probe_for_call), and hands that probe to its tile's render call: all tiles are rendered
    under the route and precision table of THIS frame — measured on the same rays by the same deterministic kernels on every rank — and a ray the estimate tripwire
    marks is rendered once more by the rank that owns it (Renderer.render_rays): no state, no exchange, and the N-rank frame is the 1-rank frame bit for bit."""
    gt_values = dict(gt_values or {})
    if edit.get("edit_intrinsic") and edit.get("edit_roughness") and edit.get("edit_roughness_by_img") and "edit_roughness" in gt_values:
        # the one override whose value depends on the reference's chunking of the FLAT frame (ibl_nerf_renderer.py:394-395 inside batchify_rays' chunks of
        # `chunk` = 32 768 rays, :735-756): resolved once on the whole frame, then sliced like any other row image
        from .renderer import resolve_edit_roughness, _dev_f32
        m = _dev_f32(gt_values["edit_intrinsic_mask"], renderer.device).reshape(H * W, -1)[:, 0]
        img = _dev_f32(gt_values["edit_roughness"], renderer.device).reshape(H * W, -1)[:, 0]
        gt_values["_edit_roughness_resolved"] = resolve_edit_roughness(m, img, edit.get("chunk", 1024 * 32))
    render_kw = {k: v for k, v in edit.items() if k != "chunk"}
    probe = None
    if H * W >= renderer.ROUTE_MIN_RAYS and not (renderer._route_imposed() and renderer._policy_imposed()):
        probe = frame_probe_for_call(renderer, H, W, K, c2w, near, far, gt_values)

    def tile(rows):
        ro, rd = renderer.get_rays_strided(H, W, K, c2w, rows.start, rows.step, len(rows))
        return renderer.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), near, far, slice_gt_rows(gt_values, W, rows), probe=probe, alarm_sync=alarm_sync(group), **render_kw)
    return render_frame_sharded(tile, H, W, keys, group, partition)


def alarm_sync(group=None):
    """The `alarm_sync` argument of Renderer.render_rays for a tile of a sharded frame, or None without a process group: (marked rays, rays, tripwire bits) of this rank's tile
    -> the frame's (sums and bitwise or over the ranks; one small all-reduce).  The renderer's alarm — escalate the route and render the call again when the marks say the
    route does not fit — is then decided on the frame's numbers by every rank alike: the tiles stay one frame.  (The common case costs one 3-integer exchange per frame.)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None

    def sync(marked, total, bits):
        dev = "cpu" if dist.get_backend(group) == "gloo" else "cuda"
        # bits as one count per bit: a SUM all-reduce then carries the bitwise OR too
        t = torch.tensor([marked, total] + [(bits >> b) & 1 for b in range(5)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, group=group)
        v = t.tolist()
        return int(v[0]), int(v[1]), sum((1 << b) for b in range(5) if v[2 + b] > 0)
    return sync
