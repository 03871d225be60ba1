"""Image-shaped gt_values for BASELINE.json configs 4 and 5 as deterministic functions of the pixel (no dataset on the GPU box).

configs/IBL-NeRF/kitchen/edit_intrinsic.txt:8-16 (one edit object: roughness 0, normal from an image) and
configs/IBL-NeRF/living-room-2/object_insert.txt:8-14 (four inserted objects: roughness / albedo / irradiance lists, depth and
normal images) read their masks from the dataset's `*_edit_mask.png` / `*_insert_mask.png` images (load_mitsuba.py), object i at
grey level 10 (i + 1).  Here the images are analytic, so that make_golden.py can hand the reference the rows of any seeded pixel set
and a GPU test can build the whole 800 x 800 frame from the same function.  Pure numpy; used by tests/golden/make_golden.py and the
GPU tests only."""
import numpy as np

# editing / inserting kwargs exactly as the two shipped configs set them
EDIT_CFG4 = dict(edit_intrinsic=True, num_edit_objects=1, edit_roughness=True, edit_normal=True, edit_normal_by_img=True,
                 editing_target_roughness_list=[0.0])
INSERT_CFG5 = dict(insert_object=True, num_insert_objects=4, inserting_target_roughness_list=[1, 1, 1, 1],
                   inserting_target_albedo_list=[0.870588, 0.3215686, 0.443137254, 0.05, 0.05, 0.05, 0.2, 0.2, 0.2, 0.05, 0.05, 0.05],
                   inserting_target_irradiance_list=[0.5, 0.1, 0.2, 0.2])


def _level(k):
    """Grey level of object k as an 8-bit image read by load_mitsuba.py hands it over: 10 (k + 1) / 255 in float32."""
    return np.float32(10 * (k + 1)) / np.float32(255)


def _xy(pix, W, H):
    i = (pix % W).astype(np.float32) / np.float32(W)
    j = (pix // W).astype(np.float32) / np.float32(H)
    return i, j


def edit_rows(pix, W=800, H=800):
    """gt_values rows of config 4 for pixel ids `pix`: one object (an ellipse over the middle of the frame, 18 % of the pixels) whose
    normals come from an image (a rippled surface, stored as (n + 1) / 2 like the dataset's normal PNGs)."""
    pix = np.asarray(pix, dtype=np.int64)
    i, j = _xy(pix, W, H)
    inside = ((i - 0.55) / 0.30) ** 2 + ((j - 0.45) / 0.19) ** 2 < 1.0
    mask = np.where(inside, _level(0), np.float32(0)).astype(np.float32)
    nx = 0.35 * np.sin(18.0 * i) * np.cos(7.0 * j)
    ny = 0.35 * np.cos(11.0 * j + 3.0 * i)
    nz = np.sqrt(np.maximum(1.0 - nx * nx - ny * ny, 0.0))
    n = np.stack([nx, ny, nz], -1)
    return {"edit_intrinsic_mask": np.repeat(mask[:, None], 3, 1).astype(np.float32),
            "edit_normal": (0.5 * n + 0.5).astype(np.float32)}


_SPHERES = [(0.30, 0.30, 0.13, 1.2), (0.72, 0.28, 0.10, 1.6), (0.35, 0.72, 0.15, 1.4), (0.70, 0.68, 0.12, 1.9)]   # (cx, cy, radius, centre depth)


def insert_rows(pix, W=800, H=800):
    """gt_values rows of config 5 for pixel ids `pix`: four spheres (discs in the image) with their own depth and normal images."""
    pix = np.asarray(pix, dtype=np.int64)
    i, j = _xy(pix, W, H)
    mask = np.zeros(pix.shape, np.float32)
    depth = np.full(pix.shape, 1.5, np.float32)
    n = np.tile(np.array([0.0, 0.0, 1.0], np.float32), (pix.size, 1))
    for k, (cx, cy, rad, zc) in enumerate(_SPHERES):
        dx, dy = (i - cx) / rad, (j - cy) / rad
        r2 = dx * dx + dy * dy
        hit = r2 < 1.0
        dz = np.sqrt(np.maximum(1.0 - r2, 0.0))
        mask = np.where(hit, _level(k), mask)
        depth = np.where(hit, zc - 0.3 * rad * dz, depth).astype(np.float32)
        nk = np.stack([dx, -dy, dz], -1)
        n = np.where(hit[:, None], nk, n)
    return {"object_insert_mask": np.repeat(mask[:, None], 3, 1).astype(np.float32),
            "object_insert_depth": depth[:, None].astype(np.float32),
            "object_insert_normal": (0.5 * n + 0.5).astype(np.float32)}
