"""Reader for the reference's on-disk Mitsuba scene layout: the test-time subset
of `MitsubaDataset` / `NerfDataset` (src/dataset/dataset_mitsuba.py:8-137,
src/dataset/dataset_interface.py:12-300) that `test.py:36-73` and `render_decomp_path`
(ibl_nerf_renderer.py:819-868) touch:

    <basedir>/transforms_<split>.json      frames[i] = {fov_degree, transform 4x4 (Mitsuba: +Z forward)}
    <basedir>/min_max_depth.json           {min_depth, max_depth}  (load_depth_range_from_file)
    <basedir>/train/1.png                  read once for the image size
    <basedir>/<split>/<n>.png              n = skip*i + 1   (or n = editing_idx)
    <basedir>/<split>/<n>_edit_intrinsic_mask.png, _edit_albedo.png, _edit_normal.png,
                      _edit_roughness.png, _edit_irradiance.png, _edit_depth.npy
    <basedir>/<split>/<n>_insert_mask.png, _insert_depth.npy, _insert_normal.png
    <basedir>/<split>/<n>_{normal,albedo,roughness,irradiance,diffuse,specular}.png, <n>_depth.npy

and the two other layouts `load_dataset` knows (dataset_interface.py:316-331):

    colmap        <basedir>/transforms.json {camera_angle_x, h, w, frames[i] = {file_path, transform_matrix}},
                  <basedir>/images/<basename of file_path>; every 8th frame is the test split, the other seven of
                  each group of eight the train split (dataset_colmap.py:36-41); poses are used as stored
    mitsuba_eval  <basedir>/{rgb,diffuse,specular,irradiance,roughness,albedo}_%03d.png (dataset_mitsuba_eval.py:40-55):
                  image sets for the metric scripts, no cameras

Images are decoded with PIL as 8-bit RGB / 255 (the reference's cv2.imread + BGR->RGB yields the same
array for 8-bit PNGs; alpha is dropped by both).  `image_scale`: 1 and 0.5 (the value of configs/real/) follow the shipped configs; any
other scale goes through `resize_linear`, OpenCV's general bilinear path restated from its published algorithm (unpinned: no cv2 here);
0.5 follows cv2.resize's documented INTER_LINEAR behaviour for an exact 2x reduction (a 2x2 box mean,
rounded half up for 8-bit data) — cv2 is not in this image, so that one step is restated from OpenCV's source
and NOT pinned by a run.  (Round 3: the prior images and prefiltered targets of the training losses are read too.)  The prior images used only by the training losses were not
built: they raise.  Nothing here touches the GPU until `to_tensor`."""
from __future__ import annotations

import json
import math
import os

import numpy as np


def resize_half(a):
    """cv2.resize(a, None, fx=0.5, fy=0.5) (default INTER_LINEAR): for an exact 2x reduction OpenCV switches to its
    area filter, i.e. the mean of each 2x2 cell — (sum + 2) >> 2 for 8-bit data, sum * 0.25 for float."""
    h, w = a.shape[:2]
    if h % 2 or w % 2:
        raise NotImplementedError("image_scale 0.5 is built for even image sizes only (got %dx%d)" % (w, h))
    cells = a.reshape((h // 2, 2, w // 2, 2) + a.shape[2:])
    if a.dtype == np.uint8:
        return ((cells.astype(np.uint32).sum(axis=(1, 3)) + 2) >> 2).astype(np.uint8)
    return (cells.astype(np.float32).sum(axis=(1, 3)) * np.float32(0.25)).astype(a.dtype)


def _round_half_even(x):
    return int(np.rint(x))


def resize_linear(a, scale):
    """cv2.resize(a, None, fx=scale, fy=scale) with the default INTER_LINEAR for any other scale, restated from OpenCV's published
    algorithm (imgproc/resize.cpp: dsize = round(size * scale); source coordinate (d + 0.5) / scale - 0.5 in float, floor + fraction,
    the fraction zeroed where the tap leaves the image on the x axis, rows replicated on the y axis; 8-bit images in fixed point:
    coefficients rounded to 1/2048, horizontal pass in int32, vertical pass ((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2;
    float images in float).  UNPINNED: the build image has no cv2 to run against — no shipped config uses such a scale
    (all use 1, configs/real 0.5)."""
    h, w = a.shape[:2]
    dw, dh = _round_half_even(w * scale), _round_half_even(h * scale)
    if dw < 1 or dh < 1:
        raise ValueError("image_scale %r leaves no pixels of a %dx%d image" % (scale, w, h))
    inv = 1.0 / scale

    def taps(n_dst, n_src, zero_frac_at_border):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * inv - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if zero_frac_at_border:
            lo, hi = s < 0, s >= n_src - 1
            f = np.where(lo | hi, np.float32(0), f)
            s = np.where(lo, 0, np.where(hi, n_src - 1, s))
        return s, f

    sx, fx = taps(dw, w, True)
    sy, fy = taps(dh, h, False)
    clip = lambda v, n: np.clip(v, 0, n - 1)
    x0, x1 = sx, clip(sx + 1, w)
    y0, y1 = clip(sy, h), clip(sy + 1, h)
    src = a.reshape(h, w, -1)
    if a.dtype == np.uint8:
        q = lambda c: np.clip(np.rint(c.astype(np.float32) * np.float32(2048)), -32768, 32767).astype(np.int64)
        a0, a1, b0, b1 = q(1 - fx), q(fx), q(1 - fy), q(fy)
        s64 = src.astype(np.int64)
        rows = s64[:, x0] * a0[None, :, None] + s64[:, x1] * a1[None, :, None]              # [h, dw, c], scaled by 2048
        out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
        out = np.clip(out, 0, 255).astype(np.uint8)
    else:
        s32 = src.astype(np.float32)
        rows = s32[:, x0] * (1 - fx)[None, :, None] + s32[:, x1] * fx[None, :, None]
        out = (rows[y0] * (1 - fy)[:, None, None] + rows[y1] * fy[:, None, None]).astype(a.dtype)
    return out.reshape((dh, dw) + a.shape[2:])


def _scaled(a, scale):
    if scale == 1:
        return a
    if scale == 0.5 and a.shape[0] % 2 == 0 and a.shape[1] % 2 == 0:
        return resize_half(a)                    # the exact 2x reduction: OpenCV's area fast path
    return resize_linear(a, scale)


def load_image_from_path(path, scale=1):
    """utils/image_utils.py:39-47."""
    from PIL import Image
    return _scaled(np.asarray(Image.open(path).convert("RGB")), scale).astype(np.float32) / np.float32(255.0)


def load_numpy_from_path(path, scale=1):
    """utils/image_utils.py:58-64."""
    return _scaled(np.load(path), scale).astype(np.float32)


# sample key -> (file suffix, loader kind, flag attribute, keep only channel 0)
_PER_VIEW = [
    ("image", "%d.png", "img", "load_image", False),
    ("normal", "%d_normal.png", "img", "load_normal", False),
    ("albedo", "%d_albedo.png", "img", "load_albedo", False),
    ("roughness", "%d_roughness.png", "img", "load_roughness", True),
    ("depth", "%d_depth.npy", "npy", "load_depth", False),
    ("irradiance", "%d_irradiance.png", "img", "load_irradiance", False),
    ("diffuse", "%d_diffuse.png", "img", "load_diffuse_specular", False),
    ("specular", "%d_specular.png", "img", "load_diffuse_specular", False),
]
_EDIT = [
    ("edit_albedo", "%d_edit_albedo.png", "img", "load_edit_albedo", False),
    ("edit_normal", "%d_edit_normal.png", "img", "load_edit_normal", False),
    ("edit_roughness", "%d_edit_roughness.png", "img", "load_edit_roughness", True),
    ("edit_irradiance", "%d_edit_irradiance.png", "img", "load_edit_irradiance", False),
    ("edit_depth", "%d_edit_depth.npy", "npy", "load_edit_depth", False),
]
# keys get_resized_normal_albedo hands to render_decomp as gt_values (dataset_interface.py:99-160)
_GT_KEYS = [("albedo", "load_albedo"), ("normal", "load_normal"), ("irradiance", "load_irradiance"),
            ("roughness", "load_roughness"), ("depth", "load_depth"),
            ("edit_intrinsic_mask", "load_edit_intrinsic_mask"), ("edit_albedo", "load_edit_albedo"),
            ("edit_normal", "load_edit_normal"), ("edit_depth", "load_edit_depth"),
            ("edit_roughness", "load_edit_roughness"), ("edit_irradiance", "load_edit_irradiance"),
            ("object_insert_mask", "object_insert"), ("object_insert_depth", "object_insert"),
            ("object_insert_normal", "object_insert")]


class NerfDataset:
    """Shared part of the readers (dataset_interface.py:12-300): flags, bulk load, tensors, camera matrix, gt rows."""

    def __init__(self, name, basedir, **kwargs):
        g = kwargs.get
        self.name = name
        self.basedir = basedir
        self.scene_name = basedir.split("/")[-1]
        self.scale = g("image_scale", 1)
        self.split = g("split", "train")
        self.coarse_radiance_number = g("coarse_radiance_number")
        self.near, self.far = g("near_plane", 1), g("far_plane", 10)           # dataset_interface.py:64-65
        self.editing_idx = g("editing_idx", None)
        self.load_image = g("load_image", True)
        for f in ("load_normal", "load_albedo", "load_roughness", "load_depth", "load_diffuse_specular", "load_irradiance",
                  "load_priors", "load_edit_intrinsic_mask", "load_edit_albedo", "load_edit_normal", "load_edit_roughness",
                  "load_edit_irradiance", "load_edit_depth", "object_insert"):
            setattr(self, f, bool(g(f, False)))
        self.prior_type = g("prior_type", "bell")
        self.prior_irradiance_mean = 0.7                                           # dataset_interface.py:44 (a Mitsuba scene reads it from avg_irradiance.json)
        self.coarse_resize_scale = 4                                               # :63
        self.prefiltered_images = []
        self.full_data_loaded = False
        self._lists = {}
        self.poses = []

    def _set_size(self, original_width, original_height):
        self.original_width, self.original_height = original_width, original_height
        self.height = int(self.original_height * self.scale)
        self.width = int(self.original_width * self.scale)
        self.focal = .5 * self.width / np.tan(0.5 * self.camera_angle_x)

    def load_all_data(self, num_of_workers=1, editing_idx=None):
        """dataset_interface.py:206-254 (sequential: the reference's DataLoader only parallelises I/O)."""
        if self.full_data_loaded:
            return
        for i in range(len(self)):
            for k, v in self[i].items():
                if k == "pose":
                    self.poses.append(v)
                else:
                    self._lists.setdefault(k, []).append(v)
        self.full_data_loaded = True

    def to_tensor(self, device):
        """Stack per-view arrays into [n_views, H, W, C] tensors on `device` (dataset_interface.py:256-300)."""
        import torch
        put = lambda lst: torch.from_numpy(np.stack(lst, 0)).to(device)
        if self.poses is not None and len(self.poses) > 0:
            self.poses = put(self.poses)
        self._lists = {k: put(v) for k, v in self._lists.items()}
        self._device = device      # the prefiltered radiance targets rgb_1.. (:259-265) are built on first use: only a training step reads them,
        self._prefiltered = None   # and test.py's flow (coarse_radiance_number = 3 on any image size) must not pay or fail for them

    @property
    def prefiltered_images(self):
        if getattr(self, "_prefiltered", None) is None:
            if "image" not in self._lists or len(self._lists["image"]) == 0:
                return []
            self._prefiltered = [self.get_coarse_images(i + 1).to(getattr(self, "_device", "cpu")) for i in range(self.coarse_radiance_number or 0)]
        return self._prefiltered

    @prefiltered_images.setter
    def prefiltered_images(self, value):
        self._prefiltered = value or None

    def get_coarse_images(self, level):
        """dataset_interface.py:162-176: every image reduced `level` times by coarse_resize_scale (integer division of the unscaled size) and
        brought back to (height, width), both with torchvision's antialiased bilinear Resize — the targets of the coarse radiance heads."""
        import torch
        import torch.nn.functional as F
        imgs = self._lists["image"]
        sh, sw = int(self.height / self.scale), int(self.width / self.scale)
        for _ in range(level):
            sh, sw = sh // self.coarse_resize_scale, sw // self.coarse_resize_scale
        out = []
        for i in range(len(imgs)):
            t = torch.as_tensor(imgs[i]).permute(2, 0, 1)[None]
            t = F.interpolate(t, size=(sh, sw), mode="bilinear", antialias=True, align_corners=False)
            t = F.interpolate(t, size=(self.height, self.width), mode="bilinear", antialias=True, align_corners=False)
            out.append(t[0].permute(1, 2, 0))
        return torch.stack(out, 0)

    def get_info(self, image_index, u, v):
        """dataset_interface.py:178-197: what a training step reads at pixel(s) (u, v) of one view — rgb, the prefiltered rgb_k, the loaded
        intrinsics, and under load_priors the prior albedo and (one channel) prior irradiance."""
        L = self._lists
        info = {"rgb": L["image"][image_index][v, u, :]}
        for i in range(self.coarse_radiance_number or 0):
            info["rgb_%d" % (i + 1)] = self.prefiltered_images[i][image_index][v, u, :]
        for key, flag, sl in (("albedo", "load_albedo", True), ("normal", "load_normal", True), ("roughness", "load_roughness", False),
                              ("depth", "load_depth", False), ("irradiance", "load_irradiance", True)):
            if getattr(self, flag):
                info[key] = L[key][image_index][v, u, :] if sl else L[key][image_index][v, u]
        if self.load_priors:
            info["prior_albedo"] = L["prior_albedo"][image_index][v, u, :]
            info["prior_irradiance"] = L["prior_irradiance"][image_index][v, u, 0]
        return info

    @property
    def images(self):
        return self._lists.get("image", [])

    def get_focal_matrix(self):
        return np.array([[self.focal, 0, 0.5 * self.width], [0, self.focal, 0.5 * self.height], [0, 0, 1]]).astype(np.float32)

    def get_near_far_plane(self):
        return {"near": self.near, "far": self.far}

    def get_resized_normal_albedo(self, resize_factor, i):
        """gt_values of view i (dataset_interface.py:99-160).  At resize_factor 1 (every shipped config;
        render_decomp_path also passes 0 when called without one) torchvision's Resize to the image's
        own size returns its input, so the maps go through unchanged; other factors use the same
        antialiased bilinear filter (`torch.nn.functional.interpolate(..., antialias=True)`)."""
        out = {}
        for key, flag in _GT_KEYS:
            if not getattr(self, flag) or key not in self._lists:
                continue
            x = self._lists[key][i]
            if resize_factor not in (0, 1):
                import torch
                t = torch.as_tensor(x).permute(2, 0, 1)[None]
                t = torch.nn.functional.interpolate(t, size=(self.height // resize_factor, self.width // resize_factor),
                                                    mode="bilinear", antialias=True, align_corners=False)
                x = t[0].permute(1, 2, 0)
            out[key] = x
        if self.load_priors and "prior_albedo" in self._lists:
            # dataset_interface.py:121-125 permutes the resized [C, h, w] priors with (2, 0, 1) where every other map uses (1, 2, 0): they come
            # out as [w, C, h].  Nothing on the render path reads them; mirrored so that the dict is the reference's.
            import torch
            import torch.nn.functional as F
            for key in ("prior_albedo", "prior_irradiance"):
                t = torch.as_tensor(self._lists[key][i]).permute(2, 0, 1)
                if resize_factor not in (0, 1):
                    t = F.interpolate(t[None], size=(self.height // resize_factor, self.width // resize_factor), mode="bilinear", antialias=True,
                                      align_corners=False)[0]
                out[key] = t.permute(2, 0, 1)
        return out

    def __str__(self):
        return "\n".join(["[Dataset]", "\t- type : %s" % self.name, "\t- split : %s" % self.split,
                          "\t- scale : %s" % str(self.scale),
                          "\t- size (raw) : %d x %d" % (self.original_width, self.original_height),
                          "\t- size : %d x %d" % (self.width, self.height), "\t- image number : %d" % len(self)])


class MitsubaDataset(NerfDataset):
    def __init__(self, basedir, **kwargs):
        super().__init__("mitsuba", basedir, **kwargs)
        g = kwargs.get
        if g("load_depth_range_from_file", False):                                # dataset_mitsuba.py:12-16
            with open(os.path.join(basedir, "min_max_depth.json")) as fp:
                f = json.load(fp)
            self.near, self.far = f["min_depth"] * 0.9, f["max_depth"] * 1.1
        if self.load_priors:                                                       # :18-21
            with open(os.path.join(basedir, "avg_irradiance.json")) as fp:
                self.prior_irradiance_mean = json.load(fp)["mean_" + self.prior_type]
        with open(os.path.join(basedir, "transforms_{}.json".format(self.split))) as fp:
            self.meta = json.load(fp)
        self.skip = 1 if self.split == "train" else g("skip", 1)
        self.camera_angle_x = float(self.meta["frames"][0]["fov_degree"]) / 180.0 * math.pi
        from PIL import Image
        with Image.open(os.path.join(basedir, "train/1.png")) as im:
            self._set_size(*im.size)

    def __len__(self):
        return len(self.meta["frames"][::self.skip]) if self.editing_idx is None else 1

    def __getitem__(self, index):
        if not 0 <= index < len(self):
            raise IndexError(index)
        n = (self.skip * index + 1) if self.editing_idx is None else self.editing_idx
        frame = self.meta["frames"][self.editing_idx - 1] if self.editing_idx is not None else self.meta["frames"][::self.skip][index]
        d = os.path.join(self.basedir, self.split)
        sample = {}

        def read(table):
            for key, pat, kind, flag, ch0 in table:
                if getattr(self, flag):
                    p = os.path.join(d, pat % n)
                    if kind == "img":
                        a = load_image_from_path(p, self.scale)
                        sample[key] = a[..., 0:1] if ch0 else a
                    else:
                        sample[key] = load_numpy_from_path(p, self.scale)[..., None]

        read(_PER_VIEW)
        if self.load_priors:                                                       # :66-67, :99-101: <n>_<prior_type>_r.png / _s.png
            sample["prior_albedo"] = load_image_from_path(os.path.join(d, "%d_%s_r.png" % (n, self.prior_type)), self.scale)
            sample["prior_irradiance"] = load_image_from_path(os.path.join(d, "%d_%s_s.png" % (n, self.prior_type)), self.scale)
        if self.load_edit_intrinsic_mask:                                          # dataset_mitsuba.py:105-117
            sample["edit_intrinsic_mask"] = load_image_from_path(os.path.join(d, "%d_edit_intrinsic_mask.png" % n), self.scale)
            read(_EDIT)
        if self.object_insert:                                                     # :119-122
            sample["object_insert_mask"] = load_image_from_path(os.path.join(d, "%d_insert_mask.png" % n), self.scale)
            sample["object_insert_normal"] = load_image_from_path(os.path.join(d, "%d_insert_normal.png" % n), self.scale)
            sample["object_insert_depth"] = load_numpy_from_path(os.path.join(d, "%d_insert_depth.npy" % n), self.scale)[..., None]
        pose = np.array(frame["transform"]).astype(np.float32)
        pose[:3, 0] *= -1                                                          # Mitsuba: camera forward is +Z (:128-130)
        pose[:3, 2] *= -1
        sample["pose"] = pose
        return sample


class ColmapDataset(NerfDataset):
    """Real scenes reconstructed with COLMAP (configs/real/*, dataset_colmap.py:7-69)."""

    def __init__(self, basedir, **kwargs):
        super().__init__("colmap", basedir, **kwargs)
        with open(os.path.join(basedir, "transforms.json")) as fp:
            self.meta = json.load(fp)
        self.skip = 1 if self.split == "train" else kwargs.get("skip", 1)
        self.camera_angle_x = float(self.meta["camera_angle_x"])
        self._set_size(self.meta["w"], self.meta["h"])
        n = len(self.meta["frames"])
        if self.split == "train":                                                   # frames 8i+1 .. 8i+7
            idx = [8 * i + j + 1 for i in range(n // 8 + 1) for j in range(7)]
        elif self.split in ("val", "test"):                                        # frames 8i
            idx = [8 * i for i in range(n // 8 + 1)]
        else:
            raise AttributeError("split %r has no index list" % self.split)       # the reference never sets index_list
        self.index_list = [i for i in idx if i < n]

    def __len__(self):
        return len(self.index_list)

    def __getitem__(self, index):
        if not 0 <= index < len(self):
            raise IndexError(index)
        frame = self.meta["frames"][::self.skip][self.index_list[index]]
        sample = {}
        name = os.path.split(frame["file_path"])[-1]
        if self.load_image:
            sample["image"] = load_image_from_path(os.path.join(self.basedir, "images", name), self.scale)
        if self.load_priors:                                                       # <name>_<prior_type>_r.png / _s.png beside the image (:49-60)
            sample["prior_albedo"] = load_image_from_path(os.path.join(self.basedir, "images", name[:-4] + "_%s_r.png" % self.prior_type), self.scale)
            sample["prior_irradiance"] = load_image_from_path(os.path.join(self.basedir, "images", name[:-4] + "_%s_s.png" % self.prior_type), self.scale)
        sample["pose"] = np.array(frame["transform_matrix"]).astype(np.float32)   # no axis flips (:62-63)
        return sample


class MitsubaEvalDataset(NerfDataset):
    """Rendered image sets for the metric scripts (dataset_mitsuba_eval.py:20-59); there are no cameras."""
    _KINDS = ("image", "diffuse", "specular", "irradiance", "roughness", "albedo")

    def __init__(self, basedir, **kwargs):
        super().__init__("mitsuba_eval", basedir, **kwargs)
        import glob
        self.load_diffuse_specular = True
        self.file_n = len(glob.glob(os.path.join(basedir, "specular_*.png")))

    def __len__(self):
        return self.file_n

    def __getitem__(self, index):
        if not 0 <= index < len(self):
            raise IndexError(index)
        sample = {}
        for key in self._KINDS:
            stem = "rgb" if key == "image" else key
            sample[key] = load_image_from_path(os.path.join(self.basedir, "%s_%03d.png" % (stem, index)), 1)
        if "monte_carlo" in self.basedir:                                          # :52-53
            sample["albedo"] = np.power(sample["albedo"], 1 / 2.2)
        return sample


def load_dataset(dataset_type, basedir, **kwargs):
    """dataset_interface.py:316-331."""
    if dataset_type == "mitsuba":
        return MitsubaDataset(basedir, **kwargs)
    if dataset_type == "mitsuba_eval":
        return MitsubaEvalDataset(basedir, **kwargs)
    if dataset_type == "colmap":
        return ColmapDataset(basedir, **kwargs)
    raise ValueError("Unknown dataset type: %s" % dataset_type)
