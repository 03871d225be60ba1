"""Per-view render + export loop: mirror of `render_decomp_path`
(src/nerf_models/ibl_nerf_renderer.py:819-910) and of the host-side helpers it uses
(`to8b` nerf_renderer_helper.py:10, `depth_to_normal_image_space`
utils/depth_to_normal_utils.py:9-46).  The render itself is the HIP path (`renderer.render_decomp`);
the export mapping is numpy on the host, exactly like the reference's `.cpu().numpy()` branch.
PNG files are written with PIL (imageio is what the reference uses; the bytes on disk are the same
8-bit arrays)."""
from __future__ import annotations

import os

import numpy as np

from . import renderer as R


def to8b(x):
    """nerf_renderer_helper.py:10 — truncation, not rounding."""
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def _normalize(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def depth_to_normal_image_space(depth_map, pose, K):
    """utils/depth_to_normal_utils.py:9-46: back-project depth along NORMALISED camera rays, central
    differences with edge padding, normal = normalize(cross(vb, va)).  numpy, host side."""
    depth = np.asarray(depth_map, dtype=np.float32)
    pose = np.asarray(pose, dtype=np.float32)
    K = np.asarray(K, dtype=np.float32)
    H, W = depth.shape
    i = np.broadcast_to(np.arange(W, dtype=np.float32)[None, :], (H, W))
    j = np.broadcast_to(np.arange(H, dtype=np.float32)[:, None], (H, W))
    dirs = np.stack([(i - K[0, 2]) / K[0, 0], -(j - K[1, 2]) / K[1, 1], -np.ones_like(i)], -1).astype(np.float32)
    dirs = dirs / np.maximum(np.linalg.norm(dirs, axis=-1, keepdims=True), 1e-12).astype(np.float32)
    rays_d = np.sum(dirs[..., None, :] * pose[:3, :3], -1, dtype=np.float32)
    position = (pose[:3, -1] + rays_d * depth[..., None]).astype(np.float32)
    padded = np.pad(position, ((1, 1), (1, 1), (0, 0)), "edge")
    va = _normalize(padded[1:-1, 2:, :] - padded[1:-1, :-2, :])
    vb = _normalize(padded[2:, 1:-1, :] - padded[:-2, 1:-1, :])
    return _normalize(np.cross(vb, va, axis=-1)).astype(np.float32)


# (result key, output name) in the order render_decomp_path appends them (ibl_nerf_renderer.py:870-900)
def export_plan(coarse_radiance_number=3):
    plan = [("color_map", "rgb"), ("radiance_map", "radiance")]
    for k in range(coarse_radiance_number):
        plan += [("radiance_map_%d" % (k + 1), "radiance_%d" % (k + 1)),
                 ("reflected_coarse_radiance_map_%d" % (k + 1), "reflected_coarse_radiance_%d" % (k + 1))]
    plan += [("irradiance_map", "irradiance"), ("max_irradiance_map", "max_irradiance"),
             ("min_irradiance_map", "min_irradiance"), ("albedo_map", "albedo"),
             ("reflected_radiance_map", "reflected_radiance"), ("prefiltered_reflected_map", "prefiltered_reflected"),
             ("roughness_map", "roughness"), ("specular_map", "specular"), ("diffuse_map", "diffuse"),
             ("n_dot_v_map", "n_dot_v"), ("inferred_normal_map", "inferred_normal_map"),
             ("target_normal_map", "target_normal_map"), ("target_binormal_map", "target_binormal_map"),
             ("target_tangent_map", "target_tangent_map"), ("visibility_average_map", "visibility_average_map"),
             ("inferred_depth_map", "inferred_disp"), ("disp_map", "disp"), ("depth_map", "depth"),
             ("target_depth_map", "target_depth")]
    return plan


def map_for_export(key_name, out_name, image, far):
    """append_result's value mapping (ibl_nerf_renderer.py:848-853): normals -> (n+1)/2, any key
    containing 'depth' -> 1 / max(1e-10, depth / (0.1 far))."""
    image = np.asarray(image, dtype=np.float32)
    if "normal" in out_name or "tangent" in out_name:
        return ((image + np.float32(1)) * np.float32(0.5)).astype(np.float32)
    if "depth" in key_name:
        image = (image / np.float32(far * 0.1)).astype(np.float32)
        return (np.float32(1.0) / np.maximum(np.float32(1e-10), image)).astype(np.float32)
    return image


def write_png(path, img8):
    from PIL import Image
    a = np.asarray(img8)
    if a.ndim == 3 and a.shape[-1] == 1:
        a = a[..., 0]
    Image.fromarray(a).save(path)


def render_decomp_path(dataset_test, hwf, K, chunk, render_kwargs, savedir=None, render_factor=0, gt_values=None,
                       render_fn=None, export_workers=8, views=None, **kwargs):
    """Drop-in for ibl_nerf_renderer.py:819-910.  `dataset_test` needs `.poses` (iterable of c2w),
    `.far` and `.get_resized_normal_albedo(render_factor, i)` like the reference's NerfDataset.
    Returns {out_name: array [n_views, H, W, ...]} and writes `<out_name>_{i:03d}.png` to savedir.
    PNG encoding (1.2 s of host time per 800x800 view, against 2.3 s of GPU time) runs on
    `export_workers` threads while the next view renders; every file is on disk when this returns.
    `export_workers=0` writes inline like the reference.  `views` (iterable of view indices, default all)
    restricts the loop to this process's share of the views (`dist.view_indices`); file names keep the
    global view index.  With the default `render_fn` the coarse pass only evaluates density (`coarse_outputs=False`): no
    exported map comes from it and the exported ones do not change by a bit."""
    from concurrent.futures import ThreadPoolExecutor
    if render_fn is None:
        render_fn = R.render_decomp
        # every exported map is a last-pass map (append_result reads un-suffixed keys only, :870-900), and those are
        # bit-identical without the coarse pass's own maps: skip them (a quarter of a view's time) unless the caller says otherwise
        if "coarse_outputs" not in render_kwargs and "coarse_outputs" not in kwargs:
            kwargs = dict(kwargs, coarse_outputs=False)
    pool = ThreadPoolExecutor(max_workers=export_workers) if (savedir is not None and export_workers > 0) else None
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]]).astype(np.float32)   # :832-836 (K argument is rebuilt)
    results = {}

    def finish(key_name, out_name, index, img):
        """append_result's host side (ibl_nerf_renderer.py:846-858): value mapping, 8-bit PNG."""
        img = map_for_export(key_name, out_name, img, dataset_test.far)
        if savedir is not None:
            write_png(os.path.join(savedir, (out_name + "_{:03d}.png").format(index)), to8b(img))
        return img

    def append_result(res_i, key_name, index, out_name):
        if key_name not in res_i or res_i[key_name] is None:
            return
        img = res_i[key_name]
        img = img.detach().cpu().numpy() if hasattr(img, "detach") else np.asarray(img)
        if pool is not None:             # mapping and encoding overlap the next view's render
            results.setdefault(out_name, []).append(pool.submit(finish, key_name, out_name, index, img))
        else:
            results.setdefault(out_name, []).append(finish(key_name, out_name, index, img))

    if savedir is not None:
        os.makedirs(savedir, exist_ok=True)
    plan = export_plan(render_kwargs.get("coarse_radiance_number", 3))
    try:
        for i in (range(len(dataset_test.poses)) if views is None else views):
            c2w = dataset_test.poses[i]
            gt = dataset_test.get_resized_normal_albedo(render_factor, i)
            gt = {k: v.reshape(-1, v.shape[-1]) for k, v in gt.items()}
            c2w34 = c2w[:3, :4]
            res_i = dict(render_fn(H, W, K, chunk=chunk, c2w=c2w34, gt_values=gt, **render_kwargs, **kwargs))
            for key_name, out_name in plan:
                append_result(res_i, key_name, i, out_name)
            if "depth_map" in res_i:
                d = res_i["depth_map"]
                d = d.detach().cpu().numpy() if hasattr(d, "detach") else np.asarray(d)
                c = c2w34.detach().cpu().numpy() if hasattr(c2w34, "detach") else np.asarray(c2w34)
                if pool is not None:
                    results.setdefault("normal_from_depth", []).append(pool.submit(
                        lambda d=d, c=c, i=i: finish("normal_map_from_depth_map", "normal_from_depth", i, depth_to_normal_image_space(d, c, K))))
                else:
                    res_i["normal_map_from_depth_map"] = depth_to_normal_image_space(d, c, K)
                    append_result(res_i, "normal_map_from_depth_map", i, "normal_from_depth")
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
    if pool is not None:                 # .result() re-raises a failed mapping / write
        results = {k: [f.result() for f in v] for k, v in results.items()}
    return {k: np.stack(v, 0) for k, v in results.items()}


def psnr(pred, gt, data_range=1.0):
    """10 log10(range^2 / MSE) — the definition piq.psnr uses in evaluation/calculate_metrics.py:13-33."""
    mse = float(np.mean((np.asarray(pred, np.float64) - np.asarray(gt, np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(data_range ** 2 / mse)


def _gaussian_kernel(size=11, sigma=1.5):
    c = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-(c ** 2) / (2.0 * sigma ** 2))
    k = np.outer(g, g)
    return k / k.sum()


def ssim(pred, gt, data_range=1.0, kernel_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03, downsample=True):
    """Structural similarity of two [H,W,C] (or [H,W]) images as `piq.ssim` computes it with its defaults —
    the call of evaluation/calculate_metrics.py:31 (`piq` is an unpinned entry of the reference's
    requirements.txt and is absent here, so this restates piq's published algorithm: inputs / data_range;
    average-pool by f = max(1, round(min(H,W)/256)); 11x11 Gaussian window, sigma 1.5, 'valid' convolution
    per channel; ssim = mean over window positions of l*cs, then mean over channels)."""
    from scipy.signal import fftconvolve
    x = np.asarray(pred, np.float64) / data_range
    y = np.asarray(gt, np.float64) / data_range
    if x.ndim == 2:
        x, y = x[..., None], y[..., None]
    if x.shape != y.shape:
        raise ValueError("ssim: shapes differ %r vs %r" % (x.shape, y.shape))
    f = max(1, round(min(x.shape[:2]) / 256))
    if f > 1 and downsample:                                  # F.avg_pool2d(kernel_size=f): floor, no padding
        Hc, Wc = (x.shape[0] // f) * f, (x.shape[1] // f) * f
        pool = lambda a: a[:Hc, :Wc].reshape(Hc // f, f, Wc // f, f, -1).mean((1, 3))
        x, y = pool(x), pool(y)
    if min(x.shape[:2]) < kernel_size:
        raise ValueError("ssim: image smaller than the %dx%d window" % (kernel_size, kernel_size))
    k = _gaussian_kernel(kernel_size, kernel_sigma)
    c1, c2 = k1 ** 2, k2 ** 2
    vals = []
    for ch in range(x.shape[-1]):
        a, b = x[..., ch], y[..., ch]
        conv = lambda im: fftconvolve(im, k, mode="valid")
        mu_a, mu_b = conv(a), conv(b)
        s_aa, s_bb, s_ab = conv(a * a) - mu_a ** 2, conv(b * b) - mu_b ** 2, conv(a * b) - mu_a * mu_b
        cs = (2.0 * s_ab + c2) / (s_aa + s_bb + c2)
        ss = (2.0 * mu_a * mu_b + c1) / (mu_a ** 2 + mu_b ** 2 + c1) * cs
        vals.append(ss.mean())
    return float(np.mean(vals))


def calculate_metrics(gt_path, pred_path, target="rgb", n_views=100):
    """evaluation/calculate_metrics.py:10-33, mitsuba branch: `<pred>/<target>_{i:03d}.png` against
    `<gt>/test/{i+1}.png`, 8-bit RGB / 255.  Returns per-view lists of ssim / psnr / mse."""
    from PIL import Image
    load = lambda p: np.asarray(Image.open(p).convert("RGB"), dtype=np.float32) / np.float32(255.0)
    out = {"ssim": [], "psnr": [], "mse": []}
    for i in range(n_views):
        pred = load(os.path.join(pred_path, "%s_%03d.png" % (target, i)))
        gt = load(os.path.join(gt_path, "test", "%d.png" % (i + 1)))
        out["ssim"].append(ssim(pred, gt))
        out["psnr"].append(psnr(pred, gt))
        out["mse"].append(float(np.mean((pred.astype(np.float64) - gt.astype(np.float64)) ** 2)))
    return out
