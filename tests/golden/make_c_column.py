#!/usr/bin/env python3
"""The YARDSTICK column of the launch-scale parity reports (VERDICT r4 next-1 a): what an actual fp32 implementation of the path attains against the
reference's own float32 render on the very rays of each launch-scale fixture.

    python tests/golden/make_c_column.py [fixture ...]        (default: all seven; needs only this repo — the fixtures and oracle/csrc)

For every fixture tests/golden/<name>.npz the C restatement (oracle/csrc: plain fp32, its own GEMM order, no knowledge of the HIP kernels) renders the fixture's
rays; per map the per-ray relative error against the reference's `out__<map>` (max over the map's channels over the map's largest value — the metric of
tests/test_gpu_launch_scale.py per_ray) is stored as float16 of 2^14 x the value in tests/golden/c_restatement_column.npz (`<fixture>/<map>` [n]; `weights`: the
fixture's subsampled rows), beside a JSON summary (rays above 1e-3, 99.9th percentile, worst ray).  The GPU tests hold the HIP path to THIS: no more rays above the
north-star 1e-3 than the fp32 restatement has, plus a small allowance — instead of the number of rays the reference's own sensitivity yardsticks flag.
Data only: arrays of errors; nothing of the reference travels.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg  # noqa: E402
_pkg.load()
from conftest import load_golden, load_lut_rgb  # noqa: E402
import iblnerf_cpu as OC  # noqa: E402

FIXTURES = ["fitted_launch16k", "fitted_edit_cfg4", "fitted_insert_cfg5", "fitted_posed4k", "fitted2_launch4k", "fitted2_posed4k", "fitted3_launch4k", "fitted3_posed4k",
            "fitted_launch64k"]
MAPS = ["depth_map", "albedo_map", "roughness_map", "irradiance_map", "target_normal_map", "n_dot_v_map", "weights", "prefiltered_reflected_map", "color_map",
        "specular_map", "reflected_radiance_map", "reflected_coarse_radiance_map_1", "reflected_coarse_radiance_map_2", "reflected_coarse_radiance_map_3", "diffuse_map",
        "radiance_map"]
MAPS = MAPS + [k + "0" for k in MAPS]
SCALE = 2.0 ** 14
OUT = os.path.join(HERE, "c_restatement_column.npz")
OUT_JSON = os.path.join(HERE, "c_restatement_column.json")


def per_ray(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(float(np.nanmax(np.abs(ref))), 1e-30)
    with np.errstate(invalid="ignore"):
        return np.nanmax(np.abs(np.asarray(got, dtype=np.float64).reshape(ref.shape) - ref).reshape(len(ref), -1), -1) / scale


def fixture_rays(g):
    if "rays_o" in g.files:
        return g["rays_o"], g["rays_d"]
    f = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))          # the compact 65 536-ray fixture: rays from the pixel ids (bench view)
    K = np.array([[f, 0, 400], [0, f, 400], [0, 0, 1]], dtype=np.float32)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    ro, rd = OC.get_rays(800, 800, K, c2w)
    return ro.reshape(-1, 3)[g["pix"]], rd.reshape(-1, 3)[g["pix"]]


def column(name, lut):
    g, sdc, sdf, gt, edit = load_golden(name)
    ro, rd = fixture_rays(g)
    res = OC.render_rays(sdc, sdf, ro, rd, float(g["near"]) if "near" in g.files else 0.5, float(g["far"]) if "far" in g.files else 8.0, lut, 64, 128, gt, edit)
    we = int(g["weights_every"]) if "weights_every" in g.files else 1
    arrays, summary = {}, {}
    for k in MAPS:
        if "out__" + k not in g.files or k not in res:
            continue
        got = res[k][::we] if k.startswith("weights") else res[k]
        e = per_ray(got, g["out__" + k])
        arrays["%s/%s" % (name, k)] = np.minimum(e * SCALE, 65504.0).astype(np.float16)
        summary[k] = {"rays": int(len(e)), "above_1e-3": int((e > 1e-3).sum()), "p999": float(np.nanpercentile(e, 99.9)), "max": float(np.nanmax(e))}
    return arrays, summary


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or [n for n in FIXTURES if os.path.exists(os.path.join(HERE, n + ".npz"))]
    lut = load_lut_rgb()
    arrays = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    summary = json.load(open(OUT_JSON)) if os.path.exists(OUT_JSON) else {}
    for n in names:
        a, s = column(n, lut)
        arrays = {k: v for k, v in arrays.items() if not k.startswith(n + "/")}
        arrays.update(a)
        summary[n] = s
        print(n, json.dumps({k: (v["above_1e-3"], "%.1e" % v["p999"], "%.1e" % v["max"]) for k, v in s.items()}), flush=True)
    np.savez_compressed(OUT, **arrays)
    json.dump({"what": "C restatement (oracle/csrc, fp32) vs the reference's float32 render on each launch-scale fixture's rays: per map rays above 1e-3, p99.9, worst ray",
               "isa": OC.isa(), **{k: v for k, v in summary.items() if k not in ("what", "isa")}}, open(OUT_JSON, "w"), indent=1)
    print("wrote", OUT, "%.2f MB" % (os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main()
