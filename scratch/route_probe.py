"""Round 5, first GPU contact of the route table / predicted offsets / tripwire: the same rays under the default route, round 4's offsets (offsets_estimate_all) and
every sample refined (coarse_density_all_points); route text, slot units, selection counts, differences between the routes, frame times."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _pkg
_pkg.load()
from conftest import load_lut_rgb
from ibl_nerf_amd import renderer as R, checkpoint as ck, dist as D

lut = load_lut_rgb()
fl = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
out = {}
for which in (sys.argv[1:] or ["fitted", "fitted2"]):
    f = np.load(os.path.join(ROOT, "tests", "golden", which + "_ckpt.npz"))
    res, rep = {}, {}
    for label, routing in (("default", ()), ("estimate_all", ("offsets_estimate_all",)), ("all_points", ("coarse_density_all_points",))):
        r = R.Renderer(64, 128, mlp_precision="f16x3_mxfp6x", query_routing=routing)
        r.load_weights(0, ck.blob_to_state_dict(f["coarse"])); r.load_weights(1, ck.blob_to_state_dict(f["fine"])); r.load_lut(lut)
        D.decide_on_frame(r, 800, 800, K, c2w, 0.5, 8.0)
        ro, rd = r.get_rays(800, 800, K, c2w)
        ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        if label == "default":
            print(which, "route:", r.route)
            print(r.describe_route(), flush=True)
        idx = torch.as_tensor(np.sort(np.random.RandomState(3).permutation(640000)[:65536]), device=ro.device)
        m = r.render_rays(ro[idx].contiguous(), rd[idx].contiguous(), 0.5, 8.0)
        torch.cuda.synchronize()
        res[label] = {k: v.cpu() for k, v in m.items()}
        sel = r.last_selection()
        slots = r.last_slot_units()
        t0 = time.perf_counter()
        for _ in range(2):
            r.render_rays(ro, rd, 0.5, 8.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        rep[label] = dict(selection=sel, slot_units_per_ray=slots / 65536, executed_gflop_per_ray=r.last_executed_flops() / 640000 / 1e9, frame_ms=1e3 * dt, rays_per_s=640000 / dt,
                          trips=r.trips, route=r.get_route())
        print(which, label, json.dumps(rep[label]), flush=True)
        del r
    for other in ("estimate_all", "all_points"):
        d = {}
        for k in res["default"]:
            a, b = res["default"][k].double(), res[other][k].double()
            e = (a - b).abs().nan_to_num(0.0)
            d[k] = (float(e.max() / b.abs().nan_to_num(0.0).max().clamp_min(1e-30)), int((e.reshape(e.shape[0], -1).amax(-1) > 0).sum()))
        print(which, "default vs", other, {k: ("%.1e" % v[0], v[1]) for k, v in d.items() if v[1]}, flush=True)
    out[which] = rep
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "route_probe.json"), "w"), indent=1)
