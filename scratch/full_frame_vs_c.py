"""Whole 800x800 frame (BASELINE configs[1], fitted checkpoint, default precision mode) on the HIP path against the C restatement of the
reference path (oracle/csrc, all host threads): per-ray error distribution of every map over all 640 000 rays.
    python scratch/full_frame_vs_c.py [n_rows] [fitted|fitted2] [plain|edit|insert]      (defaults: 800 = the whole frame, fitted, plain)
edit / insert: BASELINE configs 4 / 5 (the shipped kwargs on tests/frame_overrides.py's analytic images)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import _pkg; pkg = _pkg.load()
import torch
import iblnerf_cpu as OC
from conftest import load_lut_rgb

import frame_overrides as FO
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 800
which = sys.argv[2] if len(sys.argv) > 2 else "fitted"
config = sys.argv[3] if len(sys.argv) > 3 else "plain"
f = np.load(os.path.join(ROOT, "tests", "golden", which + "_ckpt.npz"))
ck = pkg.checkpoint
sdc, sdf = ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])
lut = load_lut_rgb()
fl = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
r = pkg.Renderer(64, 128)
r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
ro, rd = r.get_rays(800, 800, K, c2w)
r0 = (800 - rows) // 2
ro, rd = ro.reshape(800, 800, 3)[r0:r0 + rows].reshape(-1, 3), rd.reshape(800, 800, 3)[r0:r0 + rows].reshape(-1, 3)
pix = np.arange(r0 * 800, (r0 + rows) * 800)
gt, edit = {}, {}
if config == "edit":
    gt, edit = FO.edit_rows(pix), dict(FO.EDIT_CFG4)
elif config == "insert":
    gt, edit = FO.insert_rows(pix), dict(FO.INSERT_CFG5)
gt_d = {k: torch.from_numpy(v).cuda() for k, v in gt.items()}
torch.cuda.synchronize(); t0 = time.time()
got = r.render_rays(ro, rd, 0.5, 8.0, gt_d, **edit)
torch.cuda.synchronize(); t_gpu = time.time() - t0
got = {k: v.cpu().numpy() for k, v in got.items()}
t0 = time.time()
ref = OC.render_rays(sdc, sdf, ro.cpu().numpy(), rd.cpu().numpy(), 0.5, 8.0, lut, gt=gt, edit=edit)
t_cpu = time.time() - t0
n = ro.shape[0]
print("checkpoint %s, config %s" % (which, config))
print("rays %d   HIP %.2f s (%.0f rays/s)   C restatement %.1f s (%.0f rays/s, %d threads, %s)" % (n, t_gpu, n / t_gpu, t_cpu, n / t_cpu, OC.usable_cpus(), OC.isa()))
out = {}
for k in ref:
    a, b = got[k].astype(np.float64).reshape(n, -1), ref[k].astype(np.float64).reshape(n, -1)
    e = np.abs(a - b).max(-1) / max(np.abs(b).max(), 1e-30)
    out[k] = dict(p50=float(np.percentile(e, 50)), p99=float(np.percentile(e, 99)), p999=float(np.percentile(e, 99.9)), p9999=float(np.percentile(e, 99.99)), max=float(e.max()),
                  over_1e3=int((e > 1e-3).sum()), over_2e4=int((e > 2e-4).sum()))
    print("%-36s p50 %.1e  p99 %.1e  p99.9 %.1e  p99.99 %.1e  max %.1e   rays > 1e-3: %d   > 2e-4: %d" % (k, out[k]["p50"], out[k]["p99"], out[k]["p999"], out[k]["p9999"], out[k]["max"], out[k]["over_1e3"], out[k]["over_2e4"]))
mse = float(np.mean((got["color_map"].astype(np.float64) - ref["color_map"]) ** 2))
print("color PSNR %.1f dB" % (10 * np.log10(1 / max(mse, 1e-30))))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(rays=n, t_gpu=t_gpu, t_cpu=t_cpu, threads=OC.usable_cpus(), maps=out), open(os.path.join(ROOT, "gpurun_out", "full_frame_vs_c_%s_%s.json" % (which, config)), "w"), indent=1)
