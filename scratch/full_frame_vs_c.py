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
routing = tuple(a[8:] for a in sys.argv if a.startswith("routing="))      # e.g. routing=fine_main_precise
r = pkg.Renderer(64, 128, query_routing=routing or 0)
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
print("checkpoint %s, config %s, query_routing %s" % (which, config, routing or "default"))
print("rays %d   HIP %.2f s (%.0f rays/s)   C restatement %.1f s (%.0f rays/s, %d threads, %s)" % (n, t_gpu, n / t_gpu, t_cpu, n / t_cpu, OC.usable_cpus(), OC.isa()))
out = {}
for k in ref:
    a, b = got[k].astype(np.float64).reshape(n, -1), ref[k].astype(np.float64).reshape(n, -1)
    e = np.abs(a - b).max(-1) / max(np.abs(b).max(), 1e-30)
    out[k] = dict(p50=float(np.percentile(e, 50)), p99=float(np.percentile(e, 99)), p999=float(np.percentile(e, 99.9)), p9999=float(np.percentile(e, 99.99)), max=float(e.max()),
                  over_1e3=int((e > 1e-3).sum()), over_2e4=int((e > 2e-4).sum()))
    print("%-36s p50 %.1e  p99 %.1e  p99.9 %.1e  p99.99 %.1e  max %.1e   rays > 1e-3: %d   > 2e-4: %d" % (k, out[k]["p50"], out[k]["p99"], out[k]["p999"], out[k]["p9999"], out[k]["max"], out[k]["over_1e3"], out[k]["over_2e4"]))
if len(sys.argv) > 4 and sys.argv[4] == "param":
    # the same frame by the C restatement with both checkpoints rounded to 22-bit mantissas (f16(w) + f16(w - f16(w)): how the three-product kernels hold the weights)
    def r22(sd):
        o = {}
        for k_, v in sd.items():
            hi = v.astype(np.float16).astype(np.float32)
            o[k_] = (hi + (v - hi).astype(np.float16).astype(np.float32)).astype(np.float32)
        return o
    t0 = time.time()
    ref22 = OC.render_rays(r22(sdc), r22(sdf), ro.cpu().numpy(), rd.cpu().numpy(), 0.5, 8.0, lut, gt=gt, edit=edit)
    print("\nC restatement with 22-bit parameters against the C restatement with the checkpoint as it is (%.0f s); HIP against the exact one in brackets:" % (time.time() - t0))
    for k in ("depth_map", "albedo_map", "roughness_map", "irradiance_map", "radiance_map", "weights", "target_normal_map", "n_dot_v_map", "depth_map0", "target_normal_map0", "weights0"):
        b = ref[k].astype(np.float64).reshape(n, -1)
        sc = max(np.abs(b).max(), 1e-30)
        ep = np.abs(ref22[k].astype(np.float64).reshape(n, -1) - b).max(-1) / sc
        eh = np.abs(got[k].astype(np.float64).reshape(n, -1) - b).max(-1) / sc
        both = int(((ep > 1e-3) & (eh > 1e-3)).sum())
        print("%-22s p99 %.1e (%.1e)  p99.9 %.1e (%.1e)  p99.99 %.1e (%.1e)  rays > 1e-3: %d (%d; %d in both)   HIP rays beyond max(1e-3, 8x the parameter sensitivity): %d" % (
            k, np.percentile(ep, 99), np.percentile(eh, 99), np.percentile(ep, 99.9), np.percentile(eh, 99.9), np.percentile(ep, 99.99), np.percentile(eh, 99.99),
            int((ep > 1e-3).sum()), int((eh > 1e-3).sum()), both, int((eh > np.maximum(1e-3, 8 * ep)).sum())))
mse = float(np.mean((got["color_map"].astype(np.float64) - ref["color_map"]) ** 2))
print("color PSNR %.1f dB" % (10 * np.log10(1 / max(mse, 1e-30))))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(rays=n, t_gpu=t_gpu, t_cpu=t_cpu, threads=OC.usable_cpus(), maps=out), open(os.path.join(ROOT, "gpurun_out", "full_frame_vs_c_%s_%s.json" % (which, config)), "w"), indent=1)
