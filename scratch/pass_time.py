"""Per-ray kernels of a launch, timed by the bench's own profile hooks is too coarse: rocprofv3 --kernel-trace --stats around one frame.
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o f -- python3 scratch/pass_time.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _pkg; pkg = _pkg.load()
from conftest import load_lut_rgb
f = np.load(os.path.join(ROOT, "tests", "golden", "fitted_ckpt.npz"))
ck = pkg.checkpoint
r = pkg.Renderer(64, 128)
r.load_weights(0, ck.blob_to_state_dict(f["coarse"])); r.load_weights(1, ck.blob_to_state_dict(f["fine"])); r.load_lut(load_lut_rgb())
fl = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
ro, rd = r.get_rays(800, 800, K, np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32))
for _ in range(2):
    m = r.render_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), 0.5, 8.0)
torch.cuda.synchronize()
print("sum", float(m["color_map"].double().sum()), float(m["weights"].double().sum()), float(m["target_normal_map"].double().sum()))
