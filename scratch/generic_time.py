"""Rays/s of the layer-by-layer path (csrc/generic_mlp.hip) on seeded random-init networks of architectures outside the built one."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg; _pkg.load()
import torch
import bench as Bn
from ibl_nerf_amd import checkpoint as ck, renderer as R
K, c2w = Bn.camera()
for arch in ((9, 256, 10, 4), (10, 384, 12, 5), (8, 512, 10, 4)):
    r = R.Renderer(64, 128)
    r.load_weights(0, ck.synthetic_arch_state_dict(1, arch)); r.load_weights(1, ck.synthetic_arch_state_dict(2, arch)); r.load_lut(Bn.load_lut())
    ro, rd = r.get_rays(800, 800, K, c2w, 380, 40)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    r.render_rays(ro[:4096], rd[:4096], 0.5, 8.0); torch.cuda.synchronize()
    t0 = time.perf_counter(); r.render_rays(ro, rd, 0.5, 8.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fl = r.last_executed_flops()
    print("IBLNeRF%s: %d rays in %.2f s = %.0f rays/s, %.1f TFLOP/s of fp32 MFMA (peak 157)" % (arch, ro.shape[0], dt, ro.shape[0] / dt, fl / dt / 1e12), flush=True)
