"""Thread scaling of the C restatement on this host (which thread count should bench.py's cpu_baseline use?)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import iblnerf_cpu as OC
from conftest import load_golden, load_lut_rgb
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a", " affinity:", len(os.sched_getaffinity(0)), " cpu_count:", os.cpu_count(), flush=True)
lut = load_lut_rgb()
g, sdc, sdf, gt, edit = load_golden("fitted_launch16k")
for th in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128, 256]:
    n = min(16380, 12 * th * 4)
    OC.render_rays(sdc, sdf, g["rays_o"][:12 * th], g["rays_d"][:12 * th], 0.5, 8.0, lut, n_threads=th)
    t0 = time.time(); OC.render_rays(sdc, sdf, g["rays_o"][:n], g["rays_d"][:n], 0.5, 8.0, lut, n_threads=th); dt = time.time() - t0
    print("threads %3d  %5d rays  %.2f s  %.0f rays/s  %.1f rays/s/thread" % (th, n, dt, n / dt, n / dt / th), flush=True)
