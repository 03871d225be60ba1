#!/bin/bash
# samples power / clocks with rocm-smi while the TRUNK benchmark loops (run ON the GPU box; IBLNERF_GRID / IBLNERF_LIB select CUs / build)
( for i in $(seq 1 24); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk|mclk" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/power_samples.txt &
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
import _pkg; _pkg.load()
from ibl_nerf_amd import renderer as R, checkpoint as ck
sd = ck.synthetic_state_dict(0)
pts = torch.rand((65536, 128, 3), device='cuda') * 8 - 4
r = R.Renderer(64, 0, max_rays_per_launch=64); r.load_weights(0, sd)
import time
torch.cuda.synchronize()
t, n = time.time(), 0
while time.time() - t < 9:
    r.network_query(pts, None, 0)
    torch.cuda.synchronize()
    n += 1
print("launch %.2f ms" % ((time.time() - t) / n * 1e3))
PY
wait
sed -n 8,10p gpurun_out/power_samples.txt | sed 's/.*sclk clock level: 1: (\([0-9]*\)Mhz).*Power (W): \([0-9.]*\).*/sclk \1 MHz  \2 W/'
