#!/bin/bash
# samples power / clocks with rocm-smi while the TRUNK benchmark loops (run ON the GPU box)
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
t = time.time()
while time.time() - t < 9:
    r.network_query(pts, None, 0)
torch.cuda.synchronize()
PY
wait
cat gpurun_out/power_samples.txt | head -30
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
