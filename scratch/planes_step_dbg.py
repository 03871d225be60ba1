"""Round 5: train_step_planes.npz, warmup phase — where does the fine network's positions_linears.1 gradient differ from the reference's?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _pkg; _pkg.load()
from conftest import GOLDEN, rel_linf, load_lut_rgb
import test_gpu_training as TT
import train_loss as TL
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
name = sys.argv[1] if len(sys.argv) > 1 else "train_step_planes"
phase = sys.argv[2] if len(sys.argv) > 2 else "warmup"
G = np.load(os.path.join(GOLDEN, name + ".npz"))
planes = G["near"].ndim > 0 and G["near"].size > 1
nets, kw, K, rays = TT._setup(dict(G, near=np.float32(0), far=np.float32(0)) if planes else G, lut, phase)
if planes:
    kw.update(near=torch.from_numpy(G["near"]).cuda(), far=torch.from_numpy(G["far"]).cuda())
approx = phase != "warmup"
res = R.render_decomp(800, 800, K, chunk=int(G["chunk"]), rays=rays, gt_values={}, approximate_radiance=approx, **kw)
for k in ("z_std", "weights", "weights0", "depth_map", "depth_map0"):
    a, b = res[k].detach().cpu().numpy(), G["%s__out__%s" % (phase, k)]
    e = np.abs(a - b).reshape(len(a), -1).max(-1) / np.abs(b).max()
    print(k, "rel_linf %.2e" % e.max(), "rays above 1e-3:", np.flatnonzero(e > 1e-3), e[e > 1e-3])
loss = TL.total_loss(torch, res, {k[8:]: G[k] for k in G.files if k.startswith("target__")}, approx)
print("loss", float(loss), float(G[phase + "__loss"]))
loss.backward()
worst, zero = TT._grads_against(G, phase, nets)
for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:8]:
    print("%-40s %.2e" % (k, v))
