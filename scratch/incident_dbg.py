"""Round 5: use_gradient_for_incident_radiance — the incident part of the gradient (step with the flag minus the plain step) against the reference's, per tensor."""
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
GI = np.load(os.path.join(GOLDEN, "train_step_incident.npz")); GP = np.load(os.path.join(GOLDEN, "train_step.npz"))
teacher = "teacher" in sys.argv
def run(flag, G):
    nets, kw, K, rays = TT._setup(G, lut, "full")
    kw["use_gradient_for_incident_radiance"] = flag
    if teacher:
        kw["teacher_maps"] = {k[11:]: torch.from_numpy(G[k]).cuda() for k in G.files if k.startswith("full__out__") and k[11:].startswith(("n_dot_v_map", "reflected_", "target_normal_map", "target_depth_map"))}
    res = R.render_decomp(800, 800, K, chunk=int(G["chunk"]), rays=rays, gt_values={}, approximate_radiance=True, **kw)
    TL.total_loss(torch, res, {k[8:]: G[k] for k in G.files if k.startswith("target__")}, True).backward()
    return {t + "." + n: p.grad.cpu().numpy() for t, net in (("c", nets[0]), ("f", nets[1])) for n, p in net.named_parameters()}
gi, gp = run(True, GI), run(False, GP)
rows = []
for k in gi:
    t, n = k.split(".", 1)
    ri, rp = GI["full__grad_%s__%s" % (t, n)], GP["full__grad_%s__%s" % (t, n)]
    sc = np.abs(ri).max()
    rows.append((k, np.abs(gi[k] - ri).max() / sc, np.abs(gp[k] - rp).max() / sc, np.abs((gi[k] - gp[k]) - (ri - rp)).max() / sc, np.abs(ri - rp).max() / sc))
print("%-46s %9s %9s %9s %9s" % ("tensor", "with flag", "plain", "incident", "|incident|"))
for r_ in sorted(rows, key=lambda r_: -r_[1])[:24]:
    print("%-46s %9.2e %9.2e %9.2e %9.2e" % r_)
