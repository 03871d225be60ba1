"""Round 5: this path's reflected-ray inputs and dL/d env per pass (teacher maps on), for comparison with scratch/incident_ref_dump.py's."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _pkg; _pkg.load()
from conftest import GOLDEN, load_lut_rgb
import test_gpu_training as TT
import train_loss as TL
from ibl_nerf_amd import renderer as R
lut = load_lut_rgb()
GI = np.load(os.path.join(GOLDEN, "train_step_incident.npz"))
nets, kw, K, rays = TT._setup(GI, lut, "full")
kw["use_gradient_for_incident_radiance"] = True
kw["teacher_maps"] = {k[11:]: torch.from_numpy(GI[k]).cuda() for k in GI.files if k.startswith("full__out__") and k[11:].startswith(("n_dot_v_map", "reflected_", "target_normal_map", "target_depth_map"))}
dump, o_cdb, o_rob = {}, R.Renderer.composite_direct_backward, R.Renderer.ray_outputs_backward
def cdb(self, raw, z, rd, dm, dw=None, full=False):
    if full:
        i = len([k for k in dump if k.startswith("z")])
        dump["z%d" % i], dump["rd%d" % i], dump["raw%d" % i], dump["denv%d" % i] = z.cpu().numpy(), rd.cpu().numpy(), raw.cpu().numpy(), dm[:, 7:19].reshape(-1, 4, 3).cpu().numpy()
    return o_cdb(self, raw, z, rd, dm, dw, full)
R.Renderer.composite_direct_backward = cdb
res = R.render_decomp(800, 800, K, chunk=int(GI["chunk"]), rays=rays, gt_values={}, approximate_radiance=True, **kw)
TL.total_loss(torch, res, {k[8:]: GI[k] for k in GI.files if k.startswith("target__")}, True).backward()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "incident_my.npz"), **dump)
print(sorted(dump))
