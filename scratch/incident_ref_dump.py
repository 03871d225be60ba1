"""Round 5 (build container only: imports the reference): the reference's step with use_gradient_for_incident_radiance — per pass the reflected rays' inputs of
raw2outputs_simple and dL/d(its four output maps), for scratch/incident_cmp.py."""
import os, sys, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), ROOT]
import make_golden as MG
import train_loss as TL
torch, R, M, Hh = MG.import_reference()
lut = MG.load_lut(torch)
G = np.load(os.path.join(ROOT, "tests", "golden", "train_step_incident.npz"))
tmp = tempfile.mkdtemp()
try:
    kw, _, *_ = M.create_IBLNeRF(MG.reference_args(tmp, 128))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
sd_c, sd_f = MG.fitted_state_dicts()
kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
kw["network_fine"].load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
kw.update(near=0.5, far=8.0, pytest=True, brdf_lut=lut, use_gradient_for_incident_radiance=True)
calls = []
f0 = R.raw2outputs_simple
def spy(raw, z_vals, rays_d, **k):
    rad, coarse = f0(raw, z_vals, rays_d, **k)
    maps = [rad] + list(coarse)
    for m in maps:
        m.retain_grad()
    calls.append(dict(z=z_vals.detach().numpy().copy(), rd=rays_d.detach().numpy().copy(), raw=raw.detach().numpy().copy(), maps=maps))
    return rad, coarse
R.raw2outputs_simple = spy
K = np.array([[692.8203, 0, 400], [0, 692.8203, 400], [0, 0, 1]], dtype=np.float32)
rays = torch.from_numpy(np.stack([G["rays_o"], G["rays_d"]], 0))
with torch.enable_grad():
    res = R.render_decomp(800, 800, K, chunk=64, rays=rays, gt_values={}, approximate_radiance=True, **kw, **MG.EDIT_KEYS_OFF)
    loss = TL.total_loss(torch, res, {k[8:]: G[k] for k in G.files if k.startswith("target__")}, True)
    loss.backward()
out = {}
for i, c in enumerate(calls):
    out["z%d" % i], out["rd%d" % i], out["raw%d" % i] = c["z"], c["rd"], c["raw"]
    out["denv%d" % i] = np.stack([m.grad.numpy() for m in c["maps"]], 1)
    out["env%d" % i] = np.stack([m.detach().numpy() for m in c["maps"]], 1)
print(len(calls), "calls; loss", float(loss), float(G["full__loss"]))
np.savez("/tmp/incident_ref.npz", **out)
