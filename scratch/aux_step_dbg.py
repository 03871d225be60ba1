"""Round 5: train_step_aux.npz — per-tensor gradient distances of the auxiliary networks."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import _pkg; _pkg.load()
from conftest import GOLDEN, rel_linf, load_lut_rgb
import test_gpu_training as TT
import train_loss as TL
from torch_ref import AuxShaped
from ibl_nerf_amd import renderer as R, checkpoint as ck
lut = load_lut_rgb()
phase = sys.argv[1] if len(sys.argv) > 1 else "warmup"
G = np.load(os.path.join(GOLDEN, "train_step_aux.npz"))
nets, kw, K, rays = TT._setup(G, lut, phase)
aux = {k[5:]: AuxShaped(ck.AUX_OUT_CH[k[5:]], ck.synthetic_position_mlp(int(G[k]), ck.AUX_OUT_CH[k[5:]], 1.0)).cuda() for k in G.files if k.startswith("aux__")}
kw.update(aux, infer_normal=True)
approx = phase == "full"
res = R.render_decomp(800, 800, K, chunk=int(G["chunk"]), rays=rays, gt_values={}, approximate_radiance=approx, **kw)
for k in ("albedo_map", "irradiance_map", "roughness_map", "inferred_normal_map", "z_std", "weights"):
    for sfx in ("", "0"):
        if k + sfx in res:
            print(k + sfx, "%.2e" % rel_linf(res[k + sfx].detach().cpu().numpy(), G["%s__out__%s" % (phase, k + sfx)]))
loss = TL.total_loss(torch, res, {k[8:]: G[k] for k in G.files if k.startswith("target__")}, approx)
loss.backward()
worst, zero = TT._grads_against(G, phase, [("c", nets[0]), ("f", nets[1])] + sorted(aux.items()))
for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:40]:
    print("%-48s %.2e" % (k, v))

# --- the same aux_backward calls against torch autograd (float64) on the very inputs the step hands them
print("\n--- aux_backward vs autograd on the step's own (pts, upstream) ---")
from torch_ref import embed
import ibl_nerf_amd.renderer as RR
orig = RR.Renderer.aux_backward
def spy(self, name, pts, dout, grad_scale=None):
    g = orig(self, name, pts, dout, grad_scale)
    net = AuxShaped(ck.AUX_OUT_CH[name], {k: v.detach().cpu() for k, v in aux[name].state_dict().items()}).double().cuda()
    p = pts.reshape(-1, 3).double()
    with torch.enable_grad():
        out = net(embed(p, 10))
        (out * dout.reshape(out.shape).double()).sum().backward()
    rep = {n: rel_linf(g[n].cpu().numpy(), q.grad.cpu().numpy()) for n, q in net.named_parameters()}
    d = dout.reshape(-1, dout.shape[-1])
    top = sorted(rep.items(), key=lambda kv: -kv[1])[:3]
    print("%-14s n_pts %6d  |up| max %.2e  rows above 1e-3 max: %5d   worst %s" % (name, p.shape[0], float(d.abs().max()), int((d.abs().amax(-1) > 1e-3 * d.abs().max()).sum()),
                                                                                  ", ".join("%s %.1e" % kv for kv in top)))
    return g
RR.Renderer.aux_backward = spy
for m in list(aux.values()) + list(nets):
    m.zero_grad()
res = R.render_decomp(800, 800, K, chunk=int(G["chunk"]), rays=rays, gt_values={}, approximate_radiance=approx, **kw)
TL.total_loss(torch, res, {k[8:]: G[k] for k in G.files if k.startswith("target__")}, approx).backward()
