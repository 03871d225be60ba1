"""Gradient parity of a training step against the reference's loss.backward() (train_step.npz), default routing vs fine_main_precise."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import GOLDEN, load_lut_rgb
import test_gpu_training as TT, train_loss as TL
from ibl_nerf_amd import renderer as R
G = np.load(os.path.join(GOLDEN, "train_step.npz")); lut = load_lut_rgb()
for phase in ("warmup", "full"):
    for routing in (0, ("fine_main_precise",)):
        nets, kw, K, rays = TT._setup(G, lut, phase)
        res = R.render_decomp(800, 800, K, chunk=int(G["chunk"]), rays=rays, gt_values={}, approximate_radiance=phase != "warmup", query_routing=routing, **kw)
        tg = {k[8:]: G[k] for k in G.files if k.startswith("target__")}
        loss = TL.total_loss(torch, res, tg, phase != "warmup"); loss.backward()
        worst = {}
        for tag, net in (("c", nets[0]), ("f", nets[1])):
            for name, prm in net.named_parameters():
                ref = G["%s__grad_%s__%s" % (phase, tag, name)]
                scale = float(np.abs(ref).max())
                if name.endswith(".bias") and ref.size <= 3:
                    scale = max(scale, float(np.abs(G["%s__grad_%s__%s" % (phase, tag, name[:-4] + "weight")]).max()))
                worst[tag + "." + name] = float(np.abs(prm.grad.cpu().numpy() - ref).max()) / max(scale, 1e-30)
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
        fw = {k: float(np.abs(res[k].detach().cpu().numpy() - G["%s__out__%s" % (phase, k)]).max() / np.abs(G["%s__out__%s" % (phase, k)]).max()) for k in ("weights", "albedo_map", "depth_map")}
        print(phase, routing, "loss err %.1e" % abs(float(loss.detach()) / float(G[phase + "__loss"]) - 1), "median grad err %.1e" % np.median(list(worst.values())), "worst:", " ".join("%s %.1e" % kv for kv in top), "| fwd", fw)
