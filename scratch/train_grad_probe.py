"""Worst parameter-gradient errors of a training step against the reference's loss.backward() (tests/test_gpu_training.py) with the coarse pass's density on the
15-slot form (default) and on round 3's 22-bit operands (IBLNERF_ROUTE_COARSE_MAIN_22BIT).   python scratch/train_grad_probe.py"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_lut_rgb, GOLDEN
import test_gpu_training as T, train_loss as TL
from ibl_nerf_amd import renderer as R
G = np.load(os.path.join(GOLDEN, "train_step.npz")); lut = load_lut_rgb()
for routing in ((), ("coarse_main_22bit",)):
    for phase in ("full", "frozen"):
        nets, kw, K, rays = T._setup(G, lut, phase)
        kw["query_routing"] = routing
        res = R.render_decomp(800, 800, K, chunk=int(G["chunk"]), rays=rays, gt_values={}, approximate_radiance=True, **kw)
        tg = {k[8:]: G[k] for k in G.files if k.startswith("target__")}
        loss = TL.total_loss(torch, res, tg, True); loss.backward()
        worst = {}
        for tag, net in (("c", nets[0]), ("f", nets[1])):
            for name, prm in net.named_parameters():
                ref = G["%s__grad_%s__%s" % (phase, tag, name)]
                got = np.zeros_like(ref) if prm.grad is None else prm.grad.cpu().numpy()
                scale = float(np.abs(ref).max())
                if scale == 0: continue
                if name.endswith(".bias") and ref.size <= 3:
                    scale = max(scale, float(np.abs(G["%s__grad_%s__%s" % (phase, tag, name[:-4] + "weight")]).max()))
                worst[tag + "." + name] = float(np.abs(got - ref).max()) / scale
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
        zs = float(np.abs(res["z_std"].detach().cpu().numpy() - G[phase + "__out__z_std"]).max())
        print(routing or "default", phase, "loss rel %.1e  z_std abs %.1e " % (abs(float(loss.detach()) - float(G[phase + "__loss"])) / float(G[phase + "__loss"]), zs), " ".join("%s %.2e" % kv for kv in top), flush=True)
