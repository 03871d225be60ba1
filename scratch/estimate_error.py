"""How wrong is a plain-f16 trunk (one f16 product per MAC, fp32 accumulate) as a density ESTIMATE on the fitted checkpoints?  torch emulation on coarse-grid points
(+ eps offsets) of seeded rays: distribution of |sigma_f16 - sigma_fp32|, and the cases that matter to k_select_points: fp32 density > 0 with an estimate below -margin.
    python scratch/estimate_error.py"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from ibl_nerf_amd import checkpoint as ck
dev = "cuda" if torch.cuda.is_available() else "cpu"

def enc(x):
    out = [x]
    for k in range(10):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)

def trunk(sd, pts, rnd):
    q = (lambda t: t.half().float()) if rnd == "f16" else (lambda t: t.double()) if rnd == "f64" else (lambda t: t)
    e = enc(pts)
    h = e
    for i in range(8):
        W, b = sd["positions_linears.%d.weight" % i], sd["positions_linears.%d.bias" % i]
        h = torch.relu((q(h) @ q(W).T).float() + b)
        if i == 4:
            h = torch.cat([e, h], -1)
    return ((q(h) @ q(sd["sigma_linear.weight"]).T).float() + sd["sigma_linear.bias"])[..., 0]

fl = 0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0))
g = torch.Generator().manual_seed(0)
for which in ("fitted", "fitted2"):
    f = np.load(os.path.join(ROOT, "tests", "golden", which + "_ckpt.npz"))
    sd = {k: torch.as_tensor(v, device=dev) for k, v in ck.blob_to_state_dict(f["coarse"]).items()}
    n = 16384
    px = torch.rand(n, 2, generator=g) * 800
    d = torch.stack([(px[:, 0] - 400) / fl, -(px[:, 1] - 400) / fl, -torch.ones(n)], -1).to(dev)
    z = torch.linspace(0.5, 8.0, 64, device=dev)
    pts = d[:, None, :] * z[None, :, None]
    up = torch.tensor([0.0, 1.0, 0.0], device=dev).expand_as(d)
    right = torch.cross(d, up, dim=-1)
    allp = torch.cat([pts, pts + 0.01 * right[:, None, :], pts - 0.01 * right[:, None, :]], 0).reshape(-1, 3)
    s32, s16, s64 = trunk(sd, allp, "f32"), trunk(sd, allp, "f16"), trunk(sd, allp, "f64")
    err = (s16 - s64).abs()
    print(which, "points %d  sigma range %.1f .. %.1f" % (allp.shape[0], float(s64.min()), float(s64.max())))
    for p in (50, 99, 99.9, 99.99, 100):
        print("   |f16 - fp64| p%-6g %.3e     |fp32 - fp64| %.3e" % (p, float(torch.quantile(err[:1 << 24].double(), p / 100)) if p < 100 else float(err.max()),
              float(torch.quantile((s32 - s64).abs()[:1 << 24].double(), p / 100)) if p < 100 else float((s32 - s64).abs().max())))
    for m in (0.25, 0.5, 1.0, 2.0):
        bad = (s64 > 0) & (s16 < -m)
        print("   density > 0 with an f16 estimate below -%.2f: %d      selected at this margin: %.4f" % (m, int(bad.sum()), float((s16 > -m).float().mean())))
    big = s64 > 10
    print("   relative error where density > 10: max %.3e  p99.9 %.3e" % (float((err[big] / s64[big]).max()), float(torch.quantile((err[big] / s64[big]).double()[:1 << 24], 0.999))))
    worst = err.argmax()
    print("   worst point: fp64 %.4f  fp32 %.4f  f16 %.4f" % (float(s64[worst]), float(s32[worst]), float(s16[worst])))
