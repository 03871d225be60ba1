"""Can the MAIN ray's density estimates choose which epsilon-offset samples need an estimate of their own?  

CPU experiment on the C restatement (fp32): n seeded pixels of the bench view (and a posed camera), coarse pass -> fine z -> the fine network's
densities at the main samples and at the four offset copies.  Then, per offset sample (v, r, s):
  today     relevant = sigma > -2 and the copy's OWN conservative transmittance in front of it > 1e-10   (k_select_points)
  rule A    candidate = max(main sigma at s-1, s, s+1) > -M2  and the MAIN ray's conservative transmittance in front of s-1 > 1e-12
  rule B    candidate = max(main sigma at s-1, s, s+1) > -M2                                (no transmittance of the main ray)
  rule C    candidate = s <= cut(r): every sample up to the main ray's saturation index (+1)  (no emptiness criterion: the per-ray adaptive z-cut)
and reports: share of candidates, share of today's relevant samples a rule misses, and how many rays they sit on.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import iblnerf_cpu as CPU          # noqa: E402
import iblnerf_oracle as O         # noqa: E402

F32 = np.float32
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
MARGIN = 2.0


def cons_T(sig, dists, margin=MARGIN):
    a = 1.0 - np.exp(-np.maximum(sig * 0.75 - margin, 0.0) * dists)
    om = (1.0 - a) + 1e-10
    T = np.cumprod(np.concatenate([np.ones_like(om[..., :1]), om], -1).astype(np.float64), -1)[..., :-1]
    return T


def blob_sd(blob):
    import _pkg
    ck = _pkg.load().checkpoint
    return ck.blob_to_state_dict(blob)


def run(ckpt, posed, seed):
    f = np.load(os.path.join(ROOT, "tests", "golden", ckpt + "_ckpt.npz"))
    sd_c, sd_f = blob_sd(f["coarse"]), blob_sd(f["fine"])
    rng = np.random.RandomState(seed)
    fl = F32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
    K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=F32)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(F32)
    if posed:
        q, _ = np.linalg.qr(np.eye(3) + 0.25 * rng.randn(3, 3))
        q = (q * np.sign(np.linalg.det(q))).astype(F32)
        c2w = np.concatenate([q, np.array([[0.35], [-0.25], [0.5]], F32)], 1).astype(F32)
    ro, rd = CPU.get_rays(800, 800, K, c2w)
    pix = rng.choice(640000, N, replace=False)
    ro, rd = ro.reshape(-1, 3)[pix], rd.reshape(-1, 3)[pix]
    zc = O.coarse_z(0.5, 8.0, 64, N)
    out = {}
    for name, sd, z in (("coarse", sd_c, zc), ("fine", sd_f, None)):
        if z is None:
            mids = (F32(0.5) * (zc[:, 1:] + zc[:, :-1])).astype(F32)
            zs = CPU.sample_pdf(mids, out["w_c"][:, 1:-1], 128)
            z = np.sort(np.concatenate([zc, zs], -1), -1)
        S = z.shape[1]
        pts = (ro[:, None, :] + rd[:, None, :] * z[:, :, None]).astype(F32)
        sig = CPU.network_query(sd, pts, None)[..., 0]
        dists = O.ray_dists(z, rd)
        if name == "coarse":
            out["w_c"] = O.alpha_weights(sig, dists)
        eps = F32(0.01)
        up0 = np.broadcast_to(np.array([0, 1, 0], dtype=F32), rd.shape)
        right = O.cross(rd, up0)
        up = O.cross(right, rd)
        offs = [eps * right, -(eps * right), eps * up, -(eps * up)]
        sig4 = np.stack([CPU.network_query(sd, (pts + o[:, None, :]).astype(F32), None)[..., 0] for o in offs], 0)     # [4, N, S]
        T_own = cons_T(sig4, dists[None])
        today = (sig4 > -MARGIN) & (T_own > 1e-10)
        T_main = cons_T(sig, dists)
        # neighbourhood max of the main ray's density
        pad = np.pad(sig, ((0, 0), (1, 1)), constant_values=-1e30)
        nb = np.maximum(np.maximum(pad[:, :-2], pad[:, 1:-1]), pad[:, 2:])
        T_before_prev = np.concatenate([np.ones_like(T_main[:, :1]), T_main[:, :-1]], -1)     # transmittance in front of s-1
        print("%s %s %s: relevant today %.3f of the offset samples; main ray relevant %.3f" % (
            ckpt, "posed" if posed else "frontal", name, today.mean(), ((sig > -MARGIN) & (T_main > 1e-8)).mean()))
        print("   main-ray sigma percentiles in front of the first relevant sample: ", end="")
        front = np.cumsum((sig > -MARGIN), 1) == 0
        print(np.round(np.percentile(sig[front], [50, 90, 99, 99.9, 100]), 2) if front.any() else "-")
        for M2 in (3.0, 4.0, 6.0, 8.0):
            for rule, cand in (("A", (nb > -M2) & (T_before_prev > 1e-12)), ("B", nb > -M2)):
                c4 = np.broadcast_to(cand[None], today.shape)
                miss = today & ~c4
                # what a missed sample could do: its weight on the copy's own ray (true arithmetic)
                w4 = np.stack([O.alpha_weights(sig4[v], dists) for v in range(4)], 0)
                print("   rule %s M2=%.0f: candidates %.3f, missed %d of %d relevant (%.2e), on %d rays; largest weight of a missed sample %.2e; sum of missed weights per ray max %.2e" % (
                    rule, M2, c4.mean(), miss.sum(), today.sum(), miss.sum() / max(1, today.sum()), miss.any(0).any(-1).sum(),
                    (w4 * miss).max(), (w4 * miss).sum(-1).max()))
        # rule C: per-ray adaptive cut from the main ray's transmittance; which copies do NOT saturate by then (need a second chunk)
        alive_main = T_before_prev > 1e-12
        cut = alive_main.sum(1)                                    # samples [0, cut) are estimated for every copy
        idx = np.arange(S)[None, :]
        first = idx < cut[:, None]
        T_at_cut = np.take_along_axis(T_own, np.minimum(cut, S - 1)[None, :, None].repeat(4, 0), 2)[..., 0]
        T_at_cut = np.where(cut[None] >= S, 0.0, T_at_cut)
        second = T_at_cut > 1e-12                                  # copies still alive behind the main ray's cut: estimate the rest for them
        n_est = 4 * first.sum() + (second * (S - cut)[None]).sum()
        own_alive = (np.concatenate([np.ones_like(T_own[..., :1]), T_own[..., :-1]], -1) > 1e-12).sum()
        print("   rule C (adaptive cut): estimates %.3f of the offset samples (copies needing the second chunk: %.4f); ideal per-copy own-T marching %.3f; fixed cuts today ~" % (
            n_est / today.size, second.mean(), own_alive / today.size))
        sys.stdout.flush()


for ckpt in ("fitted", "fitted2"):
    for posed in (False, True):
        run(ckpt, posed, 7)
