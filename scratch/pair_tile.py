"""Does the stream pair pay for an 8-rank tile (80 000 rays) or a 4-rank tile (160 000)?"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as Bn, _pkg
pkg = _pkg.load()
from ibl_nerf_amd import renderer as R, dist as D
torch.cuda.set_device(0)
lut = Bn.load_lut(); K, c2w = Bn.camera()
sdc, sdf = Bn.load_checkpoint("fitted")
r = R.Renderer(64, 128, max_rays_per_launch=327680)
r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
probe = D.frame_probe_for_call(r, 800, 800, K, c2w, Bn.NEAR, Bn.FAR)
fo, fd = r.get_rays(800, 800, K, c2w)
for world in (8, 4, 2):
    rr = D.tile_row_indices(800, 1, world, "interleaved")
    ts = slice(rr.start, rr.stop, rr.step)
    to, td = fo[ts].reshape(-1, 3).contiguous(), fd[ts].reshape(-1, 3).contiguous()
    for thr in (1 << 30, 32768, 1 << 30, 32768):
        r.PAIR_MIN_RAYS = thr
        for _ in range(2): r.render_rays(to, td, Bn.NEAR, Bn.FAR, probe=probe)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): r.render_rays(to, td, Bn.NEAR, Bn.FAR, probe=probe)
        torch.cuda.synchronize()
        print("tile of %d ranks (%d rays): %s %.1f ms" % (world, to.shape[0], "pair" if thr < 1 << 29 else "one ", (time.perf_counter() - t0) / 4 * 1e3), flush=True)
