"""Thirty frames on the stream pair: the caching allocator's reserved memory must settle (outputs are allocated on the side streams, concatenated on the caller's)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as Bn, _pkg
pkg = _pkg.load()
from ibl_nerf_amd import renderer as R, dist as D
torch.cuda.set_device(0)
lut = Bn.load_lut(); K, c2w = Bn.camera()
sdc, sdf = Bn.load_checkpoint("fitted2")
r = R.Renderer(64, 128, max_rays_per_launch=327680)
r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
ro, rd = r.get_rays(800, 800, K, c2w); ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
seen = []
for i in range(30):
    out = r.render_rays(ro, rd, Bn.NEAR, Bn.FAR)
    torch.cuda.synchronize()
    seen.append((torch.cuda.memory_reserved() >> 20, torch.cuda.memory_allocated() >> 20))
    del out
print("reserved / allocated MiB per frame:", seen[:3], "...", seen[-3:], "free HBM GiB", torch.cuda.mem_get_info()[0] >> 30)
assert seen[-1][0] == seen[5][0], "reserved memory still growing"
print("ok")
