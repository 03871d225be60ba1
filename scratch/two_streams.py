"""Would two launch sets on two streams (two contexts) overlap one set's per-ray helper kernels and launch tails with the other's MLP kernels?
The fitted frame: one context, two launch sets in a row, against two contexts with one launch set each on two streams."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as Bn, _pkg
pkg = _pkg.load()
from ibl_nerf_amd import renderer as R
torch.cuda.set_device(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "fitted"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sdc, sdf = Bn.load_checkpoint(kind)
lut = Bn.load_lut()
K, c2w = Bn.camera()
def mk():
    r = R.Renderer(64, 128, max_rays_per_launch=327680)
    r.load_weights(0, sdc); r.load_weights(1, sdf); r.load_lut(lut)
    return r
rs = [mk() for _ in range(NS)]
r1, r2 = rs[0], rs[1]
ro, rd = r1.get_rays(800, 800, K, c2w)
ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
n = ro.shape[0]
idx = torch.linspace(0, n - 1, 4096, device=ro.device).long()
route = r1.decide_route(ro[idx].contiguous(), rd[idx].contiguous(), Bn.NEAR, Bn.FAR)
r1.policy = {"decision": "fast", "imposed": True}; r1._set_routing(0)
for r in rs[1:]:
    r.set_route(route); r.policy = {"decision": "fast", "imposed": True}; r._set_routing(0)
half = n // 2
A = (ro[:half].contiguous(), rd[:half].contiguous()); B = (ro[half:].contiguous(), rd[half:].contiguous())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
streams = [torch.cuda.Stream() for _ in range(NS)]
cuts = [n * i // NS for i in range(NS + 1)]
parts = [(ro[cuts[i]:cuts[i + 1]].contiguous(), rd[cuts[i]:cuts[i + 1]].contiguous()) for i in range(NS)]

def many():
    for r in rs: r.range_check = "lazy"
    try:
        outs = []
        for r, st, pt in zip(rs, streams, parts):
            with torch.cuda.stream(st):
                outs.append(r._render(pt[0], pt[1], Bn.NEAR, Bn.FAR, None, {}, on_range="ignore"))
    finally:
        for r in rs: r.range_check = "eager"
    for st in streams: torch.cuda.current_stream().wait_stream(st)
    return outs

def sequential():
    a = r1._render(A[0], A[1], Bn.NEAR, Bn.FAR, None, {}, on_range="ignore")
    b = r1._render(B[0], B[1], Bn.NEAR, Bn.FAR, None, {}, on_range="ignore")
    return a, b

def concurrent():
    # (_render reads the range flags = a host sync: make both contexts lazy for the experiment)
    r1.range_check = r2.range_check = "lazy"
    try:
        with torch.cuda.stream(s1):
            a = r1._render(A[0], A[1], Bn.NEAR, Bn.FAR, None, {}, on_range="ignore")
        with torch.cuda.stream(s2):
            b = r2._render(B[0], B[1], Bn.NEAR, Bn.FAR, None, {}, on_range="ignore")
    finally:
        r1.range_check = r2.range_check = "eager"
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    return a, b

for name, fn in (("sequential", sequential), ("two streams", concurrent), ("%d streams" % NS, many), ("sequential", sequential), ("two streams", concurrent), ("%d streams" % NS, many)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = fn()
    torch.cuda.synchronize()
    print("%-12s %.1f ms per frame" % (name, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
sa, sb = sequential(); ca, cb = concurrent(); torch.cuda.synchronize()
same = all(torch.equal(sa[0][k].nan_to_num(7.0), ca[0][k].nan_to_num(7.0)) and torch.equal(sb[0][k].nan_to_num(7.0), cb[0][k].nan_to_num(7.0)) for k in sa[0])
print("bit-identical:", same)
