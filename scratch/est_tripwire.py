"""At which synthetic gain does the plain-f16 estimate fail its first-launch check (api.cpp check_estimates)?"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import _pkg; _pkg.load()
from conftest import load_lut_rgb
from ibl_nerf_amd import renderer as R, checkpoint as ck
lut = load_lut_rgb()
g = torch.Generator().manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=-1).cuda()
o = (0.1 * torch.randn(4096, 3, generator=g)).cuda()
for gain in (1.0, 1.5, 2.0, 3.0, 4.0):
    for bias in (0.3, -2.0):
        r = R.Renderer(64, 128, max_rays_per_launch=4096, mlp_precision="f16x3_mxfp6x")
        r.load_weights(0, ck.synthetic_state_dict(0, gain=gain, sigma_bias=bias)); r.load_weights(1, ck.synthetic_state_dict(1, gain=gain, sigma_bias=bias)); r.load_lut(lut)
        m = r.render_rays(o, d, 0.5, 8.0)
        print("gain %.1f bias %.1f: policy %s %s  selection %s  fallbacks %d  finite %s" % (gain, bias, r.estimate_policy(0), r.estimate_policy(1), r.last_selection(), r.range_fallbacks,
              bool(torch.isfinite(m["depth_map"]).all())), flush=True)
