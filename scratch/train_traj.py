"""Loss trajectory of N training steps (512 rays, fitted checkpoint, seeded pixels and targets, Adam lr 5e-4): the HIP path on the GPU, or — `--ref`, build container only — the
reference's own modules on the CPU.  Same rays, same targets; the stochastic draws differ.   python scratch/train_traj.py [--ref] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import train_loss as TL
ref = "--ref" in sys.argv
steps = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 30
n = 512
fl = np.float32(0.5 * 800 / np.tan(0.5 * np.deg2rad(60.0)))
K = np.array([[fl, 0, 400], [0, fl, 400], [0, 0, 1]], dtype=np.float32)
rng = np.random.RandomState(0)
pix = rng.permutation(640000)[:n]
i, j = (pix % 800).astype(np.float32), (pix // 800).astype(np.float32)
d = np.stack([(i - 400) / fl, -(j - 400) / fl, -np.ones_like(i)], -1).astype(np.float32)
tgn = TL.targets(rng, n)
if ref:
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as MG, tempfile, shutil
    torch_, R, M, Hh = MG.import_reference()
    tmp = tempfile.mkdtemp()
    kw, _, _, _, grad_vars, opt = M.create_IBLNeRF(MG.reference_args(tmp, 128)); shutil.rmtree(tmp, ignore_errors=True)
    sdc, sdf = MG.fitted_state_dicts()
    kw["network_fn"].load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdc.items()})
    kw["network_fine"].load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdf.items()})
    kw.update(near=0.5, far=8.0); kw["brdf_lut"] = MG.load_lut(torch)
    rays = torch.from_numpy(np.stack([np.zeros_like(d), d], 0)); tg = {k: torch.from_numpy(v) for k, v in tgn.items()}
    extra = MG.EDIT_KEYS_OFF
    torch.set_num_threads(8)
else:
    import _pkg; _pkg.load()
    import bench as Bn
    from ibl_nerf_amd import renderer as R
    sdc, sdf = Bn.load_checkpoint("fitted")
    nets = Bn._trainable_module(sdc).cuda(), Bn._trainable_module(sdf).cuda()
    opt = torch.optim.Adam([p for net in nets for p in net.parameters()], lr=5e-4)
    kw = dict(network_fn=nets[0], network_fine=nets[1], N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=0.0, brdf_lut=torch.from_numpy(Bn.load_lut()).cuda(), lut_coefficient="F",
              gamma_correct=True, correct_depth_for_prefiltered_radiance_infer=True, epsilon=0.01, use_radiance_linear=False, lindisp=False, near=0.5, far=8.0,
              target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient_epsilon", max_rays_per_launch=1024)
    rays = torch.from_numpy(np.stack([np.zeros_like(d), d], 0)).cuda(); tg = {k: torch.from_numpy(v).cuda() for k, v in tgn.items()}
    extra = {}
if "--seed" in sys.argv:
    torch.manual_seed(int(sys.argv[sys.argv.index("--seed") + 1]))
for it in range(steps):
    res = R.render_decomp(800, 800, K, chunk=32768, rays=rays, gt_values={}, approximate_radiance=True, **kw, **extra)
    loss = TL.total_loss(torch, res, tg, True)
    opt.zero_grad(); loss.backward()
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for g in opt.param_groups for p in g["params"] if p.grad is not None)))
    opt.step()
    line = "step %2d loss %.4f |grad| %.3e" % (it, float(loss.detach()), gn)
    if not ref:
        r = R.renderer_for(dict(kw, _lazy_range_check=True))
        line += " skipped %d scale 2^%d" % (getattr(r, "skipped_steps", 0), int(np.log2(getattr(r, "_grad_scale", r.GRAD_SCALE_INIT))))
    print(line, flush=True)
    if not ref and "--debug" in sys.argv and getattr(r, "skipped_steps", 0) >= 1:
        break
if not ref and "--debug" in sys.argv:
    r = R.renderer_for(dict(kw, _lazy_range_check=True))
    orig_rb = r._run_backward
    def spy_rb(up, launch, out, grad, grad_scale, who):
        pre = r.range_bits()
        orig_rb(up, launch, out, grad, grad_scale, who)
        torch.cuda.synchronize()
        print("  %s: flags before %d, upstream max %.3e finite %s, scale 2^%d, ok %s, grad finite %s, out finite %s, flags after %d" % (
            who, pre, float(up.abs().max()), bool(torch.isfinite(up).all()), int(np.log2(r.last_grad_scale)), bool(r.last_backward_ok), bool(torch.isfinite(grad).all()),
            bool(torch.isfinite(out).all()), r.range_bits()), flush=True)
    r._run_backward = spy_rb
    print(" params finite:", all(bool(torch.isfinite(p).all()) for net in nets for p in net.parameters()), "adam m finite:", all(bool(torch.isfinite(st["exp_avg"]).all()) for st in opt.state.values()))
    for it in range(3):
        res = R.render_decomp(800, 800, K, chunk=32768, rays=rays, gt_values={}, approximate_radiance=True, **kw)
        print(" after render: flags", r.range_bits(), "maps finite:", {k: bool(torch.isfinite(v).all()) for k, v in res.items() if not bool(torch.isfinite(v).all())})
        loss = TL.total_loss(torch, res, tg, True)
        opt.zero_grad(); loss.backward()
        gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for g in opt.param_groups for p in g["params"] if p.grad is not None)))
        print(" debug step %d loss %.4f |grad| %.3e" % (it, float(loss), gn), flush=True)
        opt.step()
