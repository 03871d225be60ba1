#!/usr/bin/env python3
"""Headline benchmark: rays/s of the forward/inference render path (64 coarse + 128 fine samples)
on the 800x800 "Kitchen" test view of BASELINE.json configs[1], synthetic checkpoint and camera
(SURVEY.md §8 d: no dataset or checkpoint ships with the reference).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one full frame (640 000 rays): every rank renders its row tile of the frame and the
ranks reassemble it with one RCCL all-gather (strong scaling: the frame is fixed, N divides it).
Inputs (weights, LUT, camera) are resident in HBM before the timed region.  Rank 0 prints ONE
JSON line.  The `roofline` object times the dominant kernel (the fused MLP) with HIP events on
the launch stream; `cpu_baseline` times the C restatement of the path (oracle/csrc, OpenMP on the
host's cores) on a bounded sample of the same rays and reports, beside the colour PSNR, the per-ray error
percentiles of the HIP frame on that sample for every intrinsic channel (`parity_on_sample`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (before the first HIP call of the process: the renderer's two streams need hardware queues of their own beside RCCL's — ibl-nerf_amd/__init__.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import _pkg  # noqa: E402

H = W = 800
FOV_DEG = 60.0
NEAR, FAR = 0.5, 8.0
N_SAMPLES, N_IMPORTANCE = 64, 128
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
# algorithmic FLOPs per camera ray, SURVEY.md 8(d) mode (i) as csrc/api.cpp prices it (2 x nn.Linear MACs, every sample of every query once):
# (64 + 192) full + (256 + 768) trunk + 128 reflected evaluations; mode (ii) (--inference-min): the coarse pass 64 trunk evaluations only
F_FULL, F_TRUNK, F_REFL = 1591552.0, 982528.0, 1458944.0
F_ALG_PER_RAY = (N_SAMPLES + N_SAMPLES + N_IMPORTANCE) * F_FULL + 4 * (N_SAMPLES + N_SAMPLES + N_IMPORTANCE) * F_TRUNK + 2 * N_SAMPLES * F_REFL
F_ALG_PER_RAY_MIN = N_SAMPLES * F_TRUNK + (N_SAMPLES + N_IMPORTANCE) * F_FULL + 4 * (N_SAMPLES + N_IMPORTANCE) * F_TRUNK + N_SAMPLES * F_REFL
# per mlp_precision: (dtype string, kernels, matrix-core products per algorithmic MAC)
MODES = {
    "auto": ("per checkpoint, measured at load (renderer.calibrate): the fast table f16x3_mxfp6x — f16 hi/lo splits x3 products (~2^-22 per operand) where errors are amplified "
             "(coarse grid's offset queries, auxiliary networks, the offset copies' own-selection samples), the coarse pass's density — it places the fine samples — in "
             "EXACT fp32 (fp32 MFMA, the reference's own summation order), f16 + 2x MX-fp6 (~2^-16) for the "
             "main queries' other channels, layers 2-7 of the fine offsets and the reflected-ray queries — or, where that leaves the checkpoint's per-sample weights / maps "
             "beyond the calibration limits against it, f16x3_mxfp6 (x3 f16 products for every query but the reflected-ray ones); fp32 accumulate.  Since round 4 "
             "each query runs as a plain-f16 density ESTIMATE on its samples (round 5: of the offset copies only those outside the main ray's relevant range) and in the "
             "precision named here only on the samples that can carry a weight (k_select_points, k_range_points; config.route_table; roofline.executed)",
             "ibl::mxk16::mlp_kernel<TRUNK> (estimates) + ibl::mxk::mlp_kernel<TRUNK_X_LIST|FULL_LIST|REFL_LIST> + ibl::f16x3k::mlp_kernel<TRUNK_LIST> + k_trunk_fp32 (relevant samples)",
             "1 f16 product per estimated sample; on the relevant samples fp32 products for the coarse density, 3 f16 products for the coarse grid's offset copies, "
             "1 f16 + 2 block-scaled fp6 products elsewhere in the fast table (layers 0-1 of the fine offsets: 3 f16), 3 f16 products in the safe one"),
    "f16x3_mxfp6": ("f16 hi/lo splits x3 products (~2^-22 per operand) for every query but the reflected-ray ones, which run "
                    "f16 + 2x MX-fp6 residual products (~2^-16); fp32 accumulate",
                    "ibl::f16x3k::mlp_kernel + ibl::mxk::mlp_kernel", "3 f16 MFMA products; 1 f16 + 2 block-scaled fp6 products in the reflected-ray queries"),
    "f16x3_mxfp6x": ("f16 hi/lo splits x3 products (~2^-22 per operand) where errors are amplified — the coarse pass's main query (it places the fine samples), "
                     "auxiliary networks, the coarse grid's offset queries; the fine pass's offset queries on the fast kernel's mixed trunk form (layers 0-1 as three "
                     "f16 products, layers 2-7 as f16 + 2x MX-fp6); f16 + 2x MX-fp6 (~2^-16) for the fine pass's main query and the reflected-ray queries; fp32 accumulate",
                     "ibl::f16x3k::mlp_kernel + ibl::mxk::mlp_kernel<TRUNK_X>", "3 f16 MFMA products in the coarse pass; 1 f16 + 2 block-scaled fp6 products in layers 2-7 of the fine "
                     "offsets, in the fine main query and in the reflected-ray queries"),
    "f16x3_main": ("f16 hi/lo splits x3 products for the main, auxiliary and coarse-grid offset queries; f16 + 2x MX-fp6 residual products "
                   "for the fine pass's offset queries and the reflected-ray queries; fp32 accumulate",
                   "ibl::f16x3k::mlp_kernel + ibl::mxk::mlp_kernel", "3 f16 MFMA products in the precise queries, 1 f16 + 2 block-scaled fp6 products in the others"),
    "f16x3": ("f16x3 (f16 hi/lo split, 3 MFMA products, ~2^-22 per operand, fp32 accumulate)", "ibl::f16x3k::mlp_kernel", "3 f16 MFMA products"),
    "f16_mxfp6": ("f16 + 2x MX-fp6 residual products, fp32 accumulate (fp32 operands to ~2^-16)", "ibl::mxk::mlp_kernel",
                  "1 f16 + 2 block-scaled fp6 MFMA products (= 1.5 bf16-rate products)"),
    "f16_mixed": ("f16 + 2x MX-fp6 residual products for the sample-placing and normal queries, plain f16 for the others, fp32 accumulate",
                  "ibl::mxk::mlp_kernel + ibl::mxk16::mlp_kernel",
                  "1 f16 + 2 block-scaled fp6 MFMA products in the coarse main and offset queries, 1 f16 product in the fine main and reflected queries"),
    "bf16x3": ("bf16x3 (bf16 hi/lo split, 3 MFMA products, ~2^-17 per operand, fp32 accumulate)", "ibl::mlp_kernel", "3 bf16 MFMA products"),
}


def load_lut():
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "ibl_brdf_lut.png")).convert("RGB"), dtype=np.float32)
    return np.ascontiguousarray((img / np.float32(255.0)).transpose(2, 0, 1))


def camera():
    f = np.float32(0.5 * W / np.tan(0.5 * np.deg2rad(FOV_DEG)))
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=np.float32)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    return K, c2w


def pmc_traffic(mode):
    """HBM bytes per MLP launch from a COMMITTED rocprofv3 PMC pass (profiles/<round>_<tag>/pmc.json, written by
    profiles/summarize.py: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc runs), averaged over the launches of
    the kernel variants — not a counter read in this run.  The newest profile is the one named by profiles/LATEST (one line:
    a directory name), else the most recently modified pmc.json.  None if no profile is committed."""
    import glob
    latest = os.path.join(ROOT, "profiles", "LATEST")
    files = []
    if os.path.exists(latest):
        cand = os.path.join(ROOT, "profiles", open(latest).read().strip(), "pmc.json")
        files = [cand] if os.path.exists(cand) else []
    if not files:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc.json")), key=os.path.getmtime)
    if not files:
        return None, None
    d = json.load(open(files[-1]))["derived"]
    num = den = 0.0
    for k, v in d.items():
        if "mlp_kernel" in k and "hbm_read_bytes" in v and "hbm_write_bytes" in v and v.get("calls"):
            num += v["calls"] * (v["hbm_read_bytes"] + v["hbm_write_bytes"])
            den += v["calls"]
    return (num / den if den else None), os.path.relpath(files[-1], ROOT)


def load_checkpoint(kind):
    """The two networks of the benchmark: "fitted" = the checkpoint with surfaces that tests/golden/fit_checkpoint.py fitted with the
    reference's own modules (tests/golden/fitted_ckpt.npz: the stand-in for BASELINE configs[1]'s "pretrained checkpoint"; no real one
    ships with the reference), "synthetic" = round 1's seeded random-init networks (fog).  The work per ray is the same either way
    (fixed sample counts, no early termination); the operand statistics the matrix cores see are not."""
    from ibl_nerf_amd import checkpoint as ck
    if kind in ("fitted", "fitted2", "fitted3"):    # (fitted3: round 5's hold-out scene; fitted2: the second, sharper scene of tests/golden/fit_checkpoint.py — the checkpoint on which mlp_precision="auto" decides "safe")
        f = np.load(os.path.join(ROOT, "tests", "golden", kind + "_ckpt.npz"))
        return ck.blob_to_state_dict(f["coarse"]), ck.blob_to_state_dict(f["fine"])
    return ck.synthetic_state_dict(0, 1.0), ck.synthetic_state_dict(1, 1.0)


# north_star's intrinsic channels (+ depth): compared on the CPU sample's rays in every bench line
PARITY_KEYS = ("albedo_map", "roughness_map", "target_normal_map", "irradiance_map", "prefiltered_reflected_map", "depth_map")


def _cpu_worker(job):
    """The C restatement (oracle/iblnerf_cpu.h: gcc + OpenMP, fp32) in a process of its own: a short calibration batch, then as many seeded
    pixels as fill about `seconds` of wall time on all of the host's threads."""
    kind, seconds = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import iblnerf_cpu as OC
    threads = OC.usable_cpus()
    _pkg.load()
    sdc, sdf = load_checkpoint(kind)
    lut = load_lut()
    K, c2w = camera()
    ro, rd = OC.get_rays(H, W, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    pix = np.random.RandomState(0).permutation(H * W)
    n0 = 12 * threads                                                     # one task (12 rays) per thread
    t0 = time.perf_counter()
    OC.render_rays(sdc, sdf, ro[pix[-n0:]], rd[pix[-n0:]], NEAR, FAR, lut, N_SAMPLES, N_IMPORTANCE, n_threads=threads)
    rate = n0 / (time.perf_counter() - t0)
    n = int(min(H * W // 2, max(n0, rate * seconds)) // (12 * threads) * (12 * threads)) or n0
    sel = pix[:n]
    t0 = time.perf_counter()
    res = OC.render_rays(sdc, sdf, ro[sel], rd[sel], NEAR, FAR, lut, N_SAMPLES, N_IMPORTANCE, n_threads=threads)
    dt = time.perf_counter() - t0
    return dt, sel, {k: res[k] for k in PARITY_KEYS + ("color_map",)}, OC.isa(), threads


def reference_cpu_record(kind):
    """The reference's own PyTorch-CPU figure for this checkpoint, read from the file tests/golden/time_reference_cpu.py wrote in the build
    container (the reference cannot travel to the GPU box; its measured timing can).  None if the file is missing."""
    path = os.path.join(ROOT, "tests", "golden", "reference_cpu_timing.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))
    ent = rec["checkpoints"].get(kind)
    if not ent:
        return None
    threads = max(ent["threads"], key=int)
    return {"value": ent["threads"][threads]["rays_per_s"], "unit": "rays/s", "cores": int(threads), "host": rec["host"], "n_rays": rec["n_rays"],
            "weights_checksum": ent["weights_checksum"], "source": "tests/golden/reference_cpu_timing.json",
            "note": rec["what"] + " (BASELINE.md section 2b); a recorded figure: the reference cannot travel to the GPU box"}


def cpu_baseline(gpu_color_fn, kind, seconds=15.0):
    """The C restatement of the path (kind = "port": oracle/csrc, pinned to the reference's fixtures by tests/test_oracle_c.py) timed on seeded
    pixels of the same view on every hardware thread this process may use (oracle/iblnerf_cpu.py usable_cpus: affinity and cgroup quota; OpenMP, one 12-ray task at a time per thread), in
    a spawned process (never forked: this process owns a GPU context).  value = rays / wall time of the one render call."""
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(1) as pool:
        dt, idx, ref, isa, threads = pool.map(_cpu_worker, [(kind, seconds)])[0]
    got = gpu_color_fn(idx, "color_map").astype(np.float64)
    mse = float(np.mean((got - ref["color_map"].astype(np.float64)) ** 2))
    psnr = float(10 * np.log10(1.0 / max(mse, 1e-30)))
    parity = {}
    for k in PARITY_KEYS:       # per-ray |HIP - CPU| over the map's largest value: 99 % / 99.9 % / worst ray, and the share of rays above north_star's 1e-3
        b = ref[k].astype(np.float64).reshape(len(idx), -1)
        e = np.abs(gpu_color_fn(idx, k).astype(np.float64).reshape(len(idx), -1) - b).max(-1) / max(float(np.abs(b).max()), 1e-30)
        parity[k] = {"p99": float(np.percentile(e, 99)), "p999": float(np.percentile(e, 99.9)), "max": float(e.max()), "share_above_1e-3": float((e > 1e-3).mean())}
    return {"value": len(idx) / dt, "unit": "rays/s", "cores": int(threads), "kind": "port", "host_logical_cpus": os.cpu_count(),
            "implementation": "C restatement of the reference path (oracle/csrc: gcc, OpenMP, fp32, %s dense layers)" % isa,
            "reference_in_build_container": reference_cpu_record(kind),
            "sample": "%d seeded pixels of the same 800x800 view, 64+128 samples, full result dict, one call on %d OpenMP threads (%.1f s)"
                      % (len(idx), threads, dt),
            "parity_on_sample": dict(parity, note="HIP frame against the C restatement on the sample's rays, per-ray relative error per intrinsic channel "
                                                  "(DESIGN.md section 2: the tails are rays the reference itself is sensitive on; prefiltered radiance is ill-conditioned in the reference)")}, psnr


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <argv>` as a child
    (one rank per GPU, rendezvous on 127.0.0.1 at a port that is free now) and return its exit code.  The child inherits stdout / stderr,
    so rank 0's single JSON line is this command's single JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))   # dmabuf IPC: what RCCL needs on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def parity_vs_reference_fixture(frame_maps_fn):
    """The timed frame against the REFERENCE's own render at the 16 384 pixels it rendered of this very view and checkpoint (fixture
    tests/golden/fitted_launch16k.npz, made by tests/golden/make_golden.py from /root/reference in the build container: the reference's
    float32 maps and, per ray, its own sensitivity — float64-vs-float32 difference, one-ulp nudges of the coarse weights, both branches of
    sample_pdf's threshold; NOT the fourth, 22-bit-parameter column, which describes this library, not the reference).  Per intrinsic channel:
    per-ray relative error 99 % / 99.9 % / worst, rays above north_star's 1e-3; `rays_sensitive_in_reference` = the number of these rays whose
    own reference sensitivity exceeds 1e-3 / 8 (a statement about the reference's conditioning, not an allowance); and — the yardstick — `c_restatement` = the same
    statistics of the fp32 C restatement (oracle/csrc) on the same rays against the same reference render (tests/golden/c_restatement_column.json, written by
    tests/golden/make_c_column.py): what an actual fp32 implementation attains."""
    path = os.path.join(ROOT, "tests", "golden", "fitted_launch16k.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    pix = g["pix"]
    out = {"n_rays": int(len(pix)), "fixture": "tests/golden/fitted_launch16k.npz",
           "note": "HIP frame vs the reference's own float32 render (PyTorch, build container) at the pixels of the fixture; per-ray |diff| max over a map's channels "
                   "over the map's largest value; c_restatement = the fp32 C restatement against the same render on the same rays (the yardstick: what fp32 attains); "
                   "rays_sensitive_in_reference counts rays whose own fp64-vs-fp32 / one-ulp-nudge / threshold-branch sensitivity in the reference exceeds 1e-3 / 8"}
    col_path = os.path.join(ROOT, "tests", "golden", "c_restatement_column.json")
    col = json.load(open(col_path)).get("fitted_launch16k", {}) if os.path.exists(col_path) else {}
    for k in PARITY_KEYS:
        ref = g["out__" + k].astype(np.float64).reshape(len(pix), -1)
        e = np.abs(frame_maps_fn(pix, k).astype(np.float64).reshape(len(pix), -1) - ref).max(-1) / max(float(np.abs(ref).max()), 1e-30)
        f = g["floorray__" + k].astype(np.float64)
        for y in ("nudgeray__", "branchray__"):
            if y + k in g.files:
                f = np.maximum(f, g[y + k].astype(np.float64))
        out[k] = {"p99": float(np.percentile(e, 99)), "p999": float(np.percentile(e, 99.9)), "max": float(e.max()), "rays_above_1e-3": int((e > 1e-3).sum()),
                  "c_restatement": ({"rays_above_1e-3": col[k]["above_1e-3"], "p999": col[k]["p999"], "max": col[k]["max"]} if k in col else None),
                  "rays_sensitive_in_reference": int((f > 1e-3 / 8).sum())}
    return out


def _trainable_module(sd):
    """An nn.Module holding one IBLNeRF network's parameters under the reference's state-dict names and registration order (nerf_models/ibl_nerf.py:45-72): what
    train.py hands render_decomp as network_fn / network_fine and its optimizer steps.  Parameters only — the forward and backward arithmetic is the library's."""
    import torch
    import torch.nn as nn

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            groups = {}
            for key, v in sd.items():
                name, leaf = key.rsplit(".", 1)
                groups.setdefault(name, {})[leaf] = v
            done = set()
            for name, wb in groups.items():          # in the state dict's own order: render_decomp's device upload flattens parameters in registration order
                parts = name.split(".")
                if len(parts) == 2 and parts[1].isdigit():
                    if parts[0] in done:
                        continue
                    done.add(parts[0])
                    members = sorted((int(k.split(".")[1]), k) for k in groups if k.split(".")[0] == parts[0] and len(k.split(".")) == 2)
                    setattr(self, parts[0], nn.ModuleList([nn.Linear(groups[k]["weight"].shape[1], groups[k]["weight"].shape[0]) for _, k in members]))
                else:
                    setattr(self, name, nn.Linear(wb["weight"].shape[1], wb["weight"].shape[0]))
            self.load_state_dict({k: torch.as_tensor(np.array(v)) for k, v in sd.items()})
            self.coarse_radiance_number = 3
    return Net()


def reference_train_record(n):
    """The reference's own training step timed on the CPU of the build container (tests/golden/time_reference_train_cpu.py; the reference cannot travel to the GPU box)."""
    path = os.path.join(ROOT, "tests", "golden", "reference_train_cpu_timing.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))
    ent = rec["rays"].get(str(n))
    if not ent:
        return None
    threads = max(ent["threads"], key=int)
    t = ent["threads"][threads]
    return {"value": t["rays_per_s"], "unit": "rays/s", "cores": int(threads), "kind": "reference", "seconds_per_step": t["seconds_per_step"],
            "render_s": t["render_s"], "loss_backward_s": t["loss_backward_s"], "adam_s": t["adam_s"], "host": rec["host"], "torch": rec["torch"],
            "sample": "%d-ray training step (render_decomp with render_kwargs_train -> losses -> loss.backward() -> Adam), fitted checkpoint, %s threads, build container "
                      "(a RECORDED figure: the reference cannot travel to the GPU box)" % (n, threads),
            "source": "tests/golden/reference_train_cpu_timing.json"}


# algorithmic FLOPs of a training step per ray: the forward as the inference path's (every sample of every query), and the backward of the two passes' MAIN queries —
# data and weight gradients, 2 x the forward's MACs each; the offset copies and the reflected rays run under no_grad in the reference (ibl_nerf_renderer.py:358-361, :442-448)
F_TRAIN_BWD_PER_RAY = 2.0 * (N_SAMPLES + N_SAMPLES + N_IMPORTANCE) * F_FULL


def train_bench(args):
    """`bench.py --train`: the training step of train.py:286-297 / :326-441 / :479-481 on the fused path — render_decomp with trainable networks (perturb = 1: stratified
    jitter + stochastic fine samples), the dataset-free losses, loss.backward() through the one autograd Function of ibl-nerf_amd/training.py, Adam — per ray count; the
    headline is the reference's default batch, N_rand = 4 096 (config_parser.py:65).  One JSON line."""
    import torch
    pkg = _pkg.load()
    from ibl_nerf_amd import renderer as R
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import train_loss as TL          # (the losses of train.py that need no dataset: shared with the fixtures made from the reference's own loss.backward())
    torch.cuda.set_device(0)
    sdc, sdf = load_checkpoint(args.checkpoint)
    lut = torch.from_numpy(load_lut()).cuda()
    K, _ = camera()
    fl = float(K[0, 0])
    sizes = [int(v) for v in args.train_rays.split(",")]
    by_rays, line = {}, None
    for n in sizes:
        nets = _trainable_module(sdc).cuda(), _trainable_module(sdf).cuda()
        opt = torch.optim.Adam([p for net in nets for p in net.parameters()], lr=5e-4, betas=(0.9, 0.999))
        kw = dict(network_fn=nets[0], network_fine=nets[1], N_samples=N_SAMPLES, N_importance=N_IMPORTANCE, perturb=1.0, raw_noise_std=0.0, brdf_lut=lut, lut_coefficient="F",
                  gamma_correct=True, correct_depth_for_prefiltered_radiance_infer=True, epsilon=0.01, use_radiance_linear=False, lindisp=False, near=NEAR, far=FAR,
                  target_normal_map_for_radiance_calculation="normal_map_from_depth_gradient_epsilon", max_rays_per_launch=max(n, 1024))
        rng = np.random.RandomState(0)
        pix = rng.permutation(H * W)[:n]
        i, j = (pix % W).astype(np.float32), (pix // W).astype(np.float32)
        d = np.stack([(i - W / 2) / fl, -(j - H / 2) / fl, -np.ones_like(i)], -1).astype(np.float32)
        rays = torch.from_numpy(np.stack([np.zeros_like(d), d], 0)).cuda()
        tg = {k: torch.from_numpy(v).cuda() for k, v in TL.targets(rng, n).items()}

        def step(**over):
            res = R.render_decomp(H, W, K, chunk=n, rays=rays, gt_values={}, approximate_radiance=True, **dict(kw, **over))
            loss = TL.total_loss(torch, res, tg, True)
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss

        def timed(**over):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                last = step(**over)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / args.steps, last

        for _ in range(max(args.warmup, 2)):
            l0 = step()
        t_warm = time.perf_counter()                      # (a power-bound chip: the first second of work runs at boost clocks no training run sees — warm it up)
        while time.perf_counter() - t_warm < 1.5 and not args.no_extras:      # (--no-extras: a profiling run)
            step()
            torch.cuda.synchronize()
        # K timed steps between synchronisations, three times over, ALTERNATING between the product's default — the forward under a route (Renderer.training_lists;
        # measured every 64 steps inside the timed region, calls of >= 1 024 rays) — and the same steps with every sample of every query evaluated (train_lists = 0:
        # round 5's forward), after 1.5 s of untimed steps; the MEDIAN of each three.  (A 512-ray step is launch- and host-bound — 5 ms of ~300 launches on a shared host — and a block of twenty
        # 4 096-ray steps lasts 0.4 s, inside the clock transients of a power-bound chip: single repetitions scatter by 30 %, and the second thing timed sees a warmer
        # chip than the first — hence the alternation.)
        dts, dts_all = [], []
        for _ in range(3):
            dt_, loss = timed()
            dts.append(dt_)
            if args.no_extras:                            # (a profiling run: the default path's kernels only)
                continue
            step(train_lists=0)
            dts_all.append(timed(train_lists=0)[0])
            step()
        dt = sorted(dts)[1]                               # the median of the three
        # the same step in three parts (untimed extra steps)
        parts = []
        for _ in range(0 if args.no_extras else 3):
            torch.cuda.synchronize(); a = time.perf_counter()
            res = R.render_decomp(H, W, K, chunk=n, rays=rays, gt_values={}, approximate_radiance=True, **kw)
            torch.cuda.synchronize(); b = time.perf_counter()
            lv = TL.total_loss(torch, res, tg, True); opt.zero_grad(); lv.backward()
            torch.cuda.synchronize(); c = time.perf_counter()
            opt.step()
            torch.cuda.synchronize(); e = time.perf_counter()
            parts.append((b - a, c - b, e - c))
        pm = np.array(parts).min(0) if parts else [None] * 3
        r = R.renderer_for(dict(kw, _lazy_range_check=True))
        f_step = F_ALG_PER_RAY + F_TRAIN_BWD_PER_RAY
        ts = r.training_state()
        by_rays[str(n)] = {"rays_per_s": n / dt, "ms_per_step": 1e3 * dt, "render_ms": pm[0] and 1e3 * pm[0], "loss_backward_ms": pm[1] and 1e3 * pm[1], "adam_ms": pm[2] and 1e3 * pm[2],
                           "frac": n / dt * f_step / 1e12 / PEAK_BF16_TFLOPS, "ms_per_step_repetitions": [1e3 * v for v in dts],
                           "ms_per_step_every_sample": 1e3 * sorted(dts_all)[1] if dts_all else None, "ms_per_step_every_sample_repetitions": [1e3 * v for v in dts_all],
                           "route": None if ts is None else {k: ts[k] for k in ("step", "measured", "events", "near_misses")},
                           "loss_first": float(l0.detach()), "loss_last": float(loss.detach()),
                           "skipped_steps": int(getattr(r, "skipped_steps", 0)), "range_fallbacks": int(r.range_fallbacks),
                           "reference_in_build_container": reference_train_record(n)}
    head = str(args.train_headline if str(args.train_headline) in by_rays else sizes[-1])
    hb = by_rays[head]
    f_step = F_ALG_PER_RAY + F_TRAIN_BWD_PER_RAY
    line = {"metric": "training rays/sec (64c+128f samples, one step = render + losses + backward + Adam)", "value": hb["rays_per_s"], "unit": "rays/s", "n_gpus": 1,
            "steps": args.steps, "warmup": max(args.warmup, 2), "ms_per_step": hb["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16 (3 MFMA products on hi/lo splits, forward and backward; f16 operand stash with a power-of-two loss scale), fp32 accumulate, fp32 Adam",
            "data": "synthetic (fitted checkpoint as the starting point, seeded pixels of the synthetic 800x800 camera, seeded targets)",
            "config": {"workload": "training step of train.py:286-297 / :479-481 at N_rand = %s rays (the reference's default batch is 4096, config_parser.py:65), 64+128 samples, "
                                   "approximate_radiance=True, perturb = 1; the forward under a route measured on the step's rays every 64 steps (estimates + lists, Renderer.training_lists: "
                                   "calls of >= 1024 rays; by_rays.*.ms_per_step_every_sample = the same steps with every sample evaluated)" % head,
                       "checkpoint": args.checkpoint, "rays_per_step": int(head), "optimizer": "torch.optim.Adam(lr 5e-4) over both networks' 92 tensors", "parallelism": "single GPU"},
            "roofline": {"bound": "mfma", "achieved": hb["rays_per_s"] * f_step / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": hb["frac"], "traffic": None,
                         "flop_per_ray": f_step, "flop_forward_per_ray": F_ALG_PER_RAY, "flop_backward_per_ray": F_TRAIN_BWD_PER_RAY,
                         "kernel": "f16x3k::mlp_kernel<FULL|TRUNK> (forward) + f16x3k::mlp_kernel<NET_BWD> (forward recompute + data gradient + operand stash) + k_wgrad (weight gradients)",
                         "note": "algorithmic FLOPs: the forward priced as the inference path's (every sample of every query, once), the backward as data + weight gradients of the two "
                                 "passes' main queries (the offset copies and reflected rays carry no gradient in the reference); three f16 products per MAC are issued, counted once; "
                                 "a step of 512 rays is launch- and latency-bound (8 000 points per MLP launch on 256 CUs), see by_rays"},
            "cpu_baseline": hb["reference_in_build_container"] or {"value": None, "unit": "rays/s", "cores": 0, "kind": "reference", "sample": "not recorded"},
            "by_rays": by_rays}
    print(json.dumps(line), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rays-per-launch", type=int, default=327680,
                    help="rays per launch set of the library (iblnerf_options.max_rays_per_launch): the frame in two sets; 65536 (ten sets) is 1.2 %% slower, one set of 655360 0.1 %% faster")
    ap.add_argument("--inference-min", action="store_true",
                    help="coarse pass evaluates density only (no coarse '0' maps): SURVEY.md §8 d mode (ii)")
    ap.add_argument("--checkpoint", choices=["fitted", "fitted2", "fitted3", "synthetic"], default="fitted",
                    help="fitted: the checkpoint with surfaces (tests/golden/fitted_ckpt.npz); synthetic: round 1's random-init networks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra frames (inference-minimum mode, other product schemes): profiling runs")
    ap.add_argument("--mlp-precision", choices=["auto", "f16x3_mxfp6x", "f16x3_mxfp6", "f16x3", "f16x3_main", "f16_mxfp6", "f16_mixed", "bf16x3"], default="auto",
                    help="matrix-core product scheme of the fused MLP kernel (include/iblnerf.h: mlp_precision); the default is "
                         "the renderer's default, the mode that holds parity on a checkpoint with surfaces")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the frame exchange: nccl (= RCCL, the default) | gloo (test hook: several ranks on one GPU, host-staged)")
    ap.add_argument("--partition", choices=["interleaved", "contiguous"], default="interleaved", help="how the frame's rows are dealt to the ranks (dist.tile_row_indices)")
    ap.add_argument("--tile-times", action="store_true", help="also time each of the 8 contiguous and 8 interleaved row tiles of the frame by itself on this one GPU "
                                                              "(the balance of an 8-rank frame, measured without an 8-GPU node) and report them as `tile_ms`")
    ap.add_argument("--train", action="store_true", help="time the TRAINING step instead (render + losses + backward + Adam; SURVEY.md 8 f-3): one JSON line of its own")
    ap.add_argument("--train-rays", default="512,1024,4096", help="--train: ray counts of the steps timed")
    ap.add_argument("--train-headline", type=int, default=4096, help="--train: the ray count `value` is quoted on (the reference's default N_rand)")
    ap.add_argument("--query-routing", default="", help="comma-separated iblnerf_options.query_routing names (A/B measurements, e.g. point_batch)")
    ap.add_argument("--single-stream", action="store_true", help="render every call on one context and one stream (Renderer.pair_streams = False): profiling runs, "
                                                                 "whose per-kernel durations should not include another stream's kernels")
    args = ap.parse_args()
    routing = [n for n in args.query_routing.split(",") if n]
    if args.train:
        return train_bench(args)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process (this process has not touched HIP and never
        # will; nothing is exec'd).  Rank 0 of the child prints the one JSON line on the stdout it inherits from us.
        return launch_ranks(args.gpus, sys.argv[1:])

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s): pass the same number to --nproc-per-node" % (args.gpus, world))
    # --backend gloo is a test hook: it lets the N>1 code path run with several ranks sharing the one GPU of a
    # test box (RCCL refuses two ranks on one device).  The driver's runs use the default, RCCL.
    backend = args.backend
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # a process group whenever torch.distributed.run launched us — also with one rank, so that `--gpus 1` under the launcher takes the
    # same pack + all-gather path (RCCL on device buffers) as the multi-GPU runs; a plain `python bench.py` has no group and no exchange
    grouped = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ
    if grouped:
        # (RCCL prints its version banner on STDOUT when the communicator comes up; this command's stdout is ONE JSON line: the banner goes to stderr)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            # bring the RCCL communicator (rings over xGMI) up before anything is timed, even with --warmup 0
            if backend == "nccl":
                probe = torch.zeros(world * 256, device="cuda")
                dist.all_gather_into_tensor(probe, torch.ones(256, device="cuda"))
                torch.cuda.synchronize()
                del probe
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    pkg = _pkg.load()
    from ibl_nerf_amd import checkpoint as ck, dist as D, renderer as R
    sdc, sdf = load_checkpoint(args.checkpoint)
    lut = load_lut()
    K, c2w = camera()
    r = R.Renderer(N_SAMPLES, N_IMPORTANCE, coarse_outputs=not args.inference_min,
                   max_rays_per_launch=args.rays_per_launch, mlp_precision=args.mlp_precision, query_routing=routing or 0)
    r.load_weights(0, sdc)
    r.load_weights(1, sdf)
    r.load_lut(lut)
    r.pair_streams = not args.single_stream
    # What a frame is rendered under — its ROUTE (which queries run as estimate + list, on which estimates: Renderer._measure_route) and, for mlp_precision="auto",
    # its precision table (FAST / SAFE: Renderer._measure_table) — is measured PER FRAME since round 6, inside the render call and inside the timed region, on 4 096
    # seeded pixels of the whole frame (the same on every rank: `probe`): the answer depends on the camera, so a view must not inherit another view's.
    # `decision_ms_per_frame` reports what that costs.
    rows = D.tile_row_indices(H, rank, world, args.partition)      # interleaved: rank, rank + world, ... (a ray's cost depends on what it sees: every rank gets the same mix)
    n_rows = len(rows)
    ro, rd = r.get_rays_strided(H, W, K, c2w, rows.start, rows.step, n_rows)      # this rank's rows only; rays resident in HBM before the timed region
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    probe = D.frame_probe_for_call(r, H, W, K, c2w, NEAR, FAR)

    def step(events=None):
        maps = r.render_rays(ro, rd, NEAR, FAR, probe=probe, alarm_sync=D.alarm_sync() if grouped else None)
        if grouped:
            if events:
                events[0].record()
            buf, _ = D.pack_maps(maps, D.EXPORT_KEYS, n_rows, W)
            if events:
                events[1].record()
            D.all_gather_frame(buf, H, W, partition=args.partition)
            if events:
                events[2].record()
        return maps

    def fence():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        maps = step()
    fence()
    dt = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the same frame with the coarse pass reduced to the density the fine sampling needs (no coarse
    # '...0' maps; SURVEY.md §8 d mode ii) — reported as an extra, never as `value`
    value_min = None
    if world == 1 and not args.inference_min and not args.no_extras:
        r2 = R.Renderer(N_SAMPLES, N_IMPORTANCE, coarse_outputs=False, max_rays_per_launch=args.rays_per_launch,
                        mlp_precision=args.mlp_precision)
        r2.load_weights(0, sdc)
        r2.load_weights(1, sdf)
        r2.load_lut(lut)
        r2.render_rays(ro, rd, NEAR, FAR, probe=probe)           # (one untimed frame: allocations, code objects, the second context of the stream pair)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        r2.render_rays(ro, rd, NEAR, FAR, probe=probe)
        torch.cuda.synchronize()
        value_min = H * W / (time.perf_counter() - t1)
        del r2

    # the same frame in the other product schemes, one frame each after one untimed frame — reported as extras, never as `value`
    by_precision = {}
    if world == 1 and args.mlp_precision == "auto" and not args.inference_min and not args.no_extras:
        for mode in ("f16x3_mxfp6x", "f16x3_mxfp6", "f16_mxfp6"):
            r3 = R.Renderer(N_SAMPLES, N_IMPORTANCE, max_rays_per_launch=args.rays_per_launch, mlp_precision=mode)
            r3.load_weights(0, sdc)
            r3.load_weights(1, sdf)
            r3.load_lut(lut)
            r3.render_rays(ro, rd, NEAR, FAR, probe=probe)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            r3.render_rays(ro, rd, NEAR, FAR, probe=probe)
            torch.cuda.synchronize()
            by_precision[mode] = H * W / (time.perf_counter() - t1)
            del r3

    # the same frame of the OTHER checkpoints (the headline is scene-dependent since round 4: what a ray costs depends on what it sees) — extras, never `value`
    by_checkpoint = {}
    if world == 1 and not args.inference_min and not args.no_extras:
        for kind in ("fitted", "fitted2", "fitted3", "synthetic"):
            if kind == args.checkpoint or (kind.startswith("fitted") and not os.path.exists(os.path.join(ROOT, "tests", "golden", kind + "_ckpt.npz"))):
                continue
            r4 = R.Renderer(N_SAMPLES, N_IMPORTANCE, max_rays_per_launch=args.rays_per_launch, mlp_precision=args.mlp_precision)
            c4, f4 = load_checkpoint(kind)
            r4.load_weights(0, c4)
            r4.load_weights(1, f4)
            r4.load_lut(lut)
            for _ in range(2):                                      # (two untimed frames: allocations, code objects; a checkpoint that answers an f16 range event by
                r4.render_rays(ro, rd, NEAR, FAR, probe=probe)      # rescaling does so in the first, and builds the second context of its stream pair in the second)
            torch.cuda.synchronize()
            trips0 = r4.trips
            t1 = time.perf_counter()
            r4.render_rays(ro, rd, NEAR, FAR, probe=probe)          # the frame as every frame is rendered: its own decision, its own tripped rays repeated
            torch.cuda.synchronize()
            by_checkpoint[kind] = {"value": H * W / (time.perf_counter() - t1), "decision": (r4.policy or {}).get("decision"), "route": r4.get_route(),
                                   "rays_repeated_by_the_tripwire": r4.trips - trips0, "probe_escalations": (r4.route or {}).get("probe_escalations"),
                                   "streams": 2 if r4._pair is not None else 1}
            del r4

    # the balance of an 8-rank frame on ONE GPU: each of the 8 contiguous bands and each of the 8 interleaved row sets of the frame rendered by itself
    tile_ms = None
    if world == 1 and args.tile_times:
        fo, fd = r.get_rays(H, W, K, c2w)
        tile_ms = {}
        for part in ("contiguous", "interleaved"):
            ms = []
            for tr in range(8):
                rr = D.tile_row_indices(H, tr, 8, part)
                ts = slice(rr.start, rr.stop, rr.step)
                to, td = fo[ts].reshape(-1, 3).contiguous(), fd[ts].reshape(-1, 3).contiguous()
                r.render_rays(to, td, NEAR, FAR, probe=probe)
                torch.cuda.synchronize()
                best = 1e30
                for _ in range(2):
                    t1 = time.perf_counter()
                    r.render_rays(to, td, NEAR, FAR, probe=probe)
                    torch.cuda.synchronize()
                    best = min(best, 1e3 * (time.perf_counter() - t1))
                ms.append(best)
            tile_ms[part] = {"ms": ms, "max_over_mean": max(ms) / (sum(ms) / len(ms))}

    # dominant kernel (fused MLP), HIP events around every launch on the launch stream (untimed extra step)
    # what the per-frame decision costs: the same measurement on its own (untimed extra)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    r._decide_for_call(ro, rd, NEAR, FAR, None, {}, probe)
    torch.cuda.synchronize()
    decision_ms = 1e3 * (time.perf_counter() - t1)
    policy = r.policy
    # (the profiled frame on ONE stream: the per-launch durations behind frac_mlp_only / executed / avg_launch_ms are then those of kernels running alone, comparable
    # with the rocprofv3 summaries; the timed frames above ran their two halves on two streams unless --single-stream)
    paired, r.pair_streams = r.pair_streams, False
    r.set_profiling(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if grouped else None
    trips_before = r.trips
    step(ev)
    torch.cuda.synchronize()
    # (the events are those of the step's LAST library call: the frame's own render — its probe renders ran before it — unless the tripwire marked rays, whose repeat
    # then is the last call; `profiled_call` says which)
    profiled_call = "frame render" if r.trips == trips_before else "repeat of %d tripped rays (the MLP-only figures below describe that call, not the frame)" % (r.trips - trips_before)
    mlp_ms, n_launch, flop = r.last_mlp_time()
    flop_executed, selection, slot_units = r.last_executed_flops(), r.last_selection(), r.last_slot_units()
    r.set_profiling(False)
    r.pair_streams = paired
    # the exchange step on its own (same untimed extra step; events on torch's current stream, where pack and all-gather are enqueued):
    # packing the export maps into one buffer, and the all-gather (host-staged under the gloo test hook)
    pack_ms = ev[0].elapsed_time(ev[1]) if grouped else None
    gather_ms = ev[1].elapsed_time(ev[2]) if grouped else None
    achieved_mlp_only = flop / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0
    value = H * W * args.steps / dt
    f_alg = F_ALG_PER_RAY_MIN if args.inference_min else F_ALG_PER_RAY
    achieved = value * f_alg / 1e12                     # TFLOP/s of the WHOLE driver-timed step: every kernel, every probe render, the exchange

    if rank == 0:
        traffic, traffic_src = pmc_traffic(args.mlp_precision)
        line = {
            "metric": "rays/sec (64c+128f samples) at 800x800 Kitchen; PSNR vs ref",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": {"bf16x3": "bf16 (3 MFMA products on hi/lo splits), fp32 accumulate"}.get(
                args.mlp_precision, "f16 (MFMA products per query class: 3x f16 on hi/lo splits, or f16 + 2x MX-fp6 residuals), fp32 accumulate"),
            "dtype_detail": MODES[args.mlp_precision][0],
            "data": ("synthetic (checkpoint fitted to an analytic scene with the reference's own modules, reference state-dict schema; synthetic pinhole camera)"
                     if args.checkpoint.startswith("fitted") else "synthetic (seeded random-init checkpoint in the reference state-dict schema, synthetic pinhole camera)"),
            "config": {"workload": "Kitchen 800x800 full test view, 64+128 samples, eps-normal + reflected pass"
                                   + (", inference-minimum coarse pass" if args.inference_min else ", full result dict incl. coarse '0' maps"),
                       "checkpoint": args.checkpoint, "rays_per_frame": H * W, "rays_per_launch": args.rays_per_launch,
                       "mlp_precision": args.mlp_precision, "policy": policy, "route": r.get_route(), "route_table": r.describe_route().splitlines(),
                       "decision_scope": "per frame, inside the timed step: route + precision table measured on 4 096 seeded pixels of the frame (the same on every rank)",
                       "decision_ms_per_frame": decision_ms, "partition": args.partition,
                       "streams": 2 if (r.pair_streams and H * W // world >= R.Renderer.PAIR_MIN_RAYS) else 1, "rays_repeated_by_the_tripwire": r.trips, "probe_escalations": r.probe_escalations,
                       **({"query_routing": routing} if routing else {}),
                       "parallelism": ("ray-tile x%d + %s all-gather" % (world, "RCCL" if backend == "nccl" else backend))
                                      if grouped else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         # frac = value x F_alg / peak: the driver-timed number, nothing else in the denominator (round 5 divided by MLP-launch time only; that figure is
                         # frac_mlp_only below)
                         "frac": achieved / PEAK_BF16_TFLOPS, "flop_per_ray": f_alg,
                         "frac_mlp_only": achieved_mlp_only / PEAK_BF16_TFLOPS, "achieved_mlp_only": achieved_mlp_only, "profiled_call": profiled_call,
                         "traffic": traffic,
                         # HBM bytes per camera ray (counter bytes of a frame's MLP launches / rays) over SURVEY 8(d)'s fused-ideal 236 B per ray:
                         # points in, raw rows out and back in; ~0.2 % of HBM bandwidth at this frame rate, so a ratio to watch, not a time bound
                         "traffic_ratio": (traffic * n_launch / (H * W // world) / 236.0) if traffic else None,
                         "traffic_note": "a committed figure, not a counter read in this run: HBM bytes per launch (reads x2-corrected + writes) of the rocprofv3 PMC pass %s; points in + raw outputs out, weights stay in L2" % traffic_src,
                         "kernel": MODES[args.mlp_precision][1] + ("" if args.mlp_precision == "auto" else "<FULL|TRUNK|REFL>"), "launches_per_step": n_launch,
                         "avg_launch_ms": mlp_ms / max(n_launch, 1), "mlp_share_of_step": mlp_ms / (1e3 * dt / args.steps),
                         "range_fallbacks": r.range_fallbacks,
                         # what the launches really evaluated: `achieved` prices every sample of every query as the reference evaluates it (SURVEY 8 d); the kernels
                         # run the coarse main / coarse-offset / reflected queries as a trunk-only density estimate everywhere and the rest on the relevant samples
                         "executed": {"tflops": flop_executed / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0,
                                      "frac": (flop_executed / (mlp_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if mlp_ms > 0 else 0.0,
                                      # ... and over the whole driver-timed step: MACs the frame's render really evaluated x 2 x frames / wall time
                                      "frac_whole_step": flop_executed * world * args.steps / dt / 1e12 / PEAK_BF16_TFLOPS,
                                      "share_of_algorithmic": flop_executed / flop if flop else None,
                                      "samples_refined": selection[0], "of_candidates": selection[1],
                                      # what governs speed on a power-bound chip (STATE.md section 2): per launch points x 64-MAC groups x the matrix slots its
                                      # product scheme spends per group (plain f16 4, f16 + 2 fp6 6, mixed trunk 7.5, three f16 products 12, 15-slot 15), per ray
                                      "slot_units_per_ray": slot_units / max(n_rows * W, 1),
                                      "slot_unit": "64 MACs of one sample x the matrix slots its product scheme spends on them (plain f16 4, f16 + 2 fp6 6, mixed trunk 7.5, 3 x f16 12, 15-slot 15, fp32 64), summed over the last step's MLP launches",
                                      "ps_per_slot_unit": (mlp_ms * 1e9 / max(slot_units, 1.0)) if mlp_ms > 0 else None,
                                      "note": "2 x nn.Linear MACs of the launches as run (estimates: trunk only; head layers, 15-slot densities and refinements: selected samples only)"},
                         "note": "algorithmic FLOPs (2 x nn.Linear MACs) counted once; per MAC the kernel issues " + MODES[args.mlp_precision][2]},
        }
        if grouped:
            line["pack_ms"], line["gather_ms"] = pack_ms, gather_ms
            line["config"]["ranks_seen"] = dist.get_world_size()
            try:
                line["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
            except Exception:       # (a torch build without the query: the line must not depend on it)
                line["rccl_version"] = None
            line["exchange"] = {"backend": "RCCL" if backend == "nccl" else backend, "bytes_per_rank": int(n_rows * W * 4 * sum(3 if k in R.MAP_KEYS_3 else 1 for k in D.EXPORT_KEYS)),
                                "note": "rank 0's pack (torch.cat of the export maps) and all-gather of one untimed extra frame; both are inside the timed step"}
        if value_min is not None:
            line["value_inference_min"] = value_min
        if by_precision:
            line["value_by_mlp_precision"] = dict(by_precision, **{args.mlp_precision: line["value"]})
        if by_checkpoint:
            line["value_by_checkpoint"] = dict({k: v["value"] for k, v in by_checkpoint.items()}, **{args.checkpoint: line["value"]})
            line["by_checkpoint"] = by_checkpoint
            fitted = {k: v for k, v in line["value_by_checkpoint"].items() if k.startswith("fitted")}
            worst = min(fitted, key=fitted.get)
            # the headline beside its worst case: the slowest of the fitted scenes (the random-init "synthetic" fog is no scene: every sample relevant, lists off)
            line["value_worst_checkpoint"] = {"checkpoint": worst, "value": fitted[worst], "ratio_to_best": fitted[worst] / max(fitted.values()),
                                              "frac": fitted[worst] * f_alg / 1e12 / PEAK_BF16_TFLOPS}
        if tile_ms:
            line["tile_ms"] = tile_ms
        if world == 1 and not args.no_cpu_baseline:
            color = maps["color_map"]

            def gpu_color(idx, key="color_map"):
                return maps[key][torch.as_tensor(idx, device=color.device)].cpu().numpy()

            if args.checkpoint == "fitted" and not args.inference_min:
                line["parity_vs_reference"] = parity_vs_reference_fixture(gpu_color)
            line["psnr_note"] = ("psnr_vs_ref_db: colour PSNR of the HIP frame against the pinned fp32 C restatement of the reference path (oracle/csrc, "
                                 "tests/test_oracle_c.py pins it to the reference's fixtures) on the CPU sample's rays - the reference itself cannot travel to this box; "
                                 "parity_vs_reference holds the comparison with the reference's own render at the pixels it rendered in the build container")
            try:
                line["cpu_baseline"], line["psnr_vs_ref_db"] = cpu_baseline(gpu_color, args.checkpoint)
            except Exception as e:      # the CPU leg must never cost the GPU line (e.g. no gcc on a host that did not receive oracle/_build/)
                line["cpu_baseline"] = {"value": None, "unit": "rays/s", "cores": 0, "kind": "port", "sample": "not measured", "error": "%s: %s" % (type(e).__name__, e),
                                        "reference_in_build_container": reference_cpu_record(args.checkpoint)}
        print(json.dumps(line), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
