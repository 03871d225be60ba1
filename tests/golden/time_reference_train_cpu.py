#!/usr/bin/env python3
"""Times the REFERENCE's own training step on the CPU of this container (build container only; the reference cannot travel to the GPU box, its measured timing can):
render_decomp with render_kwargs_train (perturb = 1, trainable network_fn / network_fine; train.py:286-297) -> the losses of train.py that need no dataset
(tests/train_loss.py, :326-441) -> loss.backward() -> Adam step (train.py:479-481), on seeded pixels of the 800x800 bench view, fitted checkpoint, approximate_radiance=True.

    python tests/golden/time_reference_train_cpu.py [n_rays ...]      (default 512)

Writes tests/golden/reference_train_cpu_timing.json: seconds per step and rays/s per ray count and thread count — what `bench.py --train` reports as
`cpu_baseline.reference_in_build_container`.  Run it on an otherwise idle container."""
import json
import os
import platform
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as MG  # noqa: E402
import train_loss as TL  # noqa: E402

ck = MG.ck


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512]
    torch, R, M, Hh = MG.import_reference()
    lut = MG.load_lut(torch)
    tmp = tempfile.mkdtemp()
    try:
        kw, _, _, _, grad_vars, optimizer = M.create_IBLNeRF(MG.reference_args(tmp, 128))      # [0] = render_kwargs_train; the reference's own Adam over its own parameter list
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert kw["perturb"] == 1.0
    sdc, sdf = MG.fitted_state_dicts()
    kw["network_fn"].load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdc.items()})
    kw["network_fine"].load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdf.items()})
    kw.update(near=0.5, far=8.0)
    kw["brdf_lut"] = lut
    cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")]
    record = {"what": "the reference's own training step on the CPU: render_decomp(render_kwargs_train: perturb = 1, chunk 32768, approximate_radiance=True, 64+128 samples) "
                      "-> losses of train.py:326-441 that need no dataset (tests/train_loss.py) -> loss.backward() -> the reference's Adam step, fitted checkpoint, seeded pixels of "
                      "the 800x800 bench view; best of the timed steps after one warm-up step",
              "script": "tests/golden/time_reference_train_cpu.py", "host": {"cpu": cpu[0] if cpu else platform.processor(), "logical_cpus": os.cpu_count()},
              "torch": torch.__version__, "weights_checksum": [ck.blob_checksum(ck.state_dict_to_blob(sdc)), ck.blob_checksum(ck.state_dict_to_blob(sdf))], "rays": {}}
    for n in sizes:
        rng = np.random.RandomState(0)
        o, d, _, focal = MG.camera_rays(rng, n)
        K = np.array([[focal, 0, 400], [0, focal, 400], [0, 0, 1]], dtype=np.float32)
        rays = torch.from_numpy(np.stack([o, d], 0))
        tg = {k: torch.from_numpy(v) for k, v in TL.targets(rng, n).items()}
        entry = record["rays"][str(n)] = {"threads": {}}
        for threads in (8,):
            torch.set_num_threads(threads)
            best, parts = 1e9, None
            for it in range(3):
                t0 = time.perf_counter()
                res = R.render_decomp(800, 800, K, chunk=32768, rays=rays, gt_values={}, approximate_radiance=True, **kw, **MG.EDIT_KEYS_OFF)
                t1 = time.perf_counter()
                loss = TL.total_loss(torch, res, tg, True)
                optimizer.zero_grad()
                loss.backward()
                t2 = time.perf_counter()
                optimizer.step()
                t3 = time.perf_counter()
                if it > 0 and t3 - t0 < best:
                    best, parts = t3 - t0, (t1 - t0, t2 - t1, t3 - t2)
                print("n %d threads %d step %d: %.2f s (render %.2f, loss + backward %.2f, Adam %.3f) loss %.4f" % (n, threads, it, t3 - t0, t1 - t0, t2 - t1, t3 - t2, float(loss)), flush=True)
            entry["threads"][str(threads)] = {"seconds_per_step": round(best, 3), "rays_per_s": round(n / best, 2),
                                              "render_s": round(parts[0], 3), "loss_backward_s": round(parts[1], 3), "adam_s": round(parts[2], 4)}
    out = os.path.join(HERE, "reference_train_cpu_timing.json")
    with open(out, "w") as f:
        json.dump(record, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
