#!/usr/bin/env python3
"""Times the REFERENCE's PyTorch-CPU render path in this container, on the weights the
fixtures use: the seeded synthetic checkpoint of bench.py and the fitted checkpoint.  Build container only.

    python tests/golden/time_reference_cpu.py [n_rays]

Writes tests/golden/reference_cpu_timing.json (rays/s per checkpoint and thread count, host description, weight checksums): the file
bench.py's `cpu_baseline.reference_in_build_container` is read from — the reference cannot travel to the GPU box, its timing can.
Run it on an otherwise idle container.
"""
import json
import platform
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

ck = MG.ck


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    torch, R, M, Hh = MG.import_reference()
    lut = MG.load_lut(torch)
    tmp = tempfile.mkdtemp()
    try:
        _, kw, *_ = M.create_IBLNeRF(MG.reference_args(tmp, 128))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kw.update(near=0.5, far=8.0)
    kw["brdf_lut"] = lut
    rng = np.random.RandomState(0)
    o, d, _, focal = MG.camera_rays(rng, n)
    K = np.array([[focal, 0, 400], [0, focal, 400], [0, 0, 1]], dtype=np.float32)
    rays = torch.from_numpy(np.stack([o, d], 0))
    sets = {"synthetic": (ck.synthetic_state_dict(0), ck.synthetic_state_dict(1)), "fitted": MG.fitted_state_dicts()}
    cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")]
    record = {"what": "the reference's own PyTorch-CPU render_decomp (approximate_radiance=True, 64+128 samples, chunk 1024) on seeded pixels of the "
                      "800x800 bench view, best of the timed runs after a 64-ray warm-up",
              "script": "tests/golden/time_reference_cpu.py", "n_rays": n, "host": {"cpu": cpu[0] if cpu else platform.processor(), "logical_cpus": os.cpu_count()},
              "torch": torch.__version__, "checkpoints": {}}
    for name, (sdc, sdf) in sets.items():
        entry = record["checkpoints"][name] = {"weights_checksum": [ck.blob_checksum(ck.state_dict_to_blob(sdc)), ck.blob_checksum(ck.state_dict_to_blob(sdf))], "threads": {}}
        kw["network_fn"].load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdc.items()})
        kw["network_fine"].load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdf.items()})
        for threads in (8, 1):
            torch.set_num_threads(threads)
            best = 1e9
            with torch.no_grad():
                R.render_decomp(800, 800, K, chunk=1024, rays=rays[:, :64], gt_values={}, approximate_radiance=True, **kw, **MG.EDIT_KEYS_OFF)
                for _ in range(2 if threads == 8 else 1):
                    t0 = time.perf_counter()
                    R.render_decomp(800, 800, K, chunk=1024, rays=rays, gt_values={}, approximate_radiance=True, **kw, **MG.EDIT_KEYS_OFF)
                    best = min(best, time.perf_counter() - t0)
            entry["threads"][str(threads)] = {"seconds": round(best, 3), "rays_per_s": round(n / best, 1)}
            print("%-34s %5d rays  %d threads  %7.2f s  %7.1f rays/s" % (name, n, threads, best, n / best), flush=True)
    with open(os.path.join(HERE, "reference_cpu_timing.json"), "w") as f:
        json.dump(record, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
