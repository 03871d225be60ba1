#!/usr/bin/env python3
"""Times the REFERENCE's PyTorch-CPU render path in this container, on the weights the
fixtures use: the seeded synthetic checkpoint of bench.py and the fitted checkpoint.  Build container only.

    python tests/golden/time_reference_cpu.py [n_rays]
"""
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
    sets = {"synthetic seeds 0/1 (bench.py)": (ck.synthetic_state_dict(0), ck.synthetic_state_dict(1)), "fitted": MG.fitted_state_dicts()}
    for name, (sdc, sdf) in sets.items():
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
            print("%-34s %5d rays  %d threads  %7.2f s  %7.1f rays/s" % (name, n, threads, best, n / best), flush=True)


if __name__ == "__main__":
    main()
