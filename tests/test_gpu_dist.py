"""The sharded path (BASELINE configs 3 and 5) on a GPU: two ranks under torch.distributed.run.  On a one-GPU box both ranks
share the device and exchange over gloo (RCCL refuses two ranks on one device; `--backend gloo`); with two or more
GPUs visible the same tests run one rank per GPU over RCCL.  The launcher is a fresh child process (nothing is exec'd from this
GPU-initialised process other than through subprocess)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _launch(script_args, timeout=900, nproc=2):
    n_gpu = torch.cuda.device_count()
    assert n_gpu >= 1, "GPU tests need a HIP device"
    backend = "gloo" if n_gpu < nproc else "nccl"          # RCCL refuses two ranks on one device: those runs exchange over gloo, host-staged
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args + ["--backend", backend]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT), backend


def test_two_rank_bench_step():
    """bench.py --gpus 2: each rank renders its 400-row tile of the 800x800 frame on the HIP path, the ranks pack and all-gather
    the export maps; rank 0 prints the one JSON line with n_gpus = 2."""
    out, backend = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["unit"] == "rays/s" and d["steps"] == 1
    assert abs(d["value"] - 640000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert ("RCCL" if backend == "nccl" else backend) in d["config"]["parallelism"] and "x2" in d["config"]["parallelism"]
    assert d["roofline"]["range_fallbacks"] == 0 and d["roofline"]["launches_per_step"] > 0


def test_eight_rank_bench_step():
    """bench.py --gpus 8 as the driver launches it on an 8-GPU node (BASELINE configs 3 and 5): eight ranks, 100-row tiles, the padded flat
    all-gather — here with the eight ranks sharing the one GPU of the test box over gloo (with eight GPUs visible: one each, RCCL)."""
    out, backend = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"], nproc=8, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and "x8" in d["config"]["parallelism"]
    assert abs(d["value"] - 640000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["pack_ms"] > 0 and d["gather_ms"] > 0 and d["exchange"]["bytes_per_rank"] == 100 * 800 * 4 * 47
    assert d["roofline"]["range_fallbacks"] == 0


def test_two_rank_frame_with_overrides_is_bit_identical():
    """dist.render_frame from two ranks under object-insertion and material-edit gt_values: the gathered frame equals the
    frame one rank renders alone, bit for bit, on every rank (tests/dist_gpu_worker.py)."""
    out, _ = _launch([os.path.join(ROOT, "tests", "dist_gpu_worker.py")])
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "DIST_OK 0" in out.stdout and "DIST_OK 1" in out.stdout


def test_eight_rank_frame_of_the_fitted_checkpoint_is_the_one_rank_frame():
    """The frame bench.py times — fitted checkpoint, 800x800, default renderer: estimates + lists, predicted offset copies, exact-fp32 coarse density — from EIGHT
    ranks (interleaved rows, every rank measuring the route on the same seeded probe pixels; here sharing the one GPU over gloo) against the same frame rendered by one
    rank in one call: every export map bit for bit on every rank, one route and one table decision everywhere (tests/dist_gpu_worker.py --fitted-frame).  Round 4's
    per-rank first-launch decisions could not promise this."""
    out, _ = _launch([os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "--fitted-frame"], nproc=8, timeout=1500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert all(("DIST_OK %d" % i) in out.stdout for i in range(8))


def test_eight_rank_frame_of_the_checkpoint_that_trips_is_the_one_rank_frame():
    """the SECOND fitted checkpoint's frame — the one whose whole frame trips the estimate wire (a positive density whose estimate lay below half the
    selection margin, on a ray or two of 640 000) — from eight ranks against one rank: bit for bit on every export map, the tiles' repeated rays adding up to the frame's
    (tests/dist_gpu_worker.py --fitted-frame --checkpoint fitted2).  Round 5's per-context margin doubling made the tripping rank render its tile twice, under other
    margins than its peers."""
    out, _ = _launch([os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "--fitted-frame", "--checkpoint", "fitted2"], nproc=8, timeout=1800)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert all(("DIST_OK %d" % i) in out.stdout for i in range(8))


def test_two_rank_view_sharded_export_is_the_one_rank_export():
    """Three views (frontal, rotated, frontal) of the first fitted checkpoint dealt to two ranks (render_views.test / dist.view_indices) against the same export from one
    rank: every exported map of every view bit for bit — each view decides its route and table on its own rays, so neither the rank nor what it rendered before matters
    (tests/dist_gpu_worker.py --views)."""
    out, _ = _launch([os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "--views"], nproc=2, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "DIST_OK 0" in out.stdout and "DIST_OK 1" in out.stdout


def test_rccl_one_rank_frame_on_device_buffers():
    """The RCCL leg on one GPU: a world-size-1 `nccl` process group (librccl loaded, communicator up) and dist.render_frame's pack ->
    all_gather_into_tensor on DEVICE buffers -> unpack, under the insert / edit gt_values — the branch of dist.all_gather_frame that
    the two-rank tests cannot reach on a one-GPU box (they exchange over gloo)."""
    out, backend = _launch([os.path.join(ROOT, "tests", "dist_gpu_worker.py")], nproc=1)
    assert backend == "nccl"
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "DIST_OK 0" in out.stdout and "BACKEND nccl" in out.stdout


def test_rccl_one_rank_bench_line():
    """bench.py --gpus 1 under torch.distributed.run: the group is initialised, the step packs and all-gathers over RCCL, and the JSON
    line carries pack_ms / gather_ms so that the first multi-GPU run explains itself."""
    out, backend = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"], nproc=1)
    assert backend == "nccl"
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and "RCCL" in d["config"]["parallelism"]
    assert d["pack_ms"] > 0 and d["gather_ms"] > 0 and d["exchange"]["backend"] == "RCCL"
    assert d["pack_ms"] + d["gather_ms"] < 0.1 * d["ms_per_step"]          # the exchange is a few per cent of a frame at most


def _plain_bench(extra, timeout=900):
    """`python bench.py ...` exactly as the driver types it: no launcher in front."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_invocation_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (the form the driver uses): bench.py starts its two ranks itself as a child
    `torch.distributed.run` and relays rank 0's one JSON line; `--gpus 1` stays a single process without a group."""
    n_gpu = torch.cuda.device_count()
    backend = "gloo" if n_gpu < 2 else "nccl"
    d = _plain_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--backend", backend])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and "x2" in d["config"]["parallelism"]
    assert d["pack_ms"] > 0 and d["gather_ms"] > 0 and "rccl_version" in d
    assert (d["rccl_version"] is not None) == (backend == "nccl")
    assert abs(d["value"] - 640000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    d1 = _plain_bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"])
    assert d1["n_gpus"] == 1 and d1["config"]["parallelism"] == "single GPU" and "pack_ms" not in d1
