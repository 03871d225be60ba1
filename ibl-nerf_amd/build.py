"""Build recipe for the HIP library (csrc/ -> ibl-nerf_amd/libiblnerf_hip.so), gfx950 only.

    python ibl-nerf_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so is built IN-TREE (git-ignored, but it travels to
the GPU box with the gpurun snapshot).  render_kernels.hip is compiled with -ffp-contract=off:
the reference's elementwise torch ops round multiply and add separately and sample positions
are sensitive to that (see the file header).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libiblnerf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-I" + os.path.join(os.path.dirname(HERE), "include")]
# (source, extra flags placed before -c[, object name]); every instantiation of the two MLP kernels is its own object
# so that they compile side by side (each takes a minute or more)
# accumulator chains of the MX kernel in arch VGPRs (see run_layer in mlp_kernel_mx.hip)
MX_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
SOURCES = [
    ("mlp_kernel.hip", ["-DIBL_VARIANT=0"], "mlp_kernel_full"),
    ("mlp_kernel.hip", ["-DIBL_VARIANT=1"], "mlp_kernel_trunk"),
    ("mlp_kernel.hip", ["-DIBL_VARIANT=2"], "mlp_kernel_refl"),
    ("mlp_kernel.hip", ["-DIBL_VARIANT=3"], "mlp_kernel_full_ci"),
    ("mlp_kernel.hip", ["-DIBL_VARIANT=4"], "mlp_kernel_refl_ci"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=0"], "mlp_kernel_f16x3_full"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=1"], "mlp_kernel_f16x3_trunk"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=2"], "mlp_kernel_f16x3_refl"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=3"], "mlp_kernel_f16x3_full_ci"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=4"], "mlp_kernel_f16x3_refl_ci"),
    ("mlp_kernel.hip", ["-DIBL_VARIANT=6"], "mlp_kernel_trunk_grad"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=6"], "mlp_kernel_f16x3_trunk_grad"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=7"], "mlp_kernel_f16x3_trunk_bwd"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=8"], "mlp_kernel_f16x3_trunk_feat"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=9"], "mlp_kernel_f16x3_trunk_bwd_feat"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=10"], "mlp_kernel_f16x3_trunk_feat2"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=11"], "mlp_kernel_f16x3_trunk_bwd_feat2"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=12"], "mlp_kernel_f16x3_net_bwd"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=16"], "mlp_kernel_f16x3_full_list"),
    ("mlp_kernel.hip", ["-DIBL_F16X3", "-DIBL_VARIANT=18"], "mlp_kernel_f16x3_trunk_list"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_VARIANT=0"], "mlp_kernel_mx_full"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_VARIANT=1"], "mlp_kernel_mx_trunk"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_VARIANT=2"], "mlp_kernel_mx_refl"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_VARIANT=5"], "mlp_kernel_mx_trunk_x"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_VARIANT=13"], "mlp_kernel_mx_trunk_p"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_VARIANT=15"], "mlp_kernel_mx_refl_list"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_VARIANT=16"], "mlp_kernel_mx_full_list"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_VARIANT=17"], "mlp_kernel_mx_trunk_x_list"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_VARIANT=3"], "mlp_kernel_mx_full_ci"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_VARIANT=4"], "mlp_kernel_mx_refl_ci"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_F16ONLY", "-DIBL_MX_VARIANT=0"], "mlp_kernel_mx16_full"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_F16ONLY", "-DIBL_MX_VARIANT=1"], "mlp_kernel_mx16_trunk"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_F16ONLY", "-DIBL_MX_VARIANT=2"], "mlp_kernel_mx16_refl"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_F16ONLY", "-DIBL_MX_VARIANT=3"], "mlp_kernel_mx16_full_ci"),
    ("mlp_kernel_mx.hip", MX_FLAGS + ["-DIBL_MX_F16ONLY", "-DIBL_MX_VARIANT=4"], "mlp_kernel_mx16_refl_ci"),
    ("render_kernels.hip", ["-ffp-contract=off"]),
    ("pack_kernels.hip", ["-ffp-contract=off"]),
    ("posdir_kernel.hip", []),
    ("range_kernel.hip", []),
    ("trunk_fp32_kernel.hip", []),
    ("generic_mlp.hip", []),
    ("wgrad_kernel.hip", []),
    ("api.cpp", ["-x", "hip"]),
    ("pack.cpp", ["-x", "hip"]),
]


def _deps(path, seen=None):
    """The source plus every local header it includes, transitively (`#include "..."`, relative to the including file)."""
    import re
    seen = seen if seen is not None else []
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.append(path)
    with open(path) as f:
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', f.read(), re.M):
            _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _digest(paths, extra):
    h = hashlib.sha256(" ".join(extra).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(verbose=True, force=False):
    """Compile every source whose inputs changed, then link.  Returns the library path."""
    os.makedirs(OBJ, exist_ok=True)
    objs, relink, jobs = [], force or not os.path.exists(LIB), []
    for entry in SOURCES:
        src, flags = entry[0], entry[1]
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, (entry[2] if len(entry) > 2 else src) + ".o")
        stamp = op + ".sha"
        dg = _digest(_deps(sp), [f for f in COMMON if not f.startswith("-I")] + flags)   # (not the -I path: the tree may live anywhere)
        old = open(stamp).read() if os.path.exists(stamp) else ""
        if force or old != dg or not os.path.exists(op):
            jobs.append(([HIPCC] + COMMON + flags + ["-c", sp, "-o", op], stamp, dg))
        objs.append(op)
    if jobs:   # the two MLP kernels take minutes each: compile the stale sources side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(job):
            cmd, stamp, dg = job
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp, "w") as f:
                f.write(dg)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as pool:
            list(pool.map(run, jobs))
        relink = True
    if relink:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
