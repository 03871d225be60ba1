"""CPU ORACLE (test infrastructure, NOT the product): builds oracle/_build/libiblnerf_cpu.so from oracle/csrc/*.c with gcc.

    python oracle/build_cpu.py          (also called by __graft_entry__.build())

The dense layer (csrc/gemm.c) is compiled three times — AVX-512, AVX2 + FMA, baseline x86-64 — and picked at run time
(iblnerf_cpu_isa()), so one library serves the build container and the GPU box's host CPU; everything else (csrc/render.c) is compiled
for baseline x86-64 with -ffp-contract=off.  A stamp of sources + flags skips the rebuild when nothing changed."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libiblnerf_cpu.so")
COMMON = ["-O2", "-fPIC", "-fopenmp", "-std=gnu11", "-Wall", "-Wno-unused-function", "-fno-math-errno"]
GEMM = [("avx512", ["-mavx512f", "-mavx2", "-mfma", "-DGEMM_VB=64", "-ffp-contract=fast"]),
        ("avx2", ["-mavx2", "-mfma", "-DGEMM_VB=32", "-ffp-contract=fast"]),
        ("base", ["-DGEMM_VB=16", "-ffp-contract=off"])]


def _digest():
    h = hashlib.sha256()
    for f in ("csrc/gemm.c", "csrc/render.c", "iblnerf_cpu.h", "../include/iblnerf.h", "build_cpu.py"):
        h.update(open(os.path.join(HERE, f), "rb").read())
    return h.hexdigest()


def build(verbose=False):
    os.makedirs(OUT, exist_ok=True)
    stamp = os.path.join(OUT, "stamp")
    dig = _digest()
    if os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    objs = []
    for isa, flags in GEMM:
        o = os.path.join(OUT, "gemm_%s.o" % isa)
        cmd = ["gcc", "-c", os.path.join(HERE, "csrc", "gemm.c"), "-o", o, "-DGEMM_ISA=" + isa] + COMMON + ["-O3"] + flags
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    o = os.path.join(OUT, "render.o")
    cmd = ["gcc", "-c", os.path.join(HERE, "csrc", "render.c"), "-o", o, "-ffp-contract=off"] + COMMON
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    objs.append(o)
    cmd = ["gcc", "-shared", "-o", LIB] + objs + ["-fopenmp", "-lm"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    open(stamp, "w").write(dig)
    return LIB


if __name__ == "__main__":
    print(build(verbose="-q" not in sys.argv))
