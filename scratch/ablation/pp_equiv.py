import importlib.util, os, re, subprocess, sys
spec = importlib.util.spec_from_file_location("b", "/root/repo/ibl-nerf_amd/build.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
pairs = {"mlp_kernel_mx.hip": "/tmp/mx_stripped.hip", "mlp_kernel.hip": "/tmp/k_stripped.hip"}
def pp(path, flags, incdir):
    cmd = [b.HIPCC] + [f for f in b.COMMON if f != "-O3"] + flags + ["-I", incdir, "-E", "--cuda-device-only", "-x", "hip", path, "-o", "-"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    txt = "\n".join(l for l in out.stdout.split("\n") if not l.startswith("#"))
    # keep only what comes from the kernel file itself: everything after the last system-header chunk is hard to isolate, so compare whole streams
    return re.sub(r"\s+", " ", txt)
bad = 0
for entry in b.SOURCES:
    src, flags = entry[0], entry[1]
    if src not in pairs:
        continue
    a = pp(os.path.join(b.CSRC, src), flags, b.CSRC)
    c = pp(pairs[src], flags, b.CSRC)
    same = a == c
    print(src, " ".join(flags), "IDENTICAL" if same else "DIFFERENT", len(a))
    bad += not same
sys.exit(bad)
