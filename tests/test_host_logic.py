"""CPU-only tests of the host side: the C-ABI library loads and exports what include/iblnerf.h
declares, the weight packer's MFMA fragment layout reproduces the oracle MLP when the MFMA lane
semantics are emulated in numpy, the encoder's sin/cos meets its error bound, checkpoint I/O and
the create_IBLNeRF mirror follow the reference's rules.  No compute call touches a GPU."""
import ctypes as C
import os
import re
import tempfile

import numpy as np
import pytest

import iblnerf_oracle as O
from conftest import ROOT
from ibl_nerf_amd import binding as B
from ibl_nerf_amd import checkpoint as ck
from ibl_nerf_amd import dist as D
from ibl_nerf_amd import model as M


@pytest.fixture(scope="module")
def lib():
    return B.load_library()


def test_library_exports_every_declared_symbol(lib):
    import glob
    hdr = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))))      # the boundary (iblnerf.h) and the measurement hooks (iblnerf_experimental.h)
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(iblnerf_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(B.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.iblnerf_blob_floats() == ck.N_PARAMS == 798994


def test_struct_sizes_match_header_layout(tmp_path):
    """ctypes mirrors vs the real header, measured by compiling it with gcc."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "iblnerf.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu",'
                   'sizeof(iblnerf_options), sizeof(iblnerf_overrides), sizeof(iblnerf_maps), sizeof(iblnerf_outputs),'
                   'offsetof(iblnerf_overrides, d_mask), offsetof(iblnerf_overrides, irradiance_list),'
                   'offsetof(iblnerf_outputs, z_std));return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(B.Options), C.sizeof(B.Overrides), C.sizeof(B.Maps), C.sizeof(B.Outputs),
            B.Overrides.d_mask.offset, B.Overrides.irradiance_list.offset, B.Outputs.z_std.offset]
    assert got == want


def test_integration_stub_matches_options_struct():
    """The ctypes stub printed in INTEGRATION.md must declare the same iblnerf_options fields as binding.py
    (a shorter struct there would make iblnerf_create read past the caller's memory)."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = md[md.index("class Options(C.Structure)"):md.index("opt = Options()")]
    assert re.findall(r'\("(\w+)", C\.c_(?:int32|float)\)', stub) == [n for n, _ in B.Options._fields_]


def test_ctypes_structs_mirror_the_header():
    """Every struct of include/iblnerf.h against its ctypes mirror in binding.py (which oracle/iblnerf_cpu.py shares): field names in order,
    array lengths, and the kind of each field (int32 / float / pointer) — a missing or reordered field would shift every later one."""
    hdr = open(os.path.join(ROOT, "include", "iblnerf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    mirrors = {"iblnerf_options": B.Options, "iblnerf_overrides": B.Overrides, "iblnerf_maps": B.Maps, "iblnerf_outputs": B.Outputs,
               "iblnerf_sampling": B.Sampling, "iblnerf_taps": B.Taps, "iblnerf_stage_inputs": B.StageInputs}
    seen = 0
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(\w+);", hdr, flags=re.S):
        if name not in mirrors:
            continue
        seen += 1
        want = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(?:const )?(int32_t|float|iblnerf_maps|unsigned char)\s*(\*?)\s*(.*)$", decl)
            assert m, (name, decl)
            base, ptr, rest = m.groups()
            for item in rest.split(","):
                item = item.strip()
                star = ptr == "*" or item.startswith("*")
                am = re.match(r"\*?\s*(\w+)(?:\[(\d+)\])?$", item)
                assert am, (name, item)
                kind = "ptr" if star else base
                want.append((am.group(1), kind, int(am.group(2) or 0)))
        got = []
        for fname, ftype in mirrors[name]._fields_:
            n = getattr(ftype, "_length_", 0)
            el = getattr(ftype, "_type_", ftype) if n else ftype
            kind = {C.c_int32: "int32_t", C.c_float: "float", C.c_void_p: "ptr", B.Maps: "iblnerf_maps"}[el]
            got.append((fname, kind, n))
        assert got == want, (name, [a for a, b in zip(got, want) if a != b][:3], len(got), len(want))
    assert seen == len(mirrors)


def test_no_gpu_fails_loudly(lib):
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    o = B.default_options()
    ctx = C.c_void_p()
    assert lib.iblnerf_create(C.byref(o), C.byref(ctx)) == -3          # IBLNERF_ERR_HIP, no CPU fallback
    assert b"no usable HIP device" in lib.iblnerf_last_error(None)
    from ibl_nerf_amd import renderer as R
    with pytest.raises(B.IblNerfError):
        R.Renderer(64, 128)


def test_options_validation(lib):
    o = B.default_options()
    assert (o.n_samples, o.n_importance, o.gamma_correct, o.coarse_outputs) == (64, 128, 1, 1)
    o.n_samples, o.n_importance = 200, 128
    ctx = C.c_void_p()
    assert lib.iblnerf_create(C.byref(o), C.byref(ctx)) == -1
    o = B.default_options()
    o.max_rays_per_launch = 3_000_000                      # 4 * 3e6 * 192 points would overflow the 2^31 launch limit
    assert lib.iblnerf_create(C.byref(o), C.byref(ctx)) == -1
    assert b"2^31" in lib.iblnerf_last_error(None)


# --------------------------------------------------------------------------------------------
# encoder sin/cos (csrc/sincos_enc.h)
# --------------------------------------------------------------------------------------------
def test_encoder_sincos_error_bound(lib):
    rng = np.random.RandomState(0)
    xs = np.concatenate([rng.uniform(-16, 16, 3000), rng.uniform(-1e-2, 1e-2, 500), [0.0, 8.0, -8.0, 3.1415927]]).astype(np.float32)
    out = np.empty(20, dtype=np.float32)
    worst = 0.0
    for x in xs:
        lib.iblnerf_encode_host(C.c_float(float(x)), 10, out.ctypes.data)
        arg = np.float64(x) * (2.0 ** np.arange(10))
        worst = max(worst, np.abs(out[0::2] - np.sin(arg)).max(), np.abs(out[1::2] - np.cos(arg)).max())
    assert worst < 2.5e-7        # ~2 ulp of 1.0; 20x below the bf16x3 operand error it feeds


# --------------------------------------------------------------------------------------------
# packer layout vs oracle MLP, MFMA lane semantics emulated in numpy
# --------------------------------------------------------------------------------------------
KS_BYTES, CH = 2048, 16
CH_L0, CH_L1, CH_L5, CH_L6, CH_L7, CH_FEAT, CH_ALB, CH_IRR, CH_VIEW, CH_AR = 0, 2, 34, 44, 52, 60, 68, 72, 76, 85
CH_G7, CH_G6, CH_G5, CH_G4, CH_G0 = 97, 105, 113, 123, 155      # backward stream of the trunk (layout.h)
CH_GV, CH_GF, CH_GA, CH_GH = 157, 165, 173, 185                 # ... of feature_linear / views_linears.0, and the K-concatenated head tiles
TAB_BIAS, TAB_SIG, TAB_ROUGH, TAB_ALB, TAB_IRR, TAB_RAD, TAB_AR, TAB_SCALAR = 0, 3200, 3456, 3712, 4096, 4224, 4992, 6144


def bf16_rne(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def split(x):
    hi = bf16_rne(x)
    return hi.astype(np.float64), bf16_rne((x - hi).astype(np.float32)).astype(np.float64)


def acc_feature(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def enc_ref_index(sl, h, pph):
    n = 2 * pph
    if sl < n:
        m = pph * h + (sl >> 1)
        return 3 + 6 * (m // 3) + (3 if sl & 1 else 0) + (m % 3)
    if sl == n:
        return 0 if h == 0 else 2
    if sl == n + 1:
        return 1 if h == 0 else -1
    return -1


class Emu:
    """out[p,row] = sum_{h,e} A[(row,h),e] * B[(p,h),e] per k-step, three products on (hi,lo) splits."""

    def __init__(self, stream, tab):
        ks = np.frombuffer(stream, dtype=np.uint16).reshape(-1, 2, 64, 8)
        f = (ks.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        self.Ah, self.Al = f[:, 0].reshape(-1, 2, 32, 8), f[:, 1].reshape(-1, 2, 32, 8)  # [ks][h][row][e]
        self.tab = tab

    def frag_act(self, feat):              # [P,256] -> hi,lo [P, 16 ksteps, h, e]
        idx = np.empty((16, 2, 8), dtype=np.int64)
        for j in range(16):
            for h in range(2):
                for e in range(8):
                    idx[j, h, e] = 32 * (j >> 1) + acc_feature(8 * (j & 1) + e, h)
        return split(feat[:, idx])

    def frag_enc(self, embedded, pph, nk):  # reference-order embedding [P, 3+6L] -> [P, nk, h, e]
        idx = np.array([[[enc_ref_index(8 * jj + e, h, pph) for e in range(8)] for h in range(2)] for jj in range(nk)])
        v = np.where(idx >= 0, embedded[:, np.maximum(idx, 0)], 0.0).astype(np.float32)
        return split(v)

    def ksteps(self, ks0, Bh, Bl):         # k-steps ks0.. over fragments [P, n, h, e] -> [P, 32 rows]
        n = Bh.shape[1]
        Ah, Al = self.Ah[ks0:ks0 + n], self.Al[ks0:ks0 + n]
        return (np.einsum("khre,pkhe->pr", Ah, Bh) + np.einsum("khre,pkhe->pr", Ah, Bl)
                + np.einsum("khre,pkhe->pr", Al, Bh))

    def lane_vec(self, off, ntiles):       # lane-layout table -> plain feature vector
        t = self.tab[off:off + ntiles * 32].reshape(ntiles, 2, 16)
        v = np.empty(ntiles * 32)
        for tt in range(ntiles):
            for h in range(2):
                for r in range(16):
                    v[32 * tt + acc_feature(r, h)] = t[tt, h, r]
        return v

    def layer(self, chunk0, ntiles, act, enc=None, bias_tile=0, relu=True):
        nke = 0 if enc is None else enc[0].shape[1]
        per = nke + (16 if act is not None else 0)
        out = []
        for t in range(ntiles):
            ks = chunk0 * CH + t * per
            o = 0.0
            if enc is not None:
                o = o + self.ksteps(ks, *enc)
            if act is not None:
                o = o + self.ksteps(ks + nke, *act)
            out.append(o)
        out = np.concatenate(out, 1) + self.lane_vec(TAB_BIAS + bias_tile * 32, ntiles)
        out = out.astype(np.float32)
        return np.maximum(out, 0) if relu else out

    def back(self, chunk0, ntiles, dz):    # one backward layer: rows = the layer's input features, K = dZ of its 256 outputs, no bias
        a = self.frag_act(dz.astype(np.float32))
        return np.concatenate([self.ksteps((chunk0 + t) * CH, *a) for t in range(ntiles)], 1).astype(np.float32)

    def back_cat(self, chunk0, operands):
        """One K-concatenated backward layer (layout.h: CH_GA, CH_GH): 8 row tiles, per tile the k-steps of every operand in order;
        operands = [dZ [P, 16 * nk] ...] (128-wide ones: nk = 8)."""
        frs, nks = [], []
        for dz in operands:
            nk = dz.shape[1] // 16
            full = np.zeros((dz.shape[0], 256), np.float32)
            full[:, :dz.shape[1]] = dz
            hi, lo = self.frag_act(full)
            frs.append((hi[:, :nk], lo[:, :nk]))
            nks.append(nk)
        per = sum(nks)
        out = []
        for t in range(8):
            ks, o = chunk0 * CH + t * per, 0.0
            for (hi, lo), nk in zip(frs, nks):
                o = o + self.ksteps(ks, hi, lo)
                ks += nk
            out.append(o)
        return np.concatenate(out, 1).astype(np.float32)

    def head_input_gradients(self, pts, dirs, draw):
        """The head part of the VAR_NET_BWD program on the packed stream: FULL's forward, the 128-wide layers' dZ from the head tables, the two
        K-concatenated tiles and the transposed views / feature layers -> (dL/dh2 before the views layer's pass bits, dL/dh7 before the trunk's)."""
        emb = O.embed(pts, 10)
        pe = self.frag_enc(emb, 15, 4)
        h = self.layer(CH_L0, 8, None, pe, 0)
        for l in range(1, 5):
            h = self.layer(CH_L1 + 8 * (l - 1), 8, self.frag_act(h), None, 8 * l)
        h = self.layer(CH_L5, 8, self.frag_act(h), pe, 40)
        h = self.layer(CH_L6, 8, self.frag_act(h), None, 48)
        h7 = self.layer(CH_L7, 8, self.frag_act(h), None, 56)
        a7 = self.frag_act(h7)
        feat = self.layer(CH_FEAT, 8, a7, None, 64, relu=False)
        albf = self.layer(CH_ALB, 4, a7, None, 72)
        irrf = self.layer(CH_IRR, 4, a7, None, 76)
        de = self.frag_enc(O.embed(dirs, 4), 6, 2)
        h2 = self.layer(CH_VIEW, 8, self.frag_act(feat), de, 80)
        a2 = self.frag_act(h2)
        d = draw.astype(np.float64)
        dF = []
        for k in range(3):
            f = self.layer(CH_AR + 4 * k, 4, a2, None, 88 + 4 * k)
            wk = np.stack([self.lane_vec(TAB_AR + (3 * k + c) * 128, 4) for c in range(3)], 0)          # [3, 128]
            dF.append(((d[:, 9 + 3 * k:12 + 3 * k] @ wk) * (f > 0)).astype(np.float32))
        rad = np.stack([self.lane_vec(TAB_RAD + c * 256, 8) for c in range(3)], 0)
        dh2 = self.back_cat(CH_GA, [dF[2], np.concatenate([dF[0], dF[1]], 1)]) + d[:, 6:9] @ rad         # [ARF.2 | ARF.0 | ARF.1] + radiance_linear
        dzv = (dh2 * (h2 > 0)).astype(np.float32)
        dfeat = self.back(CH_GV, 8, dzv)
        alb = np.stack([self.lane_vec(TAB_ALB + c * 128, 4) for c in range(3)], 0)
        dfa = ((d[:, 1:4] @ alb) * (albf > 0)).astype(np.float32)
        dfi = ((d[:, 5:6] @ self.lane_vec(TAB_IRR, 4)[None, :]) * (irrf > 0)).astype(np.float32)
        dh7 = self.back_cat(CH_GH, [np.concatenate([dfa, dfi], 1), dfeat]) \
            + d[:, 0:1] * self.lane_vec(TAB_SIG, 8)[None, :] + d[:, 4:5] * self.lane_vec(TAB_ROUGH, 8)[None, :]
        assert np.abs(self.back(CH_GF, 8, dfeat) - self.back_cat(CH_GH, [np.zeros((dfeat.shape[0], 256), np.float32), dfeat])).max() <= 1e-6   # CH_GF = the feature_linear part of CH_GH
        return dh2, dh7

    def density_gradient(self, pts):
        """The VAR_TRUNK_GRAD program on the packed stream: forward keeping the pass masks, dZ(l-1) = (W(l)^T dZ(l)) * mask(l-1) on the
        transposed chunks, encoding tiles in slot order (row (r&3) + 8(r>>2) + 4h of tile t = slot 16t + r of half h), chain rule of the encoding."""
        emb = O.embed(pts, 10)
        pe = self.frag_enc(emb, 15, 4)
        hs = [self.layer(CH_L0, 8, None, pe, 0)]
        for l in range(1, 5):
            hs.append(self.layer(CH_L1 + 8 * (l - 1), 8, self.frag_act(hs[-1]), None, 8 * l))
        hs.append(self.layer(CH_L5, 8, self.frag_act(hs[-1]), pe, 40))
        hs.append(self.layer(CH_L6, 8, self.frag_act(hs[-1]), None, 48))
        hs.append(self.layer(CH_L7, 8, self.frag_act(hs[-1]), None, 56))
        dz = self.lane_vec(TAB_SIG, 8)[None, :] * (hs[7] > 0)
        dz = self.back(CH_G7, 8, dz) * (hs[6] > 0)
        dz = self.back(CH_G6, 8, dz) * (hs[5] > 0)
        full = self.back(CH_G5, 10, dz)
        genc = full[:, 256:].copy()
        dz = full[:, :256] * (hs[4] > 0)
        for l in range(4, 0, -1):
            dz = self.back(CH_G4 + 8 * (4 - l), 8, dz) * (hs[l - 1] > 0)
        genc = genc + self.back(CH_G0, 2, dz)
        grad = np.zeros((pts.shape[0], 3))
        for t in range(2):
            for row in range(32):
                h, r = (row >> 2) & 1, (row & 3) + 4 * (row >> 3)
                ref = enc_ref_index(16 * t + r, h, 15)
                if ref < 0:
                    assert np.all(genc[:, 32 * t + row] == 0)          # pad slot: a zero row of the stream
                    continue
                gcol = genc[:, 32 * t + row].astype(np.float64)
                if ref < 3:
                    grad[:, ref] += gcol
                    continue
                k, rem = divmod(ref - 3, 6)                             # [sin(2^k x) x3, cos(2^k x) x3]
                c, f = rem % 3, 2.0 ** k
                grad[:, c] += gcol * (f * emb[:, ref + 3] if rem < 3 else -f * emb[:, ref - 3])
        return grad.astype(np.float32)

    def forward(self, pts, dirs):
        pe = self.frag_enc(O.embed(pts, 10), 15, 4)
        h = self.layer(CH_L0, 8, None, pe, 0)
        for l in range(1, 5):
            h = self.layer(CH_L1 + 8 * (l - 1), 8, self.frag_act(h), None, 8 * l)
        h = self.layer(CH_L5, 8, self.frag_act(h), pe, 40)
        h = self.layer(CH_L6, 8, self.frag_act(h), None, 48)
        h7 = self.layer(CH_L7, 8, self.frag_act(h), None, 56)
        sc = self.tab[TAB_SCALAR:TAB_SCALAR + 18]
        sigma = h7 @ self.lane_vec(TAB_SIG, 8) + sc[0]
        if dirs is None:
            return sigma[:, None].astype(np.float32)
        a7 = self.frag_act(h7)
        feat = self.layer(CH_FEAT, 8, a7, None, 64, relu=False)
        albf = self.layer(CH_ALB, 4, a7, None, 72)
        irrf = self.layer(CH_IRR, 4, a7, None, 76)
        de = self.frag_enc(O.embed(dirs, 4), 6, 2)
        h2 = self.layer(CH_VIEW, 8, self.frag_act(feat), de, 80)
        a2 = self.frag_act(h2)
        cols = [sigma]
        cols += [albf @ self.lane_vec(TAB_ALB + c * 128, 4) + sc[1 + c] for c in range(3)]
        cols += [h7 @ self.lane_vec(TAB_ROUGH, 8) + sc[4], irrf @ self.lane_vec(TAB_IRR, 4) + sc[5]]
        cols += [h2 @ self.lane_vec(TAB_RAD + c * 256, 8) + sc[6 + c] for c in range(3)]
        for k in range(3):
            f = self.layer(CH_AR + 4 * k, 4, a2, None, 88 + 4 * k)
            cols += [f @ self.lane_vec(TAB_AR + (3 * k + c) * 128, 4) + sc[9 + 3 * k + c] for c in range(3)]
        return np.stack(cols, 1).astype(np.float32)


@pytest.mark.parametrize("gain", [1.0, 1.6])
def test_packed_stream_reproduces_oracle_mlp(lib, gain):
    sd = ck.synthetic_state_dict(seed=5, gain=gain)
    blob = ck.state_dict_to_blob(sd)
    stream = np.zeros(lib.iblnerf_stream_bytes(), dtype=np.uint8)
    tab = np.zeros(lib.iblnerf_table_floats(), dtype=np.float32)
    assert stream.size == (97 + 60 + 16 + 28) * 32768 and tab.size == 6176     # the network's layers + the backward streams (trunk; feature / views layers)
    rc = lib.iblnerf_pack_weights_host(blob.ctypes.data, blob.size, stream.ctypes.data, stream.size, tab.ctypes.data, tab.size)
    assert rc == 0
    assert lib.iblnerf_pack_weights_host(blob.ctypes.data, blob.size - 1, stream.ctypes.data, stream.size, tab.ctypes.data, tab.size) == -1
    rng = np.random.RandomState(3)
    pts = rng.uniform(-8, 8, (24, 3)).astype(np.float32)
    dirs = rng.uniform(-1.2, 1.2, (24, 3)).astype(np.float32)
    emu = Emu(stream.tobytes(), tab)
    ref = O.mlp_forward(sd, O.embed(pts, 10), O.embed(dirs, 4))
    got = emu.forward(pts, dirs)
    # three bf16 products on hi/lo splits: ~2^-17 per operand; 8 layers deep -> a few 1e-5 absolute
    assert np.abs(got - ref).max() <= (4e-5 if gain == 1.0 else 4e-4), np.abs(got - ref).max()
    ref_s = O.mlp_forward(sd, O.embed(pts, 10))
    assert np.abs(emu.forward(pts, None) - ref_s).max() <= (2e-5 if gain == 1.0 else 2e-4)
    # transpose / permutation detector: a wrong K order or row map gives O(0.1) errors, not 1e-5
    assert np.abs(ref).max() > 0.05
    # the backward stream (density-gradient query): the chain on the packed transposed chunks against the oracle's written-out backward
    # (itself pinned on the reference's autograd, test_autograd_normal_modes)
    sg, gg = O.density_gradient(sd, pts)
    ge = emu.density_gradient(pts)
    err = np.abs(ge - gg).max(-1) / np.abs(gg).max()
    assert np.median(err) <= (2e-5 if gain == 1.0 else 2e-4) and err.max() <= 1e-2, (np.median(err), err.max())   # (a pass bit may flip at 2^-17 operands)
    assert np.abs(gg).max() > 0.5
    # ... and the head layers' transposes (K-concatenated tiles of the whole-network backward): dL/dh2 and dL/dh7 for random dL/d raw against the
    # plain matrices
    if gain == 1.0:
        draw = rng.uniform(-1, 1, (24, 18)).astype(np.float32)
        dh2, dh7 = emu.head_input_gradients(pts, dirs, draw)
        W = lambda n: sd[n + ".weight"].astype(np.float64)
        h = O.embed(pts, 10); e0 = h
        for i in range(8):
            h = np.maximum(O._lin(sd, "positions_linears.%d" % i, h), 0)
            if i == 4:
                h = np.concatenate([e0, h], -1)
        h7 = h
        feat = O._lin(sd, "feature_linear", h7)
        zv = O._lin(sd, "views_linears.0", np.concatenate([feat, O.embed(dirs, 4)], -1))
        h2 = np.maximum(zv, 0)
        d = draw.astype(np.float64)
        r2 = d[:, 6:9] @ W("radiance_linear")
        for k in range(3):
            fk = O._lin(sd, "additional_radiance_feature_linear.%d" % k, h2)
            r2 = r2 + ((d[:, 9 + 3 * k:12 + 3 * k] @ W("additional_radiance_linear.%d" % k)) * (fk > 0)) @ W("additional_radiance_feature_linear.%d" % k)
        assert np.abs(dh2 - r2).max() <= 2e-5 * max(1.0, np.abs(r2).max()), np.abs(dh2 - r2).max()
        r7 = ((r2 * (zv > 0)) @ W("views_linears.0"))[:, :256] @ W("feature_linear") + d[:, 0:1] @ W("sigma_linear") + d[:, 4:5] @ W("roughness_linear")
        fa, fi = O._lin(sd, "albedo_feature_linear", h7), O._lin(sd, "irradiance_feature_linear", h7)
        r7 = r7 + ((d[:, 1:4] @ W("albedo_linear")) * (fa > 0)) @ W("albedo_feature_linear") + ((d[:, 5:6] @ W("irradiance_linear")) * (fi > 0)) @ W("irradiance_feature_linear")
        assert np.abs(dh7 - r7).max() <= 2e-5 * max(1.0, np.abs(r7).max()), np.abs(dh7 - r7).max()
        assert np.abs(r7).max() > 0.05 and np.abs(r2).max() > 0.05


# --------------------------------------------------------------------------------------------
# f16 + MX-fp6 stream (csrc/layout_mx.h): same check, the three MFMA forms of a K=64 block emulated
# --------------------------------------------------------------------------------------------
FP6_VAL = np.array([(m / 8.0 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 1)) * (-1.0 if s else 1.0)
                    for s in range(2) for e in range(4) for m in range(8)])
FP6_MAG = FP6_VAL[:32]


def fp6_round(x):
    """round-to-nearest-even onto the e2m3 grid, saturating at 7.5 (what v_cvt_scalef32_pk32_fp6_f16 does)."""
    a = np.minimum(np.abs(x), 7.5)
    step = np.where(a < 2, 0.125, np.where(a < 4, 0.25, 0.5))
    return np.sign(x) * np.minimum(np.rint(a / step) * step, 7.5)


def fp6_codes(dwords):
    """[..., 6] uint32 -> [..., 32] e2m3 values (slot j at bits [6j, 6j+6))."""
    bits = np.unpackbits(dwords.view(np.uint8), axis=-1, bitorder="little").reshape(dwords.shape[:-1] + (32, 6))
    return FP6_VAL[(bits * (1 << np.arange(6))).sum(-1)]


class EmuMX(Emu):
    MX_CH = dict(L0=0, L1=2, L5=34, L6=44, L7=52, FEAT=60, ALB=68, IRR=72, VIEW=76, AR=86)

    def __init__(self, stream, tab):
        blk = np.frombuffer(stream, dtype=np.uint8).reshape(-1, 8192)
        self.A16 = blk[:, :4096].copy().view(np.float16).reshape(-1, 4, 2, 32, 8).astype(np.float64)     # [blk][j][h][row][e]
        sc = blk[:, 7168:7424].copy().view(np.uint32).reshape(-1, 2, 32)                                   # [blk][h][row]

        def fp6(a_off, b_off, byte):
            d = np.concatenate([blk[:, a_off:a_off + 1024].copy().view(np.uint32).reshape(-1, 64, 4),
                                blk[:, b_off:b_off + 512].copy().view(np.uint32).reshape(-1, 64, 2)], -1)
            scale = 2.0 ** (((sc >> (8 * byte)) & 255).astype(np.float64) - 127)
            return fp6_codes(d).reshape(-1, 2, 32, 32) * scale[..., None]                                  # [blk][h][row][jj]
        self.W6, self.R6 = fp6(4096, 6144, 0), fp6(5120, 6656, 1)
        assert not blk[:, 7424:].any()
        self.tab = tab

    @staticmethod
    def operands(v):
        """v [P, nb, h, 32] fp32 values of each lane's block -> (f16, fp6 of the f16, fp6 of the residual), as
        the kernel builds them (finish_block in mlp_kernel_mx.hip)."""
        v = v.astype(np.float32)
        xh = v.astype(np.float16).astype(np.float32)
        xl = (v - xh).astype(np.float16).astype(np.float64)
        mx = np.maximum(np.abs(v).max(-1, keepdims=True).astype(np.float64), 2.0 ** -100)
        e = np.floor(np.log2(mx))
        sh, sl = 2.0 ** (e - 2), 2.0 ** (e - 14)
        return xh.astype(np.float64), fp6_round(xh / sh) * sh, fp6_round(xl / sl) * sl

    def frag_act(self, feat):
        idx = np.array([[[32 * (2 * b + (jj >> 4)) + acc_feature(8 * ((jj >> 3) & 1) + (jj & 7), h) for jj in range(32)]
                         for h in range(2)] for b in range(4)])
        return self.operands(feat[:, idx])

    def frag_enc(self, embedded, pph, nk=None):
        idx = np.array([[[enc_ref_index(jj, h, pph) for jj in range(32)] for h in range(2)]])
        return self.operands(np.where(idx >= 0, embedded[:, np.maximum(idx, 0)], 0.0))

    def blocks(self, b0, ops):
        xh, x6, l6 = ops
        n = xh.shape[1]
        A16 = self.A16[b0:b0 + n].transpose(0, 2, 3, 1, 4).reshape(n, 2, 32, 32)                          # [blk][h][row][jj = 8j + e]
        return (np.einsum("bhrj,pbhj->pr", A16, xh) + np.einsum("bhrj,pbhj->pr", self.W6[b0:b0 + n], l6)
                + np.einsum("bhrj,pbhj->pr", self.R6[b0:b0 + n], x6))

    def layer(self, chunk0, ntiles, act, enc=None, bias_tile=0, relu=True):
        per = (1 if enc is not None else 0) + (4 if act is not None else 0)
        out = []
        for t in range(ntiles):
            b = chunk0 * 4 + t * per
            o = 0.0
            if enc is not None:
                o = o + self.blocks(b, enc)
                b += 1
            if act is not None:
                o = o + self.blocks(b, act)
            out.append(o)
        out = (np.concatenate(out, 1) + self.lane_vec(TAB_BIAS + bias_tile * 32, ntiles)).astype(np.float32)
        return np.maximum(out, 0) if relu else out

    # ---- the 15-slot form (VAR_TRUNK_P, mlp_kernel_mx.hip): every trunk block as (network block, residual block 98 * 4 + r) ----
    @staticmethod
    def operands_p(v):
        """v [P, nb, h, 32] fp32 -> (Xh, Xl, fp6 Xh, fp6 Xl, fp6 X3) as split3_pair / finish_block_p / run_layer_p build them."""
        v = v.astype(np.float32)
        xh = v.astype(np.float16).astype(np.float32)
        r = v - xh                                                       # exact in fp32
        xl = r.astype(np.float16).astype(np.float32)
        x3 = (r - xl).astype(np.float64)                                 # exact; the kernel packs it as bf16 before the fp6 conversion
        x3 = (x3.astype(np.float32).view(np.uint32) + 0x7fff + ((x3.astype(np.float32).view(np.uint32) >> 16) & 1) & 0xffff0000).view(np.float32).astype(np.float64)
        mx = np.maximum(np.abs(v).max(-1, keepdims=True).astype(np.float64), 2.0 ** -100)
        e = np.floor(np.log2(mx))
        sh, sl, st = 2.0 ** (e - 2), 2.0 ** (e - 14), 2.0 ** np.maximum(e - 25, -28)
        return (xh.astype(np.float64), xl.astype(np.float64), fp6_round(xh / sh) * sh, fp6_round(xl.astype(np.float64) / sl) * sl, fp6_round(x3 / st) * st)

    def blocks_p(self, b0, ops):
        xh, xl, x6, l6, t6 = ops
        n = xh.shape[1]
        sel = lambda a, o: a[o + b0:o + b0 + n]
        R = 98 * 4
        A16 = lambda o: sel(self.A16, o).transpose(0, 2, 3, 1, 4).reshape(n, 2, 32, 32)
        Wh, Wl = A16(0), A16(R)
        ein = lambda W, X: np.einsum("bhrj,pbhj->pr", W, X)
        return (ein(Wh, xh) + ein(Wh, xl) + ein(Wl, xh)                  # three f16 products
                + ein(sel(self.W6, 0), t6) + ein(sel(self.R6, 0), l6) + ein(sel(self.W6, R), x6))   # fp6(Wh) fp6(X3) + fp6(Wl) fp6(Xl) + fp6(W3) fp6(Xh)

    def layer_p(self, chunk0, act, enc, bias_tile):
        per = (1 if enc is not None else 0) + (4 if act is not None else 0)
        out = []
        for t in range(8):
            b = chunk0 * 4 + t * per
            o = 0.0
            if enc is not None:
                o = o + self.blocks_p(b, enc)
                b += 1
            if act is not None:
                o = o + self.blocks_p(b, act)
            out.append(o)
        return np.maximum((np.concatenate(out, 1) + self.lane_vec(TAB_BIAS + bias_tile * 32, 8)).astype(np.float32), 0)

    def forward_p(self, pts):
        c = self.MX_CH
        act = lambda f: self.operands_p(f[:, np.array([[[32 * (2 * b + (jj >> 4)) + acc_feature(8 * ((jj >> 3) & 1) + (jj & 7), h) for jj in range(32)]
                                                         for h in range(2)] for b in range(4)])])
        idx = np.array([[[enc_ref_index(jj, h, 15) for jj in range(32)] for h in range(2)]])
        emb = O.embed(pts, 10)
        pe = self.operands_p(np.where(idx >= 0, emb[:, np.maximum(idx, 0)], 0.0))
        h = self.layer_p(c["L0"], None, pe, 0)
        for l in range(1, 5):
            h = self.layer_p(c["L1"] + 8 * (l - 1), act(h), None, 8 * l)
        h = self.layer_p(c["L5"], act(h), pe, 40)
        h = self.layer_p(c["L6"], act(h), None, 48)
        h7 = self.layer_p(c["L7"], act(h), None, 56)
        return (h7 @ self.lane_vec(TAB_SIG, 8) + self.tab[TAB_SCALAR])[:, None].astype(np.float32)

    def forward(self, pts, dirs):
        c = self.MX_CH
        pe = self.frag_enc(O.embed(pts, 10), 15)
        h = self.layer(c["L0"], 8, None, pe, 0)
        for l in range(1, 5):
            h = self.layer(c["L1"] + 8 * (l - 1), 8, self.frag_act(h), None, 8 * l)
        h = self.layer(c["L5"], 8, self.frag_act(h), pe, 40)
        h = self.layer(c["L6"], 8, self.frag_act(h), None, 48)
        h7 = self.layer(c["L7"], 8, self.frag_act(h), None, 56)
        sc = self.tab[TAB_SCALAR:TAB_SCALAR + 18]
        sigma = h7 @ self.lane_vec(TAB_SIG, 8) + sc[0]
        if dirs is None:
            return sigma[:, None].astype(np.float32)
        a7 = self.frag_act(h7)
        feat = self.layer(c["FEAT"], 8, a7, None, 64, relu=False)
        albf = self.layer(c["ALB"], 4, a7, None, 72)
        irrf = self.layer(c["IRR"], 4, a7, None, 76)
        h2 = self.layer(c["VIEW"], 8, self.frag_act(feat), self.frag_enc(O.embed(dirs, 4), 6), 80)
        a2 = self.frag_act(h2)
        cols = [sigma]
        cols += [albf @ self.lane_vec(TAB_ALB + k * 128, 4) + sc[1 + k] for k in range(3)]
        cols += [h7 @ self.lane_vec(TAB_ROUGH, 8) + sc[4], irrf @ self.lane_vec(TAB_IRR, 4) + sc[5]]
        cols += [h2 @ self.lane_vec(TAB_RAD + k * 256, 8) + sc[6 + k] for k in range(3)]
        for k in range(3):
            f = self.layer(c["AR"] + 4 * k, 4, a2, None, 88 + 4 * k)
            cols += [f @ self.lane_vec(TAB_AR + (3 * k + q) * 128, 4) + sc[9 + 3 * k + q] for q in range(3)]
        return np.stack(cols, 1).astype(np.float32)


@pytest.mark.parametrize("gain", [1.0, 1.6])
def test_packed_mx_stream_reproduces_oracle_mlp(lib, gain):
    sd = ck.synthetic_state_dict(seed=5, gain=gain)
    blob = ck.state_dict_to_blob(sd)
    stream = np.zeros(lib.iblnerf_stream_bytes_mx(), dtype=np.uint8)
    tab = np.zeros(lib.iblnerf_table_floats(), dtype=np.float32)
    assert stream.size == (98 + 60) * 32768            # the network's 98 chunks + the 240 residual blocks of the trunk (layout_mx.h: CH_RES)
    assert lib.iblnerf_pack_weights_host_mx(blob.ctypes.data, blob.size, stream.ctypes.data, stream.size, tab.ctypes.data, tab.size) == 0
    assert lib.iblnerf_pack_weights_host_mx(blob.ctypes.data, blob.size, stream.ctypes.data, stream.size - 1, tab.ctypes.data, tab.size) == -1
    emu = EmuMX(stream.tobytes(), tab)
    # the f16 part of one block is the f16 rounding of the right weights (positions_linears.2, tile 3, block 1)
    W = sd["positions_linears.2.weight"]
    blk = (2 + 8 + 3) * 4 + 1
    for h in range(2):
        for jj in (0, 9, 31):
            feat = 32 * (2 + (jj >> 4)) + acc_feature(8 * ((jj >> 3) & 1) + (jj & 7), h)
            assert np.array_equal(emu.A16[blk, jj >> 3, h, :, jj & 7], W[96:128, feat].astype(np.float16).astype(np.float64))
            assert np.abs(emu.W6[blk, h, :, jj] - W[96:128, feat]).max() <= 0.07 * np.abs(W[96:128]).max()   # 4 significant bits, block-scaled
    # residual blocks (the mixed TRUNK form's third product): f16(W - f16 W) of layer 1, tile 3, block 1 and of layer 0, tile 5, slot for slot
    W1, W0 = sd["positions_linears.1.weight"], sd["positions_linears.0.weight"]
    rb1, rb0 = 98 * 4 + 8 + 4 * 3 + 1, 98 * 4 + 5
    for h in range(2):
        for jj in (0, 9, 31):
            feat = 32 * (2 + (jj >> 4)) + acc_feature(8 * ((jj >> 3) & 1) + (jj & 7), h)
            w = W1[96:128, feat]
            want = (w - w.astype(np.float16).astype(np.float32)).astype(np.float16).astype(np.float64)
            assert np.array_equal(emu.A16[rb1, jj >> 3, h, :, jj & 7], want) and np.abs(want).max() > 0
            ref_i = enc_ref_index(jj, h, 15)
            w0 = W0[160:192, ref_i] if ref_i >= 0 else np.zeros(32, np.float32)
            assert np.array_equal(emu.A16[rb0, jj >> 3, h, :, jj & 7], (w0 - w0.astype(np.float16).astype(np.float32)).astype(np.float16).astype(np.float64))
    raw_blocks = np.frombuffer(stream.tobytes(), dtype=np.uint8).reshape(-1, 8192)
    # a residual block: the f16 area (Wl), the fp6(W) area with its scale byte (W3 = what two f16 terms leave of the weight), nothing else
    assert not raw_blocks[98 * 4:, 5120:6144].any() and not raw_blocks[98 * 4:, 6656:7168].any()
    assert not (raw_blocks[98 * 4:, 7168:7424].copy().view(np.uint32) >> 8).any()
    # ... residual block r belongs to trunk block r: positions_linears.6, tile 2, block 3 (and the third term itself, to fp6 precision of its block)
    W6_ = sd["positions_linears.6.weight"]
    nb6 = 44 * 4 + 4 * 2 + 3
    for h in range(2):
        feat = np.array([32 * (6 + (jj >> 4)) + acc_feature(8 * ((jj >> 3) & 1) + (jj & 7), h) for jj in range(32)])
        w = W6_[64:96][:, feat]                                            # [row][jj]
        wh = w.astype(np.float16).astype(np.float32)
        wl = (w - wh).astype(np.float16).astype(np.float32)
        got_l = emu.A16[98 * 4 + nb6][:, h].transpose(1, 0, 2).reshape(32, 32)
        assert np.array_equal(got_l, wl.astype(np.float64))
        w3 = (w - wh - wl).astype(np.float64)
        assert np.abs(w3).max() > 0
        assert np.abs(emu.W6[98 * 4 + nb6, h] - w3).max() <= 0.07 * np.abs(w3).max(-1).max()
    rng = np.random.RandomState(3)
    pts = rng.uniform(-8, 8, (24, 3)).astype(np.float32)
    dirs = rng.uniform(-1.2, 1.2, (24, 3)).astype(np.float32)
    ref = O.mlp_forward(sd, O.embed(pts, 10), O.embed(dirs, 4))
    got = emu.forward(pts, dirs)
    # f16 main term + fp6 residual terms: ~2^-16 per operand
    assert np.abs(got - ref).max() <= (4e-5 if gain == 1.0 else 4e-4), np.abs(got - ref).max()
    assert np.abs(emu.forward(pts, None) - O.mlp_forward(sd, O.embed(pts, 10))).max() <= (2e-5 if gain == 1.0 else 2e-4)
    # the 15-slot form against the same network in FLOAT64 (same float32 embedding): below what float32 arithmetic itself leaves (the fp32 oracle),
    # and several times below the three-f16-product form (the precise kernel's arithmetic, emulated here as the 15-slot form without its fp6 terms)
    sd64 = {k: v.astype(np.float64) for k, v in sd.items()}
    x64 = O.embed(pts, 10).astype(np.float64)
    h = x64
    for l in range(8):
        h = np.maximum((np.concatenate([x64, h], -1) if l == 5 else h) @ sd64["positions_linears.%d.weight" % l].T + sd64["positions_linears.%d.bias" % l], 0)
    s64 = h @ sd64["sigma_linear.weight"].T + sd64["sigma_linear.bias"]
    e_p = np.abs(emu.forward_p(pts) - s64).max()
    e_32 = np.abs(O.mlp_forward(sd, O.embed(pts, 10)) - s64).max()
    blocks_p = EmuMX.blocks_p
    try:
        EmuMX.blocks_p = lambda self, b0, ops: blocks_p(self, b0, ops[:2] + tuple(np.zeros_like(o) for o in ops[2:]))
        e_3 = np.abs(emu.forward_p(pts) - s64).max()
    finally:
        EmuMX.blocks_p = blocks_p
    assert e_p <= 1.5 * e_32 + 1e-7 and e_p <= 0.5 * e_3, (e_p, e_32, e_3)


# --------------------------------------------------------------------------------------------
# checkpoint + factory mirror
# --------------------------------------------------------------------------------------------
def test_checkpoint_roundtrip_and_discovery():
    torch = pytest.importorskip("torch")
    sdc, sdf = ck.synthetic_state_dict(10), ck.synthetic_state_dict(11, gain=1.3)
    blob = ck.state_dict_to_blob(sdc)
    back = ck.blob_to_state_dict(blob)
    assert list(back.keys()) == list(sdc.keys())
    assert all(np.array_equal(back[k], sdc[k]) for k in sdc)
    assert sdc["sigma_linear.bias"][0] == np.float32(0.3)
    with pytest.raises(ValueError):
        ck.blob_to_state_dict(blob[:-1])
    bad = dict(sdc)
    bad["albedo_linear.weight"] = np.zeros((3, 64), np.float32)
    with pytest.raises(ValueError):
        ck.state_dict_to_blob(bad)
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "exp"))
        ck.save_checkpoint(os.path.join(d, "exp", "000100.tar"), 100, sdc, sdf)
        ck.save_checkpoint(os.path.join(d, "exp", "200000.tar"), 200000, sdf, sdc)
        open(os.path.join(d, "exp", "args.txt"), "w").write("x")
        assert ck.find_checkpoint(d, "exp").endswith("200000.tar")            # lexicographically last '*tar*'
        assert ck.find_checkpoint(d, "exp", target_load_N_iter=100).endswith("000100.tar")
        assert ck.find_checkpoint(d, "exp", ft_path="/x/y.tar") == "/x/y.tar"
        step, c, f = ck.load_checkpoint(os.path.join(d, "exp", "000100.tar"))
        assert step == 100 and torch.is_tensor(c["sigma_linear.weight"])
        assert np.array_equal(ck.state_dict_to_blob(c), blob)
        # the factory follows the same discovery rule and returns reference-shaped kwargs
        args = M.default_args(basedir=d, expname="exp", no_reload=False)
        train, test, start, _, _, _ = M.create_IBLNeRF(args)
        assert start == 200000 and test["perturb"] is False and test["raw_noise_std"] == 0 and train["perturb"] == 1.0
        assert np.array_equal(ck.state_dict_to_blob(test["network_fn"].state_dict()), ck.state_dict_to_blob(sdf))
        assert np.array_equal(ck.state_dict_to_blob(test["network_fine"].state_dict()), blob)
        for k in ("network_query_fn", "N_samples", "N_importance", "lut_coefficient", "gamma_correct", "epsilon",
                  "target_normal_map_for_radiance_calculation", "correct_depth_for_prefiltered_radiance_infer"):
            assert k in test
    with tempfile.TemporaryDirectory() as d3:
        os.makedirs(os.path.join(d3, "exp"))
        kw3 = M.create_IBLNeRF(M.default_args(basedir=d3, infer_depth=True))[1]       # a PositionDirectionMLP (ibl_nerf.py:293-297)
        assert kw3["depth_mlp"].out_ch == 1 and kw3["infer_depth"] is True and kw3["visibility_mlp"] is None
        assert list(kw3["depth_mlp"].state_dict())[-2:] == ["final_linear.weight", "final_linear.bias"]
        assert ck.posdir_blob(kw3["depth_mlp"].state_dict()).size == 644865
    with tempfile.TemporaryDirectory() as d2:
        os.makedirs(os.path.join(d2, "exp"))
        assert M.create_IBLNeRF(M.default_args(basedir=d2, infer_normal=True))[1]["normal_mlp"].out_ch == 3
    # a smaller architecture is a container of its own shapes (evaluated inside the built one at upload: checkpoint.embed_architecture); a larger one is refused
    small = M.IBLNeRF(D=6, W=128, input_ch=39, input_ch_views=15)
    assert small.arch == (6, 128, 6, 2) and ck.arch_of(small.state_dict()) == (6, 128, 6, 2)
    small.load_state_dict(ck.synthetic_arch_state_dict(3, (6, 128, 6, 2)))
    with pytest.raises(ValueError):
        small.load_state_dict(ck.synthetic_state_dict(3))
    # (round 6: a LARGER one is a container too — evaluated layer by layer, csrc/generic_mlp.hip; what stays refused: D = 5, odd or huge widths, other skips / radiance counts)
    for arch_kw, arch in ((dict(W=512), (8, 512, 10, 4)), (dict(D=9), (9, 256, 10, 4)), (dict(input_ch=69), (8, 256, 11, 4)), (dict(D=12, W=384, input_ch_views=33), (12, 384, 10, 5))):
        big = M.IBLNeRF(**arch_kw)
        assert big.arch == arch and ck.is_generic_arch(arch) and not ck.is_member_of_built(arch)
        big.load_state_dict(ck.synthetic_arch_state_dict(5, arch))
        assert ck.arch_blob(big.state_dict()).size == sum(o * i + o for _, o, i in ck.arch_schema(*arch))
    with pytest.raises(NotImplementedError):
        M.IBLNeRF(W=512, is_color_independent_to_direction=True)
    for bad in (dict(D=5), dict(W=257), dict(W=8192), dict(D=40), dict(input_ch=70), dict(coarse_radiance_number=2), dict(skips=(3,))):
        with pytest.raises(NotImplementedError):
            M.IBLNeRF(**bad)
    # ... the auxiliary networks likewise (create_IBLNeRF builds them with the same netdepth / netwidth / multires, ibl_nerf.py:293-323)
    with tempfile.TemporaryDirectory() as d3:
        os.makedirs(os.path.join(d3, "exp"))
        kw4 = M.create_IBLNeRF(M.default_args(basedir=d3, netdepth=6, netwidth=128, multires=6, multires_views=2, infer_albedo_separate=True, infer_normal=True, infer_depth=True))[1]
    assert kw4["albedo_mlp"].arch == (6, 128, 6) and kw4["normal_mlp"].out_ch == 3 and kw4["depth_mlp"].arch == (6, 128, 6, 2)
    sd_small = ck.synthetic_position_mlp(4, 3, 1.0, (6, 128, 6))
    kw4["albedo_mlp"].load_state_dict(sd_small)
    big = ck.embed_position_mlp(sd_small)
    assert big["out_linears.weight"].shape == (3, 256) and ck.embed_position_mlp(big) is big and ck.aux_channel_blob(sd_small, 1).size == ck.N_PARAMS
    pd = ck.embed_position_direction_mlp(kw4["depth_mlp"].state_dict())
    assert list(pd) == [n + t for n, _, _ in ck.posdir_schema(1) for t in (".weight", ".bias")] and np.array_equal(pd["views_linears.3.weight"][:64, :64], np.eye(64, dtype=np.float32)) and float(np.abs(pd["views_linears.3.weight"]).sum()) == 64.0      # (D // 2 = 3 view layers of W // 2 = 64: the fourth is the identity on the live units)
    assert ck.posdir_blob(kw4["depth_mlp"].state_dict()).size == 644865
    with pytest.raises(NotImplementedError):
        M.PositionMLP(D=8, W=512)


def test_unsupported_flags_raise():
    from ibl_nerf_amd import renderer as R
    base = dict(approximate_radiance=True)
    R._check_supported(base)
    R._check_supported(dict(base, lindisp=True, use_radiance_linear=True))     # built flag variants
    R._check_supported(dict(base, calculate_albedo_from_gt=True, calculate_roughness_from_gt=True,
                            calculate_irradiance_from_gt=True, depth_map_from_ground_truth=True))
    R._check_supported(dict(base, white_bkgd=True, retraw=True, use_environment_map=True))   # dead flags in the reference too
    with pytest.raises(TypeError):
        R._check_supported(dict(base, infer_normal=True))                          # needs the normal_mlp kwarg
    R._check_supported(dict(base, infer_normal=True, normal_mlp=object(), target_normal_map_for_radiance_calculation="inferred_normal_map"))
    R._check_supported(dict(base, infer_normal=True, infer_normal_at_surface=True, normal_mlp=object()))
    with pytest.raises(TypeError):
        R._check_supported(dict(base, infer_depth=True))                            # needs the depth_mlp kwarg
    R._check_supported(dict(base, infer_depth=True, depth_mlp=object()))
    with pytest.raises(ValueError):
        R._check_supported(dict(base, target_normal_map_for_radiance_calculation="bogus"))   # ibl_nerf_renderer.py:375
    R._check_supported(dict(base, perturb=1.0, pytest=True))                         # training-time sampling: built
    R._check_supported(dict(base, raw_noise_std=1.0))                                  # density noise: built


def test_tile_rows_partition():
    for H in (1, 7, 100, 800, 801):
        for world in (1, 2, 3, 8):
            spans = [D.tile_rows(H, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == H
            assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1
    with pytest.raises(ValueError):
        D.tile_rows(10, 2, 2)
    # the interleaved partition (the default since round 5): rank r takes rows r, r + world, ...; together every row once, sizes within one of each other
    for H in (1, 7, 100, 800, 801):
        for world in (1, 2, 3, 8):
            tiles = [list(D.tile_row_indices(H, r, world)) for r in range(world)]
            assert sorted(sum(tiles, [])) == list(range(H)) and all(t == list(range(r, H, world)) for r, t in enumerate(tiles))
            assert max(map(len, tiles)) - min(map(len, tiles)) <= 1 and len(tiles[0]) == max(map(len, tiles))
            assert [list(D.tile_row_indices(H, r, world, "contiguous")) for r in range(world)] == [list(range(a, a + n)) for a, n in (D.tile_rows(H, r, world) for r in range(world))]
    with pytest.raises(ValueError):
        D.tile_row_indices(10, 0, 2, "diagonal")
    # slice_gt_rows with a row range = the rows of the [H, W] image
    import numpy as _np
    img = _np.arange(6 * 4 * 3).reshape(6 * 4, 3)
    assert _np.array_equal(D.slice_gt_rows({"m": img}, 4, range(1, 6, 2))["m"], img.reshape(6, 4, 3)[1::2].reshape(-1, 3))
    assert _np.array_equal(D.slice_gt_rows({"m": img}, 4, 2, 3)["m"], img[8:20])


def test_auxiliary_network_blob_and_checkpoint(tmp_path):
    """PositionMLP (src/networks/MLP.py) -> one IBLNeRF-schema blob per output channel: trunk copied, out_linears row in the
    place of sigma_linear, zeros elsewhere; the oracle's trunk + sigma head on that blob equals the PositionMLP's channel."""
    aux = ck.synthetic_position_mlp(5, 3)
    assert list(aux)[:2] == ["positions_linears.0.weight", "positions_linears.0.bias"] and list(aux)[-2:] == ["out_linears.weight", "out_linears.bias"]
    pts = np.random.RandomState(0).uniform(-3, 3, (5, 7, 3)).astype(np.float32)
    want = O.position_mlp_query(aux, pts)
    for ch in range(3):
        sd = ck.blob_to_state_dict(ck.aux_channel_blob(aux, ch))
        assert np.array_equal(sd["positions_linears.5.weight"], aux["positions_linears.5.weight"])
        assert np.array_equal(sd["sigma_linear.weight"][0], aux["out_linears.weight"][ch]) and not sd["radiance_linear.weight"].any()
        assert np.abs(O.network_query(sd, pts, None)[..., 0] - want[..., ch]).max() <= 2e-6      # sgemv vs one column of sgemm
    with pytest.raises(ValueError):
        ck.aux_channel_blob(aux, 3)
    with pytest.raises(KeyError):
        ck.aux_channel_blob(ck.synthetic_state_dict(0), 0)
    os.makedirs(tmp_path / "exp")
    ck.save_checkpoint(str(tmp_path / "exp" / "000007.tar"), 7, ck.synthetic_state_dict(1), ck.synthetic_state_dict(2),
                       aux={"albedo_mlp": aux})
    _, kw, start, *_ = M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), no_reload=False, infer_albedo_separate=True,
                                                           infer_roughness_separate=True))
    assert start == 7 and kw["irradiance_mlp"] is None and kw["roughness_mlp"].out_ch == 1     # built, not in the checkpoint
    assert np.array_equal(kw["albedo_mlp"].state_dict()["out_linears.bias"], aux["out_linears.bias"])
    with pytest.raises(ValueError):
        kw["roughness_mlp"].load_state_dict(aux)                                               # 3 rows into a 1-channel network


def test_weights_key_cannot_recur():
    """renderer_for's cache key: models built one after another from different seeds — each freed before the next
    is built, so CPython hands out the same id() and array addresses again — must never share a key; a reload of the same
    object must change its key."""
    import torch
    from ibl_nerf_amd import renderer as R
    from torch_ref import RefShaped
    keys = []
    for seed in range(6):
        m = M.IBLNeRF()
        m.load_state_dict(ck.synthetic_state_dict(seed))
        keys.append(R._weights_key(m))
        del m
    assert len(set(keys)) == 6 and len({k[0] for k in keys}) == 6
    m = M.IBLNeRF()
    k0 = R._weights_key(m)
    assert R._weights_key(m) == k0
    m.load_state_dict(ck.synthetic_state_dict(3))
    assert R._weights_key(m) != k0 and R._weights_key(m)[0] == k0[0]
    nets = []
    for seed in range(3):                                  # nn.Modules rebuilt and reloaded: same version sums, other tokens
        n = RefShaped(ck.synthetic_state_dict(seed))
        nets.append(R._weights_key(n))
        del n
    assert len(set(nets)) == 3
    n = RefShaped(ck.synthetic_state_dict(0))
    k1 = R._weights_key(n)
    with torch.no_grad():
        n.sigma_linear.bias.add_(1.0)                      # an optimizer step
    assert R._weights_key(n) != k1


def test_create_iblnerf_checkpoint_key_errors(tmp_path):
    """ibl_nerf.py:355-376: elapsed_time comes from the checkpoint; a checkpoint without network_fine_state_dict
    (N_importance > 0) or without normal_mlp (infer_normal) is a KeyError, never a render with placeholder weights."""
    os.makedirs(tmp_path / "a")
    ck.save_checkpoint(str(tmp_path / "a" / "000003.tar"), 3, ck.synthetic_state_dict(1), ck.synthetic_state_dict(2), elapsed_time=12.5)
    ret = M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), expname="a", no_reload=False))
    assert ret[2] == 3 and ret[3] == 12.5
    with pytest.raises(KeyError):
        M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), expname="a", no_reload=False, infer_normal=True))
    os.makedirs(tmp_path / "b")
    ck.save_checkpoint(str(tmp_path / "b" / "000003.tar"), 3, ck.synthetic_state_dict(1))          # coarse only
    with pytest.raises(KeyError):
        M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), expname="b", no_reload=False))
    assert M.create_IBLNeRF(M.default_args(basedir=str(tmp_path), expname="b", no_reload=False, N_importance=0))[1]["network_fine"] is None


def test_training_ray_section_reproduces_the_reference_maps(lut):
    """training._ray_outputs — the differentiable torch restatement of the RAY-sized part of raw2outputs that the fused training step's
    backward differentiates (ibl_nerf_renderer.py:258, :412-474, :480-525) — on the reference's own recorded raw rows of the fitted
    checkpoint (fixture fitted_plain, fine pass): linear direct maps by the plain-torch compositing, the reference's n.v and reflected-ray
    maps as constants -> every output map of the reference's result dict."""
    torch = pytest.importorskip("torch")
    from conftest import load_golden, teacher_pass
    from ibl_nerf_amd import training as T
    from torch_ref import composite_direct
    g, _, _, _, _ = load_golden("fitted_plain")
    tp = teacher_pass(g, "f")
    k = tp["k"]
    rd = torch.from_numpy(g["rays_d"][:k])
    lin, _ = composite_direct(torch.from_numpy(tp["raw"]), torch.from_numpy(tp["z"]), rd)
    flags = dict(gamma_correct=True, use_radiance_linear=False, lut_coefficient="F", correct_depth=True)
    env = torch.stack([T._ungamma(torch.from_numpy(g["out__" + n][:k]), True) for n in
                       ("reflected_radiance_map", "reflected_coarse_radiance_map_1", "reflected_coarse_radiance_map_2", "reflected_coarse_radiance_map_3")], 1)
    consts = dict(n_dot_v=torch.from_numpy(g["out__n_dot_v_map"][:k]), env=env, lut=torch.from_numpy(lut), depth0=0.5 * (float(g["near"]) + float(g["far"])))
    x = lin.clone().requires_grad_(True)
    out = T._ray_outputs(x, consts, flags)
    for key in ("color_map", "specular_map", "diffuse_map", "prefiltered_reflected_map", "albedo_map", "irradiance_map", "radiance_map", "radiance_map_2",
                "roughness_map", "disp_map", "depth_map", "acc_map"):
        ref = g["out__" + key][:k]
        err = np.abs(out[key].detach().numpy().reshape(ref.shape) - ref).max() / np.abs(ref).max()
        assert err <= 2e-5, (key, err)
    # ... and it is differentiable where the reference is: roughness reaches color_map through the LUT coordinate, Fresnel, metallic and the mip remainder
    (dx,) = torch.autograd.grad(out["color_map"].sum(), x)
    assert float(dx[:, 5].abs().max()) > 0 and float(dx[:, 2:5].abs().max()) > 0 and float(dx[:, 6].abs().max()) > 0
    assert float(dx[:, 0].abs().max()) == 0.0 and float(dx[:, 7:].abs().max()) == 0.0      # depth is detached in the mip level (:455); radiance does not feed color_map


@pytest.mark.parametrize("name", ["train_step_edit", "train_step_insert"])
def test_override_rows_are_the_reference_overrides(name):
    """training.override_rows (round 5) — the edit / insert overrides of raw2outputs as ray-sized (mask, value) rows for a training step's backward — against the
    reference's own output maps of a step rendered with them (fixtures train_step_edit / train_step_insert, tests/frame_overrides.py's images at the step's pixels):
    on the masked rays the reference's roughness_map / albedo_map / irradiance_map / target_depth_map ARE the rows' values (albedo through its gamma, irradiance
    through output_f), elsewhere they are not; outside approximate_radiance only the depth rows exist.  And the backward's two halves: substituting the rows into
    the maps and zeroing the overridden entries' gradient is autograd through where(mask, value, x)."""
    import json
    import types
    torch = pytest.importorskip("torch")
    from conftest import GOLDEN
    from ibl_nerf_amd import training as T
    G = np.load(os.path.join(GOLDEN, name + ".npz"))
    edit = json.loads(str(G["edit_kwargs"]))
    gt = {k[4:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("gt__")}
    r = types.SimpleNamespace(device=torch.device("cpu"))
    n = int(G["rays_o"].shape[0])
    rows = T.override_rows(r, n, gt, edit, True, int(G["chunk"]))
    assert sorted(rows) == (["albedo", "depth", "roughness"] if name.endswith("edit") else ["albedo", "depth", "irradiance", "roughness"])
    assert sorted(T.override_rows(r, n, gt, edit, False, int(G["chunk"]))) == ["depth"]                  # :253-256 are outside `if approximate_radiance`
    assert T.override_rows(r, n, gt, {}, True) == {}
    gamma = lambda v: (v + 1e-12) ** (1.0 / 2.2)
    for sfx in ("", "0"):
        out = {k: torch.from_numpy(G["full__out__" + k + sfx]) for k in ("roughness_map", "albedo_map", "irradiance_map", "target_depth_map", "depth_map")}
        m, v = rows["roughness"]
        assert 5 <= int(m.sum()) <= n - 5 and torch.equal(out["roughness_map"][m], v[m]) and not torch.equal(out["roughness_map"][~m], v[~m])
        m, v = rows["albedo"]
        assert float((out["albedo_map"][m] - gamma(v[m])).abs().max()) <= 1e-6
        m, v = rows["depth"]
        assert torch.equal(out["target_depth_map"][m], v[m]) and torch.equal(out["depth_map"][m], v[m])      # target_depth_map IS depth_map: the assignment lands in both (:250-256)
        if "irradiance" in rows:
            m, v = rows["irradiance"]
            assert int(m.sum()) > 0 and float((out["irradiance_map"][m].reshape(-1) - gamma(v[m])).abs().max()) <= 1e-6
    # the backward's two halves against autograd through the masked assignment
    rng = np.random.RandomState(0)
    x = torch.from_numpy(rng.uniform(0.1, 0.9, (n, 19)).astype(np.float32)).double().requires_grad_(True)
    rows64 = {k: (m, v.double()) for k, (m, v) in rows.items()}
    flags = dict(gamma_correct=True, use_radiance_linear=False, lut_coefficient="F", correct_depth=True)
    keys = ("albedo_map", "roughness_map", "irradiance_map", "depth_map", "target_depth_map", "disp_map", "radiance_map")
    up = {k: torch.from_numpy(rng.randn(n, 3 if k in ("albedo_map", "radiance_map") else 1)).double() for k in keys}
    x_eff, _ = T._apply_overrides(x, None, rows64)
    outs = T._ray_outputs(x_eff, None, flags)
    (g_auto,) = torch.autograd.grad([outs[k] for k in keys], x, [up[k].reshape(outs[k].shape) for k in keys])
    xe = x_eff.detach().requires_grad_(True)
    outs2 = T._ray_outputs(xe, None, flags)
    (g_eff,) = torch.autograd.grad([outs2[k] for k in keys], xe, [up[k].reshape(outs2[k].shape) for k in keys])
    g_mine = T._zero_overridden(g_eff.clone(), None, rows64)
    assert torch.allclose(g_mine, g_auto, rtol=0, atol=1e-12) and float(g_auto[rows["roughness"][0], 5].abs().max()) == 0.0 and float(g_auto[:, 7:10].abs().min()) > 0


def test_bench_plain_multi_gpu_invocation_spawns_a_child_launcher(monkeypatch):
    """`python bench.py --gpus N` (N > 1) without a launcher in front — the form the driver types: bench.py starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same argv>` as a CHILD (subprocess.run, never exec: on the GPU pool an exec from
    a process that touched HIP is fatal, and this parent must not touch HIP at all) and returns the child's exit code."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1"])
    had_cuda = "torch" in sys.modules and sys.modules["torch"].cuda.is_initialized()
    assert bench.main() == 7                                  # the child's return code, relayed
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "2", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "torch" in sys.modules:
        assert sys.modules["torch"].cuda.is_initialized() == had_cuda      # the parent did not initialise the GPU
    # under a launcher (WORLD_SIZE set) a mismatching --gpus is refused, not re-launched
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit):
        bench.main()
    assert seen["cmd"] is cmd
