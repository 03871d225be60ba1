// Layout probes for the gfx950 MX path: v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 e2m3 operands, e8m0 block scales),
// v_cvt_scalef32_2xpk16_fp6_f32 and v_cvt_pk_f16_f32.  Hypotheses are checked against host arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void k_mfma(const unsigned* a6, const unsigned* b6, const unsigned* sa, const unsigned* sb, float* out, int opsel) {
    const int l = threadIdx.x;
    i32x8 a = {}, b = {};
    for (int i = 0; i < 6; ++i) { a[i] = a6[l * 6 + i]; b[i] = b6[l * 6 + i]; }
    f32x16 c = {};
    if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, (int)sa[l], 0, (int)sb[l]);
    else if (opsel == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 1, (int)sa[l], 1, (int)sb[l]);
    else if (opsel == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 2, (int)sa[l], 2, (int)sb[l]);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 3, (int)sa[l], 3, (int)sb[l]);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}
__global__ void k_cvt6(const float* in, unsigned* out, float scale) {
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x * 32 + i]; b[i] = in[threadIdx.x * 32 + 16 + i]; }
    u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}
__global__ void k_cvt16(const float* in, h2* out) {
    f32x2 v = {in[2 * threadIdx.x], in[2 * threadIdx.x + 1]};
    out[threadIdx.x] = __builtin_convertvector(v, h2);
}

static double fp6_val(int c) {   // e2m3, bias 1
    int s = c >> 5, e = (c >> 3) & 3, m = c & 7;
    double v = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * std::ldexp(1.0, e - 1);
    return s ? -v : v;
}
static int fp6_enc(double x) {   // RNE to e2m3, saturating
    int s = x < 0; double a = std::fabs(x);
    int best = 0; double bd = 1e30;
    for (int c = 0; c < 32; ++c) { double d = std::fabs(fp6_val(c) - a); if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; } }
    return best | (s << 5);
}
static void set_slot(unsigned* regs, int j, int code) {   // hypothesis: slot j at bits [6j, 6j+6) of the 192-bit lane vector
    int bit = 6 * j;
    for (int q = 0; q < 6; ++q) { int bb = bit + q; if ((code >> q) & 1) regs[bb >> 5] |= 1u << (bb & 31); }
}
static int get_slot(const unsigned* regs, int j) {
    int code = 0, bit = 6 * j;
    for (int q = 0; q < 6; ++q) { int bb = bit + q; code |= ((regs[bb >> 5] >> (bb & 31)) & 1) << q; }
    return code;
}

int main() {
    srand(3);
    // ---- MFMA: A[32][64], B[64][32] as fp6 codes, block scales per (row|col, k-half)
    std::vector<int> A(32 * 64), B(64 * 32);
    for (auto& x : A) x = rand() & 63;
    for (auto& x : B) x = rand() & 63;
    for (int opsel = 0; opsel < 4; ++opsel) {
        std::vector<unsigned> a6(64 * 6, 0), b6(64 * 6, 0), sa(64), sb(64);
        int ea[64], eb[64];
        for (int l = 0; l < 64; ++l) {
            ea[l] = 124 + rand() % 7; eb[l] = 124 + rand() % 7;
            unsigned junk = ((unsigned)rand() << 16) ^ rand();
            sa[l] = (junk & ~(0xffu << (8 * opsel))) | ((unsigned)ea[l] << (8 * opsel));
            junk = ((unsigned)rand() << 16) ^ rand();
            sb[l] = (junk & ~(0xffu << (8 * opsel))) | ((unsigned)eb[l] << (8 * opsel));
            for (int j = 0; j < 32; ++j) {
                set_slot(&a6[l * 6], j, A[(l & 31) * 64 + 32 * (l >> 5) + j]);   // lane l: row l&31, k = 32*(l>>5)+j
                set_slot(&b6[l * 6], j, B[(32 * (l >> 5) + j) * 32 + (l & 31)]); // lane l: col l&31
            }
        }
        unsigned *da, *db, *dsa, *dsb; float* dout;
        hipMalloc(&da, 64 * 24); hipMalloc(&db, 64 * 24); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dout, 64 * 64);
        hipMemcpy(da, a6.data(), 64 * 24, hipMemcpyHostToDevice); hipMemcpy(db, b6.data(), 64 * 24, hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout, opsel);
        std::vector<float> out(64 * 16);
        hipMemcpy(out.data(), dout, 64 * 64, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
                double ref = 0;
                for (int k = 0; k < 64; ++k) {
                    int h = k >> 5;
                    ref += fp6_val(A[row * 64 + k]) * std::ldexp(1.0, ea[row + 32 * h] - 127) * fp6_val(B[k * 32 + col]) * std::ldexp(1.0, eb[col + 32 * h] - 127);
                }
                maxerr = std::fmax(maxerr, std::fabs(ref - out[l * 16 + r])); maxref = std::fmax(maxref, std::fabs(ref));
            }
        printf("mfma fp6 opsel=%d: max|err| = %.3g (max|ref| = %.3g)  %s\n", opsel, maxerr, maxref, maxerr <= 1e-4 * maxref ? "LAYOUT HYPOTHESIS OK" : "MISMATCH");
    }
    // ---- cvt fp6
    {
        std::vector<float> in(64 * 32);
        for (auto& x : in) x = ((float)rand() / RAND_MAX * 2 - 1) * 40.0f;
        float* din; unsigned* dout; hipMalloc(&din, in.size() * 4); hipMalloc(&dout, 64 * 24);
        hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
        for (float scale : {1.0f, 8.0f, 0.25f, 12.0f}) {
            hipLaunchKernelGGL(k_cvt6, dim3(1), dim3(64), 0, 0, din, dout, scale);
            std::vector<unsigned> out(64 * 6); hipMemcpy(out.data(), dout, 64 * 24, hipMemcpyDeviceToHost);
            int se; std::frexp(scale, &se); double div = std::ldexp(1.0, se - 1);   // 2^floor(log2 scale)
            int bad_seq = 0, bad_il = 0;
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 32; ++j) {
                    int got = get_slot(&out[l * 6], j);
                    int exp_seq = fp6_enc(in[l * 32 + j] / div);                         // slot j <- element j (a0..a15,b0..b15)
                    int exp_il = fp6_enc(in[l * 32 + ((j & 1) * 16 + (j >> 1))] / div);    // interleaved a0,b0,a1,b1,...
                    bad_seq += got != exp_seq; bad_il += got != exp_il;
                }
            printf("cvt_scalef32_2xpk16_fp6_f32 scale=%g: mismatches sequential=%d interleaved=%d of 2048\n", scale, bad_seq, bad_il);
            if (scale == 1.0f) { printf("  lane0 in: "); for (int j = 0; j < 6; ++j) printf("%.3f ", in[j]); printf(" codes: "); for (int j = 0; j < 6; ++j) printf("%d(%.3f) ", get_slot(&out[0], j), fp6_val(get_slot(&out[0], j))); printf("\n"); }
        }
    }
    // ---- cvt f16
    {
        float h[8] = {1.00048828125f /*1+2^-11: tie*/, 1.00146484375f /*1+3*2^-11: tie*/, 70000.f, -70000.f, 65519.f, 65520.f, 1e-8f, 3.0e-8f};
        float* din; h2* dout; hipMalloc(&din, 32); hipMalloc(&dout, 16);
        hipMemcpy(din, h, 32, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_cvt16, dim3(1), dim3(4), 0, 0, din, dout);
        unsigned short o[8]; hipMemcpy(o, dout, 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < 8; ++i) printf("cvt_pk_f16_f32(%.10g) = 0x%04x\n", h[i], o[i]);
    }
    return 0;
}
