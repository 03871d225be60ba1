// probe: v_cvt_scalef32_pk32_fp6_f16 slot order / scale; codegen of the f16 residual
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
__global__ void k_cvt(const float* in, unsigned* out, float scale) {
    h32 v;
    for (int i = 0; i < 32; ++i) v[i] = (_Float16)in[threadIdx.x * 32 + i];
    u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}
__global__ void k_res(const float* in, float* out, unsigned* outh) {
    f32x2 x = {in[2 * threadIdx.x], in[2 * threadIdx.x + 1]};
    h2 h = __builtin_convertvector(x, h2);
    out[2 * threadIdx.x] = x[0] - (float)h[0];
    out[2 * threadIdx.x + 1] = x[1] - (float)h[1];
    outh[threadIdx.x] = __builtin_bit_cast(unsigned, h);
}
static double fp6_val(int c) { int s = c >> 5, e = (c >> 3) & 3, m = c & 7; double v = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * std::ldexp(1.0, e - 1); return s ? -v : v; }
static int fp6_enc(double x) { int s = x < 0; double a = std::fabs(x); int best = 0; double bd = 1e30;
    for (int c = 0; c < 32; ++c) { double d = std::fabs(fp6_val(c) - a); if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; } } return best | (s << 5); }
static int get_slot(const unsigned* regs, int j) { int code = 0, bit = 6 * j; for (int q = 0; q < 6; ++q) { int bb = bit + q; code |= ((regs[bb >> 5] >> (bb & 31)) & 1) << q; } return code; }
int main() {
    srand(5);
    std::vector<float> in(64 * 32);
    for (auto& x : in) x = (float)(_Float16)(((float)rand() / RAND_MAX * 2 - 1) * 9.0f);
    float* din; unsigned* dout; hipMalloc(&din, in.size() * 4); hipMalloc(&dout, 64 * 24);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 2.0f, 0.5f, 3.5f}) {
        hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, dout, scale);
        std::vector<unsigned> out(64 * 6); hipMemcpy(out.data(), dout, 64 * 24, hipMemcpyDeviceToHost);
        int se; std::frexp(scale, &se); double div = std::ldexp(1.0, se - 1);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) bad += get_slot(&out[l * 6], j) != fp6_enc(in[l * 32 + j] / div);
        printf("cvt_scalef32_pk32_fp6_f16 scale=%g: mismatches (sequential hypothesis) = %d of 2048\n", scale, bad);
    }
    return 0;
}
