// Does v_mfma_f32_32x32x16_f16 honour f16 denormal inputs on gfx950?  (needed by the f16 hi/lo split: lo ~ 2^-12 |x|)
// hipcc --offload-arch=gfx950 -O2 scratch/probe_f16_denorm.hip -o /tmp/probe_f16_denorm && /tmp/probe_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
    // the mix conversion: residual of a value whose f16 residual is denormal
    float x = 0.1234567f; unsigned hb, lb = 0;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    f32x2 xv = {x, x};
    hb = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, f16x2));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(x), "v"(hb));
    if (threadIdx.x == 0) { out[1] = (float)__builtin_bit_cast(f16x2, lb)[0]; out[2] = x - (float)__builtin_bit_cast(f16x2, hb)[0]; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    const float tests[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1.0f}, {1.0e-6f, 1.0f}, {6.0e-8f, 1.0f}, {1.0f, 1.0e-6f}, {1.0e-6f, 1024.0f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, t[0], t[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  mfma c[0]=%.9g  expected %.9g   | mixlo residual %.9g exact %.9g\n", t[0], t[1], h[0], 16.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1], h[1], h[2]);
    }
    return 0;
}
