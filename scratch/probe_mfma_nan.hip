// Which NaN does v_mfma_f32_32x32x16_f16 produce for (+inf) + (-inf) and 0 * inf?  (sign decides whether an integer-max ReLU keeps it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(int mode, unsigned* out) {
    f16x8 a, b;
    const _Float16 inf = (_Float16)INFINITY;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)0.0f; }
    f32x16 c = {0};
    if (mode == 0) { b[0] = inf; b[1] = -inf; }                       // w*(+inf) + w*(-inf) inside one MFMA
    if (mode == 1) { b[0] = inf; a[0] = (_Float16)0.0f; }             // 0 * inf
    if (mode == 2) { b[0] = inf; }                                    // two MFMAs: +inf then -inf accumulated
    if (mode == 3) { b[0] = inf; a[0] = (_Float16)-1.0f; b[1] = inf; }  // (-1)*inf + 1*inf
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (mode == 2) { b[0] = -inf; c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    if (threadIdx.x == 0) out[0] = __builtin_bit_cast(unsigned, c[0]);
}
int main() {
    unsigned* d; hipMalloc(&d, 64);
    for (int m = 0; m < 4; ++m) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, m, d);
        unsigned h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("mode %d: c[0] bits = 0x%08x\n", m, h);
    }
    return 0;
}
