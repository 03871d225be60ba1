// Sustained rate under power of the candidate product scheme vs the shipped one, per K=64 block of a 32x32 tile:
//   shipped  : 12 x v_mfma_f32_32x32x16_bf16        (hi*hi, hi*lo, lo*hi for 4 k-steps)
//   candidate:  4 x v_mfma_f32_32x32x16_f16  +  2 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 e2m3 or fp8 e4m3)
// one wave per SIMD, one dependent accumulator chain, random operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0 shipped bf16x3, 1 f16 + fp6, 2 f16 + fp8, 3 f16 only (4 MFMAs), 4 fp6 only (2 MFMAs)
__global__ __launch_bounds__(256, 1) void k(const float* seed, const int* bits, float* out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 ab[4], bb[4];
    f16x8 ah[4], bh[4];
    for (int q = 0; q < 4; q++)
        for (int e = 0; e < 8; e++) {
            float x = seed[(tid * 64 + q * 16 + e) & 65535], y = seed[(tid * 64 + q * 16 + 8 + e) & 65535];
            ab[q][e] = (__bf16)x; bb[q][e] = (__bf16)y; ah[q][e] = (_Float16)x; bh[q][e] = (_Float16)y;
        }
    i32x8 a8[2], b8[2];
    for (int q = 0; q < 2; q++)
        for (int e = 0; e < 8; e++) { a8[q][e] = bits[(tid * 32 + q * 8 + e) & 65535]; b8[q][e] = bits[(tid * 32 + 16 + q * 8 + e) & 65535]; }
    const int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;   // e8m0 = 127 -> scale 1
    f32x16 acc = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 12; j++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[j & 3], bb[(j * 3 + blk) & 3], acc, 0, 0, 0);
            } else {
                if (MODE != 4) {
#pragma unroll
                    for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bh[(j + blk) & 3], acc, 0, 0, 0);
                }
                if (MODE == 1 || MODE == 4) {
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[0], b8[blk & 1], acc, 2, 2, 0, sa, 0, sb);
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[1], b8[(blk + 1) & 1], acc, 2, 2, 0, sa, 0, sb);
                } else if (MODE == 2) {
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[0], b8[blk & 1], acc, 0, 0, 0, sa, 0, sb);
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[1], b8[(blk + 1) & 1], acc, 0, 0, 0, sa, 0, sb);
                }
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc[r];
    out[tid] = s;
}

static int g_grid = 256;
template <int MODE> double run(const float* d_seed, const int* d_bits, float* d_out, int iters, const char* name) {
    hipLaunchKernelGGL(k<MODE>, dim3(g_grid), dim3(256), 0, 0, d_seed, d_bits, d_out, iters / 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(g_grid), dim3(256), 0, 0, d_seed, d_bits, d_out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double blocks = 1024.0 * iters * 4;   // K=64 blocks of a 32x32 tile, 1024 waves
    printf("%-34s %8.2f ms   %.2f ns per K=64 block per wave   algorithmic %.0f TFLOP/s\n", name, ms, ms * 1e6 / (iters * 4.0),
           blocks * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12);
    return ms;
}

int main(int argc, char** argv) {
    if (argc > 1) g_grid = atoi(argv[1]);
    printf("grid = %d workgroups (one per CU)\n", g_grid);
    std::vector<float> h(65536); srand(1);
    for (auto& x : h) x = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    std::vector<int> b(65536);
    for (auto& x : b) x = (int)(((unsigned)rand() << 16) ^ (unsigned)rand()) & 0x77777777 | 0x11111111;   // random mantissas, no NaN patterns in fp8
    float *d_seed, *d_out; int* d_bits;
    hipMalloc(&d_seed, 65536 * 4); hipMalloc(&d_bits, 65536 * 4); hipMalloc(&d_out, 256 * 256 * 4);
    hipMemcpy(d_seed, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_bits, b.data(), 65536 * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = run<0>(d_seed, d_bits, d_out, iters, "shipped: 12 bf16 MFMA");
        double t1 = run<1>(d_seed, d_bits, d_out, iters, "candidate: 4 f16 + 2 MX-fp6");
        double t2 = run<2>(d_seed, d_bits, d_out, iters, "candidate: 4 f16 + 2 MX-fp8");
        run<3>(d_seed, d_bits, d_out, iters, "4 f16 only");
        run<4>(d_seed, d_bits, d_out, iters, "2 MX-fp6 only");
        printf("speed-up fp6 %.2fx   fp8 %.2fx\n", t0 / t1, t0 / t2);
    }
    return 0;
}
