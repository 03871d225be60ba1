// Issue cost of the MX kernel's filler instructions beside dependent MFMAs, one wave per SIMD:
// N fillers of one kind between consecutive v_mfma_f32_32x32x16_f16 of one accumulator chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(const float* seed, float* out, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)seed[threadIdx.x * 8 + e]; b[e] = (_Float16)seed[4096 + threadIdx.x * 8 + e]; }
    f32x16 acc = {}, other;
    for (int r = 0; r < 16; ++r) other[r] = seed[8192 + threadIdx.x * 16 + r];
    asm volatile("" : "+a"(other));
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed[100 + threadIdx.x + i];
    unsigned pk = 0;
    u32x16 hv; for (int i = 0; i < 16; ++i) hv[i] = __builtin_bit_cast(unsigned, seed[300 + threadIdx.x + i]) & 0x3fff3fffu;
    unsigned sinkc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                float& x = v[n & 7];
                if (KIND == 0) asm volatile("v_max_i32 %0, 0, %0" : "+v"(x));
                if (KIND == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(other[(n + j) & 15]));
                if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x), "v"(v[(n + 1) & 7]));
                if (KIND == 3) asm volatile("v_fma_mix_f32 %0, %0, 1.0, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(x) : "v"(pk));
                if (KIND == 4) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(pk) : "v"(x));
                if (KIND == 5) asm volatile("s_nop 0");
                if (KIND == 6) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(v[(n + 1) & 7]), "v"(v[(n + 2) & 7]));
                if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(v[(n + 3) & 7]));
                if (KIND == 8) { asm volatile("" : "+v"(hv)); auto r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, hv), 1.0f); asm volatile("" :: "v"(r)); }
                if (KIND == 9) { f32x16 a16, b16; for (int i = 0; i < 16; ++i) { a16[i] = v[i & 7]; b16[i] = v[(i + 3) & 7]; }
                                 auto r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a16, b16, 1.0f); sinkc ^= (unsigned)r[n % 6]; v[n & 7] += 1.0f; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)pk + (float)sinkc;
}
template <int KIND, int N> double run(const float* d_seed, float* d_out, int grid) {
    const int iters = 3000;
    hipLaunchKernelGGL((k<KIND, N>), dim3(grid), dim3(256), 0, 0, d_seed, d_out, iters / 10); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<KIND, N>), dim3(grid), dim3(256), 0, 0, d_seed, d_out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (iters * 16.0);
}
template <int KIND> void row(const float* d_seed, float* d_out, int grid, const char* name) {
    printf("%-22s ns/MFMA with 0,2,4,6,8 fillers: %5.1f %5.1f %5.1f %5.1f %5.1f\n", name, run<KIND, 0>(d_seed, d_out, grid), run<KIND, 2>(d_seed, d_out, grid),
           run<KIND, 4>(d_seed, d_out, grid), run<KIND, 6>(d_seed, d_out, grid), run<KIND, 8>(d_seed, d_out, grid));
}
int main(int argc, char** argv) {
    int grid = argc > 1 ? atoi(argv[1]) : 64;
    float *d_seed, *d_out; hipMalloc(&d_seed, 65536 * 4); hipMalloc(&d_out, 256 * 256 * 4);
    float* h = (float*)malloc(65536 * 4); srand(1); for (int i = 0; i < 65536; ++i) h[i] = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    hipMemcpy(d_seed, h, 65536 * 4, hipMemcpyHostToDevice);
    printf("grid %d (14 ns = one 32-cycle MFMA at 2.3 GHz)\n", grid);
    row<0>(d_seed, d_out, grid, "v_max_i32");
    row<1>(d_seed, d_out, grid, "v_accvgpr_read_b32");
    row<2>(d_seed, d_out, grid, "v_cvt_pk_f16_f32");
    row<3>(d_seed, d_out, grid, "v_fma_mix_f32");
    row<4>(d_seed, d_out, grid, "v_fma_mixlo_f16");
    row<5>(d_seed, d_out, grid, "s_nop 0");
    row<6>(d_seed, d_out, grid, "v_max3_i32");
    row<7>(d_seed, d_out, grid, "v_mov_b32");
    printf("%-22s ns/MFMA with 0,1,2 fillers: %5.1f %5.1f %5.1f\n", "cvt_pk32_fp6_f16", run<8, 0>(d_seed, d_out, grid), run<8, 1>(d_seed, d_out, grid), run<8, 2>(d_seed, d_out, grid));
    printf("%-22s ns/MFMA with 0,1,2 fillers: %5.1f %5.1f %5.1f\n", "cvt_2xpk16_fp6_f32", run<9, 0>(d_seed, d_out, grid), run<9, 1>(d_seed, d_out, grid), run<9, 2>(d_seed, d_out, grid));
    return 0;
}
