// How much does one LDS operand read cost a wave that is otherwise issuing dependent MFMAs (one wave per SIMD)?
// variants: no read / ds_read_b128 into VGPRs / into AGPRs / two reads / read used as the NEXT MFMA's A operand
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* seed, float* out, int iters) {
    extern __shared__ char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((float*)smem)[i] = seed[i & 16383];
    __syncthreads();
    f16x8 a, b, a2;
    for (int e = 0; e < 8; ++e) { a2[e] = (_Float16)seed[2048 + threadIdx.x * 8 + e]; a[e] = (_Float16)seed[threadIdx.x * 8 + e]; b[e] = (_Float16)seed[4096 + threadIdx.x * 8 + e]; }
    f32x16 acc = {};
    unsigned addr = (threadIdx.x & 63) * 16;
    u32x4 sink = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            const unsigned ad = addr + ((it * 16 + j) & 31) * 1024;
            if (MODE == 1 || MODE == 3) { u32x4 q; asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(ad)); sink ^= q; }
            if (MODE == 3) { u32x4 q; asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(q) : "v"(ad)); sink ^= q; }
            if (MODE == 2) { u32x4 q; asm volatile("ds_read_b128 %0, %1" : "=a"(q) : "v"(ad)); asm volatile("" :: "a"(q)); }
            if (MODE == 4) { u32x4 q; asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(ad)); a = __builtin_bit_cast(f16x8, q); }
            if (MODE == 6) { asm volatile("ds_read_b128 %0, %1" : "+v"(a) : "v"(ad)); }   // overwrites the A operand the MFMA just issued is still reading
            if (MODE == 7) { asm volatile("ds_read_b128 %0, %1" : "+v"(a2) : "v"(ad)); f16x8 t = a; a = a2; a2 = t; }   // ping-pong: writes the OTHER register set
            if (MODE == 5) { unsigned q; asm volatile("ds_read_b32 %0, %1" : "=v"(q) : "v"(ad)); sink[0] ^= q; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(sink[0] ^ sink[1] ^ sink[2] ^ sink[3]);
}
template <int MODE> void run(const float* d_seed, float* d_out, int grid, const char* name) {
    const int iters = 4000;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 65536, 0, d_seed, d_out, iters / 10); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 65536, 0, d_seed, d_out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f ms  %.1f ns per MFMA\n", name, ms, ms * 1e6 / (iters * 16.0));
}
int main(int argc, char** argv) {
    int grid = argc > 1 ? atoi(argv[1]) : 64;
    float *d_seed, *d_out; hipMalloc(&d_seed, 65536 * 4); hipMalloc(&d_out, 256 * 256 * 4);
    float* h = (float*)malloc(65536 * 4); srand(1); for (int i = 0; i < 65536; ++i) h[i] = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    hipMemcpy(d_seed, h, 65536 * 4, hipMemcpyHostToDevice);
    printf("grid %d\n", grid);
    run<0>(d_seed, d_out, grid, "dependent f16 MFMAs only");
    run<1>(d_seed, d_out, grid, "+ 1 ds_read_b128 -> VGPR per MFMA (unused)");
    run<2>(d_seed, d_out, grid, "+ 1 ds_read_b128 -> AGPR per MFMA (unused)");
    run<3>(d_seed, d_out, grid, "+ 2 ds_read_b128 -> VGPR per MFMA (unused)");
    run<4>(d_seed, d_out, grid, "+ 1 ds_read_b128 feeding the next MFMA's A");
    run<5>(d_seed, d_out, grid, "+ 1 ds_read_b32 -> VGPR per MFMA (unused)");
    run<6>(d_seed, d_out, grid, "+ 1 ds_read_b128 INTO the A regs just issued");
    run<7>(d_seed, d_out, grid, "+ 1 ds_read_b128 into the other A set (ping-pong)");
    return 0;
}
