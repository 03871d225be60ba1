// The MX kernel's inner skeleton in isolation: per K=64 block 4 f16 MFMAs + 2 scaled-fp6 MFMAs on one accumulator,
// A operands of the NEXT block read from LDS during this one (double-buffered registers), B operands constant.
// Switches (argv[2] bitmask): 1 = no LDS reads in the loop, 2 = fp6 slots replaced by f16 MFMAs,
// 4 = all reads as b128 (no b64/b32), 8 = barrier every 4 blocks, 16 = every 4 blocks the chain restarts from an LDS bias
// and the finished accumulator is pinned (tile boundary of the real kernel), 32 = B operands alternate between 4 register sets
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
struct Ops { u32x4 f[4]; u32x4 wq, rq; u32x2 wd, rd; unsigned sc; };
template <int SW>
__device__ __forceinline__ void load_ops(Ops& o, const char* blk, int lane) {
    for (int j = 0; j < 4; ++j) o.f[j] = *(const u32x4*)(blk + j * 1024 + lane * 16);
    o.wq = *(const u32x4*)(blk + 4096 + lane * 16);
    o.rq = *(const u32x4*)(blk + 5120 + lane * 16);
    if (SW & 4) { o.wd = u32x2{o.wq[0], o.wq[1]}; o.rd = u32x2{o.rq[2], o.rq[3]}; o.sc = 0x7f7f7f7f; }
    else { o.wd = *(const u32x2*)(blk + 6144 + lane * 8); o.rd = *(const u32x2*)(blk + 6656 + lane * 8); o.sc = *(const unsigned*)(blk + 7168 + lane * 4); }
}
__device__ __forceinline__ i32x8 op6(u32x4 q, u32x2 d) { i32x8 v; v[0]=q[0]; v[1]=q[1]; v[2]=q[2]; v[3]=q[3]; v[4]=d[0]; v[5]=d[1]; return v; }
template <int SW>
__global__ __launch_bounds__(256, 1) void k(const float* seed, float* out, int iters) {
    extern __shared__ char smem[];
    for (int i = threadIdx.x; i < 98304 / 4; i += 256) ((float*)smem)[i] = seed[i & 16383];
    for (int b = 0; b < 12; ++b) for (int i = threadIdx.x; i < 64; i += 256) *(unsigned*)(smem + b * 8192 + 7168 + i * 4) = 0x7f7f7f7fu;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f16x8 b; i32x8 b6;
    for (int e = 0; e < 8; ++e) { b[e] = (_Float16)seed[4096 + threadIdx.x * 8 + e]; b6[e] = 0x11111111 * (e + 1); }
    f32x16 acc = {}, sum = {};
    Ops cur, nxt;
    load_ops<SW>(cur, smem, lane);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 12; ++blk) {
            const char* nb = smem + ((it * 12 + blk + 1) % 12) * 8192;
            if (!(SW & 1)) { nxt.f[0] = *(const u32x4*)(nb + lane * 16); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur.f[0]), b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(SW & 1)) { nxt.f[1] = *(const u32x4*)(nb + 1024 + lane * 16); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur.f[1]), b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(SW & 1)) { nxt.f[2] = *(const u32x4*)(nb + 2048 + lane * 16); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur.f[2]), b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(SW & 1)) { nxt.f[3] = *(const u32x4*)(nb + 3072 + lane * 16); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur.f[3]), b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(SW & 1)) {
                nxt.wq = *(const u32x4*)(nb + 4096 + lane * 16);
                if (SW & 4) { nxt.wd = u32x2{nxt.wq[0], nxt.wq[1]}; nxt.sc = 0x7f7f7f7f; }
                else { nxt.wd = *(const u32x2*)(nb + 6144 + lane * 8); nxt.sc = *(const unsigned*)(nb + 7168 + lane * 4); }
            }
            if (SW & 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur.wq), b, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(op6(cur.wq, cur.wd), b6, acc, 2, 2, 0, (int)cur.sc, 1, 0x7f7f7f7f);
            __builtin_amdgcn_sched_barrier(0);
            if (!(SW & 1)) {
                nxt.rq = *(const u32x4*)(nb + 5120 + lane * 16);
                if (SW & 4) nxt.rd = u32x2{nxt.rq[2], nxt.rq[3]}; else nxt.rd = *(const u32x2*)(nb + 6656 + lane * 8);
            }
            if (SW & 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, cur.rq), b, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(op6(cur.rq, cur.rd), b6, acc, 2, 2, 1, (int)cur.sc, 0, 0x7f7f7f7f);
            __builtin_amdgcn_sched_barrier(0);
            if (!(SW & 1)) cur = nxt;
            if ((SW & 8) && (blk & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if ((SW & 16) && (blk & 3) == 3) {
                asm volatile("" : "+a"(acc));
                for (int r = 0; r < 16; ++r) sum[r] += acc[r];
                acc = *(const f32x16*)(smem + 90112 + ((it + blk) & 7) * 64 + (lane >> 5) * 64);
            }
        }
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r] + sum[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SW> void run(const float* d_seed, float* d_out, int grid) {
    const int iters = 400;
    hipFuncSetAttribute((const void*)k<SW>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipLaunchKernelGGL(k<SW>, dim3(grid), dim3(256), 98304, 0, d_seed, d_out, iters / 10); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<SW>, dim3(grid), dim3(256), 98304, 0, d_seed, d_out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("switches %2d: %.1f ns per MFMA slot%s%s%s%s\n", SW, ms * 1e6 / (iters * 72.0), SW & 1 ? "  [no LDS reads]" : "", SW & 2 ? "  [fp6 slots as f16]" : "",
           SW & 4 ? "  [b128 reads only]" : "", SW & 8 ? "  [barrier per 4 blocks]" : ""); if (SW & 16) printf("             ^ with tile boundaries (bias restart + pinned accumulator)\n");
}
int main(int argc, char** argv) {
    int grid = argc > 1 ? atoi(argv[1]) : 64;
    float *d_seed, *d_out; hipMalloc(&d_seed, 65536 * 4); hipMalloc(&d_out, 256 * 256 * 4);
    float* h = (float*)malloc(65536 * 4); srand(1); for (int i = 0; i < 65536; ++i) h[i] = ((float)rand() / RAND_MAX * 2 - 1) * 0.05f;
    hipMemcpy(d_seed, h, 65536 * 4, hipMemcpyHostToDevice);
    printf("grid %d\n", grid);
    run<1>(d_seed, d_out, grid); run<0>(d_seed, d_out, grid); run<2>(d_seed, d_out, grid); run<8>(d_seed, d_out, grid); run<16>(d_seed, d_out, grid); run<24>(d_seed, d_out, grid);
    return 0;
}
