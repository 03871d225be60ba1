// The trunk of an IBLNeRF network (positions_linears.0-7 + sigma_linear, src/nerf_models/ibl_nerf.py:154-176, :200) in EXACT fp32 on the matrix cores, for a compact
// list of points (k_select_points): the density of the coarse pass's relevant samples — the few samples per ray whose weights place the fine samples.
//
// Why a fourth product scheme (round 5).  The reference computes in fp32.  The 15-slot form (three f16 + three block-scaled fp6 products, operands to ~2^-26) lands the
// coarse pass's per-sample weights within 1e-6 .. 1.2e-5 of the reference's, the fp32 C restatement within 2e-6 at worst (16 384 rays) — and the fine samples sit where
// those weights put them: on the ray that was the launch-scale fixture's worst (normal off by 8.6e-3) the fine z moved by 4.6e-5, and with the oracle's fine z every
// product scheme, fp32 included, renders that normal to 7e-4 (scratch/outlier_rays.py).  No ray-level proxy tells such rays from their neighbours
// (scratch/outlier_census.py: depth agreement, grazing angle, far-plane mass flag 5 of 18), so the remedy is arithmetic, not routing: fp32 operands, fp32 products,
// fp32 accumulation — v_mfma_f32_32x32x2_f32, bit for bit a k-ordered fmaf chain (guide section 3), 1/16 of the f16 rate.  On ~5 of a ray's 64 coarse samples that is
// 3 % of a frame.
//
// Layout: one workgroup = 4 waves = 128 list entries, 32 per wave.  D[out][point] = W[out][k] x act[k][point]: the A operand is a weight (lane l: out = 32 tile + (l & 31),
// k = l >> 5), B an activation (k = l >> 5, point = l & 31), 8 independent accumulators of 32 outputs each per wave.  Activations live in LDS as [k][32 points] per wave
// (32 KB; a layer's outputs replace its inputs once all 8 accumulators are complete), weights are read from the state dict's own fp32 [out][in] rows (2 MB per network:
// L2-resident) in chunks of KC columns, transposed into a double-buffered [k][out] LDS tile shared by the four waves, the next chunk prefetched into registers under
// the current one's 32 MFMAs.  No packed stream, no range guard (fp32 has the range), no tables.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "sincos_enc.h"

namespace ibl {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KC = 8;                    // weight columns per staged chunk
constexpr int NPT = 32;                  // points per wave
constexpr int WIDTH = 256;
constexpr int ACT_FLOATS = WIDTH * NPT;
constexpr int SMEM_BYTES = (2 * KC * WIDTH + 4 * ACT_FLOATS) * (int)sizeof(float);      // 16 KB + 128 KB

__device__ __forceinline__ void load_chunk(const float* __restrict__ row, int ncols, int c, float (&v)[KC]) {
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        const int k = c * KC + j;
        v[j] = k < ncols ? row[k] : 0.0f;
    }
}

// acc[tile][.] += W[:, col0 : col0 + ncols] x act[0 : nk][.]   (nk = ncols rounded up to a multiple of KC: the padding columns are zero weights)
__device__ __forceinline__ void gemm(const float* __restrict__ W, int ld, int col0, int ncols, int nk, const float* act, float* wl, f32x16 (&acc)[8], int t, int lane) {
    const float* row = W + (long)t * ld + col0;
    float v[KC];
    const int n_chunk = nk / KC;
    load_chunk(row, ncols, 0, v);
    __syncthreads();                     // (the previous product's last chunk is out of every wave's hands)
    for (int c = 0; c < n_chunk; ++c) {
        float* buf = wl + (c & 1) * KC * WIDTH;
#pragma unroll
        for (int j = 0; j < KC; ++j) buf[j * WIDTH + t] = v[j];
        __syncthreads();                 // (one barrier per chunk: a wave that writes chunk c + 1 has passed barrier c, which every wave reaches only after chunk c - 1)
        if (c + 1 < n_chunk) load_chunk(row, ncols, c + 1, v);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
            const float b = act[(c * KC + kk + (lane >> 5)) * NPT + (lane & 31)];
            const float* wrow = buf + (kk + (lane >> 5)) * WIDTH + (lane & 31);
#pragma unroll
            for (int tile = 0; tile < 8; ++tile) acc[tile] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[32 * tile], b, acc[tile], 0, 0, 0);
        }
    }
}

// Every output element is ONE accumulator that takes its K products in order k = 0, 1, 2, ... from zero, the bias last — the order of an sgemm micro-kernel (what the
// reference's addmm runs: MKL / oneDNN) and of the C restatement (oracle/csrc/gemm.c): fp32 results of a fitted network's cancelling sums depend on the order at the
// 1e-4 level in raw density, and the implementations that share this order agree with the reference 2.5x more closely than one that starts the chain from the bias or
// takes the skip layer's two blocks in the other order (measured on the coarse weights: 9.5e-7 against 2.4e-6 at 99 %).
__device__ __forceinline__ void start(f32x16 (&acc)[8]) {
#pragma unroll
    for (int tile = 0; tile < 8; ++tile)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tile][i] = 0.0f;
}

__device__ __forceinline__ void finish_relu(const f32x16 (&acc)[8], const float* __restrict__ bias, float* act, int lane) {
#pragma unroll
    for (int tile = 0; tile < 8; ++tile)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = 32 * tile + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
            act[o * NPT + (lane & 31)] = fmaxf(acc[tile][i] + bias[o], 0.0f);
        }
}

// rows 0..62: [x, sin(2^0 x), cos(2^0 x), ...] (positional_embedder.py:21-34: per frequency sin xyz, cos xyz), row 63: 0
__device__ __forceinline__ void write_encoding(float* act, const float (&p)[3], int lane) {
    const int pt = lane & 31, half = lane >> 5;
    // sin / cos of fl(x 2^k) (exact) in DOUBLE, rounded once to float: within half an ulp (+ 1e-9) of the true value, i.e. what a correctly rounded sinf returns — the
    // closest any implementation gets to the reference's own (ATen's vectorised sinf, <= 1 ulp).  The fused MLP kernels' shared encoding (sincos_enc.h) is good to
    // 2.5e-7 = ~2 ulp of 1.0; through a fitted network's cancelling sums that alone moves a coarse weight by up to 4e-6 (measured, scratch notes in DESIGN.md 4.1j) —
    // as much as everything else in this kernel together.  60 double sincos per point against 491 264 fp32 MACs: 2 % of the kernel.
    for (int f = 5 * half; f < 5 * half + 5; ++f)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double sd, cd;
            sincos((double)(p[c] * (float)(1 << f)), &sd, &cd);
            act[(3 + 6 * f + c) * NPT + pt] = (float)sd;
            act[(6 + 6 * f + c) * NPT + pt] = (float)cd;
        }
    if (half == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) act[c * NPT + pt] = p[c];
    } else {
        act[63 * NPT + pt] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void k_trunk_fp32(TrunkFp32Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float* wl = smem;
    float* act = smem + 2 * KC * WIDTH + wave * ACT_FLOATS;
    const long n_total = a.n_dev != nullptr ? (long)*a.n_dev : a.n;
    const long n_groups = (n_total + 127) / 128;
    auto Wt = [&](int l) { return a.blob + a.w_off[l]; };
    auto Bs = [&](int l) { return a.blob + a.b_off[l]; };
    for (long g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const long p = g * 128 + wave * 32 + (lane & 31);
        const bool valid = p < n_total;
        float x[3] = {0.0f, 0.0f, 0.0f};
        if (valid) { x[0] = a.pts[3 * p]; x[1] = a.pts[3 * p + 1]; x[2] = a.pts[3 * p + 2]; }
        f32x16 acc[8];
        write_encoding(act, x, lane);
        start(acc);
        gemm(Wt(0), 63, 0, 63, 64, act, wl, acc, t, lane);                     // 0: x63 -> h
        finish_relu(acc, Bs(0), act, lane);
        for (int l = 1; l <= 4; ++l) {                                         // 1..4
            start(acc);
            gemm(Wt(l), WIDTH, 0, WIDTH, WIDTH, act, wl, acc, t, lane);
            finish_relu(acc, Bs(l), act, lane);
        }
        // 5: cat([x63, h]) (ibl_nerf.py:168), columns in that order: the encoding is written over h's first 64 rows — which wait in registers meanwhile — for the
        // first 63 columns, then h comes back for the other 256
        start(acc);
        {
            float keep[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) keep[j] = act[(32 * (lane >> 5) + j) * NPT + (lane & 31)];
            write_encoding(act, x, lane);
            gemm(Wt(5), 63 + WIDTH, 0, 63, 64, act, wl, acc, t, lane);
#pragma unroll
            for (int j = 0; j < 32; ++j) act[(32 * (lane >> 5) + j) * NPT + (lane & 31)] = keep[j];
        }
        gemm(Wt(5), 63 + WIDTH, 63, WIDTH, WIDTH, act, wl, acc, t, lane);
        finish_relu(acc, Bs(5), act, lane);
        for (int l = 6; l <= 7; ++l) {
            start(acc);
            gemm(Wt(l), WIDTH, 0, WIDTH, WIDTH, act, wl, acc, t, lane);
            finish_relu(acc, Bs(l), act, lane);
        }
        // sigma_linear: one row; products and sum in double, one rounding, the bias last (oracle/csrc/render.c head(): the N = 1 heads of the reference run as a
        // vectorised dot product whose order is not a chain — the exact sum is the neutral choice); the two lane halves take alternate k
        const float* ws = Wt(8);
        double sd = 0.0;
        for (int k = (lane >> 5); k < WIDTH; k += 2) sd = fma((double)ws[k], (double)act[k * NPT + (lane & 31)], sd);
        sd += __shfl_xor(sd, 32);
        const float s = (float)sd + Bs(8)[0];
        if (valid && lane < 32) a.out[(a.out_index != nullptr ? (long)a.out_index[p] : p) * a.out_stride] = s;
    }
}

}  // namespace

hipError_t launch_trunk_fp32(const TrunkFp32Args& a, int n_cu, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    static bool attr_set[64] = {};   // per device: the attribute is, and one process may hold contexts on several (iblnerf_options.device)
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_trunk_fp32), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const long groups = (a.n + 127) / 128;
    const unsigned grid = (unsigned)(groups < n_cu ? groups : n_cu);
    hipLaunchKernelGGL(k_trunk_fp32, dim3(grid), dim3(256), SMEM_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace ibl
