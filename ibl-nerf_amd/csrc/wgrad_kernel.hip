// Weight gradient of the trunk (positions_linears.0..7, sigma_linear) from the operand stash the backward form of the fused MLP
// kernel leaves behind (mlp_kernel.hip, VAR_TRUNK_BWD; layout.h: STASH_*).  Replaces what torch.autograd accumulates into
// `.grad` of those parameters for the gradient-carrying trunk query of a training step (train.py:479-481 through
// ibl_nerf.py:154-176).
//
//   dW(l)[m][n] = sum_p dZ(l)[p][m] * X(l-1)[p][n]          a GEMM whose reduction runs over the POINTS
//   db(l)[m]    = sum_p dZ(l)[p][m]
//   d sigma_linear.weight[n] = sum_p dL/dsigma[p] * X(7)[p][n],   d sigma_linear.bias = sum_p dL/dsigma[p]
//
// The stash holds the fused kernel's own fragments: a lane has 8 FEATURES of one point; the matrix core wants 8 POINTS of one
// feature per lane for both operands.  The transposition is done BY the matrix core, in registers: with the fragments of a
// 32-feature tile as the A operand (row = point) and a 0/1 selection matrix as B,   D[p][c] = sum_k A[p][k] S[k][c]   puts
// feature c of the tile into lane c (two v_mfma_f32_32x32x16_f16 per 32 x 32 block, exact: f16 values times 1.0 in fp32).
// Lane c then holds its feature for points (r&3) + 8(r>>2) + 4h, r = 0..15 — some fixed permutation of the 32 points, the same
// for dZ and for X, and a contraction does not care in which order its terms come.  Registers 0..7 / 8..15 convert (exactly)
// back to the f16 operand fragments of two K = 16 steps.  No LDS, no scattered access: 16-byte coalesced loads only.
//
// One wave owns a 128 x 128 block of dW(l) (4 x 4 tiles in the accumulator file) and walks its share of the 32-point wave
// groups: per group 16 transposing + 32 product MFMAs.  Partial sums per split go to a scratch buffer; k_wgrad_reduce adds the
// splits and writes the reference's state-dict layout ([out][in] row-major, encoding columns back in embedder order).
// Operands are the f16 `hi` fragments (2^-11 per operand, random over >= thousands of points per sum); accumulation is fp32.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "layout.h"

namespace ibl {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// 32 features x 32 points of one stash activation, tile t: fragments j = 2t, 2t+1  ->  fp32 [feature lane][16 points] ...
__device__ __forceinline__ f32x16 transpose_issue(const f16x8& f0, const f16x8& f1, const f16x8& sel0, const f16x8& sel1) {
    f32x16 d = {0};
    d = MFMA16(f0, sel0, d);
    d = MFMA16(f1, sel1, d);
    return d;
}
// ... -> two operand fragments (K = 16 points each), exact
__device__ __forceinline__ void to_operands(const f32x16& d, f16x8 (&op)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 x = {d[8 * ks + 2 * q], d[8 * ks + 2 * q + 1]};
            v[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));
        }
        op[ks] = __builtin_bit_cast(f16x8, v);
    }
}

// MT x NT tiles per wave.  Hidden GEMMs (256 columns): MT = NT = 4, waves 2 x 2.  Encoding GEMMs (64 columns): MT = 2, NT = 2, waves 4 x 1.
// Schedule of one 32-point group (one wave per SIMD, nothing else hides latency): the next group's fragments are requested first; the
// dZ tiles are transposed back to back; then per X tile n: convert tile n, ISSUE the transposition of tile n+1, run the 2 MT products
// of tile n — so a transposition's result is due only after a tile's worth of products.  The sched_barriers keep that order (left
// alone, the compiler batches all transpositions first and spills the prefetched fragments).
template <int MT, int NT>
__device__ __forceinline__ void wgrad_block(const WgradArgs& a, const WgradGemm& gm, int wave, int lane, int split) {
    const int mt0 = (NT == 4) ? MT * (wave >> 1) : 2 * wave;     // NT == 4: waves 2 x 2 over (2 MT) x 8 tiles; else 4 x 1
    const int nt0 = (NT == 4) ? 4 * (wave & 1) : 0;
    const int h = lane >> 5, i = lane & 31;
    // selection fragments: B[k = (h, e)][col i] = 1 where the tile-local feature of slot (s, h, e) is i
    f16x8 sel[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) sel[s][e] = (i == (e & 3) + 8 * (2 * s + (e >> 2)) + 4 * h) ? (_Float16)1.0f : (_Float16)0.0f;

    f32x16 C[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) C[m][n] = f32x16{0};
    // db(l) = sum_p dZ(l)[p]: the transposed dZ tiles have feature m in lane m and 16 of the 32 points in its registers, so the bias
    // gradient is a by-product (one wave column per layer keeps it: gm.bias_part >= 0 and this wave's first column tile is 0)
    float bsum[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) bsum[m] = 0.0f;
    const bool keeps_bias = gm.bias_part >= 0 && nt0 == 0;

    const long per = (a.wave_groups + a.n_split - 1) / a.n_split;
    const long g0 = split * per, g1 = g0 + per < a.wave_groups ? g0 + per : a.wave_groups;
    const char* zbase = a.stash + stash_offset(gm.dz_what, a.wave_groups, 0) + (size_t)(2 * mt0) * 1024 + lane * 16;
    const long zstride = STASH_ACT_BYTES;
    const char* xbase = a.stash + stash_offset(gm.x_what, a.wave_groups, 0) + (size_t)(2 * nt0) * 1024 + lane * 16;
    const long xstride = gm.x_what == STASH_ENC ? STASH_ENC_BYTES : gm.x_what == STASH_DENC ? STASH_DENC_BYTES : STASH_ACT_BYTES;

    f16x8 fz[MT][2], fx[NT][2], nz[MT][2], nx[NT][2];
    auto load = [&](long g, f16x8 (&z)[MT][2], f16x8 (&x)[NT][2]) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int s = 0; s < 2; ++s) z[m][s] = *reinterpret_cast<const f16x8*>(zbase + g * zstride + (2 * m + s) * 1024);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int s = 0; s < 2; ++s) x[n][s] = *reinterpret_cast<const f16x8*>(xbase + g * xstride + (2 * n + s) * 1024);
    };
    if (g0 < g1) load(g0, fz, fx);
    for (long g = g0; g < g1; ++g) {
        load(g + 1 < g1 ? g + 1 : g, nz, nx);      // the next group's fragments travel while this group's products run
        __builtin_amdgcn_sched_barrier(0);
        f16x8 A[MT][2];
        {
            f32x16 d[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) d[m] = transpose_issue(fz[m][0], fz[m][1], sel[0], sel[1]);
            f32x16 dx = transpose_issue(fx[0][0], fx[0][1], sel[0], sel[1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m) to_operands(d[m], A[m]);
            if (keeps_bias) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float t = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += d[m][r];
                    bsum[m] += t;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f16x8 B[2];
                to_operands(dx, B);
                __builtin_amdgcn_sched_barrier(0);
                if (n + 1 < NT) dx = transpose_issue(fx[n + 1 < NT ? n + 1 : n][0], fx[n + 1 < NT ? n + 1 : n][1], sel[0], sel[1]);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    C[m][n] = MFMA16(A[m][0], B[0], C[m][n]);
                    C[m][n] = MFMA16(A[m][1], B[1], C[m][n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) { fz[m][0] = nz[m][0]; fz[m][1] = nz[m][1]; }
#pragma unroll
        for (int n = 0; n < NT; ++n) { fx[n][0] = nx[n][0]; fx[n][1] = nx[n][1]; }
    }
    if (keeps_bias) {   // lane (i, h) holds feature 32(mt0 + m) + i for the points of its half: the halves are added by the reduce kernel
#pragma unroll
        for (int m = 0; m < MT; ++m) a.partial[(size_t)split * a.partial_stride + gm.bias_part + h * 256 + 32 * (mt0 + m) + i] = bsum[m];
    }
    // partial[split][gemm]: [256 rows][ncols]; accumulator register r of lane (i, h) = row (r&3) + 8(r>>2) + 4h, column i of its tile
    float* part = a.partial + (size_t)split * a.partial_stride + gm.part_off;
    const int ncols = gm.ncols;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                part[(size_t)(32 * (mt0 + m) + (r & 3) + 8 * (r >> 2) + 4 * h) * ncols + 32 * (nt0 + n) + i] = C[m][n][r];
}

__global__ __launch_bounds__(256, 1) void k_wgrad(WgradArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const WgradGemm gm = a.gemm[blockIdx.x];
    if (gm.nrows == 128) wgrad_block<2, 4>(a, gm, wave, lane, blockIdx.y);
    else if (gm.ncols == 64) wgrad_block<2, 2>(a, gm, wave, lane, blockIdx.y);
    else if (gm.ncols == 32) wgrad_block<2, 1>(a, gm, wave, lane, blockIdx.y);
    else wgrad_block<4, 4>(a, gm, wave, lane, blockIdx.y);
}

// sum of the splits -> the reference's [out][in] row-major weight gradient inside the state-dict blob
__global__ void k_wgrad_reduce(WgradArgs a) {
    const WgradGemm gm = a.gemm[blockIdx.y];
    const bool enc = gm.enc_pairs > 0;
    const int ncols = gm.ncols;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= gm.nrows * ncols) return;
    const int m = idx / ncols, n = idx - m * ncols;
    int col = n;
    if (enc) {   // tile-local feature -> encoding slot -> embedder column (layout.h: enc_ref_index); pad slots have no column
        const int t = n >> 5, nl = n & 31;
        const int h = (nl >> 2) & 1, s = nl >> 4, e = (nl & 3) + 4 * ((nl >> 3) & 1);
        col = enc_ref_index(8 * (2 * t + s) + e, h, gm.enc_pairs);
        if (col < 0) return;
    }
    float acc = 0.0f;
    for (int s = 0; s < a.n_split; ++s) acc += a.partial[(size_t)s * a.partial_stride + gm.part_off + idx];
    a.grad[gm.blob_off + (size_t)m * gm.in_dim + gm.col_base + col] += acc * a.unscale;   // (+=: the caller zeroes the blob and may walk the points in pieces)
    if (gm.bias_part >= 0 && n == 0) {      // the layer's bias gradient: both lane halves of every split
        float b = 0.0f;
        for (int s = 0; s < a.n_split; ++s) {
            const float* bp = a.partial + (size_t)s * a.partial_stride + gm.bias_part;
            b += bp[m] + bp[256 + m];
        }
        a.grad[gm.bias_off + m] += b * a.unscale;
    }
}

// d sigma_linear.weight = sum_p dL/dsigma[p] X(7)[p], d sigma_linear.bias = sum_p dL/dsigma[p]  (the biases of the trunk layers come out of
// k_wgrad).  block = quarter of the 16 k-steps; thread = (k-step, lane) of the fragment layout; grid.y splits the groups.
__global__ __launch_bounds__(256) void k_head_grad(WgradArgs a, const float* __restrict__ dsigma, long n_pts) {
    const int what = 8, jq = blockIdx.x & 3;
    const int lane = threadIdx.x & 63, j = 4 * jq + (threadIdx.x >> 6);
    const int h = lane >> 5, i = lane & 31;
    const bool head = what == 8;
    if (head && dsigma == nullptr) return;      // the caller keeps sigma_linear (upstream gradient given on the trunk features)
    const char* base = a.stash + stash_offset(head ? STASH_X + 7 : STASH_DZ + what, a.wave_groups, 0) + j * 1024 + lane * 16;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float accb = 0.0f;
    for (long g = blockIdx.y; g < a.wave_groups; g += gridDim.y) {
        const f16x8 f = *reinterpret_cast<const f16x8*>(base + g * STASH_ACT_BYTES);
        float w = a.unscale;          // dZ carries the gradient scale; X(7) and dL/dsigma do not
        if (head) {
            const long p = g * 32 + i;
            w = p < n_pts ? dsigma[p] : 0.0f;
            accb += w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += w * (float)f[e];
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {       // over the 32 points of a lane half
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], m);
        accb += __shfl_xor(accb, m);
    }
    if (i == 0) {
        float* dst = a.grad + a.sigma_w_off;
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(dst + 32 * (j >> 1) + acc_feature(8 * (j & 1) + e, h), acc[e]);
        if (head && j == 0 && h == 0) atomicAdd(a.grad + a.sigma_b_off, accb);
    }
}

}  // namespace

// the N = 1/3 heads of the whole-network backward (WgradArgs::head): block = (head, quarter of the 16 k-steps), thread = (k-step, lane)
__global__ __launch_bounds__(256) void k_heads_grad(WgradArgs a, const float* __restrict__ up, long n_pts) {
    const WgradArgs::Head hd = a.head[blockIdx.x >> 2];
    const int jq = blockIdx.x & 3;
    const int lane = threadIdx.x & 63, j = 4 * jq + (threadIdx.x >> 6);
    if (j >= hd.n_ksteps) return;
    const int h = lane >> 5, i = lane & 31;
    const char* base = a.stash + stash_offset(hd.x_what, a.wave_groups, 0) + j * 1024 + lane * 16;
    float acc[3][8], accb[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[c][e] = 0.0f;
    for (long g = blockIdx.y; g < a.wave_groups; g += gridDim.y) {
        const f16x8 f = *reinterpret_cast<const f16x8*>(base + g * STASH_ACT_BYTES);
        const long p = g * 32 + i;
        float w[3] = {0.f, 0.f, 0.f};
        if (p < n_pts)
            for (int c = 0; c < hd.nc; ++c) w[c] = up[p * RAW_CH + hd.ch0 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            accb[c] += w[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[c][e] += w[c] * (float)f[e];
        }
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[c][e] += __shfl_xor(acc[c][e], m);
            accb[c] += __shfl_xor(accb[c], m);
        }
    if (i == 0) {
        const int width = 16 * hd.n_ksteps;          // 256 or 128 input features
        for (int c = 0; c < hd.nc; ++c) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(a.grad + hd.w_off + (long)c * width + 32 * (j >> 1) + acc_feature(8 * (j & 1) + e, h), acc[c][e]);
            if (j == 0 && h == 0) atomicAdd(a.grad + hd.b_off + c, accb[c]);
        }
    }
}

hipError_t launch_wgrad(const WgradArgs& a, const float* dsigma, long n_pts, hipStream_t s) {
    hipLaunchKernelGGL(k_wgrad, dim3(a.n_gemm, a.n_split), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(256, a.n_gemm), dim3(256), 0, s, a);
    if (a.n_head > 0) hipLaunchKernelGGL(k_heads_grad, dim3(4 * a.n_head, 64), dim3(256), 0, s, a, dsigma, n_pts);
    else if (dsigma != nullptr) hipLaunchKernelGGL(k_head_grad, dim3(4, 64), dim3(256), 0, s, a, dsigma, n_pts);
    return hipGetLastError();
}

}  // namespace ibl
