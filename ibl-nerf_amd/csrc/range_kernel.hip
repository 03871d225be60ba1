// Per-activation magnitude of an IBLNeRF network on a sample of points (iblnerf_layer_ranges): the largest |value| of every 256- / 128-wide
// activation of IBLNeRF.forward (src/nerf_models/ibl_nerf.py:154-210) — the eight trunk layers, feature_linear's output, the albedo / irradiance
// feature layers, views_linears.0 and the three additional-radiance feature layers.  No reference counterpart: the reference computes in fp32,
// whose range no activation leaves; the f16 product schemes of the fused MLP kernels hold activations as f16 pairs (< 65504).  With these maxima the
// host rescales the network by powers of two (ibl-nerf_amd/checkpoint.py scale_activations: relu(t x) = t relu(x), so scaling a layer's weights,
// biases and its consumers' columns by powers of two is EXACT) until every activation fits, instead of giving the call to the 2^-17 bf16x3 kernels.
// A measurement, not a product path: plain fp32 VALU (one v_fma_f32 per MAC), thread = output neuron, 16 points per workgroup, activations in
// LDS as [k][point]; weights read in the state-dict's own [out][in] layout (a few thousand points: ~1 ms).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "sincos_enc.h"

namespace ibl {

namespace {

constexpr int P = 16;
constexpr int W = 256, W2 = 128, E_P = 63, E_D = 27;

// y[t] = act(b[t] + sum_k Wm[t][k] in[k][.]) for thread t < n_out -> out rows out_row0 + t; returns the thread's largest |y| (after the activation)
template <bool RELU>
__device__ __forceinline__ float layer(const float* __restrict__ wm, const float* __restrict__ bias, int n_in, int n_out, const float* in, float* out,
                                       int out_row0, int t, int n_valid) {
    float mx = 0.0f;
    if (t < n_out) {
        float acc[P];
        const float b = bias[t];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p] = b;
        const float* row_w = wm + (long)t * n_in;
        for (int k = 0; k < n_in; ++k) {
            const float w = row_w[k];
            const float4* row = reinterpret_cast<const float4*>(in + k * P);
#pragma unroll
            for (int q = 0; q < P / 4; ++q) {
                const float4 v = row[q];
                acc[4 * q + 0] = fmaf(w, v.x, acc[4 * q + 0]);
                acc[4 * q + 1] = fmaf(w, v.y, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(w, v.z, acc[4 * q + 2]);
                acc[4 * q + 3] = fmaf(w, v.w, acc[4 * q + 3]);
            }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float v = RELU ? fmaxf(acc[p], 0.0f) : acc[p];
            out[(out_row0 + t) * P + p] = v;
            if (p < n_valid) mx = fmaxf(mx, fabsf(v));
        }
    }
    return mx;
}

__device__ __forceinline__ void embed(const float v[3], int n_freq, float* lds, int row0, int p) {
    for (int c = 0; c < 3; ++c) lds[(row0 + c) * P + p] = v[c];
    for (int c = 0; c < 3; ++c) {
        const TurnPair tp = to_turns(v[c]);
        for (int f = 0; f < n_freq; ++f) {
            float s, co;
            sincos_turns(tp, (float)(1 << f), &s, &co);
            lds[(row0 + 3 + 6 * f + c) * P + p] = s;
            lds[(row0 + 6 + 6 * f + c) * P + p] = co;
        }
    }
}

// the workgroup's largest value -> d_max[slot] (non-negative floats order like their bit patterns)
__device__ __forceinline__ void publish(float mx, unsigned* d_max, int slot, unsigned* red) {
    unsigned b = __builtin_bit_cast(unsigned, mx);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)b, o);
        b = other > b ? other : b;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = red[0];
        for (int w = 1; w < 4; ++w) m = red[w] > m ? red[w] : m;
        atomicMax(d_max + slot, m);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_layer_ranges(LayerRangeArgs a) {
    __shared__ __attribute__((aligned(16))) float bufA[(E_P + W) * P];     // [x63 | h]
    __shared__ __attribute__((aligned(16))) float bufB[(W + E_D) * P];     // [h | dir27]
    __shared__ __attribute__((aligned(16))) float bufC[W * P];             // h7
    __shared__ unsigned red[4];
    const int t = threadIdx.x;
    const long r0 = (long)blockIdx.x * P;
    const int n_valid = (int)(a.n - r0 < P ? a.n - r0 : P);
    if (t < P) {
        const long r = r0 + t < a.n ? r0 + t : a.n - 1;
        const float o[3] = {a.pts[3 * r], a.pts[3 * r + 1], a.pts[3 * r + 2]};
        const float d[3] = {a.dirs[3 * r], a.dirs[3 * r + 1], a.dirs[3 * r + 2]};
        embed(o, 10, bufA, 0, t);
        embed(d, 4, bufB, W, t);
    }
    __syncthreads();
    unsigned* dm = reinterpret_cast<unsigned*>(a.d_max);
    auto Wt = [&](int l) { return a.blob + a.w_off[l]; };
    auto Bs = [&](int l) { return a.blob + a.b_off[l]; };
    float* hA = bufA + E_P * P;
    // blob layers (checkpoint.SCHEMA): 0-7 positions_linears, 8 views_linears.0, 9 feature_linear, 10 sigma, 11 albedo_feature, 12 albedo, 13 roughness,
    // 14 irradiance_feature, 15 irradiance, 16 radiance, 17-19 additional_radiance_feature.k, 20-22 additional_radiance_linear.k
    float m = layer<true>(Wt(0), Bs(0), E_P, W, bufA, bufA, E_P, t, n_valid);                 // 0: x63 -> hA
    __syncthreads(); publish(m, dm, 0, red);
    for (int i = 1; i <= 4; ++i) {                                                            // 1..4: hA -> bufB -> hA -> bufB -> hA
        m = (i & 1) ? layer<true>(Wt(i), Bs(i), W, W, hA, bufB, 0, t, n_valid) : layer<true>(Wt(i), Bs(i), W, W, bufB, bufA, E_P, t, n_valid);
        __syncthreads(); publish(m, dm, i, red);
    }
    m = layer<true>(Wt(5), Bs(5), E_P + W, W, bufA, bufB, 0, t, n_valid);                     // 5: cat([x63, h]) -> bufB
    __syncthreads(); publish(m, dm, 5, red);
    m = layer<true>(Wt(6), Bs(6), W, W, bufB, bufA, E_P, t, n_valid);                         // 6: bufB -> hA
    __syncthreads(); publish(m, dm, 6, red);
    m = layer<true>(Wt(7), Bs(7), W, W, hA, bufC, 0, t, n_valid);                             // 7: hA -> bufC = h7
    __syncthreads(); publish(m, dm, 7, red);
    m = layer<true>(Wt(11), Bs(11), W, W2, bufC, bufA, E_P, t, n_valid);                      // albedo_feature_linear(h7)
    __syncthreads(); publish(m, dm, 9, red);
    m = layer<true>(Wt(14), Bs(14), W, W2, bufC, bufA, E_P, t, n_valid);                      // irradiance_feature_linear(h7)
    __syncthreads(); publish(m, dm, 10, red);
    const float* h2 = bufC;                                                                   // what the radiance heads read: h7 when colour-independent
    if (!a.color_independent) {
        m = layer<false>(Wt(9), Bs(9), W, W, bufC, bufB, 0, t, n_valid);                      // feature_linear (no activation): -> bufB rows 0..255, dirs behind
        __syncthreads(); publish(m, dm, 8, red);
        m = layer<true>(Wt(8), Bs(8), W + E_D, W, bufB, bufA, E_P, t, n_valid);               // views_linears.0(cat([feature, dir27])) -> hA
        __syncthreads(); publish(m, dm, 11, red);
        h2 = hA;
    }
    for (int k = 0; k < 3; ++k) {                                                             // additional_radiance_feature_linear.k(h2)
        m = layer<true>(Wt(17 + k), Bs(17 + k), W, W2, h2, bufB, 0, t, n_valid);
        __syncthreads(); publish(m, dm, 12 + k, red);
    }
}

}  // namespace

hipError_t launch_layer_ranges(const LayerRangeArgs& a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_layer_ranges, dim3((unsigned)((a.n + P - 1) / P)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ibl
