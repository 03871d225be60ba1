// PositionDirectionMLP (src/networks/MLP.py:32-74) for `infer_depth`: render_rays evaluates the depth_mlp ONCE PER RAY, at the ray
// origin with the normalised view direction, and returns relu(out[..., 0]) as inferred_depth_map (ibl_nerf_renderer.py:722-726).
//
// 640 000 evaluations per frame against 2.5e8 of the main networks: 0.3 % of the work, so this is a plain fp32 kernel (one
// v_fma_f32 per MAC, fp32 operands: no split-precision question) and not another instantiation of the MFMA kernel.  One workgroup
// of 256 threads carries 16 rays through the 14 layers: thread = output neuron, 16 accumulators in registers, the layer's input
// activations in LDS as [k][ray] (one broadcast ds_read_b128 serves 4 rays), weights transposed on upload to [k][out] so a wave reads
// 256 contiguous bytes per k (L2-resident: 2.4 MB).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "sincos_enc.h"

namespace ibl {

namespace {

constexpr int P = 16;            // rays per workgroup (two activation buffers of (63 + 256) x P and (256 + 27) x P floats: 38 KB of LDS)
constexpr int W = 256, W2 = 128, E_P = 63, E_D = 27;

// y[o] = act(b[o] + sum_k Wt[k][o] * in[k][.])  for thread o < n_out; in = LDS [n_in][P]; result -> out LDS [o][P] at row offset o0
template <bool RELU>
__device__ __forceinline__ void layer(const float* __restrict__ wt, const float* __restrict__ bias, int n_in, int n_out,
                                      const float* in, float* out, int out_row0, int t) {
    if (t < n_out) {
        float acc[P];
        const float b = bias[t];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p] = b;
        for (int k = 0; k < n_in; ++k) {
            const float w = wt[(long)k * n_out + t];
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
        float4* dst = reinterpret_cast<float4*>(out + (out_row0 + t) * P);
#pragma unroll
        for (int q = 0; q < P / 4; ++q) {
            float4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            dst[q] = v;
        }
    }
}

// embed_L(x) = [x, sin(2^0 x), cos(2^0 x), ...] (positional_embedder.py:21-34) of one 3-vector into rows row0.. of an LDS [.][P] array
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

__global__ __launch_bounds__(256) void k_posdir_mlp(PosDirArgs a) {
    // [x63 | h256] (the skip layer's input is cat([input_pts, h]), MLP.py:64-65), a second 256-row buffer, the direction encoding
    __shared__ __attribute__((aligned(16))) float bufA[(E_P + W) * P];
    __shared__ __attribute__((aligned(16))) float bufB[(W + E_D) * P];
    const int t = threadIdx.x;
    const long r0 = (long)blockIdx.x * P;
    if (t < P) {
        const long r = r0 + t < a.n ? r0 + t : a.n - 1;     // tail rays repeat the last one (never stored)
        const float o[3] = {a.pts[3 * r], a.pts[3 * r + 1], a.pts[3 * r + 2]};
        float d[3] = {a.dirs[3 * r], a.dirs[3 * r + 1], a.dirs[3 * r + 2]};
        if (a.normalize_dirs) {                              // viewdirs = rays_d / ||rays_d|| (ibl_nerf_renderer.py:795)
            const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
            d[0] /= nrm; d[1] /= nrm; d[2] /= nrm;
        }
        embed(o, 10, bufA, 0, t);
        embed(d, 4, bufB, W, t);
    }
    __syncthreads();
    const float* w = a.weights;
    auto next = [&](int n_in, int n_out) { const float* p = w; w += (long)n_in * n_out + n_out; return p; };   // [Wt | bias]
    float* hA = bufA + E_P * P;      // h rows of bufA
    // positions_linears.0: x63 -> hA
    const float* l = next(E_P, W);
    layer<true>(l, l + E_P * W, E_P, W, bufA, bufA, E_P, t);
    __syncthreads();
    // positions_linears.1..4: hA -> bufB -> hA -> bufB -> hA
    for (int i = 1; i <= 4; ++i) {
        l = next(W, W);
        if (i & 1) layer<true>(l, l + W * W, W, W, hA, bufB, 0, t);
        else layer<true>(l, l + W * W, W, W, bufB, bufA, E_P, t);
        __syncthreads();
    }
    // positions_linears.5: cat([x63, h]) = bufA rows 0..318 -> bufB
    l = next(E_P + W, W);
    layer<true>(l, l + (E_P + W) * W, E_P + W, W, bufA, bufB, 0, t);
    __syncthreads();
    l = next(W, W);                  // .6: bufB -> hA
    layer<true>(l, l + W * W, W, W, bufB, bufA, E_P, t);
    __syncthreads();
    l = next(W, W);                  // .7: hA -> bufB
    layer<true>(l, l + W * W, W, W, hA, bufB, 0, t);
    __syncthreads();
    l = next(W, W);                  // feature_linear (no activation): bufB -> hA ... but the view layer reads cat([feature, dirs])
    layer<false>(l, l + W * W, W, W, bufB, bufA, E_P, t);
    __syncthreads();
    // move the feature under the direction rows of bufB: bufB rows 0..255 = feature, 256..282 = dirs (MLP.py:68)
    for (int i = t; i < W * P; i += 256) bufB[i] = hA[i];
    __syncthreads();
    l = next(W + E_D, W2);           // views_linears.0: 283 -> 128
    layer<true>(l, l + (W + E_D) * W2, W + E_D, W2, bufB, bufA, 0, t);
    __syncthreads();
    for (int i = 1; i <= 3; ++i) {   // views_linears.1..3: 128 -> 128 (bufA rows 0..127 <-> bufB rows 0..127)
        l = next(W2, W2);
        if (i & 1) layer<true>(l, l + W2 * W2, W2, W2, bufA, bufB, 0, t);
        else layer<true>(l, l + W2 * W2, W2, W2, bufB, bufA, 0, t);
        __syncthreads();
    }
    // final_linear: 128 -> out_ch, from bufB (after views_linears.3); one thread per (ray, channel)
    l = next(W2, a.out_ch);
    for (int i = t; i < P * a.out_ch; i += 256) {
        const int p = i / a.out_ch, c = i % a.out_ch;
        float acc = l[W2 * a.out_ch + c];
        for (int k = 0; k < W2; ++k) acc = fmaf(l[k * a.out_ch + c], bufB[k * P + p], acc);
        if (a.relu_out) acc = fmaxf(acc, 0.0f);              // F.relu(inferred_depth_map[..., 0]) (:724)
        if (r0 + p < a.n) a.out[(r0 + p) * a.out_ch + c] = acc;
    }
}

}  // namespace

hipError_t launch_posdir_mlp(const PosDirArgs& a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_posdir_mlp, dim3((unsigned)((a.n + P - 1) / P)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ibl
