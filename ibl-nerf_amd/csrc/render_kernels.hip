// Per-ray kernels of the render path (everything except the MLP): ray generation, packed point
// batches, alpha compositing with wavefront scans, epsilon-normal, split-sum shading, inverse-CDF
// fine sampling.  fp32 throughout; this file is compiled with -ffp-contract=off so that
// "multiply then add" stays two roundings exactly where the reference's elementwise torch ops
// round twice (sample positions feed a 2^9 frequency multiplier in the encoding).
//
// Work decomposition: ONE WAVEFRONT PER RAY, samples spread over lanes (S = 64 -> 1 per lane,
// S = 192 -> 3 consecutive per lane), transmittance by a 64-lane multiplicative scan, all
// per-ray sums by butterfly reductions.  These kernels are HBM/latency-trivial next to the MLP
// (about 22 KB read per ray against >1 GFLOP of MFMA work).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "gen_points.h"
#include "kernels.h"

namespace ibl {

namespace {

constexpr int MAX_NPL = 4;   // samples per lane: S <= 256

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float srgb(float x, int on) { return on ? powf(x + 1e-12f, (float)(1.0 / 2.2)) : x; }
// output mapping of radiance-like maps: gamma(ldr(x)); on = bit0 gamma_correct, bit1 use_radiance_linear
// (tonemap_reinherd x/(x+1), ibl_nerf_renderer.py:30-31, :480-487)
__device__ __forceinline__ float out_map(float x, int on) { return srgb((on & 2) ? x / (x + 1.0f) : x, on & 1); }
// radiance_f (:192-197): sigmoid, or ReLU under use_radiance_linear
__device__ __forceinline__ float radiance_f(float x, int linear) { return linear ? fmaxf(x, 0.0f) : sigmoidf_(x); }

// torch.linspace(start, end, steps)[i] in fp32 (ATen CPU kernel: symmetric fill, one fma each)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - 1 - i), end);
}

// Front-to-back compositing weights of one ray (ibl_nerf_renderer.py:203-206, 241-245):
//   dist_k = (z_{k+1} - z_k) * |d|, last = 1e10 * |d|;  alpha = 1 - exp(-relu(sigma) * dist)
//   T_k = prod_{j<k} (1 - alpha_j + 1e-10)  — accumulated in double and rounded per prefix, which
//   is what ATen's CPU cumprod does for float tensors;  w = alpha * T.
// Lane owns samples lane*NPL .. lane*NPL+NPL-1.
template <int NPL>
__device__ __forceinline__ void ray_weights(const float (&sigma)[NPL], const float (&z)[NPL], const float (&zn)[NPL],
                                            float norm, int S, int lane, float (&w)[NPL], float* visibility = nullptr) {
    float alpha[NPL];
    double om[NPL];
    double lane_prod = 1.0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        const float dist = (s == S - 1 ? 1e10f : (zn[i] - z[i])) * norm;
        float a = 1.0f - expf(-fmaxf(sigma[i], 0.0f) * dist);
        if (s >= S) a = 0.0f;
        alpha[i] = a;
        om[i] = s < S ? (double)((1.0f - a) + 1e-10f) : 1.0;
        lane_prod *= om[i];
    }
    // exclusive multiplicative scan over lanes
    double incl = lane_prod;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(incl, d);
        if (lane >= d) incl *= o;
    }
    double T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        w[i] = alpha[i] * (float)T;
        T *= om[i];
    }
    if (visibility != nullptr) *visibility = (float)__shfl(incl, 63);   // cumprod(...)[:, -1]: the product over every sample (raw2outputs_depth, :140-141)
}

// d depth / d raw_s of depth = sum_s w_s z_s — what autograd returns through relu, exp, cumprod and the sum in
// get_normal_from_depth_gradient(_direction) (normal_from_depth.py:39-47, :124-132):
//   d depth / d alpha_s = T_s z_s - (sum_{i>s} w_i z_i) / (1 - alpha_s + 1e-10);   d alpha_s / d raw_s = dist_s exp(-raw_s dist_s) [raw_s > 0]
// in double (the reference's float32 backward differs from it by its own rounding).
template <int NPL>
__device__ __forceinline__ void ray_depth_grad(const float (&sigma)[NPL], const float (&z)[NPL], const float (&zn)[NPL],
                                               float norm, int S, int lane, double (&G)[NPL]) {
    double alpha[NPL], om[NPL], da[NPL], T[NPL];
    double lane_prod = 1.0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        const double dist = (double)((s == S - 1 ? 1e10f : (zn[i] - z[i])) * norm);
        const double sr = sigma[i] > 0.0f ? (double)sigma[i] : 0.0;
        const double e = exp(-sr * dist);
        alpha[i] = s < S ? 1.0 - e : 0.0;
        om[i] = s < S ? (1.0 - alpha[i]) + 1e-10 : 1.0;
        da[i] = (s < S && sigma[i] > 0.0f) ? dist * e : 0.0;
        lane_prod *= om[i];
    }
    double incl = lane_prod;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(incl, d);
        if (lane >= d) incl *= o;
    }
    double t = __shfl_up(incl, 1);
    if (lane == 0) t = 1.0;
    double lane_wz = 0.0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        T[i] = t;
        lane_wz += alpha[i] * t * (double)z[i];
        t *= om[i];
    }
    double sfx = lane_wz;                      // inclusive suffix sum over lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_down(sfx, d);
        if (lane + d < 64) sfx += o;
    }
    double after = __shfl_down(sfx, 1);        // samples of the lanes behind this one
    if (lane == 63) after = 0.0;
#pragma unroll
    for (int i = NPL - 1; i >= 0; --i) {
        G[i] = (T[i] * (double)z[i] - after / om[i]) * da[i];
        after += alpha[i] * T[i] * (double)z[i];
    }
}

__device__ __forceinline__ void cross3(const float (&a)[3], const float (&b)[3], float (&c)[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void normalize3(float (&v)[3]) {   // F.normalize(dim=-1, eps=1e-12)
    const float n = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    v[0] /= n; v[1] /= n; v[2] /= n;
}
// right = d x (0,1,0); up = right x d   (normal_from_depth.py:143-147)
__device__ __forceinline__ void right_up(const float (&d)[3], float (&right)[3], float (&up)[3]) {
    const float up0[3] = {0.0f, 1.0f, 0.0f};
    cross3(d, up0, right);
    cross3(right, d, up);
}

// ------------------------------------------------------------------------------------------
// rows row0, row0 + row_step, ... (n_rows of them, W pixels each) — or, pixels != nullptr, the n_rows listed flat pixel indices row * W + col (a probe: any subset of a
// frame).  Every pixel's ray is computed by itself: a tile's or a probe's rays are the frame's, bit for bit.
__global__ void k_get_rays(int W, int row0, int row_step, long n_rows, const long long* __restrict__ pixels, Camera cam, float* __restrict__ ro, float* __restrict__ rd) {
    const float* K = cam.K;
    const float* c2w = cam.c2w;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (pixels ? n_rows : n_rows * W)) return;
    const int col = pixels ? (int)(pixels[idx] % W) : (int)(idx % W), row = pixels ? (int)(pixels[idx] / W) : row0 + row_step * (int)(idx / W);
    const float dir[3] = {((float)col - K[2]) / K[0], -((float)row - K[5]) / K[4], -1.0f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rd[3 * idx + c] = (dir[0] * c2w[4 * c + 0] + dir[1] * c2w[4 * c + 1]) + dir[2] * c2w[4 * c + 2];
        ro[3 * idx + c] = c2w[4 * c + 3];
    }
}

__global__ void k_coarse_z(float near, float far, int S, int lindisp, float* __restrict__ z) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= S) return;
    const float t = linspace_at(0.0f, 1.0f, S, i);
    z[i] = lindisp ? 1.0f / (1.0f / near * (1.0f - t) + 1.0f / far * t)      // ibl_nerf_renderer.py:674
                   : near * (1.0f - t) + far * t;                            // :672
}

// Stratified jitter of the coarse grid (ibl_nerf_renderer.py:678-692, perturb > 0): mids, upper = cat(mids, z[-1]),
// lower = cat(z[0], mids), z' = lower + (upper - lower) * t_rand — per ray.
__global__ void k_jitter_z(const float* __restrict__ z, int S, const float* __restrict__ t_rand, long R, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * S) return;
    const int k = (int)(idx % S);
    const float lower = k == 0 ? z[0] : 0.5f * (z[k] + z[k - 1]);
    const float upper = k == S - 1 ? z[S - 1] : 0.5f * (z[k + 1] + z[k]);
    out[idx] = lower + (upper - lower) * t_rand[idx];
}

// The coarse grid of rays with their own near / far planes (ibl_nerf_renderer.py:668-674 with near, far of shape [n, 1]) and, with t_rand, its
// stratified jitter (:678-692) — k_coarse_z and k_jitter_z per ray.
__global__ void k_ray_grid(const float* __restrict__ near, const float* __restrict__ far, int S, int lindisp, const float* __restrict__ t_rand, long R,
                           float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * S) return;
    const long r = idx / S;
    const int k = (int)(idx - r * S);
    const float nr = near[r], fr = far[r];
    auto zk = [&](int i) {
        const float t = linspace_at(0.0f, 1.0f, S, i);
        return lindisp ? 1.0f / (1.0f / nr * (1.0f - t) + 1.0f / fr * t) : nr * (1.0f - t) + fr * t;
    };
    if (t_rand == nullptr) { out[idx] = zk(k); return; }
    const float z0 = zk(k);
    const float lower = k == 0 ? z0 : 0.5f * (z0 + zk(k - 1));
    const float upper = k == S - 1 ? z0 : 0.5f * (zk(k + 1) + z0);
    out[idx] = lower + (upper - lower) * t_rand[idx];
}

// normal_from_depth.py:64-67: F.normalize(rays_d +- eps * right), F.normalize(rays_d +- eps * up)
__device__ __forceinline__ void tilted_dirs(const float* d, const float* right, const float* up, float eps, float nd[4][3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float er = eps * right[c], eu = eps * up[c];
        nd[0][c] = d[c] + er;
        nd[1][c] = d[c] - er;
        nd[2][c] = d[c] + eu;
        nd[3][c] = d[c] - eu;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) normalize3(nd[v]);
}

__global__ void k_make_points(int mode, const float* __restrict__ origin, const float* __restrict__ dir,
                              const float* __restrict__ z, int z_stride, float eps, long R, int S,
                              float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (r, s)
    if (idx >= R * S) return;
    const long r = idx / S;
    const int s = (int)(idx - r * S);
    const float zz = z[(long)z_stride * r + s];
    const float o[3] = {origin[3 * r], origin[3 * r + 1], origin[3 * r + 2]};
    const float d[3] = {dir[3 * r], dir[3 * r + 1], dir[3 * r + 2]};
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = o[c] + d[c] * zz;
    if (mode == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * idx + c] = p[c];
        return;
    }
    float right[3], up[3];
    right_up(d, right, up);
    if (mode == 2) {   // four rays from the same origin, directions tilted by +-eps*right / +-eps*up and normalised
        float nd[4][3];
        tilted_dirs(d, right, up, eps, nd);
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int c = 0; c < 3; ++c) out[3 * (v * R * S + idx) + c] = o[c] + nd[v][c] * zz;
        return;
    }
    // mode 1: the same device function the MLP kernels' input stage calls (gen_points.h), so a batch and a generated point are one arithmetic
    PointGen g;
    g.rays_o = origin; g.rays_d = dir; g.z = z; g.z_stride = z_stride; g.S = S; g.RS = (unsigned)(R * S); g.eps = eps;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float q[3];
        gen_offset_point(g, (unsigned)(v * R * S + idx), q[0], q[1], q[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[3 * (v * R * S + idx) + c] = q[c];
    }
}

// F.grid_sample(lut[1,3,512,512], bilinear, zeros padding, align_corners=True) at
// (2 n.v - 1, 2 rough - 1)  (ibl_nerf_renderer.py:418-421); returns channels 0 and 1.
__device__ __forceinline__ void lut_fetch(const float* __restrict__ lut, float ndv, float rough, float& e0, float& e1) {
    constexpr int N = 512;
    const float gx = 2.0f * ndv - 1.0f, gy = 2.0f * rough - 1.0f;
    const float x = ((gx + 1.0f) / 2.0f) * (float)(N - 1);
    const float y = ((gy + 1.0f) / 2.0f) * (float)(N - 1);
    const float x0 = floorf(x), y0 = floorf(y);
    e0 = 0.0f;
    e1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int dx = k & 1, dy = k >> 1;
        const float xi = x0 + (float)dx, yi = y0 + (float)dy;
        const float wx = dx ? (x - x0) : (x0 + 1.0f - x);
        const float wy = dy ? (y - y0) : (y0 + 1.0f - y);
        if (xi >= 0.0f && xi <= (float)(N - 1) && yi >= 0.0f && yi <= (float)(N - 1)) {
            const int o = (int)yi * N + (int)xi;
            const float wgt = wx * wy;
            e0 += lut[o] * wgt;
            e1 += lut[N * N + o] * wgt;
        }
    }
}

__device__ __forceinline__ void store3(float* p, long r, const float (&v)[3], int g) {
    if (p) { p[3 * r] = out_map(v[0], g); p[3 * r + 1] = out_map(v[1], g); p[3 * r + 2] = out_map(v[2], g); }
}

// target_depth_map IS depth_map (one tensor, ibl_nerf_renderer.py:250) unless depth_map_from_ground_truth replaces it
// (:251-252): an edited depth (:253-256) shows in depth_map / disp / the mip level only in the aliased case.
__device__ __forceinline__ float target_depth(const OverrideArgs& ov, long r, bool mask_all, float& depth) {
    float tdepth = depth;
    if (ov.gt_depth != nullptr) tdepth = ov.gt_depth[r];
    if (mask_all && ((ov.mode == 1 && ov.edit_depth) || ov.mode == 2)) {
        tdepth = ov.depth_img[(long)r * ov.depth_stride];
        if (ov.gt_depth == nullptr) depth = tdepth;
    }
    return tdepth;
}

// State record handed from pass A to pass B (floats): 0-2 albedo, 3 rough, 4 / 12 / 13 irradiance (r, g, b), 5-7 fresnel,
// 8-10 specular coefficient, 11 mip level.

// Pass A: raw2outputs up to the reflected-ray set-up (ibl_nerf_renderer.py:200-440).
template <int NPL>
__global__ __launch_bounds__(256) void k_pass_a(PassAArgs a, PassOutputs out, int gamma) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const int S = a.S;
    const float o[3] = {a.rays_o[3 * r], a.rays_o[3 * r + 1], a.rays_o[3 * r + 2]};
    const float d[3] = {a.rays_d[3 * r], a.rays_d[3 * r + 1], a.rays_d[3 * r + 2]};
    const float norm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float* zrow = a.z + (long)a.z_stride * r;

    float z[NPL], zn[NPL], sig[NPL], w[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        z[i] = s < S ? zrow[s] : 0.0f;
        zn[i] = s + 1 < S ? zrow[s + 1] : 0.0f;
        sig[i] = s < S ? a.raw[((long)r * S + s) * RAW_CH] : 0.0f;
        if (a.noise != nullptr && s < S) sig[i] = sig[i] + a.noise[(long)r * S + s];   // raw[..., 0] + noise (:242)
    }
    ray_weights<NPL>(sig, z, zn, norm, S, lane, w);

    float depth = 0.f, acc = 0.f, ch[17];
#pragma unroll
    for (int c = 0; c < 17; ++c) ch[c] = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        if (s < S) {
            a.weights[(long)r * S + s] = w[i];
            if (out.weights) out.weights[(long)r * S + s] = w[i];
            depth += w[i] * z[i];
            acc += w[i];
            // (a sample whose weight is exactly zero — alpha = 0: nine in ten on a scene with surfaces — adds +0 to every sum whatever its row holds: the row is not read;
            // 885 MB of raw rows per launch set of the fine pass otherwise)
            if (w[i] != 0.0f) {
                const float* row = a.raw + ((long)r * S + s) * RAW_CH;
#pragma unroll
                for (int c = 0; c < 17; ++c)   // albedo, roughness: sigmoid; irradiance, radiances: radiance_f (:281-318)
                    ch[c] += w[i] * ((c < 4 || (c == 4 && a.irradiance_sigmoid)) ? sigmoidf_(row[1 + c]) : radiance_f(row[1 + c], a.radiance_linear));
            }
        }
    }
    depth = wave_sum(depth);
    acc = wave_sum(acc);
#pragma unroll
    for (int c = 0; c < 17; ++c) ch[c] = wave_sum(ch[c]);

    // inferred normal (:273-276): sum_s w_s (2 sigmoid(normal_mlp(x_s)) - 1), not normalised
    float inf[3] = {0.f, 0.f, 0.f};
    if (a.nrm_raw != nullptr && a.nrm_at_surface) {   // one evaluation at the surface point (:268-271)
#pragma unroll
        for (int c = 0; c < 3; ++c) inf[c] = 2.0f * sigmoidf_(a.nrm_raw[3 * r + c]) - 1.0f;
    } else if (a.nrm_raw != nullptr) {
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int s = lane * NPL + i;
            if (s < S) {
                const float* row = a.nrm_raw + ((long)r * S + s) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) inf[c] += w[i] * (2.0f * sigmoidf_(row[c]) - 1.0f);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) inf[c] = wave_sum(inf[c]);
    }

    // epsilon-normal depths (normal_from_depth.py:158-176): same z / dists, trunk-only sigma
    float D[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        if (a.ov.gt_normal != nullptr || a.normal_inferred || a.grad_normal) break;   // "ground_truth" / "inferred_normal_map" / the gradient modes: no offset queries were made
        float sv[NPL], wv[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int s = lane * NPL + i;
            sv[i] = s < S ? a.sig4[((long)v * a.R + r) * S + s] : 0.0f;
        }
        ray_weights<NPL>(sv, z, zn, norm, S, lane, wv);
        float dv = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) dv += wv[i] * z[i];
        D[v] = wave_sum(dv);
    }

    // the autograd normal modes (normal_from_depth.py:102-137 position, :16-52 direction): d depth / d (a, b) at a = b = 0, where the
    // sample points move by a*right + b*up (position) or by z*(a*right + b*up) (direction); sig4 = [sigma, grad sigma] rows of the
    // density-gradient query at the unshifted points
    double gda = 0.0, gdb = 0.0;
    if (a.grad_normal && a.ov.gt_normal == nullptr && !a.normal_inferred) {
        float sv[NPL], gr[NPL][3];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int s = lane * NPL + i;
            const float* row = a.sig4 + ((long)r * S + (s < S ? s : 0)) * 4;
            sv[i] = s < S ? row[0] : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) gr[i][c] = s < S ? row[1 + c] : 0.0f;
        }
        double G[NPL];
        ray_depth_grad<NPL>(sv, z, zn, norm, S, lane, G);
        float rt[3], upv[3];
        right_up(d, rt, upv);
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const double g = a.grad_normal == 2 ? G[i] * (double)z[i] : G[i];
            gda += g * ((double)gr[i][0] * rt[0] + (double)gr[i][1] * rt[1] + (double)gr[i][2] * rt[2]);
            gdb += g * ((double)gr[i][0] * upv[0] + (double)gr[i][1] * upv[1] + (double)gr[i][2] * upv[2]);
        }
        gda = wave_sum_d(gda);
        gdb = wave_sum_d(gdb);
    }

    // ---- per-ray scalar section (all lanes compute, lane 0 stores) ---------------------------
    float albedo[3] = {ch[0], ch[1], ch[2]};
    float rough = ch[3];
    float irr[3] = {ch[4], ch[4], ch[4]};
    const OverrideArgs& ov = a.ov;
    // *_from_gt substitutions come first; the edit / insert overrides then act on the substituted maps (:320-330, :378-410)
    if (ov.gt_albedo != nullptr) for (int c = 0; c < 3; ++c) albedo[c] = ov.gt_albedo[3 * r + c];
    if (ov.gt_roughness != nullptr) rough = ov.gt_roughness[r];
    if (ov.gt_irradiance != nullptr) for (int c = 0; c < 3; ++c) irr[c] = ov.gt_irradiance[3 * r + c];
    float m = 0.0f;
    bool mask_all = false;
    if (ov.mode != 0) {   // ibl_nerf_renderer.py:223-228 / :233-238
        m = ov.mask[(long)r * ov.mask_stride];
        mask_all = m > 0.0f;
    }
    // object q <=> 9(q+1)/255 < m < 11(q+1)/255
    auto in_obj = [&](int q) { return (float)(11 * (q + 1) / 255.) > m && m > (float)(9 * (q + 1) / 255.); };
    const float tdepth = target_depth(ov, r, mask_all, depth);
    // :258 torch.max(1e-10, depth / acc): a NaN quotient (an empty ray: 0 / 0) stays NaN, unlike fmaxf
    const float dq = depth / acc;
    const float disp = 1.0f / (dq != dq ? dq : fmaxf(1e-10f, dq));
    float xs[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xs[c] = o[c] + d[c] * tdepth;   // :262

    float right[3], up[3], dxv[3], dyv[3], nrm[3];
    right_up(d, right, up);
    const float two_eps = 2.0f * a.eps;
    if (a.tilted_rays) {   // end points of the four tilted rays (normal_from_depth.py:88-94)
        float nd[4][3];
        tilted_dirs(d, right, up, a.eps, nd);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dxv[c] = (o[c] + D[0] * nd[0][c]) - (o[c] + D[1] * nd[1][c]);
            dyv[c] = (o[c] + D[2] * nd[2][c]) - (o[c] + D[3] * nd[3][c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dxv[c] = two_eps * right[c] + (D[0] - D[1]) * d[c];
            dyv[c] = two_eps * up[c] + (D[2] - D[3]) * d[c];
        }
    }
    cross3(dxv, dyv, nrm);
    normalize3(nrm);
    if (a.grad_normal) {   // grad = right * dx + up * dy;  normal = F.normalize(grad - rays_d)   (:48-51, :133-136)
#pragma unroll
        for (int c = 0; c < 3; ++c) nrm[c] = (right[c] * (float)gda + up[c] * (float)gdb) - d[c];
        normalize3(nrm);
    }
    if (a.normal_inferred) {
#pragma unroll
        for (int c = 0; c < 3; ++c) nrm[c] = inf[c];
    }
    if (a.ov.gt_normal != nullptr) {   // target_normal_map_for_radiance_calculation == "ground_truth" (:370-371)
#pragma unroll
        for (int c = 0; c < 3; ++c) nrm[c] = 2.0f * a.ov.gt_normal[3 * r + c] - 1.0f;
        normalize3(nrm);
    }
    const float nrm_before[3] = {nrm[0], nrm[1], nrm[2]};   // stage boundary: what get_normal_from_depth_gradient_epsilon returned

    if (mask_all && ((ov.mode == 1 && ov.edit_normal) || ov.mode == 2)) {   // :380-382, :401-403
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = 2.0f * ov.normal_img[3 * r + c] - 1.0f;
        normalize3(g);
#pragma unroll
        for (int c = 0; c < 3; ++c) nrm[c] = g[c];
    }
    // masked assignments are applied object by object in list order, as the reference's loops do
    if (ov.mode == 1) {
        if (ov.edit_albedo) {
            if (ov.edit_albedo_by_img) {
                if (mask_all) for (int c = 0; c < 3; ++c) albedo[c] = ov.albedo_img[3 * r + c];
            } else {
                for (int q = 0; q < ov.num_objects; ++q)
                    if (in_obj(q)) for (int c = 0; c < 3; ++c) albedo[c] = ov.albedo_list[3 * q + c];
            }
        }
        if (ov.edit_roughness) {
            if (ov.rough_img != nullptr) {            // edit_roughness_by_img (:394-395)
                if (mask_all) rough = ov.rough_img[r];
            } else {
                for (int q = 0; q < ov.n_rough_list; ++q)
                    if (in_obj(q)) rough = ov.rough_list[q];
            }
        }
    } else if (ov.mode == 2) {   // :406-410
        for (int q = 0; q < ov.num_objects; ++q)
            if (in_obj(q)) {
                rough = ov.rough_list[q];
                if (ov.irr_list[q] > 0.0f) irr[0] = irr[1] = irr[2] = ov.irr_list[q];
                for (int c = 0; c < 3; ++c) albedo[c] = ov.albedo_list[3 * q + c];
            }
    }

    float ndv = ((-d[0] * nrm[0]) + (-d[1] * nrm[1])) + (-d[2] * nrm[2]);   // :412
    ndv = fminf(fmaxf(ndv, 0.0f), 1.0f);
    float e0, e1;
    lut_fetch(a.lut, ndv, rough, e0, e1);
    const float metal = 1.0f - rough;
    float F0[3], fres[3], spec[3], rdir[3];
    const float p5 = powf(fminf(fmaxf(1.0f - ndv, 0.0f), 1.0f), 5.0f);
    const float ndotd = (nrm[0] * d[0] + nrm[1] * d[1]) + nrm[2] * d[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        F0[c] = 0.04f * (1.0f - metal) + albedo[c] * metal;                       // :427
        fres[c] = F0[c] + (fmaxf(1.0f - rough, F0[c]) - F0[c]) * p5;              // microfacet.py:8-12
        spec[c] = (a.lut_coefficient_F0 ? F0[c] : fres[c]) * e0 + e1;             // :433-436
        rdir[c] = d[c] - (2.0f * ndotd) * nrm[c];                                 // :439
    }
    // the mip level reads roughness_map, which is the edited tensor only while target_roughness_map aliases it (:324-326)
    const float rough_net = ov.gt_roughness != nullptr ? ch[3] : rough;
    float level = rough_net;
    if (a.correct_depth) {
        const float depth_0 = a.near_ray != nullptr ? (a.far_ray[r] + a.near_ray[r]) * 0.5f : (a.far + a.near) * 0.5f;   // :456-460 (per-ray planes: depth_0[..., 0])
        level = fminf(fmaxf(rough_net * depth / depth_0, 0.0f), 1.0f);
    }

    if (lane == 0) {
        float* st = a.state + r * ST_FLOATS;
        st[0] = albedo[0]; st[1] = albedo[1]; st[2] = albedo[2];
        st[3] = rough; st[4] = irr[0]; st[12] = irr[1]; st[13] = irr[2];
        st[5] = fres[0]; st[6] = fres[1]; st[7] = fres[2];
        st[8] = spec[0]; st[9] = spec[1]; st[10] = spec[2];
        st[11] = level;
#pragma unroll
        for (int c = 0; c < 3; ++c) { a.refl_o[3 * r + c] = xs[c]; a.refl_d[3 * r + c] = rdir[c]; }
        // maps that do not depend on the reflected pass (ibl_nerf_renderer.py:494-525)
        const int gm = (gamma ? 1 : 0) | (a.radiance_linear ? 2 : 0);
        const float rad[3] = {ch[5], ch[6], ch[7]};
        store3(out.radiance, r, rad, gm);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float rk[3] = {ch[8 + 3 * k], ch[9 + 3 * k], ch[10 + 3 * k]};
            store3(out.radiance_k[k], r, rk, gm);
        }
        if (out.irradiance) {
            if (ov.gt_irradiance != nullptr) store3(out.irradiance, r, irr, gm);   // [R,3] in this mode
            else out.irradiance[r] = out_map(irr[0], gm);
        }
        store3(out.albedo, r, albedo, gamma ? 1 : 0);     // albedo_f: gamma only (:488)
        if (out.roughness) out.roughness[r] = rough;
        if (out.n_dot_v) out.n_dot_v[r] = ndv;
        store3(out.normal, r, nrm, 0);
        if (a.nrm_raw != nullptr) store3(out.inferred_normal, r, inf, 0);
        if (out.disp) out.disp[r] = disp;
        if (out.acc) out.acc[r] = acc;
        if (out.depth) out.depth[r] = depth;
        if (out.target_depth) out.target_depth[r] = tdepth;
        if (a.stage != nullptr) {
            float* sg = a.stage + r * STAGE_FLOATS;
            sg[0] = nrm_before[0]; sg[1] = nrm_before[1]; sg[2] = nrm_before[2];
            sg[3] = ndv; sg[4] = rough; sg[5] = e0; sg[6] = e1; sg[7] = level;
        }
    }
}

// Pass B: reflected-ray composite (raw2outputs_simple, :38-68), mip interpolation between the
// prefiltered radiances (:455-470), diffuse / specular / colour (:472-474), output mapping.
template <int NPL>
__global__ __launch_bounds__(256) void k_pass_b(PassBArgs a) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const int S = a.Sc;
    const float d[3] = {a.refl_d[3 * r], a.refl_d[3 * r + 1], a.refl_d[3 * r + 2]};
    const float norm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    float z[NPL], zn[NPL], sig[NPL], w[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        const float* zrow = a.zc + (long)a.zc_stride * r;      // z_vals_constant: one shared row, or per-ray rows under perturb
        z[i] = s < S ? zrow[s] : 0.0f;
        zn[i] = s + 1 < S ? zrow[s + 1] : 0.0f;
        sig[i] = s < S ? a.refl_raw[((long)r * S + s) * REFL_CH] : 0.0f;
    }
    ray_weights<NPL>(sig, z, zn, norm, S, lane, w);
    float maps[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) maps[c] = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        if (s < S && w[i] != 0.0f) {      // (a weightless sample adds +0 whatever its row holds: not read)
            const float* row = a.refl_raw + ((long)r * S + s) * REFL_CH;
#pragma unroll
            for (int c = 0; c < 12; ++c) maps[c] += w[i] * radiance_f(row[1 + c], a.radiance_linear);
        }
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) maps[c] = wave_sum(maps[c]);
    if (lane != 0) return;
    if (a.env_tap != nullptr) {
#pragma unroll
        for (int c = 0; c < 12; ++c) a.env_tap[12 * r + c] = maps[c];
    }

    const float* st = a.state + r * ST_FLOATS;
    const float albedo[3] = {st[0], st[1], st[2]};
    const float rough = st[3], level = st[11];
    const float irr[3] = {st[4], st[12], st[13]};
    const float metal = 1.0f - rough;
    int i1 = (int)(level * 3.0f);                       // .long() truncation (:464)
    i1 = i1 < 0 ? 0 : (i1 > 3 ? 3 : i1);
    const int i2 = i1 + 1 > 3 ? 3 : i1 + 1;
    const float rem = level * 3.0f - (float)i1;
    float pref[3], diffuse[3], specular[3], color[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pref[c] = (1.0f - rem) * maps[3 * i1 + c] + rem * maps[3 * i2 + c];
        diffuse[c] = (1.0f - st[5 + c]) * (1.0f - metal) * albedo[c] * irr[c];
        specular[c] = st[8 + c] * pref[c];
        color[c] = diffuse[c] + specular[c];
    }
    const int g = (a.gamma_correct ? 1 : 0) | (a.radiance_linear ? 2 : 0);
    const PassOutputs& out = a.out;
    store3(out.color, r, color, g);
    const float m0[3] = {maps[0], maps[1], maps[2]};
    store3(out.reflected_radiance, r, m0, g);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float mk[3] = {maps[3 + 3 * k], maps[4 + 3 * k], maps[5 + 3 * k]};
        store3(out.refl_coarse_k[k], r, mk, g);
    }
    store3(out.prefiltered, r, pref, g);
    store3(out.specular, r, specular, g);
    store3(out.diffuse, r, diffuse, g);
}

template <int NPL>
__global__ __launch_bounds__(256) void k_sigma_weights(const float* __restrict__ rays_d, const float* __restrict__ zbase,
                                                      int z_stride, const float* __restrict__ sigma, const float* __restrict__ noise, long R, int S,
                                                      float* __restrict__ weights, float* __restrict__ depth_out, float* __restrict__ vis_out) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const float norm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float* zrow = zbase + (long)z_stride * r;
    float z[NPL], zn[NPL], sig[NPL], w[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        z[i] = s < S ? zrow[s] : 0.0f;
        zn[i] = s + 1 < S ? zrow[s + 1] : 0.0f;
        sig[i] = s < S ? sigma[r * S + s] : 0.0f;
        if (noise != nullptr && s < S) sig[i] = sig[i] + noise[r * S + s];
    }
    float vis = 0.0f;
    ray_weights<NPL>(sig, z, zn, norm, S, lane, w, &vis);
    if (depth_out != nullptr) {       // raw2outputs_depth (ibl_nerf_renderer.py:118-152): depth_map and visibility beside the weights
        float dp = 0.0f;
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            if (lane * NPL + i < S) dp += w[i] * z[i];
        dp = wave_sum(dp);
        if (lane == 0) { depth_out[r] = dp; if (vis_out != nullptr) vis_out[r] = vis; }
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        if (s < S) weights[r * S + s] = w[i];
    }
}

// "Precision where it matters", per POINT (round 4).  A density estimate's last bits can reach a ray's weights only through samples that are neither
//   (a) clearly empty:   sigma_s + noise_s <= -margin  ->  relu(.) = 0 and alpha_s = 0 EXACTLY, whatever the estimate's error below the margin, nor
//   (b) behind saturation: T_s <= t_min  ->  w_s = alpha_s T_s <= t_min, and so is the sum of every weight from s on.
// Everything else is relevant: its point and its flat index r * S + s go to a compact list (order across rays: whatever the atomics give — a point's
// result does not depend on its place in a batch), which the precise query then evaluates and scatters over the estimate.  One wavefront per ray.
// OFFSETS: the "rays" are the 4 R epsilon-offset copies of the samples (normal_from_depth.py:143-160: virtual ray v R + r composites sig4[v][r][:] on ray r's own z
// and dists, no noise); their points come from gen_offset_point, the generator the TRUNK kernels' input stage uses.
constexpr int SELECT_WAVES = 8;      // rays per block of k_select_points
constexpr int AUDIT_LOG2 = 6;        // one in 64 of the samples k_select_points drops as clearly empty is refined all the same (the tripwire's audit)

// the call's running totals behind a list launch (iblnerf_last_selection / iblnerf_last_executed_flops): [2..3] one uint64 of list entries, [4..5] one double of the
// MACs x 2 the list launches evaluate on them
__global__ void k_count_selection(int* counter, double flop_per_point, double slots_per_point, int count_entries) {
    const int n = counter[0];
    if (count_entries) *reinterpret_cast<unsigned long long*>(counter + 2) += (unsigned long long)n;
    *reinterpret_cast<double*>(counter + 4) += (double)n * flop_per_point;
    *reinterpret_cast<double*>(counter + 8) += (double)n * slots_per_point;      // [8..9]: matrix-slot units (api.cpp launch_slots)
}

// Estimates in two z-chunks (api.cpp estimate_chunked): the points and flat indices r S + s of samples [s0, s1) of every (virtual) ray — FIRST: of all rays, at
// position vr (s1 - s0) + (s - s0), no counter — or, second chunk, of the rays whose transmittance behind their first s0 samples (composited conservatively from the
// estimates already there, as in k_select_points) is still above t_min; the other rays' samples [s0, s1) get the density -1e30: behind saturation, never relevant.
template <int NPL, bool OFFSETS, bool FIRST>
__global__ __launch_bounds__(64 * SELECT_WAVES) void k_chunk_points(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ zbase, int z_stride,
                                                                    float* __restrict__ sigma, const float* __restrict__ noise, long R, int S, int s0, int s1, float margin,
                                                                    float t_min, float eps, float* __restrict__ pts_out, int* __restrict__ index_out, int* __restrict__ counter) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long vr_raw = (long)blockIdx.x * SELECT_WAVES + wave;
    const bool live = vr_raw < (OFFSETS ? 4 * R : R);
    const long vr = live ? vr_raw : 0;
    const long r = OFFSETS ? vr % R : vr;
    const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const float* zrow = zbase + (long)z_stride * r;
    bool alive = live;
    if constexpr (!FIRST) {
        const float norm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        double lane_prod = 1.0;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int s = lane * NPL + i;
            if (s < s0) {
                float sg = sigma[vr * S + s];
                if (noise != nullptr) sg = sg + noise[r * S + s];
                const float dist = (zrow[s + 1] - zrow[s]) * norm;                       // (s + 1 <= s0 < S)
                const float a = 1.0f - expf(-fmaxf(sg * 0.75f - margin, 0.0f) * dist);    // the conservative transmittance of k_select_points
                lane_prod *= (double)((1.0f - a) + 1e-10f);
            }
        }
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) lane_prod *= __shfl_xor(lane_prod, dd);
        alive = live && lane_prod > (double)t_min;
    }
    int base = 0;
    const int per_ray = s1 - s0;
    if constexpr (FIRST) {
        base = (int)(vr * per_ray);
    } else {
        __shared__ int wave_total[SELECT_WAVES];
        __shared__ int block_base;
        if (lane == 0) wave_total[wave] = alive ? per_ray : 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            int sum = 0;
#pragma unroll
            for (int w = 0; w < SELECT_WAVES; ++w) sum += wave_total[w];
            block_base = sum > 0 ? atomicAdd(counter, sum) : 0;
        }
        __syncthreads();
        base = block_base;
        for (int w = 0; w < wave; ++w) base += wave_total[w];
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        if (!live || s < s0 || s >= s1) continue;
        const unsigned flat = (unsigned)(vr * S + s);
        if (!alive) {
            sigma[flat] = -1e30f;
            continue;
        }
        float p[3];
        if constexpr (OFFSETS) {
            PointGen g;
            g.rays_o = rays_o; g.rays_d = rays_d; g.z = zbase; g.z_stride = z_stride; g.S = S; g.RS = (unsigned)(R * S); g.eps = eps;
            gen_offset_point(g, flat, p[0], p[1], p[2]);
        } else {
            const float zz = zrow[s];
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = o[c] + d[c] * zz;      // (k_make_points mode 0; this file is compiled without contraction)
        }
        const long pos = (long)base + (s - s0);
        pts_out[3 * pos] = p[0];
        pts_out[3 * pos + 1] = p[1];
        pts_out[3 * pos + 2] = p[2];
        index_out[pos] = (int)flat;
    }
}

// The samples of a ray that its own selection found relevant, one more on either side: [lo, hi] — what the ray's four epsilon-offset copies (0.01 beside it) are
// PREDICTED relevant by.  main_range: k_select_points' range_out, {first, last} selected sample per ray ({S, -1}: none -> lo = S, hi = S - 1: an empty range behind the ray).
__device__ __forceinline__ void predicted_range(const int* __restrict__ main_range, long r, int S, int& lo, int& hi) {
    const int a = main_range[2 * r], b = main_range[2 * r + 1];
    if (b < a) { lo = S; hi = S - 1; }
    else { lo = max(a - 1, 0); hi = min(b + 1, S - 1); }
}

// Round 5: the offset copies without an estimate of every sample (api.cpp offsets_on_lists).  Emits, per virtual ray v R + r, the points + flat indices of
//   mode 1  its predicted range [lo, hi]           -> straight to the query's own kernel, no estimate;
//   mode 2  the samples in front of it, [0, lo)    -> density estimates;
//   mode 3  the samples behind it, (hi, S)         -> density estimates, for the copies whose transmittance behind [0, hi] — composited conservatively, as k_select_points
//                                                     does, from the front estimates and the refined densities already in `sigma` — is still above t_min; the other
//                                                     copies' samples behind get the density -1e30 (behind saturation, never relevant), exactly as k_chunk_points does.
// Every sample of every copy lands in exactly one of the three; the prediction decides what a sample costs, never what it yields.
// tier_mask / tier (mode 1 only; api.cpp offset tiers): the predicted range split by k_importance's per-sample flags of the MAIN ray — tier 1 emits the flagged samples
// (where a density error would move the copy's depth most: they go to the precise kernel), tier 0 the others; null = the whole range.
template <int NPL>
__global__ __launch_bounds__(64 * SELECT_WAVES) void k_range_points(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ zbase, int z_stride,
                                                                    float* __restrict__ sigma, const int* __restrict__ main_range, long R, int S, int mode, float margin,
                                                                    float t_min, float eps, float* __restrict__ pts_out, int* __restrict__ index_out, int* __restrict__ counter,
                                                                    const unsigned long long* __restrict__ tier_mask, int tier) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long vr_raw = (long)blockIdx.x * SELECT_WAVES + wave;
    const bool live = vr_raw < 4 * R;
    const long vr = live ? vr_raw : 0;
    const long r = vr % R;
    int lo, hi;
    predicted_range(main_range, r, S, lo, hi);
    const int s0 = mode == 1 ? lo : mode == 2 ? 0 : hi + 1;
    const int s1 = mode == 1 ? hi + 1 : mode == 2 ? lo : S;
    bool alive = live && s1 > s0;
    if (mode == 3 && alive) {        // (wave-uniform)
        const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        const float norm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        const float* zrow = zbase + (long)z_stride * r;
        double lane_prod = 1.0;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int s = lane * NPL + i;
            if (s < s0) {
                const float sg = sigma[vr * S + s];
                const float dist = (zrow[s + 1] - zrow[s]) * norm;                       // (s + 1 <= s0 < S)
                const float a = 1.0f - expf(-fmaxf(sg * 0.75f - margin, 0.0f) * dist);    // the conservative transmittance of k_select_points
                lane_prod *= (double)((1.0f - a) + 1e-10f);
            }
        }
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) lane_prod *= __shfl_xor(lane_prod, dd);
        alive = lane_prod > (double)t_min;
    }
    bool emit[NPL];
    unsigned long long masks[NPL];
    int total = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        bool e = alive && s >= s0 && s < s1;
        if (tier_mask != nullptr) e = e && (int)((tier_mask[4 * r + i] >> lane) & 1ull) == tier;
        emit[i] = e;
        masks[i] = __ballot(e);
        total += __popcll(masks[i]);
    }
    __shared__ int wave_total[SELECT_WAVES];
    __shared__ int block_base;
    if (lane == 0) wave_total[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
#pragma unroll
        for (int w = 0; w < SELECT_WAVES; ++w) sum += wave_total[w];
        block_base = sum > 0 ? atomicAdd(counter, sum) : 0;
    }
    __syncthreads();
    int base = block_base;
    for (int w = 0; w < wave; ++w) base += wave_total[w];
    PointGen g;
    g.rays_o = rays_o; g.rays_d = rays_d; g.z = zbase; g.z_stride = z_stride; g.S = S; g.RS = (unsigned)(R * S); g.eps = eps;
    int before = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        const unsigned flat = (unsigned)(vr * S + s);
        if (live && !alive && s >= s0 && s < s1) sigma[flat] = -1e30f;          // (mode 3, a copy that has saturated: behind saturation, never relevant)
        if (emit[i]) {
            float p[3];
            gen_offset_point(g, flat, p[0], p[1], p[2]);
            const long pos = (long)base + before + __popcll(masks[i] & ((1ull << lane) - 1ull));
            pts_out[3 * pos] = p[0];
            pts_out[3 * pos + 1] = p[1];
            pts_out[3 * pos + 2] = p[2];
            index_out[pos] = (int)flat;
        }
        before += __popcll(masks[i]);
    }
}

// Where would an error on an offset copy's density move its depth most?  d depth / d sigma_s = T_s dist_s exp(-sigma_s dist_s) (z_s - depth behind s): per sample of
// the MAIN ray (0.01 beside the copy: the same transmittance profile to first order) the bound T_s dist_s |depth - z_s|, from the main query's own densities.  Samples above
// `tau` are flagged (bit `lane` of mask[4 r + i] <-> sample lane NPL + i): the fine grid's offset copies evaluate them on three f16 products, the others on the mixed
// trunk form, whose 1.5e-3 in raw density then moves a copy's depth by < tau x 1.5e-3 x (unflagged samples) — with tau = 5e-6 and 192 samples below 1.5e-6, i.e. 7e-5 on
// the normal.  (The rays this is for: a soft haze of small positive density in front of the surface — every sample of it sees the whole depth behind it.)
// mode 1 (round 6, the fine MAIN query's tiers): where would an error on the main query's OWN density move its per-sample weights most?  d w_s / d sigma_s = T_s dist_s
// exp(-sigma_s dist_s), and every weight behind s moves by -w_j dist_s d sigma_s: with a relative error eps on sigma_s, at most eps T_s x exp(-x), x = sigma_s dist_s — of the
// order of the sample's own weight.  Samples whose own weight alpha_s T_s (from `sigma`: here the query's density ESTIMATES, before the list is refined) exceeds `tau` are
// flagged: the main query evaluates them on three f16 products, the others — thousands of them per ray-weight above tau, each moving a weight by < eps tau — on the fast form.
template <int NPL>
__global__ __launch_bounds__(256) void k_importance(const float* __restrict__ rays_d, const float* __restrict__ zbase, int z_stride, const float* __restrict__ sigma,
                                                    int sigma_stride, long R, int S, float tau, unsigned long long* __restrict__ mask, int mode, float margin,
                                                    const unsigned long long* __restrict__ exclude) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const float norm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float* zrow = zbase + (long)z_stride * r;
    float z[NPL], dist[NPL], alpha[NPL];
    double om[NPL], lane_prod = 1.0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        z[i] = s < S ? zrow[s] : 0.0f;
        const float zn = s + 1 < S ? zrow[s + 1] : 0.0f;
        const float sg = s < S ? sigma[(r * S + s) * (long)sigma_stride] : 0.0f;
        dist[i] = (s == S - 1 ? 1e10f : (zn - z[i])) * norm;
        alpha[i] = s < S ? 1.0f - expf(-fmaxf(sg, 0.0f) * dist[i]) : 0.0f;
        // mode 4 (the fine main query's tiers, decided on density ESTIMATES): the transmittance a sample's weight is judged by is k_select_points' CONSERVATIVE one — the
        // densities in front taken at 3/4 of their estimate less the margin — so that an estimate that overshoots what lies in front (11 % of a large density, a unit of a
        // small one) cannot hide a sample that carries weight (measured on the second checkpoint: with the plain transmittance 0.1 % of the rays kept a heavy sample on the
        // fast form and the per-sample weights stayed at 6e-4 of SAFE's whatever the threshold)
        const float a_t = (mode == 4 && s < S) ? 1.0f - expf(-fmaxf(sg * 0.75f - margin, 0.0f) * dist[i]) : alpha[i];
        om[i] = s < S ? (double)((1.0f - a_t) + 1e-10f) : 1.0;
        lane_prod *= om[i];
    }
    double incl = lane_prod;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const double o = __shfl_up(incl, dd);
        if (lane >= dd) incl *= o;
    }
    double T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.0;
    float Ts[NPL], depth = 0.0f, acc = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        Ts[i] = (float)T;
        depth += alpha[i] * Ts[i] * z[i];
        acc += alpha[i] * Ts[i];
        T *= om[i];
    }
    depth = wave_sum(depth);
    acc = wave_sum(acc);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        // mode 1: own weight above tau.  mode 3 (the fine main query's fix-up, from REFINED densities): a HAZY ray — the fast form's density carries an ABSOLUTE error of
        // ~5e-3 (raw density is a small difference of large terms), i.e. ~T_s dist_s 5e-3 on a weight whatever the sample's own: nothing on a ray that ends on a surface
        // (acc ~ 1), 14 % on a ray whose whole weight is 1.5e-3 (measured: one sample of sigma 0.034; the output lambdas' pow(x, 1 / 2.2) then shows it as 3e-3 of the
        // albedo map) — every visible sample of a ray with T_s dist_s > acc is flagged.  mode 2: both.
        const bool heavy = Ts[i] * alpha[i] > tau, hazy = Ts[i] * fminf(dist[i], 1.0f) > acc && Ts[i] > 1e-4f;
        const bool flag = s < S && (mode == 0 ? Ts[i] * fminf(dist[i], 1.0f) * fabsf(depth - z[i]) > tau : (mode == 1 || mode == 4) ? heavy : mode == 3 ? hazy : (heavy || hazy));
        const unsigned long long m = __ballot(flag);
        if (lane == 0) mask[4 * r + i] = exclude != nullptr ? m & ~exclude[4 * r + i] : m;      // (exclude: the samples an earlier mask flagged already)
    }
}

// The estimate TRIPWIRE (round 5): after a list launch has written the refined densities, every entry's refined value against the estimate that put it on the list
// (k_select_points' est_list: the selected samples and the audited ones).  A positive density whose estimate lay below -margin / 2 — half-way to being dropped as
// clearly empty — raises bit 2 of the range flag; one overshot beyond what the conservative transmittance allows for (0.75 estimate - margin > refined) bit 3:
// k_compare_estimates' rule, on every launch instead of once per checkpoint.  (A kernel of its own, 0.05 ms: inside the MLP kernels' epilogues the same comparison
// cost the fast FULL list form its last registers — 20 bytes of scratch, 5.9 -> 9.2 ms per launch.)
// Round 6: the event is recorded PER RAY — trip_rays[(flat index / S) % R] = 1 (nullable; [R] bytes of the launch's rays: a sample of virtual ray v R + r, an offset
// copy, belongs to ray r) — so that the caller can render exactly the rays whose estimates were thin once more with every sample evaluated, instead of the
// whole call under wider margins: a ray's result is then a function of the ray alone, whatever call, launch or rank it is rendered in.
__global__ void k_tripwire(const float* __restrict__ est_list, const int* __restrict__ index, const int* __restrict__ n_dev, const float* __restrict__ out, int out_stride,
                           float margin, unsigned* __restrict__ flag, unsigned char* __restrict__ trip_rays, int S, long R) {
    const long n = *n_dev;
    unsigned bits = 0u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float refined = out[(long)index[i] * out_stride], est = est_list[i];
        // bit 2: a positive density whose estimate lay below -margin / 2 — a near miss: the margin is thin on this ray.  bit 4 (with bit 2), round 6: ... below -3/4 of the
        // margin — DEEP: the margin was set to twice the deepest underestimate the probe saw (+ 0.5), so this estimate is off by 1.5 times what the probe measured, and beyond
        // -margin itself (only an AUDITED entry can be: the sample had been dropped as clearly empty and was not) it is proof that samples are being dropped wrongly.  Measured
        // on a network built to break plain-f16 estimates under a route that trusts them (tests/test_gpu_scope.py): 25 near misses of 8 192 rays, and 8 OTHER rays wrong by up
        // to 0.8 of a weight with no mark at all — near misses are the visible part of an error tail; deep ones say the tail reaches the margin.
        // bit 3: an estimate overshot beyond what the conservative transmittance allows for — also where the refined density is NOT positive (round 6): an empty sample whose
        // estimate says "opaque" costs that sample nothing (it is refined), but the transmittance behind it was composited from the estimate.
        const unsigned b = refined > 0.0f && est <= -0.75f * margin ? 20u : refined > 0.0f && est < -0.5f * margin ? 4u : (0.75f * est - margin > fmaxf(refined, 0.0f) ? 8u : 0u);
        if (b != 0u && trip_rays != nullptr) trip_rays[((long)index[i] / S) % R] = 1;
        bits |= b;
    }
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) bits |= (unsigned)__shfl_xor((int)bits, dd);
    if ((threadIdx.x & 63) == 0 && bits != 0u) atomicOr(flag, bits);
}

// Is a plain-f16 density estimate good enough for k_select_points on this network?  Two estimates of the same n samples (b: the f16 + 2 fp6 form, error < 1e-2);
// counts the samples on which `a` is half-way to a wrong decision: a positive density estimated below -margin / 2, or a density overshot by more than the
// conservative transmittance allows for (0.75 a - margin > b).
// Round 5: the first of the two (a positive density estimated below -margin / 2) became a MEASUREMENT — bad[1] receives the bits of the deepest UNDERESTIMATE: the largest
// -a among the samples whose density b is positive (and within `zone`: a large density underestimated by a few per cent is no classification question).  That is how far
// below zero this network's plain-f16 estimate puts a sample that is NOT empty; api.cpp check_estimates sets the network's selection margin from it.  (An OVERestimate of an
// empty sample only costs its refinement.)
__global__ void k_compare_estimates(const float* __restrict__ a, const float* __restrict__ b, long n, float margin, float zone, int* __restrict__ bad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool wrong = i < n && 0.75f * a[i] - margin > fmaxf(b[i], 0.0f);        // (round 6: also an EMPTY sample estimated opaque — what lies behind it would be dropped as saturated)
    const unsigned long long m = __ballot(wrong);
    if ((threadIdx.x & 63) == 0 && m != 0ull) atomicAdd(bad, __popcll(m));
    float err = (i < n && b[i] > 0.0f && b[i] <= zone) ? fmaxf(-a[i], 0.0f) : 0.0f;
    if (i < n && b[i] > 0.0f && b[i] <= zone && a[i] != a[i]) err = 1e30f;
    if (!(err < 1e30f)) err = 1e30f;                      // (a NaN / inf estimate: refuse)
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) err = fmaxf(err, __shfl_xor(err, dd));
    if ((threadIdx.x & 63) == 0 && err > 0.0f) atomicMax(reinterpret_cast<unsigned*>(bad + 1), __builtin_bit_cast(unsigned, err));      // (non-negative floats order like their bits)
}

template <int NPL, bool OFFSETS>
__global__ __launch_bounds__(64 * SELECT_WAVES) void k_select_points(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ zbase, int z_stride,
                                                      const float* __restrict__ sigma, int sigma_stride, const float* __restrict__ noise, long R, int S, float margin,
                                                      float t_min, float eps, float* __restrict__ pts_out, int* __restrict__ index_out, int* __restrict__ counter,
                                                      float* __restrict__ est_out, int est_stride, int* __restrict__ range_out, const int* __restrict__ skip_range,
                                                      float* __restrict__ est_list, const unsigned long long* __restrict__ tier_mask, int tier) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long vr_raw = (long)blockIdx.x * SELECT_WAVES + wave;      // (virtual) ray
    const bool live = vr_raw < (OFFSETS ? 4 * R : R);                // (a dead wave of the last block still takes part in the block's count)
    const long vr = live ? vr_raw : 0;
    const long r = OFFSETS ? vr % R : vr;
    const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const float norm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float* zrow = zbase + (long)z_stride * r;
    float z[NPL], sg[NPL], sg_est[NPL];
    double om[NPL];
    double lane_prod = 1.0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        z[i] = s < S ? zrow[s] : 0.0f;
        const float zn = s + 1 < S ? zrow[s + 1] : 0.0f;
        sg[i] = s < S ? sigma[(vr * S + s) * (long)sigma_stride] : -1e30f;
        sg_est[i] = sg[i];                                            // (the network's own estimate, before any density noise: what the tripwire compares)
        if (est_out != nullptr && s < S && live) est_out[(vr * S + s) * (long)est_stride] = sg[i];     // the estimate itself, as the density of the samples nobody refines
        if (noise != nullptr && s < S) sg[i] = sg[i] + noise[r * S + s];
        const float dist = (s == S - 1 ? 1e10f : (zn - z[i])) * norm;
        // the transmittance a sample is judged by is a CONSERVATIVE one: densities in front of it taken at 3/4 of their estimate less the margin, so that an
        // estimate that overshoots a large density (a plain-f16 trunk: up to 11 % measured) cannot declare what lies behind it saturated too early
        const float a = s < S ? 1.0f - expf(-fmaxf(sg[i] * 0.75f - margin, 0.0f) * dist) : 0.0f;
        om[i] = s < S ? (double)((1.0f - a) + 1e-10f) : 1.0;
        lane_prod *= om[i];
    }
    double incl = lane_prod;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const double other = __shfl_up(incl, dd);
        if (lane >= dd) incl *= other;
    }
    double T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.0;
    bool sel[NPL];
    int total = 0;
    unsigned long long masks[NPL];
    // (the audit's choice of samples, below: a hash of the RAY — direction and origin bits, which offset copy — and of the sample's depth: the same samples whatever launch,
    // call or rank the ray is rendered in; round 5 hashed the flat index within the launch)
    const unsigned audit_key = (__builtin_bit_cast(unsigned, d[0]) * 0x9E3779B1u) ^ (__builtin_bit_cast(unsigned, d[1]) * 0x85EBCA77u) ^ (__builtin_bit_cast(unsigned, d[2]) * 0xC2B2AE3Du) ^
                               (__builtin_bit_cast(unsigned, o[0]) + 0x27D4EB2Fu * __builtin_bit_cast(unsigned, o[1])) ^ (__builtin_bit_cast(unsigned, o[2]) * 0x165667B1u) ^
                               (OFFSETS ? (unsigned)(vr / R) * 0x9E3779B9u : 0u);
    // skip_range (offset copies, api.cpp offsets_on_lists): the samples [lo, hi] of the ray were predicted relevant and hold their refined density already — they
    // count for the transmittance of what lies behind them, and are not selected again
    int skip_lo = S, skip_hi = -1;
    if (skip_range != nullptr) predicted_range(skip_range, r, S, skip_lo, skip_hi);
    int first_sel = S, last_sel = -1;
    bool first_sel_candidate[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        // (a sample is judged by the transmittance the ESTIMATE gives in front of it, with a margin on the estimate itself: sigma > -margin counts as
        // possibly opaque for nobody else's T, because T only ever gets smaller by counting it)
        const bool reachable = live && s < S && T > (double)t_min && !(s >= skip_lo && s <= skip_hi);
        sel[i] = reachable && sg[i] > -margin;
        // the AUDIT (round 5): one in AUDIT_ONE_IN of the samples dropped as clearly empty goes to the list all the same.  Its refined density replaces the estimate
        // (both <= 0: alpha = 0 either way, no map changes) — and the tripwire behind the list launch (k_tripwire) sees an estimate that was GROSSLY wrong, which no
        // selected sample would show: a positive density estimated below -margin is never selected, so never refined, so never compared.  Chosen by a hash of the
        // sample itself (audit_key above): the same samples on every route, launch and rank.
        const bool audit = reachable && !sel[i] && (audit_key ^ (__builtin_bit_cast(unsigned, z[i]) * 2246822519u)) * 2654435761u >> (32 - AUDIT_LOG2) == 0u;
        first_sel_candidate[i] = sel[i];
        // tier_mask / tier (round 6, the fine main query in two tiers: api.cpp run_main_query): this pass emits only the selected samples whose k_importance flag equals
        // `tier`; the ray's range (range_out) is that of both tiers together, the audited samples ride with tier 0
        if (tier_mask != nullptr) {
            const bool flagged = ((tier_mask[4 * r + i] >> lane) & 1ull) != 0ull;
            sel[i] = sel[i] && (flagged == (tier == 1));
            sel[i] = sel[i] || (audit && tier == 0);
        } else
        sel[i] = sel[i] || audit;
        T *= om[i];
        masks[i] = __ballot(sel[i]);
        total += __popcll(masks[i]);
        if (first_sel_candidate[i]) { first_sel = min(first_sel, s); last_sel = max(last_sel, s); }      // (an audited sample does not stretch the range its copies are predicted by)
    }
    if (range_out != nullptr) {       // the ray's first and last selected sample ({S, -1}: none): what its offset copies are predicted relevant by
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            first_sel = min(first_sel, __shfl_xor(first_sel, dd));
            last_sel = max(last_sel, __shfl_xor(last_sel, dd));
        }
        if (live && lane == 0) { range_out[2 * vr] = first_sel; range_out[2 * vr + 1] = last_sel; }
    }
    // one atomic per BLOCK: every wave of a launch adding to the one counter by itself serialises in L2 (256 000 waves of the coarse grid's offset copies: 5.6 ms)
    __shared__ int wave_total[SELECT_WAVES];
    __shared__ int block_base;
    if (lane == 0) wave_total[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
#pragma unroll
        for (int w = 0; w < SELECT_WAVES; ++w) sum += wave_total[w];
        block_base = sum > 0 ? atomicAdd(counter, sum) : 0;
    }
    __syncthreads();
    int base = block_base;
    for (int w = 0; w < wave; ++w) base += wave_total[w];
    int before = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        if (sel[i]) {
            const int pos = base + before + __popcll(masks[i] & ((1ull << lane) - 1ull));
            float p[3];
            const unsigned flat = (unsigned)(vr * S + lane * NPL + i);
            if constexpr (OFFSETS) {
                PointGen g;
                g.rays_o = rays_o; g.rays_d = rays_d; g.z = zbase; g.z_stride = z_stride; g.S = S; g.RS = (unsigned)(R * S); g.eps = eps;
                gen_offset_point(g, flat, p[0], p[1], p[2]);
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) p[c] = o[c] + d[c] * z[i];      // the arithmetic of k_make_points mode 0 (this file is compiled without contraction)
            }
            pts_out[3 * (long)pos] = p[0];
            pts_out[3 * (long)pos + 1] = p[1];
            pts_out[3 * (long)pos + 2] = p[2];
            index_out[pos] = (int)flat;
            if (est_list != nullptr) est_list[pos] = sg_est[i];      // the estimate that selected (or audited) this entry: k_tripwire compares the refined density with it
        }
        before += __popcll(masks[i]);
    }
}

template <int NPL>
__global__ __launch_bounds__(256) void k_surface_points(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ zbase, int z_stride, const float* __restrict__ raw,
                                                       const float* __restrict__ noise, long R, int S, OverrideArgs ov, float* __restrict__ surf) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const float norm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float* zrow = zbase + (long)z_stride * r;
    float z[NPL], zn[NPL], sig[NPL], w[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        z[i] = s < S ? zrow[s] : 0.0f;
        zn[i] = s + 1 < S ? zrow[s + 1] : 0.0f;
        sig[i] = s < S ? raw[(r * S + s) * RAW_CH] : 0.0f;
        if (noise != nullptr && s < S) sig[i] = sig[i] + noise[r * S + s];
    }
    ray_weights<NPL>(sig, z, zn, norm, S, lane, w);
    float depth = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i)
        if (lane * NPL + i < S) depth += w[i] * z[i];
    depth = wave_sum(depth);
    const bool mask_all = ov.mode != 0 && ov.mask[r * ov.mask_stride] > 0.0f;
    const float tdepth = target_depth(ov, r, mask_all, depth);
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < 3; ++c) surf[3 * r + c] = rays_o[3 * r + c] + d[c] * tdepth;   // :262
}

// torch.sum(weights + 1e-5, -1) of one row of n floats, BIT FOR BIT as ATen's CPU kernel forms it (cascade sum of cpu/SumKernel.cpp with
// 8-float vectors: vectorized_inner_sum -> row_sum (4 interleaved partial rows; the cascade levels never fill below 512 elements) -> the
// row's tail, then the 8 vector lanes, sequentially).  Why the order matters: sample_pdf replaces denominators below 1e-5 by 1
// (nerf_renderer_helper.py:128-129), and an EMPTY bin's denominator is 1e-5 / sum = 9.994e-6 +- one fp32 ulp of the cdf (6e-8): 167 or 168
// ulps, i.e. on either side of the threshold, depending on the last bit of `sum`.  A sample then lands at the bin's edge or in its interior
// (0.04 apart on the coarse grid) — invisible unless the "empty" bin hides a thin structure (1 ray in 25 000 of the fitted checkpoint:
// depth off by 3e-3).  The emulated order is pinned against torch.sum itself in tests/test_oracle_golden.py.
// Returns the sum on every lane.  n < 512.
__device__ __forceinline__ float aten_row_sum_eps(const float* __restrict__ wts, int n, int lane) {
    const int vs = n >> 3, size_ilp = vs >> 2;
    const int c = lane & 7, k = (lane >> 3) & 3;
    float p = 0.0f;
    if (lane < 32) {
        for (int i = 0; i < size_ilp; ++i) p += wts[8 * (4 * i + k) + c] + 1e-5f;          // partial row k of vector lane c
        if (k == 0)
            for (int j = 4 * size_ilp; j < vs; ++j) p += wts[8 * j + c] + 1e-5f;           // the vectors beyond the last group of four
    }
    float P = p;                                                                            // ((p0 + p1) + p2) + p3, valid on lanes 0..7
    P += __shfl(p, c + 8);
    P += __shfl(p, c + 16);
    P += __shfl(p, c + 24);
    float fin = 0.0f;
    for (int t = 8 * vs; t < n; ++t) fin += wts[t] + 1e-5f;                                 // the row's tail, then the vector lanes in order
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) fin += __shfl(P, cc);
    return fin;
}

// sample_pdf(det=True), nerf_renderer_helper.py:91-134, for one ray held by one wavefront.
// cdf/bins live in LDS (nb <= 257).  Writes n_out samples to `dst` (LDS or global).
// u_row: this ray's n_out uniform draws (det=False, :103), or null for u = linspace(0, 1, n_out) (det=True, :100-101)
template <class Dst>
__device__ __forceinline__ void sample_pdf_wave(const float* __restrict__ wts, int nb, int n_out, int lane,
                                                float* cdf /*LDS [nb]*/, const float* bins /*LDS [nb]*/, const float* __restrict__ u_row, Dst&& put) {
    const int nw = nb - 1;
    // lane owns weights lane*NPL .. (contiguous) so the cumulative sum is lane-local + wave scan
    const int npl = (nw + 63) / 64;
    float wl[MAX_NPL + 1];
#pragma unroll
    for (int i = 0; i < MAX_NPL + 1; ++i) {
        const int k = lane * npl + i;
        wl[i] = (i < npl && k < nw) ? wts[k] + 1e-5f : 0.0f;
    }
    const float tot = aten_row_sum_eps(wts, nw, lane);   // torch.sum's own summation order: the 1e-5 threshold below sits one ulp from an empty bin
    double run = 0.0, lane_tot = 0.0;
    float pdf[MAX_NPL + 1];
#pragma unroll
    for (int i = 0; i < MAX_NPL + 1; ++i) {
        const int k = lane * npl + i;
        pdf[i] = (i < npl && k < nw) ? wl[i] / tot : 0.0f;
        lane_tot += (double)pdf[i];
    }
    double incl = lane_tot;   // inclusive scan over lanes (double accumulate == ATen CPU cumsum)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    run = incl - lane_tot;
    if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
    for (int i = 0; i < MAX_NPL + 1; ++i) {
        const int k = lane * npl + i;
        if (i < npl && k < nw) {
            run += (double)pdf[i];
            cdf[k + 1] = (float)run;
        }
    }
    __builtin_amdgcn_wave_barrier();   // same-wave LDS writes -> reads: in-order LDS queue, compiler inserts the lgkmcnt
    for (int j = lane; j < n_out; j += 64) {
        const float u = u_row != nullptr ? u_row[j] : linspace_at(0.0f, 1.0f, n_out, j);
        int lo = 0, hi = nb;   // searchsorted(right=True): number of cdf entries <= u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (!(cdf[mid] > u)) lo = mid + 1; else hi = mid;   // ATen's upper bound: a NaN cdf (NaN density) sends the search right, the sample is NaN
        }
        const int below = lo - 1 < 0 ? 0 : lo - 1;
        const int above = lo > nb - 1 ? nb - 1 : lo;
        const float c0 = cdf[below], c1 = cdf[above], b0 = bins[below], b1 = bins[above];
        float den = c1 - c0;
        if (den < 1e-5f) den = 1.0f;
        const float t = (u - c0) / den;
        put(j, b0 + t * (b1 - b0));
    }
}

__global__ __launch_bounds__(256) void k_sample_pdf(const float* __restrict__ bins, int bins_stride,
                                                   const float* __restrict__ weights, int w_stride, long R, int nb,
                                                   int n_out, const float* __restrict__ u, float* __restrict__ samples) {
    __shared__ float lds[4][2][260];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * 4 + wv;
    if (r >= R) return;
    for (int k = lane; k < nb; k += 64) lds[wv][1][k] = bins[r * bins_stride + k];
    sample_pdf_wave(weights + r * w_stride, nb, n_out, lane, lds[wv][0], lds[wv][1], u ? u + r * n_out : nullptr,
                    [&](int j, float v) { samples[r * n_out + j] = v; });
}

// z_vals_mid -> sample_pdf(weights[1:-1]) -> sort(cat([z, z_samples])) -> z_std
// (ibl_nerf_renderer.py:701-707, :718).  Sc + n_imp <= 512.
__global__ __launch_bounds__(256) void k_fine_z(const float* __restrict__ zc_base, int zc_stride, int Sc, const float* __restrict__ wc,
                                               long R, int n_imp, const float* __restrict__ u, float* __restrict__ z_fine,
                                               float* __restrict__ z_std) {
    __shared__ float lds[4][2][260];
    __shared__ float vals[4][512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * 4 + wv;
    if (r >= R) return;
    const int nb = Sc - 1;
    const float* zc = zc_base + (long)zc_stride * r;
    for (int k = lane; k < nb; k += 64) lds[wv][1][k] = 0.5f * (zc[k + 1] + zc[k]);
    for (int k = lane; k < Sc; k += 64) vals[wv][k] = zc[k];
    sample_pdf_wave(wc + r * Sc + 1, nb, n_imp, lane, lds[wv][0], lds[wv][1], u ? u + r * n_imp : nullptr,
                    [&](int j, float v) { vals[wv][Sc + j] = v; });
    __builtin_amdgcn_wave_barrier();
    const int n = Sc + n_imp;
    // z_std = std(z_samples, unbiased=False)
    double s1 = 0.0;
    for (int j = lane; j < n_imp; j += 64) s1 += (double)vals[wv][Sc + j];
    const double mean = wave_sum_d(s1) / n_imp;
    double s2 = 0.0;
    for (int j = lane; j < n_imp; j += 64) {
        const double dv = (double)vals[wv][Sc + j] - mean;
        s2 += dv * dv;
    }
    s2 = wave_sum_d(s2);
    if (lane == 0 && z_std) z_std[r] = (float)sqrt(s2 / n_imp);
    // rank sort (values only; ties keep index order; NaNs order after everything, as torch.sort places them, so that a
    // poisoned ray gives a visibly NaN row instead of stale slots)
    for (int e = lane; e < n; e += 64) {
        const float v = vals[wv][e];
        const bool vn = v != v;
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float u = vals[wv][j];
            const bool un = u != u;
            const bool before = (un || vn) ? ((!un && vn) || (un && vn && j < e)) : (u < v || (u == v && j < e));
            rank += before ? 1 : 0;
        }
        z_fine[r * n + rank] = v;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The DIRECT maps of raw2outputs on their own, forward and backward (training: what torch autograd does between the raw rows and the
// per-ray maps the losses read; ibl_nerf_renderer.py:203-206, 241-259, 281-318).  maps[r] = [depth, acc, albedo(3), roughness, irradiance,
// radiance(3), radiance_1..3 (9)] = sum_s w_s * act_c(raw[s][c]) (depth: z_s; acc: 1).  Backward: with g_s = dL/dw_s — collected from the maps
// the reference composites with `weights` (depth, acc, radiance_map: :249, :259, :306) and, optionally, from a loss on the weights themselves;
// albedo, roughness, irradiance and the three coarse radiances are composited with `weights_detached` (:246, :282-315) and send no gradient
// to the density (detach = 1; detach = 0 differentiates every map through the weights) —,
//   dL/d alpha_s = g_s T_s - (sum_{i>s} g_i w_i) / (1 - alpha_s + 1e-10),    dL/d raw_s0 = that * dist_s exp(-raw_s0 dist_s) [raw_s0 > 0],
//   dL/d raw_sc  = w_s dL/dmap_c act_c'(raw_sc)                         — the depth-gradient scan of ray_depth_grad with g in place of z.
constexpr int CM_CH = 19;
__device__ __forceinline__ int cm_map_of_channel(int c) { return c < 3 ? 2 + c : c == 3 ? 5 : c == 4 ? 6 : 7 + (c - 5); }   // raw channel 1 + c -> map slot

template <int NPL>
__global__ __launch_bounds__(256) void k_composite_fwd(const float* __restrict__ raw, const float* __restrict__ zb, const float* __restrict__ rays_d,
                                                       long R, int S, int radiance_linear, float* __restrict__ maps, float* __restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const float norm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float* zrow = zb + (long)S * r;
    float z[NPL], zn[NPL], sig[NPL], w[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        z[i] = s < S ? zrow[s] : 0.0f;
        zn[i] = s + 1 < S ? zrow[s + 1] : 0.0f;
        sig[i] = s < S ? raw[((long)r * S + s) * RAW_CH] : 0.0f;
    }
    ray_weights<NPL>(sig, z, zn, norm, S, lane, w);
    float m[CM_CH];
#pragma unroll
    for (int c = 0; c < CM_CH; ++c) m[c] = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        if (s < S) {
            if (weights) weights[(long)r * S + s] = w[i];
            m[0] += w[i] * z[i];
            m[1] += w[i];
            const float* row = raw + ((long)r * S + s) * RAW_CH;
#pragma unroll
            for (int c = 0; c < 17; ++c) m[2 + c] += w[i] * (c < 4 ? sigmoidf_(row[1 + c]) : radiance_f(row[1 + c], radiance_linear));
        }
    }
#pragma unroll
    for (int c = 0; c < CM_CH; ++c) m[c] = wave_sum(m[c]);
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < CM_CH; ++c) maps[r * CM_CH + c] = m[c];
}

template <int NPL>
__global__ __launch_bounds__(256) void k_composite_bwd(const float* __restrict__ raw, const float* __restrict__ zb, const float* __restrict__ rays_d,
                                                       long R, int S, int radiance_linear, const float* __restrict__ dmaps,
                                                       const float* __restrict__ dweights, float* __restrict__ draw, int detach) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    const float norm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float* zrow = zb + (long)S * r;
    float dm[CM_CH];
#pragma unroll
    for (int c = 0; c < CM_CH; ++c) dm[c] = dmaps[r * CM_CH + c];
    // alpha, transmittance (double prefix product, as the forward), the activations and g_s = dL/dw_s
    double alpha[NPL], om[NPL], da[NPL], T[NPL], g[NPL];
    float act[NPL][17], dact[NPL][17];
    double lane_prod = 1.0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int s = lane * NPL + i;
        const bool in = s < S;
        const float zz = in ? zrow[s] : 0.0f, zn = s + 1 < S ? zrow[s + 1] : 0.0f;
        const float* row = raw + ((long)r * S + (in ? s : 0)) * RAW_CH;
        const float sg = in ? row[0] : 0.0f;
        const float dist = (s == S - 1 ? 1e10f : (zn - zz)) * norm;
        const float af = 1.0f - expf(-fmaxf(sg, 0.0f) * dist);                      // the forward's float alpha
        alpha[i] = in ? (double)af : 0.0;
        om[i] = in ? (double)((1.0f - af) + 1e-10f) : 1.0;
        da[i] = (in && sg > 0.0f) ? (double)dist * (double)expf(-sg * dist) : 0.0;
        lane_prod *= om[i];
        double gs = in ? (double)dm[0] * zz + (double)dm[1] + (dweights ? (double)dweights[(long)r * S + s] : 0.0) : 0.0;
#pragma unroll
        for (int c = 0; c < 17; ++c) {
            const float x = in ? row[1 + c] : 0.0f;
            float a_, d_;
            if (c < 4 || !radiance_linear) { a_ = sigmoidf_(x); d_ = a_ * (1.0f - a_); }
            else { a_ = fmaxf(x, 0.0f); d_ = x > 0.0f ? 1.0f : 0.0f; }
            act[i][c] = a_;
            dact[i][c] = d_;
            if (in && (!detach || (c >= 5 && c < 8))) gs += (double)dm[2 + c] * a_;   // channels 5..7 = radiance_map: the one colour map on `weights`
        }
        g[i] = gs;
    }
    double incl = lane_prod;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const double o = __shfl_up(incl, dd);
        if (lane >= dd) incl *= o;
    }
    double t = __shfl_up(incl, 1);
    if (lane == 0) t = 1.0;
    double lane_gw = 0.0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        T[i] = t;
        lane_gw += g[i] * alpha[i] * t;
        t *= om[i];
    }
    double sfx = lane_gw;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const double o = __shfl_down(sfx, dd);
        if (lane + dd < 64) sfx += o;
    }
    double after = __shfl_down(sfx, 1);
    if (lane == 63) after = 0.0;
#pragma unroll
    for (int i = NPL - 1; i >= 0; --i) {
        const int s = lane * NPL + i;
        if (s < S) {
            float* orow = draw + ((long)r * S + s) * RAW_CH;
            orow[0] = (float)((g[i] * T[i] - after / om[i]) * da[i]);
            const float wf = (float)(alpha[i] * T[i]);
#pragma unroll
            for (int c = 0; c < 17; ++c) orow[1 + c] = wf * dm[2 + c] * dact[i][c];
        }
        after += g[i] * alpha[i] * T[i];
    }
}

template <class F>
hipError_t by_npl(int S, F&& f) {
    const int npl = (S + 63) / 64;
    switch (npl) {
        case 1: f(std::integral_constant<int, 1>{}); break;
        case 2: f(std::integral_constant<int, 2>{}); break;
        case 3: f(std::integral_constant<int, 3>{}); break;
        case 4: f(std::integral_constant<int, 4>{}); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace

__global__ void k_broadcast_rows(const float* __restrict__ row, int S, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = row[i % S];
}
// z_vals.expand([N_rays, N_samples]) materialised (ibl_nerf_renderer.py:676): one shared row -> R rows
hipError_t launch_broadcast_rows(const float* row, int S, long R, float* out, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    const long n = R * S;
    hipLaunchKernelGGL(k_broadcast_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, row, S, n, out);
    return hipGetLastError();
}

hipError_t launch_get_rays(int W, int row0, int row_step, long n_rows, const long long* pixels, const Camera& cam, float* rays_o, float* rays_d, hipStream_t s) {
    const long n = pixels ? n_rows : n_rows * W;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_get_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, row0, row_step, n_rows, pixels, cam, rays_o, rays_d);
    return hipGetLastError();
}

hipError_t launch_ray_grid(const float* near, const float* far, int S, int lindisp, const float* t_rand, long R, float* out, hipStream_t s) {
    const long n = R * S;
    hipLaunchKernelGGL(k_ray_grid, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, near, far, S, lindisp, t_rand, R, out);
    return hipGetLastError();
}

hipError_t launch_coarse_z(float near, float far, int S, int lindisp, float* z, hipStream_t s) {
    hipLaunchKernelGGL(k_coarse_z, dim3((S + 63) / 64), dim3(64), 0, s, near, far, S, lindisp, z);
    return hipGetLastError();
}

hipError_t launch_make_points(int mode, const float* origin, const float* dir, const float* z, int z_stride, float eps,
                              long R, int S, float* out, hipStream_t s) {
    const long n = R * S;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_make_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mode, origin, dir, z, z_stride,
                       eps, R, S, out);
    return hipGetLastError();
}

hipError_t launch_pass_a(const PassAArgs& a, const PassOutputs& out, int gamma, hipStream_t s) {
    if (a.R <= 0) return hipSuccess;
    const dim3 grid((unsigned)((a.R + 3) / 4));
    return by_npl(a.S, [&](auto N) {
        hipLaunchKernelGGL(k_pass_a<decltype(N)::value>, grid, dim3(256), 0, s, a, out, gamma);
    });
}

hipError_t launch_composite_direct(const float* raw, const float* z, const float* rays_d, long R, int S, int radiance_linear, float* maps,
                                   float* weights, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    const dim3 grid((unsigned)((R + 3) / 4));
    return by_npl(S, [&](auto N) {
        hipLaunchKernelGGL(k_composite_fwd<decltype(N)::value>, grid, dim3(256), 0, s, raw, z, rays_d, R, S, radiance_linear, maps, weights);
    });
}
hipError_t launch_composite_direct_backward(const float* raw, const float* z, const float* rays_d, long R, int S, int radiance_linear,
                                            const float* dmaps, const float* dweights, float* draw, hipStream_t s, int detach) {
    if (R <= 0) return hipSuccess;
    const dim3 grid((unsigned)((R + 3) / 4));
    return by_npl(S, [&](auto N) {
        hipLaunchKernelGGL(k_composite_bwd<decltype(N)::value>, grid, dim3(256), 0, s, raw, z, rays_d, R, S, radiance_linear, dmaps, dweights, draw, detach);
    });
}

hipError_t launch_sigma_weights(const float* rays_d, const float* z, int z_stride, const float* sigma, const float* noise, long R, int S,
                                float* weights, hipStream_t s, float* depth, float* visibility) {
    if (R <= 0) return hipSuccess;
    const dim3 grid((unsigned)((R + 3) / 4));
    return by_npl(S, [&](auto N) {
        hipLaunchKernelGGL(k_sigma_weights<decltype(N)::value>, grid, dim3(256), 0, s, rays_d, z, z_stride, sigma, noise, R, S, weights, depth, visibility);
    });
}

hipError_t launch_select_points(const float* rays_o, const float* rays_d, const float* z, int z_stride, const float* sigma, int sigma_stride, const float* noise,
                                long R, int S, float margin, float t_min, float* pts_out, int* index_out, int* counter, hipStream_t s, bool offsets, float eps,
                                float* est_out, int est_stride, double list_flop_per_point, int* range_out, const int* skip_range, double list_slots_per_point, float* est_list,
                                const unsigned long long* tier_mask, int tier) {
    if (R <= 0) return hipSuccess;
    const dim3 grid((unsigned)(((offsets ? 4 * R : R) + SELECT_WAVES - 1) / SELECT_WAVES)), block(64 * SELECT_WAVES);
    const hipError_t e = by_npl(S, [&](auto N) {
        if (offsets)
            hipLaunchKernelGGL((k_select_points<decltype(N)::value, true>), grid, block, 0, s, rays_o, rays_d, z, z_stride, sigma, sigma_stride, noise, R, S, margin,
                               t_min, eps, pts_out, index_out, counter, est_out, est_stride, range_out, skip_range, est_list, tier_mask, tier);
        else
            hipLaunchKernelGGL((k_select_points<decltype(N)::value, false>), grid, block, 0, s, rays_o, rays_d, z, z_stride, sigma, sigma_stride, noise, R, S, margin,
                               t_min, eps, pts_out, index_out, counter, est_out, est_stride, range_out, skip_range, est_list, tier_mask, tier);
    });
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_count_selection, dim3(1), dim3(1), 0, s, counter, list_flop_per_point, list_slots_per_point, 1);
    return hipGetLastError();
}

hipError_t launch_chunk_points(const float* rays_o, const float* rays_d, const float* z, int z_stride, float* sigma, const float* noise, long R, int S, int s0, int s1,
                               float margin, float t_min, float* pts_out, int* index_out, int* counter, hipStream_t s, bool offsets, float eps, bool first,
                               double flop_per_point, double slots_per_point) {
    if (R <= 0 || s1 <= s0) return hipSuccess;
    const dim3 grid((unsigned)(((offsets ? 4 * R : R) + SELECT_WAVES - 1) / SELECT_WAVES)), block(64 * SELECT_WAVES);
    const hipError_t e = by_npl(S, [&](auto N) {
        constexpr int NPL = decltype(N)::value;
#define IBL_CHUNK(OFF, FST) hipLaunchKernelGGL((k_chunk_points<NPL, OFF, FST>), grid, block, 0, s, rays_o, rays_d, z, z_stride, sigma, noise, R, S, s0, s1, margin, t_min, \
                                               eps, pts_out, index_out, counter)
        if (offsets) { if (first) IBL_CHUNK(true, true); else IBL_CHUNK(true, false); }
        else { if (first) IBL_CHUNK(false, true); else IBL_CHUNK(false, false); }
#undef IBL_CHUNK
    });
    if (e != hipSuccess) return e;
    if (!first) hipLaunchKernelGGL(k_count_selection, dim3(1), dim3(1), 0, s, counter, flop_per_point, slots_per_point, 0);     // (the chunk's estimates: executed MACs, not refined samples)
    return hipGetLastError();
}

hipError_t launch_importance(const float* rays_d, const float* z, int z_stride, const float* sigma, int sigma_stride, long R, int S, float tau, unsigned long long* mask,
                             hipStream_t s, int mode, float margin, const unsigned long long* exclude) {
    if (R <= 0) return hipSuccess;
    return by_npl(S, [&](auto N) {
        hipLaunchKernelGGL((k_importance<decltype(N)::value>), dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, rays_d, z, z_stride, sigma, sigma_stride, R, S, tau, mask, mode, margin,
                           exclude);
    });
}

hipError_t launch_range_points(const float* rays_o, const float* rays_d, const float* z, int z_stride, float* sigma, const int* main_range, long R, int S, int mode,
                               float margin, float t_min, float eps, float* pts_out, int* index_out, int* counter, hipStream_t s, double flop_per_point, double slots_per_point, bool count_entries,
                               const unsigned long long* tier_mask, int tier) {
    if (R <= 0) return hipSuccess;
    const dim3 grid((unsigned)((4 * R + SELECT_WAVES - 1) / SELECT_WAVES)), block(64 * SELECT_WAVES);
    const hipError_t e = by_npl(S, [&](auto N) {
        hipLaunchKernelGGL((k_range_points<decltype(N)::value>), grid, block, 0, s, rays_o, rays_d, z, z_stride, sigma, main_range, R, S, mode, margin, t_min, eps, pts_out,
                           index_out, counter, tier_mask, tier);
    });
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_count_selection, dim3(1), dim3(1), 0, s, counter, flop_per_point, slots_per_point, count_entries ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_tripwire(const float* est_list, const int* index, const int* n_dev, const float* out, int out_stride, float margin, unsigned* flag, long n_bound, hipStream_t s,
                           unsigned char* trip_rays, int S, long R) {
    if (n_bound <= 0 || flag == nullptr) return hipSuccess;
    const long blocks = (n_bound + 255) / 256;
    hipLaunchKernelGGL(k_tripwire, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, est_list, index, n_dev, out, out_stride, margin, flag, trip_rays,
                       S > 0 ? S : 1, R > 0 ? R : 1);
    return hipGetLastError();
}

hipError_t launch_compare_estimates(const float* a, const float* b, long n, float margin, float zone, int* bad, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_compare_estimates, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, n, margin, zone, bad);
    return hipGetLastError();
}

hipError_t launch_surface_points(const float* rays_o, const float* rays_d, const float* z, int z_stride, const float* raw, const float* noise,
                                 long R, int S, const OverrideArgs& ov, float* surf, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    const dim3 grid((unsigned)((R + 3) / 4));
    return by_npl(S, [&](auto N) {
        hipLaunchKernelGGL(k_surface_points<decltype(N)::value>, grid, dim3(256), 0, s, rays_o, rays_d, z, z_stride, raw, noise, R, S, ov, surf);
    });
}

hipError_t launch_pass_b(const PassBArgs& a, hipStream_t s) {
    if (a.R <= 0) return hipSuccess;
    const dim3 grid((unsigned)((a.R + 3) / 4));
    return by_npl(a.Sc, [&](auto N) { hipLaunchKernelGGL(k_pass_b<decltype(N)::value>, grid, dim3(256), 0, s, a); });
}


// ---- the ray-sized part of raw2outputs, differentiated (training: ibl_nerf_renderer.py:258, :412-474, :477-527) -----------------------------
// One thread per ray: from the pass's linear direct maps x [19] (the slots of k_composite_fwd) and the pass's no-grad quantities (n.v, the four linear
// reflected-ray maps, the LUT) it re-evaluates the shading and applies the chain rule by hand — what torch autograd does in ~160 ray-sized launches
// (training.py _ray_outputs, which stays as the test's reference).  Ties of the two maximum() calls split the gradient in halves like ATen's.
__device__ __forceinline__ float d_srgb(float y, int on) { return on ? (float)(1.0 / 2.2) * powf(y + 1e-12f, (float)(1.0 / 2.2) - 1.0f) : 1.0f; }
// d out_f(v) / dv, out_f = gamma(tonemap(v)) (:30-31, :26-27, :487)
__device__ __forceinline__ float d_out_map(float v, int on) {
    if (on & 2) {
        const float t = v + 1.0f;
        return d_srgb(v / t, on & 1) / (t * t);
    }
    return d_srgb(v, on & 1);
}

__global__ void k_ray_outputs_bwd(RayBwdArgs a, long n) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* x = a.x + 19 * r;
    float dx[19];
#pragma unroll
    for (int i = 0; i < 19; ++i) dx[i] = 0.0f;
    const int on = a.out_mode;
    // target maps (:320-330): the network's, or the ground truth as a constant; roughness_map itself (the network's) still sets the mip level (:457-460)
    const bool alb_gt = a.gt_albedo != nullptr, rough_gt = a.gt_roughness != nullptr, irr_gt = a.gt_irradiance != nullptr;
    const float depth = x[0], acc = x[1], rough_net = x[5];
    const float rough = rough_gt ? a.gt_roughness[r] : rough_net;
    auto up3 = [&](const float* g, int c) { return g ? g[3 * r + c] : 0.0f; };
    auto up1 = [&](const float* g) { return g ? g[r] : 0.0f; };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dx[7 + c] += up3(a.g_radiance, c) * d_out_map(x[7 + c], on);
#pragma unroll
        for (int k = 0; k < 3; ++k) dx[10 + 3 * k + c] += up3(a.g_radiance_k[k], c) * d_out_map(x[10 + 3 * k + c], on);
        if (!alb_gt) dx[2 + c] += up3(a.g_albedo, c) * d_srgb(x[2 + c], on & 1);     // (results["albedo_map"] = albedo_f(target_albedo_map): a constant under the flag)
    }
    if (!irr_gt) dx[6] += up1(a.g_irradiance) * d_out_map(x[6], on);
    if (!rough_gt) dx[5] += up1(a.g_roughness);
    // (an upstream gradient of exactly zero is an output the loss does not read — autograd hands the Function zeros for those — and contributes nothing, as in the
    // reference, where no backward runs through an unused output: a ray whose weights sum to zero has disp = 1 / max(1e-10, 0 / 0) = NaN there too, and its 0 x NaN here
    // poisoned dL/d depth, the whole batch's upstream maximum with it, and every later step of a training run: bench.py --train found it)
    if (a.g_disp && a.g_disp[r] != 0.0f) {                   // 1 / max(1e-10, depth / acc) (:258)
        const float q = depth / acc;
        const float share = q > 1e-10f ? 1.0f : (q == 1e-10f ? 0.5f : 0.0f);
        const float dq = -a.g_disp[r] / (fmaxf(q, 1e-10f) * fmaxf(q, 1e-10f)) * share;
        dx[0] += dq / acc;
        dx[1] += -dq * depth / (acc * acc);
    }
    dx[1] += up1(a.g_acc);
    dx[0] += up1(a.g_depth) + (a.gt_depth != nullptr ? 0.0f : up1(a.g_target_depth));
    if (a.ndv != nullptr) {
        const float ndv = a.ndv[r];
        // LUT fetch and its derivative along the roughness axis (F.grid_sample, bilinear, zeros padding, align_corners=True; :418-421)
        constexpr int N = 512;
        const float gx = 2.0f * ndv - 1.0f, gy = 2.0f * rough - 1.0f;
        const float lx = ((gx + 1.0f) / 2.0f) * (float)(N - 1), ly = ((gy + 1.0f) / 2.0f) * (float)(N - 1);
        const float x0 = floorf(lx), y0 = floorf(ly);
        float e0 = 0.f, e1 = 0.f, de0 = 0.f, de1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ddx = k & 1, ddy = k >> 1;
            const float xi = x0 + (float)ddx, yi = y0 + (float)ddy;
            const float wx = ddx ? (lx - x0) : (x0 + 1.0f - lx), wy = ddy ? (ly - y0) : (y0 + 1.0f - ly);
            if (xi >= 0.0f && xi <= (float)(N - 1) && yi >= 0.0f && yi <= (float)(N - 1)) {
                const int o = (int)yi * N + (int)xi;
                const float v0 = a.lut[o], v1 = a.lut[N * N + o];
                e0 += v0 * (wx * wy);
                e1 += v1 * (wx * wy);
                de0 += v0 * wx * (ddy ? 1.0f : -1.0f);
                de1 += v1 * wx * (ddy ? 1.0f : -1.0f);
            }
        }
        de0 *= (float)(N - 1);                               // d ly / d rough = (N - 1) / 2 * 2
        de1 *= (float)(N - 1);
        const float m = 1.0f - rough;                        // metallic (:424)
        const float one_m = 1.0f - m;
        const float p5 = powf(fminf(fmaxf(1.0f - ndv, 0.0f), 1.0f), 5.0f);
        // mip level (:453-467); the depth inside it carries no gradient (depth_map.detach())
        float level = rough_net, dlevel = 1.0f;
        if (a.correct_depth) {
            const float d0 = a.depth0_ray != nullptr ? a.depth0_ray[r] : a.depth0;
            const float v = rough_net * depth / d0;
            level = fminf(fmaxf(v, 0.0f), 1.0f);
            dlevel = (v >= 0.0f && v <= 1.0f) ? depth / d0 : 0.0f;
        }
        int i1 = (int)(level * 3.0f);
        i1 = i1 < 0 ? 0 : (i1 > 3 ? 3 : i1);
        const int i2 = i1 + 1 > 3 ? 3 : i1 + 1;
        const float rem = level * 3.0f - (float)i1;
        const float* env = a.env + 12 * r;
        float g_rough = 0.f, g_e0 = 0.f, g_e1 = 0.f, g_rem = 0.f, g_irr = 0.f;
        float g_env[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) g_env[i] = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float alb = alb_gt ? a.gt_albedo[3 * r + c] : x[2 + c];
            const float irr = irr_gt ? a.gt_irradiance[3 * r + c] : x[6];
            const float F0 = 0.04f * one_m + alb * m;                                             // :425-427
            const float av = 1.0f - rough;
            const float F1 = fmaxf(av, F0) - F0;                                                  // microfacet.py:8-12
            const float fres = F0 + F1 * p5;
            const float cb = a.lut_f0 ? F0 : fres;
            const float coef = cb * e0 + e1;                                                      // :433-436
            const float pref = (1.0f - rem) * env[3 * i1 + c] + rem * env[3 * i2 + c];            // :468-470
            const float diffuse = (1.0f - fres) * one_m * alb * irr;                              // :472
            const float spec = coef * pref;
            const float G_col = up3(a.g_color, c) * d_out_map(diffuse + spec, on);
            const float G_spec = up3(a.g_specular, c) * d_out_map(spec, on) + G_col;
            const float G_diff = up3(a.g_diffuse, c) * d_out_map(diffuse, on) + G_col;
            const float G_pref = up3(a.g_prefiltered, c) * d_out_map(pref, on) + G_spec * coef;
            const float G_coef = G_spec * pref;
            float G_fres = -G_diff * one_m * alb * irr;
            g_rough += G_diff * (1.0f - fres) * alb * irr;                                        // through (1 - metallic) = 1 - (1 - rough)
            float g_alb = G_diff * (1.0f - fres) * one_m * irr;
            g_irr += G_diff * (1.0f - fres) * one_m * alb;
            g_e0 += G_coef * cb;
            g_e1 += G_coef;
            float G_F0 = 0.0f;
            if (a.lut_f0) G_F0 = G_coef * e0; else G_fres += G_coef * e0;
            G_F0 += G_fres;
            const float G_F1 = G_fres * p5;
            if (av > F0) { g_rough -= G_F1; G_F0 -= G_F1; }                                       // F1 = (1 - rough) - F0
            else if (av == F0) { g_rough -= 0.5f * G_F1; G_F0 -= 0.5f * G_F1; }                   // (else F1 = F0 - F0: no gradient)
            g_alb += G_F0 * m;
            if (!alb_gt) dx[2 + c] += g_alb;
            g_rough += G_F0 * (0.04f - alb);                                                      // d F0 / d rough through metallic = 1 - rough
            g_rem += G_pref * (env[3 * i2 + c] - env[3 * i1 + c]);
#pragma unroll
            for (int k = 0; k < 4; ++k) g_env[3 * k + c] += (k == i1 ? G_pref * (1.0f - rem) : 0.0f) + (k == i2 ? G_pref * rem : 0.0f);      // :461-467
        }
        if (a.denv != nullptr) {
#pragma unroll
            for (int i = 0; i < 12; ++i) a.denv[12 * r + i] = g_env[i];
        }
        g_rough += g_e0 * de0 + g_e1 * de1;
        dx[5] += 3.0f * g_rem * dlevel + (rough_gt ? 0.0f : g_rough);     // (the mip level reads the network's roughness_map whatever the target is)
        if (!irr_gt) dx[6] += g_irr;
    }
    float* o = a.dx + 19 * r;
#pragma unroll
    for (int i = 0; i < 19; ++i) o[i] = dx[i];
}

hipError_t launch_ray_outputs_backward(const RayBwdArgs& a, long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ray_outputs_bwd, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, a, n);
    return hipGetLastError();
}

hipError_t launch_sample_pdf(const float* bins, int bins_stride, const float* weights, int w_stride, long R, int nb,
                             int n_out, const float* u, float* samples, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    if (nb < 2 || nb > 257 || n_out < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sample_pdf, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, bins, bins_stride, weights, w_stride,
                       R, nb, n_out, u, samples);
    return hipGetLastError();
}

hipError_t launch_fine_z(const float* zc, int zc_stride, int Sc, const float* weights_c, long R, int n_imp, const float* u, float* z_fine,
                         float* z_std, hipStream_t s) {
    if (R <= 0) return hipSuccess;
    if (Sc < 3 || Sc > 256 || Sc + n_imp > 512 || n_imp < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_fine_z, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, zc, zc_stride, Sc, weights_c, R, n_imp, u, z_fine, z_std);
    return hipGetLastError();
}

hipError_t launch_jitter_z(const float* z, int S, const float* t_rand, long R, float* out, hipStream_t s) {
    const long n = R * S;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_jitter_z, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, S, t_rand, R, out);
    return hipGetLastError();
}

}  // namespace ibl
