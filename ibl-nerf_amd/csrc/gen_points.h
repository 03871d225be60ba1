// The sample points of a query, generated where they are consumed (SURVEY.md 7.1 i): the main points o + d z
// (ibl_nerf_renderer.py:200) and the four epsilon-offset copies x +- eps right, x +- eps up of get_normal_from_depth_gradient_epsilon
// (normal_from_depth.py:143-156), as ONE device function shared by k_make_points (render_kernels.hip, which still fills the [V][R][S][3] batch
// for the queries that read one) and by the TRUNK forms of the MLP kernels, which call it in their input stage instead of reading a batch.
// No operation here may be contracted into an fma: the MLP kernels are compiled with contraction on, and positions feed a 2^9 frequency multiplier,
// so they must round like the reference's separate torch multiply and add — bit-identical points are what the stage tests feed and what the chunk- and
// tile-invariance tests rely on.  `#pragma clang fp contract(off)` at the head of each body does that (hipcc's default is fast-honor-pragmas).  NOT the
// __fmul_rn / __fadd_rn intrinsics: this toolchain defines them as plain `x * y` / `x + y` (__clang_hip_math.h) and fuses them like any other — round 3's
// version of this file did, and the up axis of rays with d_x != 0 (d_x^2 + 1 as one fma) and the base point of posed cameras (o + d z) were one ulp off the
// batch k_make_points writes (found in round 4 when a list of points from render_kernels.hip met the same kernel: 27 rays of 262 144 moved by 1e-6 .. 4.5e-5).
#pragma once
#include <hip/hip_runtime.h>

namespace ibl {

struct PointGen {
    const float* rays_o = nullptr;   // [R,3]; null = the kernel reads MlpArgs::pts
    const float* rays_d = nullptr;   // [R,3]
    const float* z = nullptr;        // z_stride = 0: one row [S] shared by all rays, else per-ray rows
    int z_stride = 0;
    int S = 0;
    unsigned RS = 0;                 // R * S: points per offset copy; point index p = v * RS + r * S + s, v = 0..3 (+right, -right, +up, -up)
    float eps = 0.0f;
};

// right = d x (0,1,0), up = right x d with torch.cross's products and differences rounded one by one (normal_from_depth.py:143-147)
__device__ __forceinline__ void gen_right_up(const float (&d)[3], float (&right)[3], float (&up)[3]) {
#pragma clang fp contract(off)
    right[0] = d[1] * 0.0f - d[2] * 1.0f;
    right[1] = d[2] * 0.0f - d[0] * 0.0f;
    right[2] = d[0] * 1.0f - d[1] * 0.0f;
    up[0] = right[1] * d[2] - right[2] * d[1];
    up[1] = right[2] * d[0] - right[0] * d[2];
    up[2] = right[0] * d[1] - right[1] * d[0];
}

// The generator's parameters as the MLP kernels read them: NOT from the by-value kernel argument (seven more scalars live across an
// input stage -> 8-layer body -> epilogue loop cost the mixed TRUNK kernel 27 more spilled SGPRs, a scratch slot and 3.4 % of its time even
// with the branch not taken), but from the kernarg segment itself, through a pointer made opaque once per iteration, so that the fields are
// loaded (scalar cache) where they are used and die there.  `offset` = offsetof(MlpArgs, gen); MlpArgs is the kernel's only argument.
typedef const __attribute__((address_space(4))) PointGen* PointGenK;   // in the kernarg segment (constant address space: scalar loads)
__device__ __forceinline__ PointGenK kernarg_point_gen(unsigned offset) {
    const __attribute__((address_space(4))) char* base = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    PointGenK g = (PointGenK)(base + offset);
    asm volatile("" : "+s"(g));
    return g;
}
__device__ __forceinline__ PointGen load_point_gen(PointGenK k) {
    PointGen g;
    g.rays_o = k->rays_o; g.rays_d = k->rays_d; g.z = k->z; g.z_stride = k->z_stride; g.S = k->S; g.RS = k->RS; g.eps = k->eps;
    return g;
}

// offset copy v of sample (r, s): o + d z, then +- eps right / +- eps up
__device__ __forceinline__ void gen_offset_point(const PointGen& g, unsigned p, float& px, float& py, float& pz) {
    const unsigned v = p / g.RS, idx = p - v * g.RS;
    const unsigned r = idx / (unsigned)g.S, s = idx - r * (unsigned)g.S;
    const float zz = g.z[(size_t)g.z_stride * r + s];
    const float o[3] = {g.rays_o[3 * (size_t)r], g.rays_o[3 * (size_t)r + 1], g.rays_o[3 * (size_t)r + 2]};
    const float d[3] = {g.rays_d[3 * (size_t)r], g.rays_d[3 * (size_t)r + 1], g.rays_d[3 * (size_t)r + 2]};
    float right[3], up[3];
    gen_right_up(d, right, up);
    const float* axis = v < 2 ? right : up;
    float q[3];
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = d[c] * zz;
            const float base = o[c] + t;
            const float e = g.eps * axis[c];
            q[c] = (v & 1u) ? base - e : base + e;
        }
    }
    px = q[0]; py = q[1]; pz = q[2];
}

}  // namespace ibl
