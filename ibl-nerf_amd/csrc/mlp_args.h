// Launch interface of the fused MLP kernels (mlp_kernel.hip, mlp_kernel_mx.hip).  Kept apart from kernels.h so that a change
// to the per-ray kernels' interface does not recompile the MLP instantiations (minutes each).
#pragma once
#include <hip/hip_runtime.h>

#include "gen_points.h"
#include "layout.h"

namespace ibl {

struct MlpArgs {
    const char* stream;   // packed weight stream of one network (STREAM_BYTES)
    const float* tables;  // TAB_FLOATS floats
    const float* pts;     // [n_pts,3]
    const float* dirs;    // [n_pts / pts_per_ray, 3] view directions (null for VAR_TRUNK)
    float* out;           // [n_pts,18] (FULL) | [n_pts] (TRUNK) | [n_pts,13] (REFL)
    int out_stride = 1;   // VAR_TRUNK only: floats between consecutive points' outputs (an auxiliary network writes one
                          // column of the main network's raw rows)
    long n_pts;
    int pts_per_ray;
    unsigned* range_flag; // f16 flavours only: set to 1 if an input or activation left the f16 range (may be null)
    // VAR_TRUNK_BWD only (the trunk's backward for a training step): upstream gradient per point and the operand stash the
    // weight-gradient kernel reads (layout.h: STASH_*)
    const float* dsigma = nullptr;   // [n_pts] dL / d sigma
    const float* dh7 = nullptr;      // or [n_pts, 256] dL / d trunk features (then dsigma is not read: the heads live with the caller)
    const float* dh2 = nullptr;      // VAR_TRUNK_BWD_FEAT2: [n_pts, 256] dL / d relu(views_linears.0) output
    float* out2 = nullptr;           // VAR_TRUNK_FEAT2: [n_pts, 256] that output (out = the trunk features)
    const float* draw = nullptr;     // VAR_NET_BWD: [n_pts, 18] dL / d raw (the network's 18 output channels, ibl_nerf.py:200-208)
    char* stash = nullptr;
    PointGen gen;                    // VAR_TRUNK / VAR_TRUNK_X only: gen.rays_o != null = the four epsilon-offset copies are generated in the input stage (pts is not read)
    // VAR_TRUNK_P only: a compact point list whose length lives in device memory (k_select_points) and whose results are scattered: point i of the list ->
    // out[out_index[i] * out_stride]; n_pts is then only the upper bound that sizes the launch
    const int* n_pts_dev = nullptr;
    const int* out_index = nullptr;
    float grad_scale = 1.0f;         // a power of two: dZ = grad_scale * true dZ everywhere (keeps small gradients out of the f16 denormals);
                                     // the point gradient is unscaled in the kernel, the weight gradient by the weight-gradient kernels
};
hipError_t launch_mlp(int variant, const MlpArgs& a, int n_cu, hipStream_t stream);      // three bf16 products (layout.h)
hipError_t launch_mlp_f16x3(int variant, const MlpArgs& a, int n_cu, hipStream_t stream);  // three f16 products, same stream layout with f16 pairs
hipError_t launch_mlp_mx(int variant, const MlpArgs& a, int n_cu, hipStream_t stream);   // f16 + MX-fp6 (layout_mx.h); also VAR_TRUNK_X, its mixed trunk form
hipError_t launch_mlp_mx16(int variant, const MlpArgs& a, int n_cu, hipStream_t stream); // plain f16 on the same stream (FULL / REFL forms only)

}  // namespace ibl
