// Internal kernel launch interface (device pointers everywhere).  The public surface is
// include/iblnerf.h; these are the pieces api.cpp strings together.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"
#include "mlp_args.h"

namespace ibl {

// --- device-side weight packer (pack_kernels.hip): blob in HBM -> both weight streams + side tables ---------------
struct PackMaps {                 // device copies of pack.cpp's build_pack_maps()
    const unsigned short* id_stream;
    const int* mx;
    const int* tab;
};
// A colour-independent network (ibl_nerf.py:192) as a member of the built architecture: feature_linear = I, views_linears.0 = [I | 0], zero biases — the view layer's
// output IS the trunk's (h >= 0 after its ReLU, so the view layer's ReLU changes nothing), which is what such a network's radiance heads read.  Written over those two
// layers of a device copy of the state dict (offsets in floats: pack.h blob_offsets(8 | 9)).
hipError_t launch_identity_embed(float* d_blob, size_t views_w, size_t views_b, size_t feat_w, size_t feat_b, hipStream_t s);
hipError_t launch_pack_weights(const float* d_blob, const PackMaps& maps, char* d_stream_bf16, char* d_stream_mx, char* d_stream_f16,
                               float* d_tab, unsigned* d_range_flag, hipStream_t s);   // either fast stream may be null

// --- per-ray kernels (render_kernels.hip) ------------------------------------------------------

// nerf_renderer_helper.py:36-45.  Rows [row0, row0+n_rows) of an H x W image.
struct Camera { float K[9]; float c2w[12]; };   // 3x3 intrinsics, 3x4 camera-to-world, row-major (kernel argument)
// rows row0, row0 + row_step, ... (n_rows of them) of a W-wide image, or — pixels != nullptr — the n_rows listed flat pixel indices
hipError_t launch_get_rays(int W, int row0, int row_step, long n_rows, const long long* pixels, const Camera& cam, float* rays_o, float* rays_d, hipStream_t s);

// ibl_nerf_renderer.py:670-672: z_k = near (1 - t_k) + far t_k, t = linspace(0,1,S)
hipError_t launch_coarse_z(float near, float far, int S, int lindisp, float* z, hipStream_t s);
// ... per ray, for rays with their own planes (near / far [R]); t_rand [R, S] or null: the stratified jitter on top
hipError_t launch_ray_grid(const float* near, const float* far, int S, int lindisp, const float* t_rand, long R, float* out, hipStream_t s);

// Point batches (rays x samples packed contiguously, [V][R][S][3]):
//   mode 0: origin + dir * z                                  (ibl_nerf_renderer.py:200, :440)
//   mode 1: V = 4 epsilon-offset copies (+-eps*right, +-eps*up) (normal_from_depth.py:143-156)
//   mode 2: V = 4 rays from the same origin along normalize(dir +- eps*right), normalize(dir +- eps*up) (:64-73)
// z_stride = 0 -> one z row shared by all rays, else per-ray rows of length z_stride.
hipError_t launch_make_points(int mode, const float* origin, const float* dir, const float* z, int z_stride,
                              float eps, long R, int S, float* out, hipStream_t s);

struct OverrideArgs {          // edit_intrinsic / insert_object branches, ibl_nerf_renderer.py:218-238, 253-256, 378-410
    int mode;                  // 0 none, 1 edit_intrinsic, 2 insert_object
    int num_objects;
    int edit_depth, edit_normal, edit_albedo, edit_albedo_by_img, edit_roughness;
    int n_rough_list;
    const float* mask;         // [R, mask_stride] (channel 0 is read)
    int mask_stride;
    const float* depth_img;    // [R] (edit_depth / object_insert_depth, channel 0)
    int depth_stride;
    const float* normal_img;   // [R,3] in [0,1]
    const float* albedo_img;   // [R,3]
    const float* gt_normal;    // [R,3] in [0,1] or null: target normal = normalize(2 v - 1) for every ray (:370-371), no eps-normal
    // calculate_{albedo,roughness,irradiance}_from_gt / depth_map_from_ground_truth (:251-252, :320-330): null = network's value
    const float* gt_albedo;      // [R,3]
    const float* gt_roughness;   // [R] (channel 0 of gt_values["roughness"])
    const float* gt_irradiance;  // [R,3]: the irradiance becomes a colour and the irradiance map has 3 channels
    const float* gt_depth;       // [R]   (channel 0 of gt_values["depth"]); target depth only: depth_map / disp stay the network's
    const float* rough_img;      // [R] edit_roughness_by_img (:394-395): the roughness a masked ray takes (resolved per chunk by the caller), or null
    float rough_list[8];
    float albedo_list[24];
    float irr_list[8];
};

constexpr int ST_FLOATS = 14;  // per-ray record handed from pass A to pass B (see render_kernels.hip)

struct PassOutputs {           // any pointer may be null (skipped).  Shapes per ray.
    float* color;              // 3
    float* radiance;           // 3
    float* radiance_k[3];      // 3 each
    float* refl_coarse_k[3];   // 3 each
    float* irradiance;         // 1
    float* reflected_radiance; // 3
    float* prefiltered;        // 3
    float* albedo;             // 3
    float* roughness;          // 1
    float* specular;           // 3
    float* diffuse;            // 3
    float* n_dot_v;            // 1
    float* normal;             // 3
    float* disp;               // 1
    float* acc;                // 1
    float* depth;              // 1
    float* target_depth;       // 1
    float* weights;            // S
    float* inferred_normal;    // 3 (normal_mlp loaded)
};

struct PassAArgs {
    const float* rays_o; const float* rays_d;   // [R,3]
    const float* z; int z_stride;               // see launch_make_points
    const float* raw;                           // [R,S,18]
    const float* sig4;                          // [4,R,S]; with grad_normal: [R,S,4] rows (sigma, d sigma / d x, y, z) of the density-gradient query
    const float* noise = nullptr;               // [R,S] added to the density before compositing (raw_noise_std > 0, :208-216, :242) or null
    const float* nrm_raw;                       // [R,S,3] normal_mlp samples or null (ibl_nerf_renderer.py:273-276)
    int nrm_at_surface;                         // nrm_raw is [R,3]: one evaluation per ray at the surface point (:268-271)
    int normal_inferred;                        // target normal = the composited normal_mlp output, as it is (:372-373)
    float* weights;                             // [R,S] (always written: sample_pdf input / output map)
    const float* lut;                           // [3,512,512]
    float near, far, eps;
    const float* near_ray = nullptr;            // per-ray planes [R] (iblnerf_sampling.d_near / d_far) or null: depth_0 of the mip level per ray
    const float* far_ray = nullptr;
    int irradiance_sigmoid;                     // an irradiance_mlp's samples take sigmoid, whatever radiance_f is (:300-303)
    int tilted_rays;                            // 0: offset-sample depths (normal_from_depth.py:139-183), 1: tilted-ray depths (:55-100)
    int grad_normal = 0;                        // 1: normal from d depth / d ray origin (normal_from_depth.py:102-137), 2: d depth / d ray direction (:16-52)
    int lut_coefficient_F0;                     // 0 -> 'F' (shipped), 1 -> 'F0'
    int correct_depth;                          // correct_depth_for_prefiltered_radiance_infer
    int radiance_linear;                        // use_radiance_linear: radiance_f = ReLU, LDR map x/(1+x) before gamma
    OverrideArgs ov;
    float* state;                               // [R, ST_FLOATS]
    float* refl_o; float* refl_d;               // [R,3] reflected-ray origin / direction
    float* stage = nullptr;                     // optional [R, STAGE_FLOATS] stage boundaries (iblnerf_composite_pass): normal before the
                                                // edit / insert overrides (3), LUT coordinates n.v and roughness (2), LUT scale and bias (2), mip level
    long R; int S;
};
constexpr int STAGE_FLOATS = 8;
hipError_t launch_pass_a(const PassAArgs& a, const PassOutputs& out, int gamma_correct, hipStream_t s);

// coarse pass of the inference-minimum mode: compositing weights only (ibl_nerf_renderer.py:203-206, 241-245)
// Surface points x = o + d * target_depth of R rays from the main query's raw rows (ibl_nerf_renderer.py:249-262), the same
// arithmetic as pass A: the input of a normal_mlp evaluated at the surface (:268-271).
struct OverrideArgs;
hipError_t launch_surface_points(const float* rays_o, const float* rays_d, const float* z, int z_stride, const float* raw, const float* noise,
                                 long R, int S, const OverrideArgs& ov, float* surf, hipStream_t s);

// the 19 direct maps of one pass from its raw rows, and their backward (render_kernels.hip: k_composite_fwd / k_composite_bwd)
hipError_t launch_composite_direct(const float* raw, const float* z, const float* rays_d, long R, int S, int radiance_linear, float* maps,
                                   float* weights, hipStream_t s);
// training: the ray-sized shading of a pass, differentiated (k_ray_outputs_bwd).  Upstream gradients per output map, null = none.
struct RayBwdArgs {
    const float* x;          // [n, 19] linear direct maps (the slots of launch_composite_direct)
    const float* ndv;        // [n] n.v of the pass, or null = approximate_radiance=False (only the direct maps' output functions)
    const float* env;        // [n, 4, 3] linear reflected-ray maps (radiance, coarse radiances 1..3)
    const float* lut;        // [3, 512, 512]
    float depth0;            // (near + far) / 2
    const float* depth0_ray = nullptr;   // ... per ray [n] (per-ray near / far planes), or nullptr
    float* denv = nullptr;               // out, optional: dL/d env [n, 4, 3] (use_gradient_for_incident_radiance: the reflected-ray maps carry a gradient)
    int out_mode;            // bit 0 gamma_correct, bit 1 use_radiance_linear (out_map)
    int lut_f0, correct_depth;
    const float *g_color, *g_radiance, *g_radiance_k[3], *g_irradiance, *g_albedo, *g_roughness, *g_specular, *g_diffuse, *g_prefiltered,
        *g_disp, *g_acc, *g_depth, *g_target_depth;
    float* dx;               // [n, 19]
    // calculate_*_from_gt / depth_map_from_ground_truth (:251-252, :320-330): the target maps the shading reads and the output maps of the same name are the ground
    // truth — constants of the backward ([n, 3], [n], [n, 3], [n]; null = the network's own map)
    const float* gt_albedo = nullptr;
    const float* gt_roughness = nullptr;
    const float* gt_irradiance = nullptr;
    const float* gt_depth = nullptr;
};
hipError_t launch_ray_outputs_backward(const RayBwdArgs& a, long n, hipStream_t s);
hipError_t launch_composite_direct_backward(const float* raw, const float* z, const float* rays_d, long R, int S, int radiance_linear,
                                            const float* dmaps, const float* dweights, float* draw, hipStream_t s, int detach = 1);
hipError_t launch_sigma_weights(const float* rays_d, const float* z, int z_stride, const float* sigma, const float* noise, long R, int S,
                                float* weights, hipStream_t s, float* depth = nullptr, float* visibility = nullptr);   // noise: [R,S] or null; depth / visibility: raw2outputs_depth's maps [R]

struct PassBArgs {
    const float* state;        // [R, ST_FLOATS]
    const float* refl_raw;     // [R,64,13]
    const float* refl_d;       // [R,3]
    const float* zc;           // coarse z (z_vals_constant): [Sc], or per-ray rows [R, Sc] with zc_stride = Sc (perturb > 0)
    int zc_stride = 0;
    int Sc;
    int gamma_correct;
    int radiance_linear;
    PassOutputs out;
    long R;
    float* env_tap = nullptr;  // [R,12] or null: the four LINEAR reflected-ray maps (radiance, coarse radiances 1..3) before tone map / gamma (training taps)
};
hipError_t launch_pass_b(const PassBArgs& a, hipStream_t s);

// nerf_renderer_helper.py:91-134 (det=True).  bins [R,nb], weights [R,nb-1] -> samples [R,n_out]
// u: [R, n_out] uniform draws (det=False) or null (det=True: u = linspace(0, 1, n_out))
hipError_t launch_sample_pdf(const float* bins, int bins_stride, const float* weights, int w_stride, long R, int nb,
                             int n_out, const float* u, float* samples, hipStream_t s);

// ibl_nerf_renderer.py:701-707, :718: mids of zc, sample_pdf on weights[:,1:-1], sort(cat), z_std
// zc: one shared row (zc_stride 0) or per-ray rows; u as in launch_sample_pdf
hipError_t launch_fine_z(const float* zc, int zc_stride, int Sc, const float* weights_c, long R, int n_imp, const float* u, float* z_fine,
                         float* z_std, hipStream_t s);
// ibl_nerf_renderer.py:678-692 (perturb > 0): z [S] shared base grid, t_rand [R,S] -> out [R,S]
hipError_t launch_jitter_z(const float* z, int S, const float* t_rand, long R, float* out, hipStream_t s);
hipError_t launch_broadcast_rows(const float* row, int S, long R, float* out, hipStream_t s);   // one shared row -> R rows

// PositionDirectionMLP (src/networks/MLP.py:32-74), one evaluation per row of pts / dirs (posdir_kernel.hip).
// weights: per layer [Wt (n_in x n_out, transposed) | bias], layers in registration order (positions_linears.0-7, feature_linear,
// views_linears.0-3, final_linear) — pack_posdir() in api.cpp.
struct PosDirArgs {
    const float* weights;
    const float* pts;      // [n,3]
    const float* dirs;     // [n,3]
    float* out;            // [n,out_ch]
    long n;
    int out_ch;
    int normalize_dirs;    // 1: dirs are rays_d, normalised here (ibl_nerf_renderer.py:795)
    int relu_out;          // 1: F.relu on the output (:724)
};
constexpr long POSDIR_FLOATS_OUT1 = 63L * 256 + 256 + 4 * (256L * 256 + 256) + (319L * 256 + 256) + 2 * (256L * 256 + 256) + (256L * 256 + 256) +
                                    (283L * 128 + 128) + 3 * (128L * 128 + 128) + (128L * 1 + 1);
// The samples whose density can reach a ray's weights through more than a negative sign or a saturated tail (render_kernels.hip: k_select_points): their points
// [n_sel, 3] and flat indices r * S + s, n_sel counted into *counter (zeroed by the caller); sigma = the density estimate, element (r, s) at (r S + s) * sigma_stride
// offsets: the rays are the 4 R epsilon-offset copies (sigma = sig4 [4][R][S], stride 1; points from gen_offset_point with `eps`)
hipError_t launch_select_points(const float* rays_o, const float* rays_d, const float* z, int z_stride, const float* sigma, int sigma_stride, const float* noise,
                                long R, int S, float margin, float t_min, float* pts_out, int* index_out, int* counter, hipStream_t s, bool offsets = false,
                                float eps = 0.0f, float* est_out = nullptr, int est_stride = 0,     // est_out: the estimate copied to element (r S + s) * est_stride
                                double list_flop_per_point = 0.0,    // what the list launches behind this selection evaluate per entry (counter[4..5] += n * that)
                                int* range_out = nullptr,            // [R][2] (not with offsets): each ray's first and last selected sample ({S, -1}: none)
                                const int* skip_range = nullptr,     // offsets: such a record of the MAIN rays — the samples predicted relevant by it ([first - 1, last + 1]) are not selected (again)
                                double list_slots_per_point = 0.0,   // ... and their matrix-slot units per entry (counter[8..9])
                                float* est_list = nullptr,           // [list length] the estimate of every list entry (for launch_tripwire)
                                const unsigned long long* tier_mask = nullptr, int tier = 0);   // only the selected samples whose k_importance flag equals `tier` (audits: tier 0)
// the estimate tripwire: after a list launch, every entry's refined density out[index[i] * out_stride] against est_list[i] (k_tripwire); raises bits 2 / 3 of *flag
// ... and marks the ray of every such entry in trip_rays [R] (nullable; entry -> ray (index / S) % R: offset copies belong to their ray)
hipError_t launch_tripwire(const float* est_list, const int* index, const int* n_dev, const float* out, int out_stride, float margin, unsigned* flag, long n_bound, hipStream_t s,
                           unsigned char* trip_rays = nullptr, int S = 1, long R = 1);

// the offset copies' samples by the main ray's relevant range (k_range_points: mode 1 the predicted range, 2 in front of it, 3 behind it for the copies still alive);
// list length at counter[0] (zeroed by the caller), executed MACs x 2 (flop_per_point per entry) added to counter[4..5], entries to counter[2..3] if count_entries
hipError_t launch_range_points(const float* rays_o, const float* rays_d, const float* z, int z_stride, float* sigma, const int* main_range, long R, int S, int mode,
                               float margin, float t_min, float eps, float* pts_out, int* index_out, int* counter, hipStream_t s, double flop_per_point, double slots_per_point, bool count_entries,
                               const unsigned long long* tier_mask = nullptr, int tier = 0);   // mode 1: only the samples whose k_importance flag equals `tier`
// per sample of the R main rays: does T_s dist_s |depth - z_s| (from the main query's densities, element (r, s) at (r S + s) * sigma_stride) exceed tau?  mask [R][4] uint64:
// bit `lane` of mask[4 r + i] <-> sample lane * NPL + i (NPL = ceil(S / 64): the layout k_range_points reads)
// mode 1: the sample's own weight alpha_s T_s exceeds tau (the fine main query's tiers, from its density estimates)
hipError_t launch_importance(const float* rays_d, const float* z, int z_stride, const float* sigma, int sigma_stride, long R, int S, float tau, unsigned long long* mask,
                             hipStream_t s, int mode = 0, float margin = 0.0f, const unsigned long long* exclude = nullptr);   // mode 4: mode 1 on density ESTIMATES, the transmittance taken conservatively (margin)

// estimates in two z-chunks: points + flat indices of samples [s0, s1) of every (virtual) ray (first: of all rays, in ray order; else: of the rays not yet saturated
// behind their first s0 samples — list length at counter[0], executed MACs added to counter[4..5]; the others' samples get the density -1e30)
hipError_t launch_chunk_points(const float* rays_o, const float* rays_d, const float* z, int z_stride, float* sigma, const float* noise, long R, int S, int s0, int s1,
                               float margin, float t_min, float* pts_out, int* index_out, int* counter, hipStream_t s, bool offsets, float eps, bool first,
                               double flop_per_point, double slots_per_point);

// networks outside the built architecture (generic_mlp.hip): layer-by-layer exact-fp32 evaluation on the matrix cores, activations in HBM
struct GenericNet {
    const float* blob = nullptr;     // the state dict flattened in registration order (checkpoint.arch_schema), fp32, device memory
    int D = 8, W = 256, L = 10, Lv = 4;
};
long generic_blob_floats(int D, int W, int L, int Lv);
size_t generic_workspace_floats(int W, int L, int Lv, long chunk);
// variant 0 FULL -> out [n][18], 1 TRUNK -> out[p * out_stride], 2 REFL -> out [n][13]; dirs [n / pts_per_ray][3] (not read by TRUNK); chunk: a multiple of pts_per_ray
hipError_t launch_generic_mlp(const GenericNet& g, int variant, const float* pts, const float* dirs, int pts_per_ray, long n, float* out, int out_stride, float* ws,
                              long chunk, hipStream_t s);

// counts into *bad the samples on which estimate `a` (plain f16) is half-way to a wrong k_select_points decision against estimate `b` (f16 + 2 fp6)
// ... and into bad[1] the bits of the largest |a - b| among the samples with |b| <= zone (bad[0..1] zeroed by the caller)
hipError_t launch_compare_estimates(const float* a, const float* b, long n, float margin, float zone, int* bad, hipStream_t s);

// iblnerf_layer_ranges (range_kernel.hip): largest |value| of each of a network's 15 wide activations on n points; blob = the fp32 state dict in device memory
struct LayerRangeArgs {
    const float* blob;
    long w_off[23], b_off[23];   // float offsets of every layer's weight / bias in the blob (pack.cpp: blob_offsets)
    const float* pts;            // [n, 3]
    const float* dirs;           // [n, 3] view direction per point (as the network takes it: rays_d, not normalised)
    long n;
    int color_independent;
    float* d_max;                // [15] (zeroed by the caller): trunk 0-7, feature_linear, albedo feature, irradiance feature, views_linears.0, additional radiance features 0-2
};
hipError_t launch_layer_ranges(const LayerRangeArgs& a, hipStream_t s);

hipError_t launch_posdir_mlp(const PosDirArgs& a, hipStream_t s);

// The trunk (positions_linears.0-7 + sigma_linear) in exact fp32 on v_mfma_f32_32x32x2_f32 (trunk_fp32_kernel.hip), whole batch or a compact list
struct TrunkFp32Args {
    const float* blob;           // the network's state dict in device memory, fp32, the reference's own [out][in] layout
    long w_off[9], b_off[9];     // float offsets of positions_linears.0-7 and sigma_linear (pack.cpp: blob_offsets, layers 0-7 and 10)
    const float* pts;            // [n, 3]
    long n;                      // points, or the bound that sizes the launch when n_dev is given
    const int* n_dev;            // a list: its length in device memory (or null)
    const int* out_index;        // a list: point i -> out[out_index[i] * out_stride] (or null: out[i * out_stride])
    float* out;
    int out_stride;
};
hipError_t launch_trunk_fp32(const TrunkFp32Args& a, int n_cu, hipStream_t s);

// Weight gradient of the trunk from the backward kernel's operand stash (wgrad_kernel.hip; layout.h: STASH_*).
struct WgradGemm {
    int dz_what, x_what;     // stash activations: dZ of the layer, its input (STASH_X + l - 1 / STASH_XF, or an encoding: STASH_ENC / STASH_DENC)
    int nrows;               // 256, or 128 (a 128-wide feature layer: its dZ stash entry uses k-steps 0..7)
    int ncols, enc_pairs;    // columns of this GEMM: 256 (an activation), 64 (the 63 position-encoding columns, enc_pairs 15), 32 (the 27 direction columns, 6)
    int in_dim, col_base;    // row length of the reference's [out][in] weight, first column this GEMM fills
    long blob_off;           // the weight inside the state-dict blob
    long part_off;           // this GEMM inside one split's partial sums
    long bias_part;          // >= 0: this GEMM also sums dZ for the layer's bias gradient: [2 lane halves][256] floats inside the split's partial sums
    long bias_off;           // ... and where that bias sits in the blob
};
struct WgradArgs {
    const char* stash;
    float* partial;          // [n_split][partial_stride]
    float* grad;             // the reference's state-dict layout (blob_floats() floats), zeroed by the caller
    long wave_groups, partial_stride;
    int n_split, n_gemm;
    WgradGemm gemm[17];
    long sigma_w_off, sigma_b_off;
    float unscale;           // 1 / MlpArgs::grad_scale
    // the N = 1/3 heads: d weight[c][j] = sum_p up[p][ch0 + c] * x[p][j], d bias[c] = sum_p up[p][ch0 + c]; x = a stashed activation
    struct Head { int x_what, n_ksteps, nc, ch0; long w_off, b_off; } head[8];
    int n_head;
};
constexpr long WGRAD_PARTIAL_FLOATS = 9 * 65536L + 2 * 16384L + 8192L + 5 * 32768L + 15 * 512L;   // 9 hidden + 2 position-encoding + 1 direction-encoding + 5 128-row GEMMs, 15 bias rows
hipError_t launch_wgrad(const WgradArgs& a, const float* dsigma, long n_pts, hipStream_t s);   // dsigma: [n] (or null), or with a.n_head > 0 the [n, 18] dL/d raw rows

}  // namespace ibl
