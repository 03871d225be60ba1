/* CPU ORACLE (test infrastructure, NOT the product) — C restatement of IBL-NeRF's forward / inference hot path on host memory:
 * SURVEY.md section 8 (b) / (d)'s `iblnerf_render_cpu`, "the same contract on host pointers".
 *
 * It lives under oracle/ on purpose: the shipped library (include/iblnerf.h, libiblnerf_hip.so) has no CPU path and fails without a
 * HIP device; nothing under ibl-nerf_amd/ loads this library.  Users: tests/ (the second, independent checker beside the numpy
 * oracle — it finishes a 65 536-ray launch in a minute where numpy needs ten) and bench.py's `cpu_baseline` leg (the CPU figure
 * measured on the GPU box's host cores, kind "port").
 *
 * Parity status: PINNED — tests/test_oracle_c.py checks it against the same reference-generated fixtures (the .npz files of tests/golden) and at
 * the same tolerances as the numpy oracle, and bit for bit against the reference's sample_pdf on the threshold-critical fixture.
 *
 * Arithmetic: float32 everywhere the reference computes in float32, double where ATen's CPU kernels accumulate in double (cumprod,
 * cumsum), torch.sum's own cascade order for the row sum inside sample_pdf; every multiply and add outside the dense layers is
 * individually rounded (-ffp-contract=off: `o + d * z` is a rounded product and a rounded sum); the dense layers (csrc/gemm.c) take
 * their K products in order with fused multiply-adds.  The structs are the shipped ABI's own (include/iblnerf.h) with HOST pointers.
 *
 * Built by oracle/build_cpu.py (gcc -O2 -fopenmp; called from __graft_entry__.build()) into oracle/_build/libiblnerf_cpu.so. */
#ifndef IBLNERF_CPU_H
#define IBLNERF_CPU_H
#include "../include/iblnerf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* replaces, on the CPU: batchify_rays -> render_rays -> raw2outputs (nerf_models/ibl_nerf_renderer.py:735-756, :629-732, :153-527) with
 * approximate_radiance=True, perturb=0, raw_noise_std=0 — the contract of iblnerf_render_rays (include/iblnerf.h) with host pointers
 * in `rays_*`, `overrides` and `outputs`, and the two networks + the LUT passed per call instead of uploaded into a context:
 * blob_coarse / blob_fine = the state-dict blob of iblnerf_upload_weights (n_floats = 798 994 each; blob_fine may be NULL when
 * n_importance == 0), lut_rgb [3,512,512].
 * Options honoured: n_samples, n_importance, epsilon, gamma_correct, lut_coefficient_f0, correct_depth_for_prefiltered_radiance,
 * coarse_outputs, lindisp, use_radiance_linear, color_independent_to_direction, normal_mode DEPTH_GRADIENT_EPSILON | GROUND_TRUTH;
 * mlp_precision / device / workspace fields are ignored (fp32 on the host).  Anything else the struct can express (other normal
 * modes, the *_from_gt rows) returns IBLNERF_ERR_INVALID — never a silently different result.
 * n_threads <= 0: omp_get_max_threads().  Returns 0 or a negative iblnerf_status; message in iblnerf_cpu_last_error(). */
int iblnerf_render_cpu(const iblnerf_options* opts, const float* blob_coarse, const float* blob_fine, size_t n_floats, const float* lut_rgb,
                       const float* rays_o, const float* rays_d, int64_t n_rays, float near_, float far_,
                       const iblnerf_overrides* overrides, const iblnerf_outputs* outputs, int n_threads);

/* network_query_fn = run_network (nerf_models/ibl_nerf.py:236-252): pts [n_rays, n_samples, 3]; dirs [n_rays, 3] -> out [n_rays,
 * n_samples, 18]; dirs NULL -> the density alone, out [n_rays, n_samples] (forward_not_freezed's early return, :175-176). */
int iblnerf_network_query_cpu(const float* blob, size_t n_floats, int color_independent, const float* pts, int64_t n_rays, int n_samples,
                              const float* dirs, float* out, int n_threads);

/* sample_pdf(det=True) (nerf_models/nerf_renderer_helper.py:91-134): bins [n, n_bins], weights [n, n_bins - 1] -> out [n, n_samples] */
int iblnerf_sample_pdf_cpu(const float* bins, const float* weights, int64_t n, int n_bins, int n_samples, float* out);

/* get_rays (nerf_models/nerf_renderer_helper.py:36-45): K [3,3], c2w [3,4] -> rays_o / rays_d [H*W, 3] */
int iblnerf_get_rays_cpu(int H, int W, const float* K, const float* c2w, float* rays_o, float* rays_d);

const char* iblnerf_cpu_last_error(void);
/* "avx512" | "avx2" | "base": the dense-layer build the running CPU selected */
const char* iblnerf_cpu_isa(void);

#ifdef __cplusplus
}
#endif
#endif
