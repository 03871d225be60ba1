/* iblnerf.h — C ABI of the MI355X-native IBL-NeRF forward/inference renderer.
 *
 * The reference (changwoonchoi/IBL-NeRF) has no FFI; its seam is a Python call signature.  Each
 * entry point below names the reference callable it replaces (paths relative to the reference's
 * src/).  A Python caller binds them with ctypes (ibl-nerf_amd/binding.py; INTEGRATION.md shows the
 * stub a reference maintainer would add to test.py / train.py).
 *
 * Conventions
 *   - every function returns 0 on success, a negative iblnerf_status on failure; the message is
 *     retrievable with iblnerf_last_error(ctx) (thread-unsafe per ctx, like the reference's
 *     single-threaded renderer).  No exceptions cross the boundary.
 *   - "d_" pointers are DEVICE pointers (fp32 unless stated), borrowed for the duration of the
 *     call and never freed by the library; "h_" pointers are host pointers.
 *   - `stream` is a hipStream_t passed as void* (e.g. torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it, nothing synchronises unless stated.
 *   - one ctx per device; calls on one ctx are serialised by the caller.
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails.
 */
#ifndef IBLNERF_H
#define IBLNERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iblnerf_ctx iblnerf_ctx;

typedef enum {
    IBLNERF_OK = 0,
    IBLNERF_ERR_INVALID = -1,    /* bad argument (the reference would raise AssertionError / ValueError) */
    IBLNERF_ERR_STATE = -2,      /* weights / LUT not uploaded yet */
    IBLNERF_ERR_HIP = -3,        /* a HIP runtime call failed */
    IBLNERF_ERR_NOMEM = -4
} iblnerf_status;

/* Effective flag set of the shipped configs (config_parser.py defaults + configs/common.txt,
 * configs/IBL-NeRF/.../IBL-NeRF.txt) as consumed through render_kwargs (nerf_models/ibl_nerf.py:380-426). */
typedef struct {
    int32_t n_samples;                 /* N_samples (64) */
    int32_t n_importance;              /* N_importance (128; 0 = coarse pass only) */
    float epsilon;                     /* epsilon_for_numerical_normal (0.01) */
    int32_t gamma_correct;             /* 1 */
    int32_t lut_coefficient_f0;        /* 0 = lut_coefficient "F" (shipped), 1 = "F0" */
    int32_t correct_depth_for_prefiltered_radiance; /* 1 */
    int32_t coarse_outputs;            /* 1 = also produce the coarse-pass "...0" maps exactly as
                                          render_rays does (ibl_nerf_renderer.py:712-713); 0 = the
                                          coarse pass only evaluates density for the fine sampling */
    int32_t max_rays_per_launch;       /* workspace is sized for this many rays (default 65536): about 43 KB per ray at 64 + 128 samples — points, raw rows, offset
                                          densities, weights, and the compacted point lists of the per-sample refinement (12 KB of it): 2.7 GB by default */
    int32_t device;                    /* HIP device ordinal */
    int32_t lindisp;                   /* 0 (shipped) | 1: sample linearly in inverse depth (ibl_nerf_renderer.py:673-674) */
    int32_t use_radiance_linear;       /* 0 (shipped, sigmoid radiance) | 1: ReLU radiance + Reinhard LDR map (:30-35, :480-483) */
    int32_t normal_mode;               /* target_normal_map_for_radiance_calculation: IBLNERF_NORMAL_DEPTH_GRADIENT_EPSILON
                                          (shipped; 4 offset queries per sample, normal_from_depth.py:139-183) or
                                          IBLNERF_NORMAL_GROUND_TRUTH (gt_values["normal"] rows, :370-371; no offset queries) or
                                          IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION_EPSILON (depths along four rays with tilted
                                          directions, normal_from_depth.py:55-100; uses epsilon_direction) or
                                          IBLNERF_NORMAL_INFERRED ("inferred_normal_map", :372-373: the composited output of the
                                          normal_mlp uploaded as IBLNERF_AUX_NORMAL, used as it is; no offset queries) or
                                          IBLNERF_NORMAL_DEPTH_GRADIENT / _DIRECTION ("normal_map_from_depth_gradient",
                                          "..._direction": normal_from_depth.py:102-137 / :16-52, the derivative of the rendered
                                          depth with respect to a shift of the ray origin / a tilt of the ray direction, which the
                                          reference takes by autograd and can therefore only run with gradients enabled; here one
                                          density-gradient query per sample (iblnerf_density_gradient) and the chain rule through
                                          the compositing, no autograd) */
    int32_t color_independent_to_direction; /* 0 (shipped) | 1: networks built with is_color_independent_to_direction
                                          (ibl_nerf.py:192): radiance heads read the trunk output, no feature / view layers */
    int32_t mlp_precision;             /* how the fp32 nn.Linear products are mapped onto the matrix cores (no reference counterpart).
                                          Error per operand / matrix-core slots per 64 MACs of a 32x32 tile:
                                          IBLNERF_MLP_BF16X3      2^-17 / 12  three bf16 products on hi/lo splits, fp32 range
                                          IBLNERF_MLP_F16X3       2^-22 / 12  three f16 products on hi/lo splits: the precise mode —
                                                                  on a checkpoint with surfaces the density head amplifies operand
                                                                  round-off ~100x and only this mode keeps grazing rays at 1e-4
                                          IBLNERF_MLP_F16_MXFP6   2^-16 /  6  one f16 product + two block-scaled fp6 residual products
                                          IBLNERF_MLP_F16X3_MXFP6 F16X3 for every query except the reflected-ray
                                                                  queries, which run F16_MXFP6: they only feed maps that are
                                                                  ill-conditioned in the reference itself (its fp64 and fp32 runs
                                                                  differ by 2e-2 .. 6e-2 there on a checkpoint with surfaces)
                                          IBLNERF_MLP_F16X3_MXFP6X (the Python default) as F16X3_MXFP6, with the fine pass's offset queries on the fast kernel's
                                                                  mixed trunk form: positions_linears.0 and .1 as three f16 products,
                                                                  the other six layers as F16_MXFP6 (the first layers set the density's
                                                                  error on a network with surfaces), and the fine pass's MAIN query
                                                                  on F16_MXFP6 (it places no samples and no difference is taken of it:
                                                                  its 2^-16 reaches the maps unamplified; the worst ray of every direct
                                                                  channel is set by the coarse pass's sample placement)
                                                                  Re-measured on a second fitted checkpoint with sharper density steps: the composited maps
                                                                  hold; the per-sample `weights` output of the fine pass, which sees that query's 2^-16
                                                                  unaveraged, reaches 1.6e-3 at the 99.9th percentile there (3.5e-4 on the first checkpoint) —
                                                                  IBLNERF_ROUTE_FINE_MAIN_PRECISE below is the option for callers who consume it
                                          IBLNERF_MLP_F16X3_MAIN  F16X3 for the queries whose results are direct channels (main query
                                                                  of both passes, auxiliary networks, iblnerf_network_query) and for
                                                                  the coarse grid's offset queries; F16_MXFP6 also for the fine
                                                                  pass's offset queries (47 % of a frame's FLOPs): 16 % faster, the
                                                                  normal's worst ray of 1024 at 1.5e-3 instead of 1.9e-4
                                          IBLNERF_MLP_F16_MIXED   F16_MXFP6 for the coarse main and offset queries, ONE plain f16
                                                                  product (2^-11) for the fine main and reflected queries: fine on
                                                                  random-init fog, 1e-2 on direct channels of a fitted checkpoint
                                          The f16 modes need inputs, weights and activations below 65504 — see iblnerf_range_status (a
                                          network with a weight beyond that runs on the bf16x3 kernel by itself) */
    float epsilon_direction;           /* epsilon_direction_for_numerical_normal (0.005): tilt of the four rays of
                                          IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION_EPSILON */
    int32_t infer_normal_at_surface;   /* 0 | 1: the IBLNERF_AUX_NORMAL network is evaluated once per ray at the surface point
                                          o + d * target_depth instead of at every sample (ibl_nerf_renderer.py:268-271) */
    int32_t query_routing;             /* 0 (default) | a set of IBLNERF_ROUTE_* bits: deviations from the per-query-class table of
                                          mlp_precision above, for A/B measurements and for tests that need one kernel form on its
                                          own.  They change results (another product scheme for a query class), which is why they
                                          are options of the context and not environment variables.  Ignored by modes that do not
                                          keep the stream a bit needs. */
    int32_t persistent_workgroups;     /* 0 (default) = one persistent workgroup per compute unit; n > 0 = launch the MLP kernels
                                          with n workgroups (scaling measurements below the chip's power limit) */
} iblnerf_options;
enum {
    IBLNERF_ROUTE_COARSE_OFFSETS_MIXED = 1,   /* F16X3_MXFP6X: the coarse grid's offset queries on the mixed trunk form too */
    IBLNERF_ROUTE_USER_TRUNK_MIXED = 2,       /* F16X3_MXFP6X: the trunk-only form of iblnerf_network_query on the mixed trunk form */
    IBLNERF_ROUTE_FINE_MAIN_PRECISE = 4,      /* F16X3_MXFP6X: the fine pass's main query back on F16X3 */
    IBLNERF_ROUTE_POINT_BATCH = 8,            /* any mode: the epsilon-offset points go through a [4][n][S][3] batch in HBM (k_make_points) instead of
                                                 being generated in the TRUNK kernels' input stage; same arithmetic, bit-identical results */
    IBLNERF_ROUTE_COARSE_MAIN_22BIT = 16,     /* F16X3_MXFP6X / F16X3_MXFP6 / F16X3_MAIN: the coarse pass's density stays that of its main query's three f16 products (22-23
                                                 bits per operand: round 3's arithmetic) instead of being evaluated on the 15-slot form — three f16 + three
                                                 block-scaled fp6 products per block, operands to ~2^-26 (VAR_TRUNK_P; the default since round 4: the density of
                                                 the coarse pass places the fine samples, and a fitted network's cancelling density sum amplifies a one-ulp
                                                 perturbation of its parameters ~300x) */
    IBLNERF_ROUTE_USER_TRUNK_P = 32,          /* the same modes: the trunk-only form of iblnerf_network_query on the 15-slot form (tests of that kernel on its own) */
    IBLNERF_ROUTE_COARSE_DENSITY_ALL_POINTS = 128, /* the 15-slot density on EVERY coarse sample instead of the relevant ones only (those that are neither clearly
                                                 empty — density estimate below -1: alpha = 0 exactly — nor behind a transmittance of 1e-8); an A/B aid: results
                                                 agree to ~1e-9 on a weight */
    IBLNERF_ROUTE_FINE_OFFSETS_PRECISE = 64,  /* F16X3_MXFP6X: the fine grid's offset queries back on F16X3 (with IBLNERF_ROUTE_FINE_MAIN_PRECISE the mode then
                                                 routes every query as F16X3_MXFP6 does: the "safe" policy of ibl-nerf_amd/renderer.py's load-time calibration) */
    IBLNERF_ROUTE_ESTIMATES_WHOLE = 512,      /* the density estimates of the offset copies and of the fine main query on every sample at once instead of in z-chunks
                                                 (the later chunks only for rays not yet saturated; results are the same bit for bit, per-sample weights to 1e-15) */
    IBLNERF_ROUTE_OFFSETS_ESTIMATE_ALL = 1024, /* round 4's route for the epsilon-offset copies: a density estimate on every sample of every copy, then the copy's own
                                                 selection.  Default since round 5 (iblnerf_route below, api.cpp offsets_on_lists): the samples the MAIN ray of the same
                                                 pass found relevant (its first to its last selected sample, one more on either side) are PREDICTED relevant for the four
                                                 copies and go to the query's own kernel without an estimate; estimates run on the rest only (in front of that range for
                                                 every copy, behind it for the copies that have not saturated by then), followed by the copy's own selection among them.
                                                 Every sample is still either evaluated by the query's kernel or judged on an estimate of its own: the prediction decides
                                                 cost, never a result (against this bit: the same samples and more refined; maps agree to ~1e-7) */
    IBLNERF_ROUTE_COARSE_DENSITY_15SLOT = 4096, /* round 4's density of the coarse pass's relevant samples: the 15-slot form (operands to ~2^-26).  Default since round 5: EXACT
                                                 fp32 on v_mfma_f32_32x32x2_f32 (csrc/trunk_fp32_kernel.hip: fp32 operands, products and accumulation — the arithmetic the
                                                 reference runs; 1/16 of the f16 rate on ~5 of a ray's 64 coarse samples).  These weights place the fine samples: the
                                                 15-slot form's 1e-6 .. 1.2e-5 on a coarse weight moved one ray's normal by 8.6e-3, fp32's 2e-6 does not */
    IBLNERF_ROUTE_NO_OFFSET_TIERS = 8192,     /* F16X3_MXFP6X, fast table: the fine grid's offset copies on the mixed trunk form everywhere (round 4).  Default since round 5:
                                                 the samples a copy's OWN selection adds outside the main ray's relevant range (a silhouette, the fringe of a haze: ~2 % of the
                                                 refined samples, where a copy's depth hangs on one or two of them) on three f16 products — 7 of the 8 normals the fast table
                                                 alone left above 1e-3 on a 65 536-ray launch; and, with iblnerf_set_offset_tier_threshold > 0, the predicted range itself in
                                                 two tiers by the bound T_s dist_s |depth - z_s| of the main ray (api.cpp plan_offsets, k_importance; off by default) */
    IBLNERF_ROUTE_FINE_TIERS = 16384,         /* F16X3_MXFP6X: the TIERED table (round 6) — between the fast table and the safe one (both IBLNERF_ROUTE_FINE_*_PRECISE bits).  The fine pass's
                                                 main query and offset copies stay on the fast forms except where an error would show: of the main query's relevant samples
                                                 those whose own weight (from the density estimates) exceeds 1e-3 — a handful per ray — and of the offset copies' predicted
                                                 range those where T_s dist_s |depth - z_s| of the main ray exceeds 5e-5 go to the three-product f16 kernels (k_importance; the
                                                 bound on what a density error there moves a weight / a copy's depth by).  Measured on 4 096-pixel probes of the checkpoint x
                                                 camera cases that need the safe table: per-sample weights and normals within 1.2e-4 of the safe table's at 99.9 %, no ray
                                                 above 1e-3, for +2..4 % of a frame where the safe table costs +15 % (ibl-nerf_amd/renderer.py decides per call: fast, tiered or safe) */
    IBLNERF_ROUTE_ESTIMATES_6SLOT = 256       /* the density ESTIMATES behind a list refinement (which samples are relevant; the density of those that are not) on the
                                                 f16 + 2 fp6 form (2^-16 per operand) instead of plain f16 (2^-11: 4 matrix slots per 64 MACs instead of 6).  An estimate
                                                 only has to be right to within the selection margin of 1.0 in raw density */
};
/* Changes options.query_routing of an existing context (no reallocation; takes effect with the next call; the caller orders it against work in
 * flight by issuing it between calls on the context's stream).  Bits that need a stream the context's mlp_precision does not keep are ignored as at
 * iblnerf_create.  Lets a caller measure two routings of one mode on the same rays and keep the cheaper one that holds its tolerance. */
int iblnerf_set_query_routing(iblnerf_ctx* ctx, int query_routing);

/* The ROUTE of a checkpoint: which queries run as "estimate everywhere + the query's own kernel on a list of the relevant samples" (csrc/api.cpp full_pass).
 * Whether that pays, and whether plain-f16 estimates are good enough, depend on the networks AND on what the rays see — so they are MEASURED on probe rays the
 * caller chooses (iblnerf_decide_route), and every render call issued while that route stands takes it whatever its size or launch split: nothing is decided
 * inside a render call.  Until a route is decided (after iblnerf_create, after every upload of network 0 / 1, after iblnerf_set_route with decided = 0) every query
 * evaluates all of its samples on its own kernel (round 3's path: correct, ~1.5x slower on a scene with surfaces).  Round 6: ibl-nerf_amd/renderer.py decides PER CALL — on
 * <= 4 096 strided rays of every eager call of at least 1 024 rays, or on the probe rays its caller hands it (dist.py / bench.py: the same seeded pixels of the frame
 * on every rank) — so that what a view is rendered under is a function of that view alone: views of one export do not inherit the first view's route, ranks that are
 * dealt different views or tiles agree, and an N-rank frame is the 1-rank frame bit for bit (round 5 decided once per checkpoint, on the first call's rays).  No reference counterpart: the reference evaluates every sample in fp32 (ibl_nerf_renderer.py:201, :446, normal_from_depth.py:158). */
typedef struct iblnerf_route {
    int32_t decided;                   /* 0: no route yet (see above); the other fields are then meaningless */
    int32_t estimates_plain_f16[2];    /* per network: its density estimates run in plain f16 (4 matrix slots per 64 MACs); 0: on the f16 + 2 fp6 form (6), because the
                                          probe found a plain-f16 estimate half-way to a wrong selection, because the tripwire fired since, or IBLNERF_ROUTE_ESTIMATES_6SLOT */
    int32_t tripped;                   /* the estimate tripwire (every list launch checks the estimates it overwrites — those of the samples it refines and of one in 64
                                          of the samples dropped as clearly empty, the audit — bits 2 / 3 of the range flag) has fired since the decision: 1 = the selection
                                          margins were doubled (an underestimate, margins below 6) or the plain-f16 estimates withdrawn, 2 = it fired on f16 + 2 fp6
                                          estimates too and the lists went off (iblnerf_range_status folds the flag in; the call that raised it must be repeated:
                                          ibl-nerf_amd/renderer.py does) */
    double coarse_share;               /* share of the probe's coarse-grid samples that were relevant (neither clearly empty nor behind saturation); lists are on for
                                          the coarse main query, the coarse grid's offset copies and the reflected rays iff it is <= 0.30.  -1: not measured (lists off) */
    double fine_main_share;            /* likewise the fine main query (on a list iff <= 0.60) */
    double fine_offsets_share;         /* likewise the fine grid's offset copies (on lists iff <= 0.85 with the main ray's prediction; without it 0.42, or 0.55 where the mode
                                          refines them on three f16 products) */
    float select_margin[2];            /* per network: a sample whose density estimate lies below -select_margin is "clearly empty" (alpha = 0 exactly whatever the estimate's
                                          error).  Measured by the probe: twice the deepest underestimate it found (how far below zero the plain-f16 estimate puts a sample
                                          whose f16 + 2 fp6 estimate is positive) + 0.5, rounded up to half a unit, at least 2; a network that would need more than 6 keeps
                                          f16 + 2 fp6 estimates at 2; doubled by a tripwire event */
    float estimate_error[2];           /* ... that deepest underestimate (-1: not measured) */
} iblnerf_route;
/* Measures and freezes the route on n_rays probe rays (device pointers; 1 024 <= n_rays <= options.max_rays_per_launch; scalar planes): one render of them whose
 * outputs are discarded, with three stream synchronisations.  Needs the networks and the LUT uploaded.  *out (nullable) receives the result. */
int iblnerf_decide_route(iblnerf_ctx* ctx, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_, float far_, iblnerf_route* out);
/* Imposes a route (e.g. the one another rank or an earlier run decided); decided = 0 withdraws it. */
int iblnerf_set_route(iblnerf_ctx* ctx, const iblnerf_route* route);
int iblnerf_get_route(iblnerf_ctx* ctx, iblnerf_route* out);
/* The route of `src` onto `dst`, exactly — ladder steps included (a margin iblnerf_escalate_route doubled keeps its plain-f16 estimates; iblnerf_set_route of the
 * struct iblnerf_get_route fills would not).  For two contexts of one checkpoint that render two halves of one call on two HIP streams (ibl-nerf_amd/renderer.py
 * _render_pair: one half's per-ray kernels and launch tails run under the other half's matrix kernels, 3.7 % of a frame): both halves under one decision. */
int iblnerf_copy_route(iblnerf_ctx* dst, const iblnerf_ctx* src);
/* The tripwire's bits in iblnerf_range_status: 4 (bit 2) a refined sample's density is positive and its estimate lay below -margin / 2 — a near miss; 8 (bit 3) an estimate
 * overshot a density beyond what the conservative transmittance allows for; 16 (bit 4, always with 4) ... below -3 margin / 4: a DEEP miss — the margin is twice the
 * deepest underestimate the probe saw, so the route's error model is off by half again; beyond -margin itself (an AUDITED sample: dropped as clearly empty, and not empty) it
 * is proof that samples are being dropped wrongly.
 * A PROBE that tripped (any of them behind iblnerf_decide_route, or behind a render of the probe rays under the route), or a call whose marks say the route does not fit it (bit
 * 4, or marks on more than a handful of its rays: ibl-nerf_amd/renderer.py's alarm): the decided route
 * climbs one step of the ladder — underestimates (no bit 3) double both selection margins (up to 6); otherwise, or beyond that, the plain-f16 estimates go (f16 + 2 fp6
 * from now on); on f16 + 2 fp6 estimates already, the lists go off (route.tripped = 2).  Render the probe again and repeat until it is clean: the route a call's rays
 * are then rendered under is a function of the probe alone.  (Round 5 climbed this ladder inside iblnerf_range_status, for good and per context: a frame's route then
 * depended on the calls — and, under sharding, the rank — that had come before.)  IBLNERF_ERR_STATE without a decided route. */
int iblnerf_escalate_route(iblnerf_ctx* ctx, int trip_bits);
/* enabled = 0: the next render calls evaluate every sample of every query on its own kernel, whatever route is decided (and raise no tripwire: there are no lists);
 * 1 (the default) gives the route back.  How the rays marked in iblnerf_outputs.trip_rays are rendered once more. */
int iblnerf_set_lists(iblnerf_ctx* ctx, int enabled);
/* enabled = 1: a TAPPED call (iblnerf_render_rays_tapped with taps: a training step's forward, train.py:286-297) evaluates its main queries under the decided route as
 * an untapped call does — estimates everywhere, the query's kernel on the relevant samples — instead of on every sample (the default, 0: the taps then hold every raw
 * row).  The tapped raw rows of the samples NOT on a list are (density estimate, zeros): a clearly empty sample (alpha = 0 exactly, a dead ReLU in front of its
 * density: every gradient through it is exactly zero) or one behind a transmittance of 1e-8 (gradients below 1e-8 of the ray's).  For callers whose backward reads
 * the live rows only (ibl-nerf_amd/training.py network_backward_live).  Without a decided route nothing changes. */
int iblnerf_set_tapped_lists(iblnerf_ctx* ctx, int enabled);
/* The route as text: one line per (pass, query class) = which kernel estimates it (or none), in which z-chunks, and which kernel evaluates the list / the whole batch.
 * Writes at most n bytes including the terminating 0; returns the length the full text needs (snprintf's convention), < 0 on error. */
int iblnerf_describe_route(iblnerf_ctx* ctx, char* buf, size_t n);
enum { IBLNERF_MLP_BF16X3 = 0, IBLNERF_MLP_F16_MXFP6 = 1, IBLNERF_MLP_F16_MIXED = 2, IBLNERF_MLP_F16X3 = 3, IBLNERF_MLP_F16X3_MXFP6 = 4, IBLNERF_MLP_F16X3_MAIN = 5, IBLNERF_MLP_F16X3_MXFP6X = 6 };
enum { IBLNERF_NORMAL_DEPTH_GRADIENT_EPSILON = 0, IBLNERF_NORMAL_GROUND_TRUTH = 1, IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION_EPSILON = 2,
       IBLNERF_NORMAL_INFERRED = 3, IBLNERF_NORMAL_DEPTH_GRADIENT = 4, IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION = 5 };

void iblnerf_default_options(iblnerf_options* o);

/* replaces: create_IBLNeRF (nerf_models/ibl_nerf.py:255-428) — model construction only */
int iblnerf_create(const iblnerf_options* opts, iblnerf_ctx** out_ctx);
void iblnerf_destroy(iblnerf_ctx* ctx);
const char* iblnerf_last_error(const iblnerf_ctx* ctx);   /* ctx may be NULL: last create() error */

/* Number of floats in one network's state-dict blob (798 994). */
size_t iblnerf_blob_floats(void);

/* replaces: model.load_state_dict(ckpt[...]) (nerf_models/ibl_nerf.py:361, :375).
 * which: 0 = network_fn (coarse), 1 = network_fine.  h_blob = every tensor of IBLNeRF.state_dict()
 * in registration order, weight [out,in] row-major then bias.  Weights are copied (re-upload after
 * an optimizer step).  Synchronous. */
int iblnerf_upload_weights(iblnerf_ctx* ctx, int which, const float* h_blob, size_t n_floats);
/* Round 6: a network OUTSIDE the built architecture — IBLNeRF(D = netdepth, W = netwidth, input_ch = 3 + 6 multires, input_ch_views = 3 + 6 multires_views, skips = [4],
 * coarse_radiance_number = 3) with 1 <= netdepth <= 32 (not 5: the reference's own forward fails there), even 2 <= netwidth <= 4096, 0 <= multires, multires_views <= 24;
 * src/nerf_models/ibl_nerf.py:14-60 and config_parser.py's --netdepth / --netwidth / --multires / --multires_views accept any.  h_blob = the state dict's tensors in
 * registration order, [out, in] row-major weight then bias per layer (ibl-nerf_amd/checkpoint.py arch_schema / arch_blob).  Such a network is evaluated by
 * csrc/generic_mlp.hip: layer by layer in exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32), activations in HBM, every sample of every query — no lists, no
 * estimates (the context holds no route), no backward (the gradient entry points return IBLNERF_ERR_STATE for this slot), not for colour-independent contexts;
 * ~1/20 of the fused kernels' rate at the built width.  Networks INSIDE the built architecture (smaller ones) are embedded exactly by the caller and uploaded with
 * iblnerf_upload_weights.  A later iblnerf_upload_weights for the same slot takes the slot back to the fused kernels. */
int iblnerf_upload_weights_arch(iblnerf_ctx* ctx, int which, const float* h_blob, size_t n_floats, int netdepth, int netwidth, int multires, int multires_views);

/* Same, with the blob already in device memory (e.g. torch.cat of the module's parameters): packed by a kernel
 * enqueued on `stream`, no host copy and no synchronisation — the cheap way to follow an optimizer step
 * (train.py:479-481).  Renders enqueued later on the same stream see the new weights; the caller orders it after
 * earlier renders (same stream, or an event).  With IBLNERF_MLP_F16_MXFP6 a weight outside the f16 range raises the
 * iblnerf_range_status flag instead of switching kernels. */
int iblnerf_upload_weights_device(iblnerf_ctx* ctx, void* stream, int which, const float* d_blob, size_t n_floats);

/* replaces: albedo_mlp / roughness_mlp / irradiance_mlp.load_state_dict (nerf_models/ibl_nerf.py:312-326, :369-374) and their
 * use in raw2outputs (ibl_nerf_renderer.py:291-303): a PositionMLP (src/networks/MLP.py:6-30) has the main network's trunk
 * shape plus out_linears [out_ch, 256]; its per-sample outputs replace the main network's albedo / roughness / irradiance
 * samples before compositing (an irradiance_mlp's always through sigmoid).  One upload per output channel: h_blob is an
 * IBLNeRF-schema blob whose positions_linears.* are the auxiliary network's and whose sigma_linear.{weight,bias} is row
 * `channel` of its out_linears (the other tensors are not read).  kind: IBLNERF_AUX_ALBEDO (channels 0..2),
 * IBLNERF_AUX_ROUGHNESS, IBLNERF_AUX_IRRADIANCE (channel 0), IBLNERF_AUX_NORMAL (channels 0..2: the normal_mlp of infer_normal,
 * ibl_nerf.py:307-310; its samples 2 sigmoid(.) - 1 are composited into maps.inferred_normal_map, ibl_nerf_renderer.py:267-276,
 * or, with options.infer_normal_at_surface, evaluated at the surface point).  The network takes effect in iblnerf_render_rays once all its
 * channels are uploaded, for both passes, until iblnerf_clear_aux.  Cost: one trunk evaluation per sample and channel. */
enum { IBLNERF_AUX_ALBEDO = 0, IBLNERF_AUX_ROUGHNESS = 1, IBLNERF_AUX_IRRADIANCE = 2, IBLNERF_AUX_NORMAL = 3 };
/* An auxiliary network on its own — network_query_fn(pts, None, albedo_mlp | roughness_mlp | irradiance_mlp | normal_mlp) (ibl_nerf_renderer.py:267-303) and, for a
 * training step (the reference registers these networks' parameters with the optimizer, ibl_nerf.py:305-323), its backward.
 *   iblnerf_aux_query     d_pts [n_pts, 3] -> d_out [n_pts, channels] raw outputs (before the sigmoid), channels = 3 (albedo, normal) or 1.
 *   iblnerf_aux_backward  dL/d(raw output `channel`) d_dout [n_pts] -> d_out [n_pts, 4] = (the channel's raw output, dL/dx, dL/dy, dL/dz) and d_grad
 *                         [iblnerf_blob_floats()] in the IBLNeRF blob layout an auxiliary channel is uploaded in: positions_linears.0-7 hold THIS CHANNEL's
 *                         contribution to the shared trunk's gradient (sum over the channels), sigma_linear holds row `channel` of out_linears.  grad_scale: as
 *                         iblnerf_trunk_backward. */
int iblnerf_aux_query(iblnerf_ctx* ctx, void* stream, int kind, const float* d_pts, int64_t n_pts, float* d_out);
int iblnerf_aux_backward(iblnerf_ctx* ctx, void* stream, int kind, int channel, const float* d_pts, int64_t n_pts, const float* d_dout,
                         float grad_scale, float* d_out, float* d_grad);
int iblnerf_upload_aux_weights(iblnerf_ctx* ctx, int kind, int channel, const float* h_blob, size_t n_floats);
int iblnerf_clear_aux(iblnerf_ctx* ctx, int kind);

/* replaces: depth_mlp.load_state_dict(ckpt['depth_mlp']) (nerf_models/ibl_nerf.py:293-297, :365-366) and its use under infer_depth
 * (ibl_nerf_renderer.py:722-726): a PositionDirectionMLP (src/networks/MLP.py:32-74: the trunk, feature_linear, four view layers of
 * width 128, final_linear) evaluated once per ray at the ray origin with the normalised ray direction; relu of its first output is
 * outputs.inferred_depth_map.  h_blob = every tensor of its state_dict() in registration order, weight [out,in] row-major then bias
 * (iblnerf_posdir_floats(out_ch) floats).  Takes effect in iblnerf_render_rays until iblnerf_clear_posdir_mlp.  (The reference also
 * builds a visibility_mlp of this class under infer_visibility, ibl_nerf.py:299-304, and never evaluates it: nothing to upload.) */
size_t iblnerf_posdir_floats(int out_ch);
int iblnerf_upload_posdir_mlp(iblnerf_ctx* ctx, const float* h_blob, size_t n_floats, int out_ch);
int iblnerf_clear_posdir_mlp(iblnerf_ctx* ctx);
/* replaces: network_query_fn(inputs, viewdirs, depth_mlp) (ibl_nerf.py:327-329) for the uploaded PositionDirectionMLP:
 * d_pts [n,3], d_viewdirs [n,3] (one direction per point, used as given) -> d_out [n,out_ch] (no activation). */
int iblnerf_posdir_query(iblnerf_ctx* ctx, void* stream, const float* d_pts, const float* d_viewdirs, int64_t n, float* d_out);

/* replaces: brdf_lut tensor of test.py:79-87.  h_rgb = float [3,512,512] (R = scale, G = bias). */
int iblnerf_upload_lut(iblnerf_ctx* ctx, const float* h_rgb);

/* Host-only helper (no GPU needed): packs a blob into the device weight-stream format
 * (csrc/layout.h).  Used by the CPU tests to check the MFMA fragment layout arithmetic. */
int iblnerf_pack_weights_host(const float* h_blob, size_t n_floats, void* h_stream, size_t stream_bytes,
                              float* h_tables, size_t table_floats);
size_t iblnerf_stream_bytes(void);
size_t iblnerf_table_floats(void);
/* Host-only: the encoder's sin/cos (csrc/sincos_enc.h) for x * 2^k, k = 0..n_freq-1, written as
 * [sin_0, cos_0, sin_1, cos_1, ...]. */
void iblnerf_encode_host(float x, int n_freq, float* h_out);

/* replaces: get_rays(H, W, K, c2w) (nerf_models/nerf_renderer_helper.py:36-45) for image rows
 * [row0, row0+n_rows).  h_K = 3x3 row-major, h_c2w = 3x4 row-major.  Outputs [n_rows*W, 3]. */
int iblnerf_get_rays(iblnerf_ctx* ctx, void* stream, int H, int W, const float* h_K, const float* h_c2w,
                     int row0, int n_rows, float* d_rays_o, float* d_rays_d);
/* The same rays for image rows row0, row0 + row_step, ... (n_rows of them; outputs [n_rows * W, 3]): the interleaved row tile of one rank of a sharded frame
 * (ibl-nerf_amd/dist.py: rank r of N takes row0 = r, row_step = N), generated by that rank alone.  Every pixel's ray is computed by itself, so a tile's rays are the
 * frame's bit for bit. */
int iblnerf_get_rays_strided(iblnerf_ctx* ctx, void* stream, int H, int W, const float* h_K, const float* h_c2w,
                             int row0, int row_step, int n_rows, float* d_rays_o, float* d_rays_d);
/* ... and for a list of pixels (d_pixels [n_pixels] int64 in device memory, flat indices row * W + col; outputs [n_pixels, 3]): the probe rays a frame's route and
 * precision table are measured on — the same seeded pixels on every rank (ibl-nerf_amd/dist.py frame_probe). */
int iblnerf_get_rays_pixels(iblnerf_ctx* ctx, void* stream, int H, int W, const float* h_K, const float* h_c2w,
                            const int64_t* d_pixels, int64_t n_pixels, float* d_rays_o, float* d_rays_d);

/* replaces: network_query_fn(inputs, viewdirs, network_fn) (nerf_models/ibl_nerf.py:327-329 ->
 * run_network :236-252).  d_pts [n_rays, n_samples, 3]; d_viewdirs [n_rays,3] or NULL.
 * d_out [n_rays, n_samples, 18], or [n_rays, n_samples, 1] when d_viewdirs is NULL (sigma only,
 * IBLNeRF.forward early return :175-176). */
int iblnerf_network_query(iblnerf_ctx* ctx, void* stream, int which, const float* d_pts, int64_t n_rays,
                          int n_samples, const float* d_viewdirs, float* d_out);

/* replaces: torch.autograd through network_query_fn(pts, None, network_fn) for d raw[..., 0] / d pts — what
 * get_normal_from_depth_gradient / _direction (nerf_models/normal_from_depth.py:102-137, :16-52) obtain from
 * depth_map.backward(), and the dgrad half of a training step's backward through the trunk (train.py:479-481).
 * One fused launch: the trunk forward keeping every ReLU's pass bits, then the chain dZ(l-1) = (W(l)^T dZ(l)) * bits(l-1) on a
 * transposed weight stream, the skip layer's encoding columns, and the derivative of the positional encoding.
 * d_pts [n_pts, 3] -> d_out [n_pts, 4] = (sigma, d sigma / d x, d sigma / d y, d sigma / d z). */
int iblnerf_density_gradient(iblnerf_ctx* ctx, void* stream, int which, const float* d_pts, int64_t n_pts, float* d_out);
/* The raw density (sigma_linear of the trunk, ibl_nerf.py:154-176, :200) of network `which` at n_pts points in EXACT fp32: fp32 operands, products and accumulation on
 * v_mfma_f32_32x32x2_f32 (csrc/trunk_fp32_kernel.hip; sin / cos of the encoding in double, rounded once) — the arithmetic of the reference's own nn.Linear chain, which
 * render_rays runs on the coarse pass's relevant samples (the densities that place the fine samples).  d_pts [n_pts, 3] -> d_out [n_pts].  Not for IBLNERF_MLP_BF16X3
 * contexts (they keep no fp32 copy of the state dict).  1/16 of the f16 matrix rate: 98 ns per point. */
int iblnerf_trunk_density_fp32(iblnerf_ctx* ctx, void* stream, int which, const float* d_pts, int64_t n_pts, float* d_out);

/* replaces: loss.backward() through the trunk-only query network_query_fn(pts, None, fn) of a training step (train.py:479-481;
 * ibl_nerf.py:236-252, 154-176): given dL / d sigma per point, the gradient with respect to the points AND to the trunk's
 * parameters (positions_linears.0-7 and sigma_linear, weights and biases).
 * d_pts [n_pts, 3], d_dsigma [n_pts]  ->  d_out [n_pts, 4] = (sigma, dL/dx, dL/dy, dL/dz);
 * d_grad: iblnerf_blob_floats() floats in the state-dict layout of iblnerf_upload_weights (zeroed here; entries of the other layers stay 0).
 * Two stages: the fused forward + backward chain of iblnerf_density_gradient, which also stashes every layer's input and dZ
 * fragments (workspace grown on demand, 15.2 KiB per point — the stash is laid out for the whole-network backward, this entry fills 8.1 KiB of it —, see iblnerf_trim), then the weight-gradient GEMMs over the points (f16 operands,
 * fp32 accumulation; the operand transposition is done by the matrix core).  Needs an mlp_precision with the f16x3 stream.
 * grad_scale: a power of two applied to dL/dsigma inside the kernels and taken out of every result — the loss scaling of f16
 * training: gradients are stashed as f16, so small ones must be lifted out of the denormals (|dZ| < 6e-5) without the largest
 * reaching 65504; an overflow raises the range flag (iblnerf_range_status) and the results are then invalid: repeat with a
 * smaller scale (ibl-nerf_amd/renderer.py: trunk_backward does that).  The range entries report WHICH: *out_of_range bit 0 = a forward
 * activation left the range (the bf16x3 answer), bit 1 = only the gradients of a backward did (the smaller-scale answer). */
int iblnerf_trunk_backward(iblnerf_ctx* ctx, void* stream, int which, const float* d_pts, int64_t n_pts, const float* d_dsigma,
                           float grad_scale, float* d_out, float* d_grad);

/* The same split one layer earlier, for callers that keep the head layers (a training step's main query, whose heads run in the
 * caller's autograd): iblnerf_trunk_features = positions_linears.0-7 of IBLNeRF.forward (ibl_nerf.py:160-170) -> d_h7 [n_pts, 256],
 * the post-ReLU trunk features every head reads; iblnerf_trunk_features_backward takes dL/dh7 [n_pts, 256] and returns d_out
 * [n_pts, 4] = (sigma, dL/dpts) and the gradients of positions_linears.0-7 (sigma_linear's stay 0: that layer is the caller's). */
int iblnerf_trunk_features(iblnerf_ctx* ctx, void* stream, int which, const float* d_pts, int64_t n_pts, float* d_h7);
int iblnerf_trunk_features_backward(iblnerf_ctx* ctx, void* stream, int which, const float* d_pts, int64_t n_pts, const float* d_dh7,
                                    float grad_scale, float* d_out, float* d_grad);

/* ... and one layer pair further (not for colour-independent networks): iblnerf_trunk_features2 also evaluates feature_linear and
 * views_linears.0 (ibl_nerf.py:193-197; the direction encoding of d_viewdirs [n_rays, 3], expanded over the n_samples points of a ray as
 * run_network does) and returns both operands of the remaining heads: d_h7 [n, 256] and d_h2 = relu(views_linears.0(...)) [n, 256],
 * n = n_rays * n_samples.  Its backward takes dL/dh7 (the part that does NOT flow through feature_linear) and dL/dh2 and adds
 * feature_linear's and views_linears.0's gradients to d_grad (79 % of the network's FLOPs differentiated on the fused kernels). */
int iblnerf_trunk_features2(iblnerf_ctx* ctx, void* stream, int which, const float* d_pts, int64_t n_rays, int n_samples,
                            const float* d_viewdirs, float* d_h7, float* d_h2);
int iblnerf_trunk_features2_backward(iblnerf_ctx* ctx, void* stream, int which, const float* d_pts, int64_t n_rays, int n_samples,
                                     const float* d_viewdirs, const float* d_dh7, const float* d_dh2, float grad_scale, float* d_out,
                                     float* d_grad);

/* replaces: loss.backward() through a training step's main query network_query_fn(pts, viewdirs, fn) (train.py:479-481; ibl_nerf.py:236-252,
 * 154-210): the WHOLE network.  d_draw [n, 18] = dL / d raw (n = n_rays * n_samples, channel order of IBLNeRF.forward) -> d_out [n, 4] =
 * (sigma, dL/dpts) and d_grad = the gradients of all 23 layers in state-dict layout.  One fused launch (forward with the ReLU pass bits and
 * an operand stash, then the backward chain: the 128-wide feature layers' gradients are formed from their heads' weights, the transposed
 * layers that meet in dL/dh2 and dL/dh7 are packed K-concatenated, the N = 1 heads on those activations enter as rank-1 terms) + the
 * weight-gradient GEMMs and head reductions.  Not for colour-independent networks.  15 KiB of workspace per point (see iblnerf_trim). */
int iblnerf_network_backward(iblnerf_ctx* ctx, void* stream, int which, const float* d_pts, int64_t n_rays, int n_samples,
                             const float* d_viewdirs, const float* d_draw, float grad_scale, float* d_out, float* d_grad);

/* Frees the workspace the fused backward entry points grow on demand (operand stash: 15.2 KiB per point of the largest piece, at most
 * 262 144 points = 4 GiB — longer calls are walked in pieces —, and the weight-gradient partial sums).  Synchronises the device.  The
 * memory is a raw hipMalloc outside torch's caching allocator: call this before a phase that needs it back (e.g. after training, before
 * a full-frame render in the same process). */
int iblnerf_trim(iblnerf_ctx* ctx);

/* replaces: the compositing of raw2outputs WITH its autograd, for a training step (ibl_nerf_renderer.py:203-206, 241-259, 281-318): the 19
 * direct maps of one pass from its raw rows —  d_maps [n_rays, 19] = [depth, acc, albedo(3), roughness, irradiance, radiance(3),
 * radiance_1..3 (9)], d_weights [n_rays, S] (optional) — and the backward: dL/d maps [n_rays, 19] (+ optionally dL/d weights) -> dL/d raw
 * [n_rays, S, 18], which is what iblnerf_network_backward takes.  One wavefront per ray, transmittance and the suffix sums of the backward as
 * double-precision wave scans.  (The shading between these maps and color_map stays with the caller: ray-sized tensors.) */
int iblnerf_composite_direct(iblnerf_ctx* ctx, void* stream, const float* d_raw, const float* d_z, const float* d_rays_d, int64_t n_rays,
                             int n_samples, float* d_maps, float* d_weights);
int iblnerf_composite_direct_backward(iblnerf_ctx* ctx, void* stream, const float* d_raw, const float* d_z, const float* d_rays_d,
                                      int64_t n_rays, int n_samples, const float* d_dmaps, const float* d_dweights, float* d_draw);
/* The backward follows the reference's stop-gradients: albedo, roughness, irradiance and the three coarse radiances are composited with
 * `weights_detached` (ibl_nerf_renderer.py:246, :282-315) — their gradient reaches their own raw channels only —, while depth, acc and
 * radiance_map (`weights`, :249, :259, :306) and a loss on the weights also reach the density.  iblnerf_composite_direct_backward_full is the
 * same backward WITHOUT the stop-gradients (every map differentiated through the weights): not what the reference computes; kept for
 * callers that build other losses. */
int iblnerf_composite_direct_backward_full(iblnerf_ctx* ctx, void* stream, const float* d_raw, const float* d_z, const float* d_rays_d,
                                           int64_t n_rays, int n_samples, const float* d_dmaps, const float* d_dweights, float* d_draw);

/* The stages of render_rays between its network queries, on their own — what a training step needs around the fused network forward /
 * backward so that nothing of size [n_rays, n_samples] is computed outside this library (train.py:285-297 -> ibl_nerf_renderer.py:629-732):
 *   iblnerf_coarse_z       z_vals of the coarse pass (:670-692): the near..far grid (lindisp as the context's option), per ray; with
 *                          d_t_rand [n_rays, N_samples] the stratified jitter of perturb > 0.  d_z [n_rays, N_samples].
 *   iblnerf_sample_points  pts = rays_o + rays_d * z (:200), d_z [n_rays, n_samples] -> d_pts [n_rays, n_samples, 3], rounded as the
 *                          reference's separate multiply and add.
 *   iblnerf_fine_z         z_vals_mid -> sample_pdf(weights[1:-1]) (det = no d_u) -> sort(cat(z, z_samples)) and z_std (:700-707, :718):
 *                          d_z_coarse [n_rays, N_samples], d_weights_coarse [n_rays, N_samples], d_u [n_rays, N_importance] or NULL ->
 *                          d_z_fine [n_rays, N_samples + N_importance], d_z_std [n_rays] (may be NULL).
 *   iblnerf_composite_sigma  raw2outputs_depth (:118-152, is_depth_only): d_sigma [n_rays, n_samples] (a trunk-only query's output) ->
 *                          d_weights [n_rays, n_samples], d_depth [n_rays], d_visibility [n_rays] (the full transmittance product). */
int iblnerf_coarse_z(iblnerf_ctx* ctx, void* stream, float near_, float far_, const float* d_t_rand, int64_t n_rays, float* d_z);
/* ... with one near / far plane per ray (render_decomp's `near` / `far` as [n, 1] tensors, ibl_nerf_renderer.py:802-805 -> :668-674): d_near, d_far [n_rays]. */
int iblnerf_coarse_z_rays(iblnerf_ctx* ctx, void* stream, const float* d_near, const float* d_far, const float* d_t_rand, int64_t n_rays, float* d_z);
int iblnerf_sample_points(iblnerf_ctx* ctx, void* stream, const float* d_rays_o, const float* d_rays_d, const float* d_z, int64_t n_rays,
                          int n_samples, float* d_pts);
int iblnerf_fine_z(iblnerf_ctx* ctx, void* stream, const float* d_z_coarse, const float* d_weights_coarse, int64_t n_rays, const float* d_u,
                   float* d_z_fine, float* d_z_std);
int iblnerf_composite_sigma(iblnerf_ctx* ctx, void* stream, const float* d_sigma, const float* d_z, const float* d_rays_d, int64_t n_rays,
                            int n_samples, float* d_weights, float* d_depth, float* d_visibility);

/* replaces: sample_pdf(bins, weights, N_samples, det=True) (nerf_models/nerf_renderer_helper.py:91-134).
 * d_bins [n_rays, n_bins], d_weights [n_rays, n_bins-1] -> d_samples [n_rays, n_out]. */
int iblnerf_sample_pdf(iblnerf_ctx* ctx, void* stream, const float* d_bins, const float* d_weights,
                       int64_t n_rays, int n_bins, int n_out, float* d_samples);

/* gt_values rows + edit/insert kwargs of test.py:115-139 consumed by raw2outputs
 * (nerf_models/ibl_nerf_renderer.py:218-238, :253-256, :378-410).  All image pointers are device
 * pointers with one row per ray. */
typedef struct {
    int32_t mode;                      /* 0 none, 1 edit_intrinsic, 2 insert_object */
    int32_t num_objects;               /* num_edit_objects / num_insert_objects (<= 8) */
    int32_t edit_depth, edit_normal, edit_albedo, edit_albedo_by_img, edit_roughness;
    int32_t n_roughness_list;          /* len(editing_target_roughness_list) */
    const float* d_mask;               /* edit_intrinsic_mask / object_insert_mask [n,3] */
    const float* d_depth;              /* edit_depth / object_insert_depth [n,1] */
    const float* d_normal;             /* edit_normal / object_insert_normal [n,3] in [0,1] */
    const float* d_albedo;             /* edit_albedo [n,3] */
    float roughness_list[8];           /* editing_/inserting_target_roughness_list */
    float albedo_list[24];             /* editing_/inserting_target_albedo_list */
    float irradiance_list[8];          /* inserting_target_irradiance_list */
    const float* d_gt_normal;          /* gt_values["normal"] [n,3] in [0,1]; required when options.normal_mode is
                                          IBLNERF_NORMAL_GROUND_TRUTH (mode may then be 0), ignored otherwise */
    /* calculate_albedo_from_gt / calculate_roughness_from_gt / calculate_irradiance_from_gt / depth_map_from_ground_truth
     * (ibl_nerf_renderer.py:320-330, :251-252): non-NULL = shade with these rows instead of the network's maps (the edit /
     * insert overrides then act on them).  With d_gt_irradiance the irradiance_map buffers are [n,3]. */
    const float* d_gt_albedo;          /* gt_values["albedo"] [n,3] */
    const float* d_gt_roughness;       /* gt_values["roughness"][..., 0] [n] */
    const float* d_gt_irradiance;      /* gt_values["irradiance"] [n,3] */
    const float* d_gt_depth;           /* gt_values["depth"][..., 0] [n] */
    /* edit_roughness_by_img (ibl_nerf_renderer.py:394-395): target_roughness_map[mask_all] = gt_values["edit_roughness"][mask_all][0] — every masked
     * pixel of a CHUNK takes the first masked row of that chunk (the reference applies raw2outputs per `chunk` rays, so this one flag makes its
     * result depend on the chunk size).  The value per ray is ray-sized host-side bookkeeping: the caller resolves it (ibl-nerf_amd/renderer.py
     * follows the reference's chunking) and hands over one roughness per ray; masked rays take their row.  Needs mode 1 and edit_roughness. */
    int32_t edit_roughness_by_img;
    const float* d_roughness;          /* [n]: the roughness every masked ray takes (read at masked rays only) */
} iblnerf_overrides;

/* The 22 non-None maps raw2outputs returns per pass (ibl_nerf_renderer.py:494-525).  Device
 * pointers, caller-allocated, fp32, one row per ray; NULL = not wanted. */
typedef struct {
    float* color_map;                       /* [n,3] */
    float* radiance_map;                    /* [n,3] */
    float* radiance_map_k[3];               /* radiance_map_1..3 [n,3] */
    float* reflected_coarse_radiance_map_k[3]; /* [n,3] */
    float* irradiance_map;                  /* [n,1] */
    float* reflected_radiance_map;          /* [n,3] */
    float* prefiltered_reflected_map;       /* [n,3] */
    float* albedo_map;                      /* [n,3] */
    float* roughness_map;                   /* [n] */
    float* specular_map;                    /* [n,3] */
    float* diffuse_map;                     /* [n,3] */
    float* n_dot_v_map;                     /* [n] */
    float* target_normal_map;               /* [n,3] */
    float* disp_map;                        /* [n] */
    float* acc_map;                         /* [n] */
    float* depth_map;                       /* [n] */
    float* target_depth_map;                /* [n] */
    float* weights;                         /* [n, S]  (S = n_samples coarse, n_samples+n_importance fine) */
    float* inferred_normal_map;             /* [n,3]; written only while an IBLNERF_AUX_NORMAL network is loaded (infer_normal) */
} iblnerf_maps;

typedef struct {
    iblnerf_maps fine;      /* un-suffixed keys (the only pass when n_importance == 0) */
    iblnerf_maps coarse;    /* "...0" keys; ignored unless options.coarse_outputs and n_importance > 0 */
    float* z_std;           /* [n] (n_importance > 0) */
    float* inferred_depth_map; /* [n]; written while a PositionDirectionMLP is uploaded (infer_depth, ibl_nerf_renderer.py:722-726); may be NULL */
    unsigned char* trip_rays;  /* [n] bytes, ZEROED BY THE CALLER, may be NULL.  The estimate tripwire (iblnerf_route below) sets byte r when a list launch of the call
                                  refined a positive density on ray r (or on one of its epsilon-offset copies, or on its reflected ray) whose estimate lay half-way to
                                  dropping it — bits 2 / 3 of iblnerf_range_status.  Such a ray's other dropped samples rest on estimates that thin: render these rays
                                  once more with iblnerf_set_lists(ctx, 0) (every sample evaluated) and overwrite their rows — ibl-nerf_amd/renderer.py does.  The mark
                                  depends on the ray alone (its own samples, its own estimates), so a ray's final result does not depend on the call, launch or rank
                                  it was rendered in: the guarantee of the reference's batchify_rays (ibl_nerf_renderer.py:735-756, :768-769) */
} iblnerf_outputs;

/* The ray-sized part of raw2outputs, differentiated (a training step's backward between the loss and iblnerf_composite_direct_backward;
 * replaces autograd through nerf_models/ibl_nerf_renderer.py:258 disparity, :412-474 LUT fetch / Fresnel / mip interpolation / diffuse +
 * specular, :477-527 tone map + gamma of every output — about 160 ray-sized launches per pass in torch — by one launch).
 * d_maps [n_rays, 19]: the pass's linear direct maps (iblnerf_composite_direct's slots).  d_n_dot_v [n_rays] and d_env [n_rays, 4, 3] (the
 * LINEAR reflected-ray maps: radiance, coarse radiances 1..3): the pass's quantities the reference computes under no_grad (:442-448) or
 * detaches (the normal: no_grad queries; depth inside the mip level: :457) — both NULL for approximate_radiance=False, where only the
 * direct maps' output functions are differentiated.  depth0 = (near + far) / 2 (:456).  d_upstream: dL/d(output map) per map in the
 * layout of iblnerf_maps, NULL = no gradient; read: color_map, radiance_map, radiance_map_k, irradiance_map ([n,1]), albedo_map,
 * roughness_map, specular_map, diffuse_map, prefiltered_reflected_map, disp_map, acc_map, depth_map, target_depth_map (the others carry no
 * gradient in the reference: target_normal_map, n_dot_v_map, the reflected maps).  d_dmaps [n_rays, 19] out.  Uses the context's LUT and
 * its gamma_correct / use_radiance_linear / lut_coefficient_f0 / correct_depth_for_prefiltered_radiance options. */
int iblnerf_ray_outputs_backward(iblnerf_ctx* ctx, void* stream, const float* d_maps, const float* d_n_dot_v, const float* d_env, float depth0,
                                 const iblnerf_maps* d_upstream, int64_t n_rays, float* d_dmaps);
/* ... under calculate_albedo_from_gt / calculate_roughness_from_gt / calculate_irradiance_from_gt / depth_map_from_ground_truth (:251-252, :320-330): the
 * d_gt_albedo [n,3] / d_gt_roughness [n] / d_gt_irradiance [n,3] / d_gt_depth [n] rows of `overrides` (NULL each = flag off) are the target maps the shading
 * reads and the output maps of the same name — constants of the backward: no gradient reaches the network's own map through them (roughness_map itself still
 * sets the mip level, :457-460).  overrides->mode must be 0 (edit / insert in a gradient-carrying render: IBLNERF_ERR_STATE); overrides = NULL is the entry above. */
int iblnerf_ray_outputs_backward_gt(iblnerf_ctx* ctx, void* stream, const float* d_maps, const float* d_n_dot_v, const float* d_env, float depth0,
                                    const iblnerf_maps* d_upstream, const iblnerf_overrides* overrides, int64_t n_rays, float* d_dmaps);
/* ... with per-ray near / far planes: the mip level's depth_0 = (near + far) / 2 is then a value per ray (:455-457, `depth_0[..., 0]`): d_depth0 [n_rays]
 * (every entry positive; read only under correct_depth_for_prefiltered_radiance with d_n_dot_v given). */
int iblnerf_ray_outputs_backward_rays(iblnerf_ctx* ctx, void* stream, const float* d_maps, const float* d_n_dot_v, const float* d_env, const float* d_depth0,
                                      const iblnerf_maps* d_upstream, const iblnerf_overrides* overrides, int64_t n_rays, float* d_dmaps);
/* ... under use_gradient_for_incident_radiance (:442-453: the reflected-ray query runs WITH gradients): also d_denv [n_rays, 4, 3] out = dL/d(the linear reflected-ray
 * maps d_env) through the mip interpolation (:461-467) — what the caller carries on through raw2outputs_simple (iblnerf_composite_direct_backward_full on the
 * reflected rays' rows: every map on the live weights, :38-66) and the reflected query's network (iblnerf_network_backward).  depth0: scalar, or d_depth0 [n_rays]. */
int iblnerf_ray_outputs_backward_env(iblnerf_ctx* ctx, void* stream, const float* d_maps, const float* d_n_dot_v, const float* d_env, float depth0, const float* d_depth0,
                                     const iblnerf_maps* d_upstream, const iblnerf_overrides* overrides, int64_t n_rays, float* d_dmaps, float* d_denv);


/* replaces: batchify_rays -> render_rays -> raw2outputs (nerf_models/ibl_nerf_renderer.py:735-756,
 * :629-732, :153-527) with approximate_radiance=True, perturb=0, raw_noise_std=0.
 * d_rays_o / d_rays_d [n_rays,3] (rays_d NOT normalised, as get_rays returns it).  Any n_rays: the
 * library walks it in workspace-sized launches (chunking never changes results, :768-769). */
int iblnerf_render_rays(iblnerf_ctx* ctx, void* stream, const float* d_rays_o, const float* d_rays_d,
                        int64_t n_rays, float near_, float far_, const iblnerf_overrides* overrides,
                        const iblnerf_outputs* outputs);

/* iblnerf_decide_route with the probe's render KEPT: outs (nullable: then exactly iblnerf_decide_route) receives the maps of the probe rays, rendered under the table in
 * effect while the route is being decided (overrides as iblnerf_render_rays takes them, nullable) — what ibl-nerf_amd/renderer.py's per-call table decision uses as its
 * FAST render, so that a call's decisions cost one probe render less.  outs->trip_rays is ignored (a probe that trips escalates its route). */
int iblnerf_decide_route_outputs(iblnerf_ctx* ctx, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_, float far_,
                                 const iblnerf_overrides* overrides, const iblnerf_outputs* outs, iblnerf_route* out);

/* Training-time sampling (perturb > 0; nerf_models/ibl_nerf_renderer.py:678-692, nerf_renderer_helper.py:98-113): the caller supplies
 * the uniform [0,1) draws, so that any generator — torch's on the device, or numpy's seeded stream of the reference's `pytest` path —
 * gives the reference's samples.  d_t_rand [n_rays, N_samples]: stratified jitter of the coarse grid (which then is per ray, also
 * for the reflected-ray samples: z_vals_constant); d_u [n_rays, N_importance]: the draws sample_pdf(det=False) inverts.  Both or
 * neither (det = (perturb == 0), :703); NULL struct = iblnerf_render_rays.
 * raw_noise_std > 0 (:208-216, :242; 0 in every shipped config): d_noise_coarse [n_rays, N_samples] / d_noise_fine [n_rays, N_samples +
 * N_importance] = the noise values already multiplied by raw_noise_std, added to the main query's density before compositing in
 * the respective pass (the reflected ray and the offset queries take none, as in the reference); NULL = none. */
typedef struct {
    const float* d_t_rand;
    const float* d_u;
    const float* d_noise_coarse;
    const float* d_noise_fine;
    /* per-ray near / far planes (render_decomp's `near` / `far` as [n, 1] tensors, ibl_nerf_renderer.py:802-805): d_near / d_far [n_rays], both or
     * neither; the scalar near_ / far_ arguments are then not read.  The coarse grid is per ray (z = near (1 - t) + far t, :668-674; the
     * reflected ray's z_vals_constant too) and so is depth_0 = (near + far) / 2 of the prefiltered-radiance mip level (:456). */
    const float* d_near;
    const float* d_far;
} iblnerf_sampling;
int iblnerf_render_rays_sampled(iblnerf_ctx* ctx, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays,
                                float near_, float far_, const iblnerf_overrides* overrides, const iblnerf_sampling* sampling,
                                const iblnerf_outputs* outputs);
/* iblnerf_render_rays_sampled that also hands out what a backward pass needs (a training step with approximate_radiance=True, train.py:295):
 * the z_vals of both passes and the main query's raw rows of both passes (after auxiliary networks wrote their columns), copied out of the
 * workspace launch by launch.  Any pointer may be NULL.  d_z_coarse [n_rays, N_samples], d_z_fine [n_rays, N_samples + N_importance],
 * d_raw_coarse [n_rays, N_samples, 18], d_raw_fine [n_rays, N_samples + N_importance, 18].  Needs options.coarse_outputs.
 * d_env_coarse / d_env_fine [n_rays, 4, 3]: the pass's four reflected-ray maps (radiance, coarse radiances 1..3; raw2outputs_simple,
 * ibl_nerf_renderer.py:38-68, :446-448) LINEAR, before tone map and gamma — the constants iblnerf_ray_outputs_backward takes (the output
 * maps reflected_radiance_map / reflected_coarse_radiance_map_k are their gamma-corrected forms). */
typedef struct {
    float* d_z_coarse;
    float* d_z_fine;
    float* d_raw_coarse;
    float* d_raw_fine;
    float* d_env_coarse;
    float* d_env_fine;
} iblnerf_taps;
int iblnerf_render_rays_tapped(iblnerf_ctx* ctx, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays,
                               float near_, float far_, const iblnerf_overrides* overrides, const iblnerf_sampling* sampling,
                               const iblnerf_outputs* outputs, const iblnerf_taps* taps);
/* sample_pdf(bins, weights, N_samples, det=False) with caller-supplied draws d_u [n_rays, n_out] (NULL = det=True). */
int iblnerf_sample_pdf_u(iblnerf_ctx* ctx, void* stream, const float* d_bins, const float* d_weights, int64_t n_rays,
                         int n_bins, int n_out, const float* d_u, float* d_samples);

/* Teacher-forced stage entry (parity tests): ONE raw2outputs pass (ibl_nerf_renderer.py:153-527, with
 * raw2outputs_simple :38-68 for the reflected ray) on CALLER-SUPPLIED network outputs — no MLP launch.  Feeding the reference's
 * recorded `raw` isolates the compositing / epsilon-normal / shading kernels from the MLP kernel's rounding, so that ill-conditioned
 * checkpoints (sharp density, wide-range weights) can be held to fp32 round-off stage by stage.  n_rays <= max_rays_per_launch.
 * The pass uses the context's options (normal mode, gamma, LUT coefficient, ...) and loaded LUT; overrides as in render_rays. */
typedef struct {
    int32_t n_samples;              /* S of this pass: N_samples (coarse) or N_samples + N_importance (fine) */
    const float* d_z;               /* [n_rays, S] z_vals of this pass */
    const float* d_raw;             /* [n_rays, S, 18] main query (network_query_fn output, :202) */
    const float* d_sigma_offsets;   /* [4, n_rays, S] density of the four offset / tilted queries (normal_from_depth.py:158-160);
                                       NULL in the ground-truth and inferred normal modes (the two depth-gradient normal modes are not
                                       served by this entry: IBLNERF_ERR_STATE) */
    const float* d_refl_raw;        /* [n_rays, N_samples, 13] reflected-ray query (:445): columns 0 and 6..17 of its raw rows */
    const float* d_normal_raw;      /* [n_rays, S, 3] normal_mlp samples ([n_rays, 3] with infer_normal_at_surface) or NULL */
    float* d_stage;                 /* optional out [n_rays, 8]: normal before the edit / insert overrides (3), LUT coordinates n.v and
                                       roughness (2), LUT scale and bias (2), mip level of the prefiltered radiance (1) */
    float* d_refl_o;                /* optional out [n_rays, 3]: x_surface (:262) */
    float* d_refl_d;                /* optional out [n_rays, 3]: reflected direction (:439) */
} iblnerf_stage_inputs;
int iblnerf_composite_pass(iblnerf_ctx* ctx, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays,
                           float near_, float far_, const iblnerf_overrides* overrides, const iblnerf_stage_inputs* in,
                           const iblnerf_maps* maps);

/* IBLNERF_MLP_F16_MXFP6 / _MIXED only.  Synchronises the device and reports in *out_of_range whether any MLP launch on
 * this ctx since the last call saw an encoded input or activation at or beyond the f16 range (65504), or whether a network
 * uploaded with iblnerf_upload_weights_device holds a weight beyond it; the flags are cleared.  When it is 1 the affected
 * outputs are invalid: render again on a ctx created with IBLNERF_MLP_BF16X3 (ibl-nerf_amd/renderer.py does this
 * automatically).  A network with an out-of-range weight runs on the bf16x3 kernel from this call on (until its next upload).
 * Always 0 for IBLNERF_MLP_BF16X3. */
int iblnerf_range_status(iblnerf_ctx* ctx, int* out_of_range);
/* The same question without synchronising (for callers that issue many small queries per step, e.g. the training hook):
 * iblnerf_render_rays / _network_query / _upload_weights_device leave a snapshot of the flags behind their launches; this
 * call looks at the newest one.  *pending = 1: its launches have not finished yet (nothing is known; *out_of_range = 0).
 * Otherwise *out_of_range tells whether anything up to and including the newest call went out of range.  Nothing is cleared:
 * follow a positive answer with iblnerf_range_status. */
int iblnerf_range_peek(iblnerf_ctx* ctx, int* out_of_range, int* pending);

/* The same flags for a consumer ON THE DEVICE: enqueues, behind everything launched so far on `stream`, a copy of the flag word (bit 0: a forward
 * activation / input / weight left the f16 range; bit 1: only the gradients of a fused backward did) into d_out (one uint32 in device memory).
 * No synchronisation, nothing is cleared.  The sync-free training path gates a step's gradients on it (a skipped step, as torch.cuda.amp's
 * GradScaler does on inf / NaN — but an f16 overflow need not leave an inf in the results: a ReLU select can mask a saturated dZ).
 * Writes 0 for IBLNERF_MLP_BF16X3. */
int iblnerf_range_flags_async(iblnerf_ctx* ctx, void* stream, uint32_t* d_out);

/* Largest magnitude of each of a network's 15 wide activations (IBLNeRF.forward, ibl_nerf.py:154-210) on a sample of points — a measurement
 * for the caller's range policy (no reference counterpart: fp32 has the range).  d_blob = the state dict as iblnerf_upload_weights_device takes it,
 * d_pts / d_dirs [n_pts, 3] (one view direction per point, as the network receives it), d_max [15] floats, written (not accumulated):
 * positions_linears.0-7, feature_linear, albedo_feature_linear, irradiance_feature_linear, views_linears.0, additional_radiance_feature_linear.0-2
 * (entries 8 and 11 stay 0 for colour-independent networks).  Plain fp32 arithmetic.  ibl-nerf_amd/renderer.py answers an f16 range event with it:
 * the network is rescaled by powers of two (exact: relu(t x) = t relu(x)) until every activation fits, and only then — or if that fails — falls
 * back to IBLNERF_MLP_BF16X3. */
int iblnerf_layer_ranges(iblnerf_ctx* ctx, void* stream, const float* d_blob, size_t n_floats, const float* d_pts, const float* d_dirs, int64_t n_pts,
                         float* d_max);

/* Host-only: the f16 (hi, lo) form of iblnerf_pack_weights_host's stream (IBLNERF_MLP_F16X3), same sizes. */
int iblnerf_pack_weights_host_f16x3(const float* h_blob, size_t n_floats, void* h_stream, size_t stream_bytes,
                                    float* h_tables, size_t table_floats);
/* Host-only: the IBLNERF_MLP_F16_MXFP6 weight-stream format (csrc/layout_mx.h), for the CPU layout tests. */
int iblnerf_pack_weights_host_mx(const float* h_blob, size_t n_floats, void* h_stream, size_t stream_bytes,
                                 float* h_tables, size_t table_floats);
size_t iblnerf_stream_bytes_mx(void);


/* Timing aid for bench.py: HIP-event time (ms) of the MLP kernels launched by the last
 * iblnerf_render_rays call on this ctx, and their count.  Enabled by iblnerf_set_profiling(ctx, 1),
 * which makes render_rays record events around every MLP launch on the launch stream. */
int iblnerf_set_profiling(iblnerf_ctx* ctx, int enabled);
int iblnerf_last_mlp_time(iblnerf_ctx* ctx, float* ms_total, int* n_launches, double* flop_algorithmic);

#ifdef __cplusplus
}
#endif
#endif /* IBLNERF_H */
