// C-ABI implementation (include/iblnerf.h): context, weight/LUT upload, workspace, and the
// render_rays orchestration that strings the kernels together.
#include "../../include/iblnerf.h"
#include "../../include/iblnerf_experimental.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"
#include "layout_mx.h"
#include "pack.h"
#include "sincos_enc.h"

using namespace ibl;

namespace {
std::string g_create_error;

// algorithmic FLOPs per point (SURVEY.md §8 d): 2 * MACs of the nn.Linear layers on the path
constexpr double FLOP_FULL = 1591552.0, FLOP_TRUNK = 982528.0, FLOP_REFL = 1458944.0;
constexpr double FLOP_FEAT_VIEW = 2.0 * (256.0 * 256.0 + 256.0 * 283.0);   // what is_color_independent_to_direction skips
}  // namespace

constexpr int N_SLOTS = 10, N_AUX = 4, N_FLAGS = 1 + N_SLOTS;
// k_select_points: a density estimate below -MARGIN is empty space whatever its error, and a sample behind a transmittance of TMIN carries, with everything behind
// it, a weight below TMIN (the transmittance taken conservatively: render_kernels.hip).  Estimate errors measured on the two fitted checkpoints (3 M coarse-grid points
// and offset copies each, scratch/estimate_error.py): plain f16 0.20 / 1.45 at worst (99.99 %: 0.12 / 0.84; 11 % of a density above 10), f16 + 2 fp6 below 1e-2.
// Widening the margin is nearly free: densities between -2 and -1 are 0.2 % of the samples.
constexpr float COARSE_SELECT_MARGIN = 2.0f, COARSE_SELECT_TMIN = 1e-8f;
// ... the margin is MEASURED, per network, by the route's probe (check_estimates): how far below zero the plain-f16 estimate puts a sample whose density is positive (the deepest
// underestimate among the probe's samples, judged against the f16 + 2 fp6 estimate), doubled — the tripwire fires at half the margin — plus half a unit, rounded up to half
// a unit, never below 2.  A network that would need more than MARGIN_MAX keeps the f16 + 2 fp6 estimates (error below 1e-2 on a network that fits that form) at the base margin.
// A later tripwire event DOUBLES the margin (up to MARGIN_MAX) before it gives the plain-f16 estimates up: evidence the probe did not have.  A wider margin costs the samples
// between -margin and -2 (0.2 % per unit on the first two fitted checkpoints; on the hold-out, whose empty space sits at -3.5, a margin of 4 makes every sample relevant and
// the lists switch themselves off — which is why the margin follows the measured underestimate and not a multiple of the largest error).
constexpr float MARGIN_MAX = 6.0f, MARGIN_ZONE = 8.0f;
// ... the offset copies' depths are differenced and divided by 2 epsilon: S x TMIN x far / (2 epsilon) bounds what the samples left at their estimate can move the
// normal by (192 x 1e-8 x 8 / 0.02 = 8e-4, measured 3.5e-4 on one ray of a frame); two more decades of transmittance cost a sample or two per copy
constexpr float OFFSET_SELECT_TMIN = 1e-10f;
// The TIERED table's thresholds (IBLNERF_ROUTE_FINE_TIERS; k_importance).  Offset copies: T_s dist_s |depth - z_s| of the main ray above 5e-5 -> three f16 products; measured
// (scratch/tier_probe.py, 4 096-pixel probes of four checkpoint x camera cases whose normals need the SAFE table: 99.9 % at 5e-4 .. 2.2e-3 against SAFE under the fast table) a
// threshold of 1e-4 leaves 7e-5 .. 1.2e-4 and no ray above 1e-3 at +1..3 % of a frame, 2e-5 the same at +2..4 %; SAFE itself costs +10 %.  Main query: own weight above 1e-3.
constexpr float TIER_TAU_OFFSETS = 5e-5f, TIER_TAU_MAIN = 2e-3f;
// estimate_chunked / k_range_points mode 3: a ray counts as saturated behind a chunk (or behind the predicted range) below this transmittance.  0 = the query's own selection
// threshold: the selection drops every sample behind it whatever its estimate, so an estimate there buys only a weight below the threshold for a sample that is dropped anyway
// (set to exactly zero instead).  Round 4 used 1e-12 for every query — two decades under the thresholds, for bit-identical weights — and paid 5.7 % of a frame for it
// (scratch/tmin_ab.py); a value > 0 overrides (iblnerf_set_select_tmin).
constexpr float CHUNK_TMIN = 0.0f;
// the fine grid's offset copies: estimate (0.53 of a TRUNK_X evaluation, 0.40 of a three-product TRUNK one) + share x that evaluation
constexpr double FINE_OFFSET_SELECT_MAX_FRACTION = 0.42, FINE_OFFSET_SELECT_MAX_FRACTION_3 = 0.55;
// ... with the main ray's prediction a sample costs an estimate OR an evaluation (offsets_on_lists): in MAC terms the lists then always pay; what remains against them are the
// lists' extra launches and scattered rows.  Fog (everything relevant: share 1.0) stays a whole-batch launch.
constexpr double OFFSET_PREDICTED_MAX_FRACTION = 0.85;
constexpr double FINE_SELECT_MAX_FRACTION = 0.6;   // the fine main query: estimate (0.33 of the whole network's time per sample; 0.18 on three products) + share x whole network
constexpr long SELECT_MIN_RAYS = 1024;        // iblnerf_decide_route's probe needs at least this many rays: a handful of rays must not fix a checkpoint's route
constexpr double SELECT_MAX_FRACTION = 0.3;   // above this share of relevant samples (measured on the first launch of a checkpoint) the refinement is not worth its estimate
constexpr long BWD_CHUNK_POINTS = 262144;   // points per piece of a fused backward: 4 GiB of operand stash (15.2 KiB per point) at most
// which fast weight streams a precision mode keeps beside the always-present bf16 (hi, lo) stream
static bool wants_mx(int prec) {
    return prec == IBLNERF_MLP_F16_MXFP6 || prec == IBLNERF_MLP_F16_MIXED || prec == IBLNERF_MLP_F16X3_MXFP6 || prec == IBLNERF_MLP_F16X3_MAIN ||
           prec == IBLNERF_MLP_F16X3_MXFP6X;
}
static bool wants_f16x3(int prec) {
    return prec == IBLNERF_MLP_F16X3 || prec == IBLNERF_MLP_F16X3_MXFP6 || prec == IBLNERF_MLP_F16X3_MAIN || prec == IBLNERF_MLP_F16X3_MXFP6X;
}
// query classes of render_rays, for the per-class choice of the product scheme
enum QueryClass { Q_MAIN_COARSE, Q_MAIN_FINE, Q_OFFSET_COARSE, Q_OFFSET_FINE, Q_REFL, Q_AUX, Q_USER,
                  Q_ESTIMATE,     // a density estimate on the fast kernel, to be refined on the relevant points (k_select_points); also the fast kernel's list forms
                  Q_LIST3 };      // a list form on the three-product f16 kernel (VAR_FULL_LIST / VAR_TRUNK_LIST of mlp_kernel.hip)
// albedo, roughness, irradiance (each channel overwrites a column of the raw rows), normal (own buffer)
constexpr int AUX_SLOT0[N_AUX] = {2, 5, 6, 7}, AUX_CHANNELS[N_AUX] = {3, 1, 1, 3}, AUX_RAW_COLUMN[3] = {1, 4, 5};

struct iblnerf_ctx {
    iblnerf_options opt;
    std::string err;
    int n_cu = 256;
    // per network
    // network slots: 0 coarse, 1 fine, 2.. one trunk-shaped stream per output channel of an auxiliary PositionMLP
    // (albedo r, g, b, roughness, irradiance), allocated on first upload
    char* d_stream[N_SLOTS] = {};
    char* d_stream_mx[N_SLOTS] = {};             // f16 + MX-fp6 form (mlp_precision F16_MXFP6, F16_MIXED, F16X3_MXFP6)
    char* d_stream_f16[N_SLOTS] = {};            // f16 (hi, lo) form of d_stream's layout (mlp_precision F16X3, F16X3_MXFP6)
    GenericNet generic[2];                        // networks 0 / 1 of an architecture OUTSIDE the built one (iblnerf_upload_weights_arch): blob != null = this slot runs
                                                  // on generic_mlp.hip — every sample of every query, layer by layer in exact fp32; no lists, no estimates, no backward
    float* generic_ws = nullptr;                  // ... its activation workspace (grown on demand)
    size_t generic_ws_floats = 0;
    float* d_blob32[2] = {};                      // networks 0 / 1 as they are: the fp32 state dict (trunk_fp32_kernel.hip reads the reference's own [out][in] rows)
    bool density_15slot = false;                  // IBLNERF_ROUTE_COARSE_DENSITY_15SLOT: round 4's coarse density (the 15-slot form also on the lists)
    // [0] activation / input range flag of the MX kernels, [1 + slot] "a weight of this slot is outside the f16 range" (device packer)
    unsigned* d_range_flag = nullptr;
    unsigned* h_range_flag = nullptr;            // pinned snapshot for iblnerf_range_peek
    hipEvent_t flag_ev = nullptr;
    bool flag_armed = false;
    bool mx_ok[N_SLOTS] = {true, true, true, true, true, true, true, true, true, true};
    unsigned short* d_map16 = nullptr;            // gather maps of the device packer (built on first use)
    int* d_map_mx = nullptr;
    int* d_map_tab = nullptr;                 // false: a weight is outside the f16 range -> that network runs on the bf16x3 kernel
    float* d_tables[N_SLOTS] = {};
    bool have_net[N_SLOTS] = {};
    bool aux_on[N_AUX] = {};                  // IBLNERF_AUX_* enabled for render_rays
    float* nrm_raw = nullptr;                 // [ws_rays, Smax, 3] normal_mlp samples (allocated with the first IBLNERF_AUX_NORMAL upload)
    float* d_lut = nullptr;
    bool have_lut = false;
    // iblnerf_options.query_routing (IBLNERF_ROUTE_*), decoded at iblnerf_create
    bool x_coarse = false, x_user = false, fine_main_precise = false;
    bool x_fine_precise = false;                  // IBLNERF_ROUTE_FINE_OFFSETS_PRECISE
    bool coarse_sigma_p = true, p_user = false;
    bool est_whole = false;                       // IBLNERF_ROUTE_ESTIMATES_WHOLE: no z-chunks (estimate_chunked)
    bool est_f16 = true;                          // density estimates behind a list refinement in plain f16 (IBLNERF_ROUTE_ESTIMATES_6SLOT: on the f16 + 2 fp6 form) ...
    bool est_checked[2] = {false, false}, est_ok[2] = {false, false};   // ... once the network's first launch has shown that they are good enough (check_estimates)
    bool est_probe = false;                       // (that check's own plain-f16 launch)
    float margin[N_SLOTS] = {2.f, 2.f, 2.f, 2.f, 2.f, 2.f, 2.f, 2.f, 2.f, 2.f};   // the selection margin of networks 0 / 1 (check_estimates; COARSE_SELECT_MARGIN until measured)
    float est_error[2] = {-1.f, -1.f};            // ... and the largest plain-f16 estimate error the probe saw near zero density (-1: not measured)
    bool deciding = false;                        // inside iblnerf_decide_route's probe render: the only place a decision of the route is taken
    bool route_decided = false;                   // iblnerf_route.decided
    int tripped = 0;                              // iblnerf_escalate_route has been applied since the decision: 1 = margins doubled / the estimates moved to f16 + 2 fp6, 2 = the lists went off
    bool lists_off = false;                       // iblnerf_set_lists(ctx, 0): every query evaluates all of its samples whatever the route says (the repeat of a tripped ray)
    bool tapped_lists = false;                    // iblnerf_set_tapped_lists(ctx, 1): a tapped call's MAIN queries follow the decided route too (the taps' unlisted rows: estimate, zeros)
    unsigned char* trip_rays = nullptr;           // the current launch's slice of iblnerf_outputs.trip_rays (k_tripwire marks the rays whose estimates were thin), or null
    long cur_R = 1;                               // ... and that launch's ray count
    double coarse_share = -1.0;                   // the probe's relevant share of the coarse grid (sel_on = it is <= SELECT_MAX_FRACTION)
    bool offsets_estimate_all = false;            // IBLNERF_ROUTE_OFFSETS_ESTIMATE_ALL: round 4's offsets (an estimate on every sample of every copy)
    int* main_range = nullptr;                    // [ws_rays][2] first / last relevant sample of each ray's main query in the current pass (k_select_points range_out)
    unsigned long long* tier_mask = nullptr;      // [2][ws_rays][4] k_importance's per-sample flags of the fine pass's main rays (offset tiers; the main query's own two masks)
    float tier_tau = 0.0f;                        // ... their threshold on T_s dist_s |depth - z_s| (iblnerf_set_offset_tier_threshold; 0 = no tiers, the default)
    bool no_offset_tiers = false;                 // IBLNERF_ROUTE_NO_OFFSET_TIERS
    bool fine_tiers = false;                      // IBLNERF_ROUTE_FINE_TIERS: the TIERED table — the fine pass's main query and offset copies on the fast forms except where
                                                  // an error would show: the samples k_importance flags go to the three-product kernels
    float tier_tau_main = 0.0f;                   // ... the main query's threshold on a sample's own weight (0: TIER_TAU_MAIN)
    bool tier_single = false;                     // (experiment, env IBLNERF_TIER_TWO_PHASE=1) the fine main query's tiers all decided on the refined densities: flagged samples evaluated twice
    float tmin_main = COARSE_SELECT_TMIN, tmin_offsets = OFFSET_SELECT_TMIN, tmin_chunk = CHUNK_TMIN;     // iblnerf_set_select_tmin (tmin_chunk 0: each query's own threshold)
    float chunk_t(float own) const { return tmin_chunk > 0.f ? tmin_chunk : own; }
    int cuts_fine[2] = {0, 0}, cuts_refl[2] = {0, 0};      // iblnerf_set_chunk_cuts (0: the built-in fractions)
    bool ci_embedded[2] = {false, false};          // colour-independent context: slot's packed streams carry the identity in place of the feature / view layers
    double slot_units = 0.0;                      // matrix-slot units of the last render call's whole-batch launches (launch_slots; list launches: sel_count[8..9])
    bool p_all_points = false;                    // IBLNERF_ROUTE_COARSE_DENSITY_ALL_POINTS: the 15-slot form on every coarse sample, not only the relevant ones
    float* sel_pts = nullptr;                     // [4 * ws_rays * Sc, 3] compact list of the relevant coarse samples' points (k_select_points; 4: the offset copies)
    int* sel_index = nullptr;                     // [4 * ws_rays * Sc] their flat indices
    float* sel_est = nullptr;                     // ... and the density estimate that put each entry on the list (k_select_points est_list -> k_tripwire)
    int* sel_count = nullptr;                     // [0] this launch's list length; [2..3] (one uint64) the running total of the render call; [4..5] (one double) the list launches' 2 x MACs; [6] check_estimates' count; [8..9] (one double) the list launches' matrix-slot units
    long sel_candidates = 0;                      // ... and how many samples were candidates
    // Whether refining only the relevant samples pays is a property of the checkpoint: ~6 % of the coarse samples are relevant on a scene with surfaces, all of
    // them in fog (a random-init or barely trained network), where estimate + refinement of everything costs more than the precise kernel alone.  Decided ONCE per
    // host upload of network 0, on the first launch's own count (one stream synchronisation per checkpoint), and frozen: results must not depend on call history.
    // the FINE main query / the fine grid's offset copies on the relevant samples only: the share of relevant samples, measured on the fine pass's first launch
    // (-1: not yet); whether it pays depends on the table (the estimate costs 0.33 / 0.53 of a fast evaluation, 0.18 / 0.40 of a three-product one)
    double fsel_fraction = -1.0, xsel_fraction = -1.0;
    bool sel_decided = false, sel_on = true;   // the coarse pass's density on the 15-slot form (VAR_TRUNK_P) | ... and the trunk-only form of iblnerf_network_query
    bool fuse_points = true;                  // the epsilon-offset points are generated inside the TRUNK kernels (IBLNERF_ROUTE_POINT_BATCH: the [4][R][S][3] batch instead)
    char* bwd_stash = nullptr;                // trunk backward: operand stash and the weight-gradient kernel's partial sums (grown on demand)
    size_t bwd_stash_bytes = 0;
    float* bwd_partial = nullptr;
    size_t bwd_partial_floats = 0;
    float* d_posdir = nullptr;                // PositionDirectionMLP of infer_depth: per layer [Wt | bias] (posdir_kernel.hip)
    int posdir_out_ch = 0;                    // 0 = none uploaded
    // workspace
    long ws_rays = 0;
    int Sc = 0, Sf = 0, Smax = 0;
    float* zc_ray = nullptr;                  // [ws_rays, Sc] jittered coarse grid (perturb > 0), allocated on first use
    float *zc = nullptr, *z_fine = nullptr, *pts = nullptr, *raw = nullptr, *sig4 = nullptr, *w_c = nullptr,
          *w_f = nullptr, *state = nullptr, *refl_o = nullptr, *refl_d = nullptr, *refl_raw = nullptr;
    // profiling
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    double flop_alg = 0.0;
    double flop_exec = 0.0;                       // 2 x MACs the forward launches of the last render call evaluated on whole batches (list launches: sel_count[4..5])

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define HIP_TRY(ctx, expr)                                                                           \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return (ctx)->fail(IBLNERF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static void apply_routing(iblnerf_ctx* c, int bits) {
    c->opt.query_routing = bits;
    c->x_coarse = (bits & IBLNERF_ROUTE_COARSE_OFFSETS_MIXED) != 0;
    c->x_user = (bits & IBLNERF_ROUTE_USER_TRUNK_MIXED) != 0;
    c->fine_main_precise = (bits & IBLNERF_ROUTE_FINE_MAIN_PRECISE) != 0;
    c->fuse_points = (bits & IBLNERF_ROUTE_POINT_BATCH) == 0;
    c->coarse_sigma_p = (bits & IBLNERF_ROUTE_COARSE_MAIN_22BIT) == 0;
    c->p_user = (bits & IBLNERF_ROUTE_USER_TRUNK_P) != 0;
    c->x_fine_precise = (bits & IBLNERF_ROUTE_FINE_OFFSETS_PRECISE) != 0;
    c->p_all_points = (bits & IBLNERF_ROUTE_COARSE_DENSITY_ALL_POINTS) != 0;
    c->est_f16 = (bits & IBLNERF_ROUTE_ESTIMATES_6SLOT) == 0;
    c->est_whole = (bits & IBLNERF_ROUTE_ESTIMATES_WHOLE) != 0;
    c->offsets_estimate_all = (bits & IBLNERF_ROUTE_OFFSETS_ESTIMATE_ALL) != 0;
    c->density_15slot = (bits & IBLNERF_ROUTE_COARSE_DENSITY_15SLOT) != 0;
    c->no_offset_tiers = (bits & IBLNERF_ROUTE_NO_OFFSET_TIERS) != 0;
    c->fine_tiers = (bits & IBLNERF_ROUTE_FINE_TIERS) != 0;
    c->tier_single = std::getenv("IBLNERF_TIER_TWO_PHASE") != nullptr;
}

extern "C" {

int iblnerf_set_query_routing(iblnerf_ctx* c, int bits) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (bits < 0 || bits > 32767) return c->fail(IBLNERF_ERR_INVALID, "set_query_routing: a set of IBLNERF_ROUTE_* bits (0..32767)");
    apply_routing(c, bits);
    return IBLNERF_OK;
}

void iblnerf_default_options(iblnerf_options* o) {
    std::memset(o, 0, sizeof *o);
    o->n_samples = 64;
    o->n_importance = 128;
    o->epsilon = 0.01f;
    o->epsilon_direction = 0.005f;
    o->infer_normal_at_surface = 0;
    o->gamma_correct = 1;
    o->lut_coefficient_f0 = 0;
    o->correct_depth_for_prefiltered_radiance = 1;
    o->coarse_outputs = 1;
    o->max_rays_per_launch = 65536;
    o->device = 0;
}

size_t iblnerf_blob_floats(void) { return blob_floats(); }
size_t iblnerf_stream_bytes(void) { return (size_t)STREAM_BYTES; }
size_t iblnerf_stream_bytes_mx(void) { return (size_t)mx::STREAM_BYTES; }

int iblnerf_pack_weights_host_mx(const float* h_blob, size_t n_floats, void* h_stream, size_t stream_bytes,
                                 float* h_tables, size_t table_floats) {
    if (!h_blob || !h_stream || !h_tables) return IBLNERF_ERR_INVALID;
    if (n_floats != blob_floats() || stream_bytes != (size_t)mx::STREAM_BYTES || table_floats != (size_t)TAB_FLOATS)
        return IBLNERF_ERR_INVALID;
    pack_network_mx(h_blob, h_stream, h_tables);
    return IBLNERF_OK;
}
size_t iblnerf_table_floats(void) { return (size_t)TAB_FLOATS; }

int iblnerf_pack_weights_host_f16x3(const float* h_blob, size_t n_floats, void* h_stream, size_t stream_bytes,
                                    float* h_tables, size_t table_floats) {
    if (!h_blob || !h_stream || !h_tables) return IBLNERF_ERR_INVALID;
    if (n_floats != blob_floats() || stream_bytes != (size_t)STREAM_BYTES || table_floats != (size_t)TAB_FLOATS)
        return IBLNERF_ERR_INVALID;
    pack_network_f16x3(h_blob, h_stream, h_tables);
    return IBLNERF_OK;
}

int iblnerf_pack_weights_host(const float* h_blob, size_t n_floats, void* h_stream, size_t stream_bytes,
                              float* h_tables, size_t table_floats) {
    if (!h_blob || !h_stream || !h_tables) return IBLNERF_ERR_INVALID;
    if (n_floats != blob_floats() || stream_bytes != (size_t)STREAM_BYTES || table_floats != (size_t)TAB_FLOATS)
        return IBLNERF_ERR_INVALID;
    pack_network(h_blob, h_stream, h_tables);
    return IBLNERF_OK;
}

void iblnerf_encode_host(float x, int n_freq, float* h_out) {
    const TurnPair t = to_turns(x);
    for (int k = 0; k < n_freq; ++k) sincos_turns(t, (float)(1 << k), &h_out[2 * k], &h_out[2 * k + 1]);
}

const char* iblnerf_last_error(const iblnerf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int iblnerf_create(const iblnerf_options* opts, iblnerf_ctx** out_ctx) {
    if (!opts || !out_ctx) { g_create_error = "null argument"; return IBLNERF_ERR_INVALID; }
    if (opts->n_samples < 3 || opts->n_samples > 256 || opts->n_importance < 0 ||
        opts->n_samples + opts->n_importance > 256 || opts->max_rays_per_launch < 1) {
        g_create_error = "unsupported sample counts: need 3 <= N_samples, N_samples + N_importance <= 256";
        return IBLNERF_ERR_INVALID;
    }
    if (opts->normal_mode != IBLNERF_NORMAL_DEPTH_GRADIENT_EPSILON && opts->normal_mode != IBLNERF_NORMAL_GROUND_TRUTH &&
        opts->normal_mode != IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION_EPSILON && opts->normal_mode != IBLNERF_NORMAL_INFERRED &&
        opts->normal_mode != IBLNERF_NORMAL_DEPTH_GRADIENT && opts->normal_mode != IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION) {
        g_create_error = "normal_mode must be IBLNERF_NORMAL_DEPTH_GRADIENT_EPSILON (0), IBLNERF_NORMAL_GROUND_TRUTH (1), "
                         "IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION_EPSILON (2), IBLNERF_NORMAL_INFERRED (3), "
                         "IBLNERF_NORMAL_DEPTH_GRADIENT (4) or IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION (5)";
        return IBLNERF_ERR_INVALID;
    }
    if (opts->query_routing < 0 || opts->query_routing > 32767 || opts->persistent_workgroups < 0) {
        g_create_error = "query_routing must be a set of IBLNERF_ROUTE_* bits (0..32767), persistent_workgroups >= 0";
        return IBLNERF_ERR_INVALID;
    }
    if (opts->mlp_precision < IBLNERF_MLP_BF16X3 || opts->mlp_precision > IBLNERF_MLP_F16X3_MXFP6X) {
        g_create_error = "mlp_precision must be IBLNERF_MLP_BF16X3 (0), _F16_MXFP6 (1), _F16_MIXED (2), _F16X3 (3), _F16X3_MXFP6 (4), _F16X3_MAIN (5) or _F16X3_MXFP6X (6)";
        return IBLNERF_ERR_INVALID;
    }
    if ((long)opts->max_rays_per_launch * 4 * (opts->n_samples + opts->n_importance) >= (1L << 31)) {
        g_create_error = "max_rays_per_launch too large: 4 * rays * samples must stay below 2^31 points per MLP launch";
        return IBLNERF_ERR_INVALID;
    }
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0 || opts->device >= n_dev) {
        g_create_error = std::string("no usable HIP device (the renderer has no CPU fallback): ") +
                         (e != hipSuccess ? hipGetErrorString(e) : "device ordinal out of range");
        return IBLNERF_ERR_HIP;
    }
    if ((e = hipSetDevice(opts->device)) != hipSuccess) { g_create_error = hipGetErrorString(e); return IBLNERF_ERR_HIP; }
    iblnerf_ctx* c = new iblnerf_ctx();
    c->opt = *opts;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, opts->device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    if (opts->persistent_workgroups > 0) c->n_cu = opts->persistent_workgroups;
    apply_routing(c, opts->query_routing);
    c->Sc = opts->n_samples;
    c->Sf = opts->n_samples + opts->n_importance;
    c->Smax = c->Sf;
    c->ws_rays = opts->max_rays_per_launch;
    const size_t R = (size_t)c->ws_rays, Sm = (size_t)c->Smax, Sc = (size_t)c->Sc;
    struct { float** p; size_t n; } bufs[] = {
        {&c->zc, Sc}, {&c->z_fine, R * Sm}, {&c->pts, 4 * R * Sm * 3}, {&c->raw, R * Sm * RAW_CH},
        {&c->sig4, 4 * R * Sm}, {&c->w_c, R * Sc}, {&c->w_f, R * Sm}, {&c->state, R * ST_FLOATS},
        {&c->refl_o, R * 3}, {&c->refl_d, R * 3}, {&c->refl_raw, R * Sc * REFL_CH}};
    for (auto& b : bufs)
        if (hipMalloc((void**)b.p, b.n * sizeof(float)) != hipSuccess) {
            g_create_error = "hipMalloc of the render workspace failed";
            iblnerf_destroy(c);
            return IBLNERF_ERR_NOMEM;
        }
    if (hipMalloc((void**)&c->sel_pts, 4 * R * Sm * 3 * sizeof(float)) != hipSuccess || hipMalloc((void**)&c->sel_index, 4 * R * Sm * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&c->sel_count, 12 * sizeof(int)) != hipSuccess || hipMemset(c->sel_count, 0, 12 * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&c->main_range, 2 * R * sizeof(int)) != hipSuccess || hipMalloc((void**)&c->sel_est, 4 * R * Sm * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&c->tier_mask, 2 * 4 * R * sizeof(unsigned long long)) != hipSuccess) {
        g_create_error = "hipMalloc of the render workspace failed";
        iblnerf_destroy(c);
        return IBLNERF_ERR_NOMEM;
    }
    for (int w = 0; w < 2; ++w)
        if (hipMalloc((void**)&c->d_stream[w], STREAM_BYTES) != hipSuccess ||
            hipMalloc((void**)&c->d_tables[w], TAB_BYTES) != hipSuccess) {
            g_create_error = "hipMalloc of the weight stream failed";
            iblnerf_destroy(c);
            return IBLNERF_ERR_NOMEM;
        }
    if (opts->mlp_precision != IBLNERF_MLP_BF16X3) {
        bool ok = hipMalloc((void**)&c->d_range_flag, N_FLAGS * sizeof(unsigned)) == hipSuccess &&
                  hipMemset(c->d_range_flag, 0, N_FLAGS * sizeof(unsigned)) == hipSuccess &&
                  hipHostMalloc((void**)&c->h_range_flag, N_FLAGS * sizeof(unsigned)) == hipSuccess &&
                  hipEventCreateWithFlags(&c->flag_ev, hipEventDisableTiming) == hipSuccess;
        if (ok) std::memset(c->h_range_flag, 0, N_FLAGS * sizeof(unsigned));
        for (int w = 0; w < 2 && ok; ++w) {
            if (wants_mx(opts->mlp_precision))
                ok = ok && hipMalloc((void**)&c->d_stream_mx[w], mx::STREAM_BYTES) == hipSuccess &&
                     hipMemset(c->d_stream_mx[w], 0, mx::STREAM_BYTES) == hipSuccess;
            if (wants_f16x3(opts->mlp_precision))
                ok = ok && hipMalloc((void**)&c->d_stream_f16[w], STREAM_BYTES) == hipSuccess &&
                     hipMemset(c->d_stream_f16[w], 0, STREAM_BYTES) == hipSuccess;
        }
        if (!ok) {
            g_create_error = "hipMalloc of the f16 weight streams failed";
            iblnerf_destroy(c);
            return IBLNERF_ERR_NOMEM;
        }
    }
    if (hipMalloc((void**)&c->d_lut, 3 * 512 * 512 * sizeof(float)) != hipSuccess) {
        g_create_error = "hipMalloc of the LUT failed";
        iblnerf_destroy(c);
        return IBLNERF_ERR_NOMEM;
    }
    *out_ctx = c;
    return IBLNERF_OK;
}

void iblnerf_destroy(iblnerf_ctx* c) {
    if (!c) return;
    if (c->zc_ray) (void)hipFree(c->zc_ray);
    float* bufs[] = {c->zc, c->z_fine, c->pts, c->raw, c->sig4, c->w_c, c->w_f, c->state, c->refl_o, c->refl_d,
                     c->refl_raw, c->d_lut, c->nrm_raw};
    for (float* b : bufs)
        if (b) (void)hipFree(b);
    for (int w = 0; w < N_SLOTS; ++w) {
        if (c->d_tables[w]) (void)hipFree(c->d_tables[w]);
        if (c->d_stream[w]) (void)hipFree(c->d_stream[w]);
        if (c->d_stream_mx[w]) (void)hipFree(c->d_stream_mx[w]);
        if (c->d_stream_f16[w]) (void)hipFree(c->d_stream_f16[w]);
        if (w < 2 && c->d_blob32[w]) (void)hipFree(c->d_blob32[w]);
    }
    if (c->sel_pts) (void)hipFree(c->sel_pts);
    if (c->sel_index) (void)hipFree(c->sel_index);
    if (c->sel_count) (void)hipFree(c->sel_count);
    if (c->main_range) (void)hipFree(c->main_range);
    if (c->sel_est) (void)hipFree(c->sel_est);
    if (c->tier_mask) (void)hipFree(c->tier_mask);
    for (int w = 0; w < 2; ++w) if (c->generic[w].blob) (void)hipFree(const_cast<float*>(c->generic[w].blob));
    if (c->generic_ws) (void)hipFree(c->generic_ws);
    if (c->d_posdir) (void)hipFree(c->d_posdir);
    if (c->bwd_stash) (void)hipFree(c->bwd_stash);
    if (c->bwd_partial) (void)hipFree(c->bwd_partial);
    if (c->d_range_flag) (void)hipFree(c->d_range_flag);
    if (c->h_range_flag) (void)hipHostFree(c->h_range_flag);
    if (c->flag_ev) (void)hipEventDestroy(c->flag_ev);
    if (c->d_map16) (void)hipFree(c->d_map16);
    if (c->d_map_mx) (void)hipFree(c->d_map_mx);
    if (c->d_map_tab) (void)hipFree(c->d_map_tab);
    for (auto& ev : c->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    delete c;
}

// Another network in slot 0 / 1: whatever was measured on the previous one is gone (both upload paths: a training context that uploads every step thus never holds
// a route, and a checkpoint loaded onto a used context is measured anew — ADVICE r4)
static void reset_route(iblnerf_ctx* c, int slot) {
    if (slot >= 2) return;
    c->route_decided = false;
    c->tripped = 0;
    c->sel_decided = false; c->sel_on = true; c->coarse_share = -1.0;
    c->fsel_fraction = c->xsel_fraction = -1.0;
    c->est_checked[0] = c->est_checked[1] = c->est_ok[0] = c->est_ok[1] = false;
    c->margin[0] = c->margin[1] = COARSE_SELECT_MARGIN;
    c->est_error[0] = c->est_error[1] = -1.f;
}

static int upload_slot(iblnerf_ctx* c, int slot, const float* h_blob, size_t n_floats, const char* who) {
    if (n_floats != blob_floats())
        return c->fail(IBLNERF_ERR_INVALID, "%s: blob has %zu floats, the IBLNeRF state dict has %zu", who, n_floats, blob_floats());
    std::vector<float> embedded;
    if (slot < 2 && c->opt.color_independent_to_direction) {
        // is_color_independent_to_direction (ibl_nerf.py:192): the radiance heads read the trunk's output, feature_linear / views_linears are unused.  The packed
        // streams carry the identity in their place (kernels.h launch_identity_embed) — the forward's _CI instantiations skip those layers, the fused backward
        // (VAR_NET_BWD, built for the full architecture) runs through them and finds h2 = h7
        embedded.assign(h_blob, h_blob + n_floats);
        size_t vw, vb, fw, fb;
        blob_offsets(8, &vw, &vb); blob_offsets(9, &fw, &fb);
        std::fill(embedded.begin() + vw, embedded.begin() + vb + 256, 0.0f);
        std::fill(embedded.begin() + fw, embedded.begin() + fb + 256, 0.0f);
        for (int o = 0; o < 256; ++o) { embedded[vw + (size_t)o * 283 + o] = 1.0f; embedded[fw + (size_t)o * 256 + o] = 1.0f; }
        h_blob = embedded.data();
    }
    std::vector<char> stream((size_t)STREAM_BYTES);
    std::vector<float> tab((size_t)TAB_FLOATS);
    pack_network(h_blob, stream.data(), tab.data());
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipDeviceSynchronize());   // a previous render may still be reading the old stream
    const bool want_mx = wants_mx(c->opt.mlp_precision), want_f16 = wants_f16x3(c->opt.mlp_precision);
    if (!c->d_stream[slot]) HIP_TRY(c, hipMalloc((void**)&c->d_stream[slot], STREAM_BYTES));          // auxiliary slots: first use
    if (!c->d_tables[slot]) HIP_TRY(c, hipMalloc((void**)&c->d_tables[slot], TAB_BYTES));
    if (want_mx && !c->d_stream_mx[slot]) HIP_TRY(c, hipMalloc((void**)&c->d_stream_mx[slot], mx::STREAM_BYTES));
    if (want_f16 && !c->d_stream_f16[slot]) HIP_TRY(c, hipMalloc((void**)&c->d_stream_f16[slot], STREAM_BYTES));
    HIP_TRY(c, hipMemcpy(c->d_stream[slot], stream.data(), STREAM_BYTES, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tables[slot], tab.data(), TAB_BYTES, hipMemcpyHostToDevice));
    if (slot < 2 && c->opt.mlp_precision != IBLNERF_MLP_BF16X3) {          // the fp32 state dict itself, for the exact-fp32 trunk of the coarse pass's density
        if (!c->d_blob32[slot]) HIP_TRY(c, hipMalloc((void**)&c->d_blob32[slot], n_floats * sizeof(float)));
        HIP_TRY(c, hipMemcpy(c->d_blob32[slot], h_blob, n_floats * sizeof(float), hipMemcpyHostToDevice));
    }
    if (want_mx || want_f16) {
        bool ok = true;                           // f16(W) must be finite: |w| < 65520 and not NaN
        for (size_t i = 0; i < n_floats && ok; ++i) ok = std::fabs(h_blob[i]) < 65504.0f;
        c->mx_ok[slot] = ok;
    }
    if (want_f16) {
        pack_network_f16x3(h_blob, stream.data(), tab.data());
        HIP_TRY(c, hipMemcpy(c->d_stream_f16[slot], stream.data(), STREAM_BYTES, hipMemcpyHostToDevice));
    }
    if (want_mx) {
        std::vector<char> smx((size_t)mx::STREAM_BYTES);
        pack_network_mx(h_blob, smx.data(), tab.data());
        HIP_TRY(c, hipMemcpy(c->d_stream_mx[slot], smx.data(), mx::STREAM_BYTES, hipMemcpyHostToDevice));
    }
    c->have_net[slot] = true;
    if (slot < 2) c->ci_embedded[slot] = !embedded.empty();
    if (slot < 2 && c->generic[slot].blob) { (void)hipFree(const_cast<float*>(c->generic[slot].blob)); c->generic[slot] = GenericNet(); }
    reset_route(c, slot);
    return IBLNERF_OK;
}

int iblnerf_upload_weights_arch(iblnerf_ctx* c, int which, const float* h_blob, size_t n_floats, int netdepth, int netwidth, int multires, int multires_views) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (which < 0 || which > 1 || !h_blob) return c->fail(IBLNERF_ERR_INVALID, "upload_weights_arch: which must be 0 or 1, blob non-null");
    if (netdepth < 1 || netdepth > 32 || netdepth == 5 || netwidth < 2 || netwidth > 4096 || (netwidth & 1) || multires < 0 || multires > 24 || multires_views < 0 || multires_views > 24)
        return c->fail(IBLNERF_ERR_INVALID, "upload_weights_arch: IBLNeRF(D=%d, W=%d, multires=%d, multires_views=%d) is outside 1 <= D <= 32 (D != 5: the reference's own "
                                            "forward fails there), even 2 <= W <= 4096, 0 <= multires, multires_views <= 24", netdepth, netwidth, multires, multires_views);
    if (c->opt.color_independent_to_direction) return c->fail(IBLNERF_ERR_STATE, "upload_weights_arch: colour-independent networks are built for the fused architecture only");
    const long want = generic_blob_floats(netdepth, netwidth, multires, multires_views);
    if ((long)n_floats != want) return c->fail(IBLNERF_ERR_INVALID, "upload_weights_arch: blob has %zu floats, IBLNeRF(D=%d, W=%d, %d / %d) has %ld", n_floats, netdepth, netwidth, multires, multires_views, want);
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipDeviceSynchronize());
    if (c->generic[which].blob) { (void)hipFree(const_cast<float*>(c->generic[which].blob)); c->generic[which] = GenericNet(); }
    float* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, n_floats * sizeof(float)));
    HIP_TRY(c, hipMemcpy(d, h_blob, n_floats * sizeof(float), hipMemcpyHostToDevice));
    c->generic[which].blob = d; c->generic[which].D = netdepth; c->generic[which].W = netwidth; c->generic[which].L = multires; c->generic[which].Lv = multires_views;
    c->have_net[which] = true;
    reset_route(c, which);
    return IBLNERF_OK;
}

int iblnerf_upload_weights(iblnerf_ctx* c, int which, const float* h_blob, size_t n_floats) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (which < 0 || which > 1 || !h_blob) return c->fail(IBLNERF_ERR_INVALID, "upload_weights: which must be 0/1, blob non-null");
    return upload_slot(c, which, h_blob, n_floats, "upload_weights");
}

int iblnerf_upload_aux_weights(iblnerf_ctx* c, int kind, int channel, const float* h_blob, size_t n_floats) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (kind < 0 || kind >= N_AUX || channel < 0 || channel >= AUX_CHANNELS[kind < 0 || kind >= N_AUX ? 0 : kind] || !h_blob)
        return c->fail(IBLNERF_ERR_INVALID, "upload_aux_weights: kind must be IBLNERF_AUX_* and channel inside its out_ch, blob non-null");
    const int rc = upload_slot(c, AUX_SLOT0[kind] + channel, h_blob, n_floats, "upload_aux_weights");
    if (rc) return rc;
    if (kind == IBLNERF_AUX_NORMAL && !c->nrm_raw)
        HIP_TRY(c, hipMalloc((void**)&c->nrm_raw, (size_t)c->ws_rays * c->Smax * 3 * sizeof(float)));
    bool all = true;
    for (int ch = 0; ch < AUX_CHANNELS[kind]; ++ch) all = all && c->have_net[AUX_SLOT0[kind] + ch];
    c->aux_on[kind] = all;                       // takes effect once every output channel is there
    return IBLNERF_OK;
}

int iblnerf_clear_aux(iblnerf_ctx* c, int kind) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (kind < 0 || kind >= N_AUX) return c->fail(IBLNERF_ERR_INVALID, "clear_aux: kind must be IBLNERF_AUX_*");
    c->aux_on[kind] = false;
    for (int ch = 0; ch < AUX_CHANNELS[kind]; ++ch) c->have_net[AUX_SLOT0[kind] + ch] = false;
    return IBLNERF_OK;
}

// Enqueues a copy of the range-flag words into pinned host memory behind everything launched so far on `s`, for iblnerf_range_peek.
static int arm_range_snapshot(iblnerf_ctx* c, hipStream_t s) {
    if (!c->d_range_flag) return IBLNERF_OK;
    HIP_TRY(c, hipMemcpyAsync(c->h_range_flag, c->d_range_flag, N_FLAGS * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipEventRecord(c->flag_ev, s));
    c->flag_armed = true;
    return IBLNERF_OK;
}

int iblnerf_upload_weights_device(iblnerf_ctx* c, void* stream, int which, const float* d_blob, size_t n_floats) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (which < 0 || which > 1 || !d_blob) return c->fail(IBLNERF_ERR_INVALID, "upload_weights_device: which must be 0/1, blob non-null");
    if (n_floats != blob_floats())
        return c->fail(IBLNERF_ERR_INVALID, "upload_weights_device: blob has %zu floats, the IBLNeRF state dict has %zu", n_floats, blob_floats());
    HIP_TRY(c, hipSetDevice(c->opt.device));
    if (!c->d_map16) {                            // first use: build the gather maps on the host, keep them on the device
        std::vector<uint16_t> m16;
        std::vector<int32_t> mmx, mtab;
        build_pack_maps(m16, mmx, mtab);
        HIP_TRY(c, hipMalloc((void**)&c->d_map16, m16.size() * 2));
        HIP_TRY(c, hipMalloc((void**)&c->d_map_mx, mmx.size() * 4));
        HIP_TRY(c, hipMalloc((void**)&c->d_map_tab, mtab.size() * 4));
        HIP_TRY(c, hipMemcpy(c->d_map16, m16.data(), m16.size() * 2, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_map_mx, mmx.data(), mmx.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_map_tab, mtab.data(), mtab.size() * 4, hipMemcpyHostToDevice));
    }
    PackMaps maps{c->d_map16, c->d_map_mx, c->d_map_tab};
    // an out-of-range weight raises this slot's own flag; iblnerf_range_status / _peek report it and pin the slot to the bf16x3
    // kernel until its next upload (the activation flag alone cannot be relied on: inf * 0 or -inf through a ReLU can hide it)
    unsigned* wflag = c->d_range_flag ? c->d_range_flag + 1 + which : nullptr;
    if (wflag) HIP_TRY(c, hipMemsetAsync(wflag, 0, sizeof(unsigned), (hipStream_t)stream));
    c->ci_embedded[which] = false;
    if (c->opt.mlp_precision != IBLNERF_MLP_BF16X3) {
        if (!c->d_blob32[which]) HIP_TRY(c, hipMalloc((void**)&c->d_blob32[which], n_floats * sizeof(float)));
        HIP_TRY(c, hipMemcpyAsync(c->d_blob32[which], d_blob, n_floats * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        if (c->opt.color_independent_to_direction) {            // (upload_slot: the identity in place of the unused feature / view layers)
            size_t vw, vb, fw, fb;
            blob_offsets(8, &vw, &vb); blob_offsets(9, &fw, &fb);
            HIP_TRY(c, launch_identity_embed(c->d_blob32[which], vw, vb, fw, fb, (hipStream_t)stream));
            d_blob = c->d_blob32[which];
            c->ci_embedded[which] = true;
        }
    }
    HIP_TRY(c, launch_pack_weights(d_blob, maps, c->d_stream[which], c->d_stream_mx[which], c->d_stream_f16[which], c->d_tables[which], wflag,
                                   (hipStream_t)stream));
    c->mx_ok[which] = true;
    c->have_net[which] = true;
    reset_route(c, which);
    return arm_range_snapshot(c, (hipStream_t)stream);
}

// (in, out) of the 14 nn.Linear layers of a PositionDirectionMLP in registration order (src/networks/MLP.py:41-49, D = 8, W = 256)
static void posdir_layers(int out_ch, int (&dims)[14][2]) {
    const int d[14][2] = {{63, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {319, 256}, {256, 256}, {256, 256},
                          {256, 256}, {283, 128}, {128, 128}, {128, 128}, {128, 128}, {128, out_ch}};
    std::memcpy(dims, d, sizeof d);
}

size_t iblnerf_posdir_floats(int out_ch) {
    int dims[14][2];
    posdir_layers(out_ch, dims);
    size_t n = 0;
    for (auto& l : dims) n += (size_t)l[0] * l[1] + l[1];
    return n;
}

int iblnerf_upload_posdir_mlp(iblnerf_ctx* c, const float* h_blob, size_t n_floats, int out_ch) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!h_blob || out_ch < 1 || out_ch > 16) return c->fail(IBLNERF_ERR_INVALID, "upload_posdir_mlp: blob non-null, 1 <= out_ch <= 16");
    if (n_floats != iblnerf_posdir_floats(out_ch))
        return c->fail(IBLNERF_ERR_INVALID, "upload_posdir_mlp: blob has %zu floats, a PositionDirectionMLP with %d output(s) has %zu", n_floats,
                       out_ch, iblnerf_posdir_floats(out_ch));
    int dims[14][2];
    posdir_layers(out_ch, dims);
    std::vector<float> packed(n_floats);
    size_t off = 0;
    for (auto& l : dims) {                        // weight [out,in] row-major -> [in][out], then the bias
        const int n_in = l[0], n_out = l[1];
        for (int o = 0; o < n_out; ++o)
            for (int i = 0; i < n_in; ++i) packed[off + (size_t)i * n_out + o] = h_blob[off + (size_t)o * n_in + i];
        std::memcpy(&packed[off + (size_t)n_in * n_out], &h_blob[off + (size_t)n_in * n_out], n_out * sizeof(float));
        off += (size_t)n_in * n_out + n_out;
    }
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipDeviceSynchronize());
    if (c->d_posdir && c->posdir_out_ch != out_ch) { (void)hipFree(c->d_posdir); c->d_posdir = nullptr; }
    if (!c->d_posdir) HIP_TRY(c, hipMalloc((void**)&c->d_posdir, n_floats * sizeof(float)));
    HIP_TRY(c, hipMemcpy(c->d_posdir, packed.data(), n_floats * sizeof(float), hipMemcpyHostToDevice));
    c->posdir_out_ch = out_ch;
    return IBLNERF_OK;
}

int iblnerf_clear_posdir_mlp(iblnerf_ctx* c) {
    if (!c) return IBLNERF_ERR_INVALID;
    c->posdir_out_ch = 0;
    return IBLNERF_OK;
}

int iblnerf_posdir_query(iblnerf_ctx* c, void* stream, const float* d_pts, const float* d_viewdirs, int64_t n, float* d_out) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n < 0 || (n > 0 && (!d_pts || !d_viewdirs || !d_out))) return c->fail(IBLNERF_ERR_INVALID, "posdir_query: bad arguments");
    if (!c->posdir_out_ch) return c->fail(IBLNERF_ERR_STATE, "posdir_query: no PositionDirectionMLP uploaded");
    if (n == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    PosDirArgs a{c->d_posdir, d_pts, d_viewdirs, d_out, (long)n, c->posdir_out_ch, 0, 0};
    HIP_TRY(c, launch_posdir_mlp(a, (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_upload_lut(iblnerf_ctx* c, const float* h_rgb) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!h_rgb) return c->fail(IBLNERF_ERR_INVALID, "upload_lut: null LUT");
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(c->d_lut, h_rgb, 3 * 512 * 512 * sizeof(float), hipMemcpyHostToDevice));
    c->have_lut = true;
    return IBLNERF_OK;
}

int iblnerf_get_rays(iblnerf_ctx* c, void* stream, int H, int W, const float* h_K, const float* h_c2w, int row0,
                     int n_rows, float* d_rays_o, float* d_rays_d) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!h_K || !h_c2w || H <= 0 || W <= 0 || row0 < 0 || n_rows < 0 || row0 + n_rows > H ||
        (n_rows > 0 && (!d_rays_o || !d_rays_d)))
        return c->fail(IBLNERF_ERR_INVALID, "get_rays: bad arguments (H=%d W=%d row0=%d n_rows=%d)", H, W, row0, n_rows);
    if (n_rows == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    Camera cam;
    std::memcpy(cam.K, h_K, sizeof cam.K);
    std::memcpy(cam.c2w, h_c2w, sizeof cam.c2w);
    HIP_TRY(c, launch_get_rays(W, row0, 1, n_rows, nullptr, cam, d_rays_o, d_rays_d, (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_get_rays_strided(iblnerf_ctx* c, void* stream, int H, int W, const float* h_K, const float* h_c2w, int row0, int row_step, int n_rows,
                             float* d_rays_o, float* d_rays_d) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!h_K || !h_c2w || H <= 0 || W <= 0 || row0 < 0 || row_step < 1 || n_rows < 0 || (n_rows > 0 && (long)row0 + (long)(n_rows - 1) * row_step >= H) ||
        (n_rows > 0 && (!d_rays_o || !d_rays_d)))
        return c->fail(IBLNERF_ERR_INVALID, "get_rays_strided: bad arguments (H=%d W=%d row0=%d row_step=%d n_rows=%d)", H, W, row0, row_step, n_rows);
    if (n_rows == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    Camera cam;
    std::memcpy(cam.K, h_K, sizeof cam.K);
    std::memcpy(cam.c2w, h_c2w, sizeof cam.c2w);
    HIP_TRY(c, launch_get_rays(W, row0, row_step, n_rows, nullptr, cam, d_rays_o, d_rays_d, (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_get_rays_pixels(iblnerf_ctx* c, void* stream, int H, int W, const float* h_K, const float* h_c2w, const int64_t* d_pixels, int64_t n_pixels,
                            float* d_rays_o, float* d_rays_d) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!h_K || !h_c2w || H <= 0 || W <= 0 || n_pixels < 0 || (n_pixels > 0 && (!d_pixels || !d_rays_o || !d_rays_d)))
        return c->fail(IBLNERF_ERR_INVALID, "get_rays_pixels: bad arguments (H=%d W=%d n_pixels=%lld)", H, W, (long long)n_pixels);
    if (n_pixels == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    Camera cam;
    std::memcpy(cam.K, h_K, sizeof cam.K);
    std::memcpy(cam.c2w, h_c2w, sizeof cam.c2w);
    HIP_TRY(c, launch_get_rays(W, 0, 1, (long)n_pixels, reinterpret_cast<const long long*>(d_pixels), cam, d_rays_o, d_rays_d, (hipStream_t)stream));
    return IBLNERF_OK;
}

// The coarse pass's density comes from the 15-slot form (VAR_TRUNK_P) where the mode keeps the fast stream and the network fits f16
static bool is_generic(const iblnerf_ctx* c, int which) { return which >= 0 && which < 2 && c->generic[which].blob != nullptr; }

static bool sigma_p_available(const iblnerf_ctx* c, int which) {
    const int prec = c->opt.mlp_precision;
    return !is_generic(c, which) && c->coarse_sigma_p && c->mx_ok[which] && c->d_stream_mx[which] != nullptr &&
           (prec == IBLNERF_MLP_F16X3_MXFP6X || prec == IBLNERF_MLP_F16X3_MXFP6 || prec == IBLNERF_MLP_F16X3_MAIN);
}

// One MLP launch = a kernel FAMILY (which product scheme, which weight stream) and a VARIANT of it (which layers, whole batch or list).  Every launch of a render
// call is named by such a pair in the route table (plan_main / plan_offsets / plan_reflected below); run_launch executes one.
enum KernelFamily { K_NONE = -1, K_BF16X3 = 0, K_F16X3, K_MX, K_MX16, K_FP32 };
struct Launch {
    int kern = K_NONE;
    int variant = 0;
    bool none() const { return kern == K_NONE; }
};
static const char* launch_name(const Launch& l) {
    static const char* fam[] = {"bf16x3", "f16x3", "mx (f16 + 2 fp6)", "mx16 (plain f16)", "fp32 MFMA"};
    static const char* var[] = {"FULL", "TRUNK", "REFL", "FULL_CI", "REFL_CI", "TRUNK_X (layers 0-1 3 x f16)", "TRUNK_GRAD", "TRUNK_BWD", "TRUNK_FEAT", "TRUNK_BWD_FEAT", "TRUNK_FEAT2",
                                "TRUNK_BWD_FEAT2", "NET_BWD", "TRUNK_P (15-slot)", "?", "REFL_LIST", "FULL_LIST", "TRUNK_X_LIST", "TRUNK_LIST"};
    static thread_local char buf[96];
    if (l.none()) return "-";
    std::snprintf(buf, sizeof buf, "%s %s", fam[l.kern], (l.variant >= 0 && l.variant <= 18) ? var[l.variant] : "?");
    return buf;
}
// 2 x the nn.Linear MACs per point of a variant (FLOP_* above)
static double variant_flops(int variant) {
    if (variant == VAR_TRUNK_GRAD) return 2.0 * FLOP_TRUNK;
    if (variant == VAR_TRUNK || variant == VAR_TRUNK_P || variant == VAR_TRUNK_X || variant == VAR_TRUNK_X_LIST || variant == VAR_TRUNK_LIST) return FLOP_TRUNK;
    return (variant_albirr(variant) ? FLOP_FULL : FLOP_REFL) - (variant_ci(variant) ? FLOP_FEAT_VIEW : 0.0);
}
// matrix-core slots per 64 MACs of a launch (one slot = one 32x32x16 f16 MFMA's time; an MX-fp6 K = 64 product counts 1): what a point really costs — STATE.md section 2,
// the quantity that governs speed on a power-bound chip.  TRUNK_X: layers 0-1 on three f16 products, 2-7 on f16 + 2 fp6 (7.5 on average)
static double launch_slots(const Launch& l) {
    switch (l.kern) {
        case K_BF16X3: case K_F16X3: return 12.0;
        case K_MX16: return 4.0;
        case K_FP32: return 64.0;         // v_mfma_f32_32x32x2_f32: 1/16 of the f16 rate
        case K_MX: return l.variant == VAR_TRUNK_P ? 15.0 : variant_trunk_x(l.variant) ? 7.5 : 6.0;
        default: return 0.0;
    }
}

// Are network `which`'s estimates on the plain-f16 estimate kernel (the one that also takes lists)?
static bool est_plain(const iblnerf_ctx* c, int which) { return which < 2 && c->est_f16 && c->est_checked[which] && c->est_ok[which] && !c->est_probe; }
// ... in z-chunks / on the offset copies' front and behind ranges (lists): both estimate kernels take them — the plain-f16 one and, for a network whose plain-f16 estimates
// were refused (probe or tripwire), the f16 + 2 fp6 TRUNK form, so that such a network pays 6 instead of 4 slots per estimate and nothing else (IBLNERF_ROUTE_ESTIMATES_WHOLE: never)
static bool est_chunks(const iblnerf_ctx* c, int which) { return which < 2 && c->est_checked[which] && !c->est_probe && !c->est_whole; }

// The kernel a query CLASS runs on under the context's mlp_precision and routing bits (include/iblnerf.h: the mode table).  `variant` is the form the caller asks for;
// the class decides the product scheme (and, for the sample-placing density, the 15-slot form).
static Launch pick_kernel(const iblnerf_ctx* c, int which, int variant, int qclass, bool generated_points) {
    // a trunk-only query of the sample-placing class (the coarse pass reduced to its density), or of the caller under IBLNERF_ROUTE_USER_TRUNK_P
    if (variant == VAR_TRUNK && !generated_points && sigma_p_available(c, which) && (qclass == Q_MAIN_COARSE || (qclass == Q_USER && c->p_user))) variant = VAR_TRUNK_P;
    if (c->opt.color_independent_to_direction) variant = variant == VAR_FULL ? VAR_FULL_CI : (variant == VAR_REFL ? VAR_REFL_CI : variant);
    // product scheme of this launch (include/iblnerf.h: mlp_precision).  A network with a weight outside the f16 range runs
    // on the bf16x3 kernel whatever the mode.
    const int prec = c->opt.mlp_precision;
    int kern = K_BF16X3;
    bool mixed_trunk = false;        // the fast kernel's TRUNK form with its first two layers as three f16 products (VAR_TRUNK_X)
    if (prec != IBLNERF_MLP_BF16X3 && c->mx_ok[which]) {
        if (prec == IBLNERF_MLP_F16X3) kern = K_F16X3;
        else if (prec == IBLNERF_MLP_F16X3_MXFP6)
            // f16 + fp6 only where its 2^-16 is below the channel's own conditioning: the reflected-ray queries (the reference's
            // own fp64-vs-fp32 runs differ by 2e-2 .. 6e-2 there on a checkpoint with surfaces)
            kern = qclass == Q_REFL ? K_MX : K_F16X3;
        else if (prec == IBLNERF_MLP_F16X3_MXFP6X) {
            // ... and the offset queries on the fast kernel's mixed TRUNK form (layers 0-1 as three f16 products): the first layers set
            // the density's error, so the normal stays that of the f16x3 kernel for +17 % matrix instructions over the fast kernel;
            // ... and the FINE pass's main query on the fast kernel: its raw rows enter the maps as weighted sums (2^-16 per operand, no
            // amplification: it places no samples and no depth difference is taken of it) — on 1 024 rays of the fitted checkpoint the
            // worst ray of every direct channel is set by the coarse pass's sample placement, with or without it
            const bool x = (qclass == Q_OFFSET_FINE && !c->x_fine_precise) || (c->x_coarse && qclass == Q_OFFSET_COARSE) || (c->x_user && qclass == Q_USER);
            // ... and the COARSE pass's main query too once its density column comes from the 15-slot form (full_pass): what is left of it are the
            // coarse pass's own albedo / roughness / irradiance / radiance samples, weighted sums like the fine pass's
            if (qclass == Q_REFL || (qclass == Q_MAIN_FINE && !c->fine_main_precise) ||
                (qclass == Q_MAIN_COARSE && variant != VAR_TRUNK && variant != VAR_TRUNK_P && sigma_p_available(c, which) && !c->fine_main_precise)) kern = K_MX;
            else if (x && variant == VAR_TRUNK) { kern = K_MX; mixed_trunk = true; }
            else kern = K_F16X3;
        } else if (prec == IBLNERF_MLP_F16X3_MAIN)
            // ... and also for the offset queries on the dense fine grid: the normal's worst ray of 1024 goes from 1.9e-4 to
            // 1.5e-3 (99.9th percentile 3e-4); on the coarse grid (spacing 0.12) the same offsets would leave 1e-3 at 96 rays
            kern = (qclass == Q_OFFSET_FINE || qclass == Q_REFL) ? K_MX : K_F16X3;
        else kern = K_MX;
        // IBLNERF_MLP_F16_MIXED: queries that neither place samples nor feed the finite-difference normal run in plain f16
        // (not with HDR radiance, whose unbounded ReLU passes the raw error through, nor when the surface point of the main
        // query is the input of a normal_mlp)
        if (prec == IBLNERF_MLP_F16_MIXED && (qclass == Q_MAIN_FINE || qclass == Q_REFL) && variant != VAR_TRUNK &&
            !c->opt.use_radiance_linear && !(c->aux_on[IBLNERF_AUX_NORMAL] && c->opt.infer_normal_at_surface))
            kern = K_MX16;
    }
    if (variant == VAR_TRUNK_P) { kern = K_MX; mixed_trunk = false; }   // (its callers checked sigma_p_available)
    if (variant == VAR_TRUNK_X_LIST) { kern = K_MX; mixed_trunk = false; }
    if (qclass == Q_LIST3) { kern = K_F16X3; mixed_trunk = false; }     // (its callers checked the f16 pair stream: d_stream_f16)
    // (likewise; the trunk-only estimates in plain f16: all an estimate has to get right is which side of -1 a raw density lies on)
    if (qclass == Q_ESTIMATE) { kern = (variant == VAR_TRUNK && which < 2 && ((c->est_f16 && c->est_checked[which] && c->est_ok[which]) || c->est_probe)) ? K_MX16 : K_MX; mixed_trunk = false; }
    // the density-gradient query exists in the three-product kernels only: f16 pairs when the mode keeps that stream, else bf16 pairs
    if (variant == VAR_TRUNK_GRAD) { kern = (prec != IBLNERF_MLP_BF16X3 && c->mx_ok[which] && c->d_stream_f16[which]) ? K_F16X3 : K_BF16X3; mixed_trunk = false; }
    Launch l;
    l.kern = kern;
    l.variant = mixed_trunk ? VAR_TRUNK_X : variant;
    return l;
}

struct MlpCall {
    const float* pts = nullptr;
    const float* dirs = nullptr;
    int pts_per_ray = 1;
    long n_pts = 0;
    float* out = nullptr;
    int out_stride = 1;
    const PointGen* gen = nullptr;
    bool count_flops = true;            // this launch's points enter the call's ALGORITHMIC count (every sample of every query, once) ...
    double flop_per_point = -1.0;       // ... priced as this (an estimate launch standing for the query it belongs to), or as the variant itself (< 0)
    const int* n_pts_dev = nullptr;     // a list: its length in device memory, n_pts = the bound that sizes the launch
    const int* out_index = nullptr;
    float trip_margin = 0.0f;           // > 0: a list (c->sel_index / c->sel_est of the k_select_points call before it) launched over the estimates that selected it — the
                                        // tripwire compares the refined densities with them afterwards (k_tripwire)
};

// A network outside the built architecture: generic_mlp.hip, whole batches only (its context never holds a route: sigma_p_available is false for it).
static int run_generic(iblnerf_ctx* c, hipStream_t s, const Launch& l, int which, const MlpCall& m) {
    const GenericNet& g = c->generic[which];
    if (m.n_pts_dev != nullptr || m.out_index != nullptr || m.gen != nullptr)
        return c->fail(IBLNERF_ERR_STATE, "internal: a list / generated-point launch reached a network of a generic architecture");
    int variant;
    if (l.variant == VAR_FULL || l.variant == VAR_FULL_CI) variant = 0;
    else if (l.variant == VAR_TRUNK || l.variant == VAR_TRUNK_P || l.variant == VAR_TRUNK_X) variant = 1;
    else if (l.variant == VAR_REFL || l.variant == VAR_REFL_CI) variant = 2;
    else return c->fail(IBLNERF_ERR_STATE, "this query (kernel variant %d: a backward or feature query) is not built for IBLNeRF(D=%d, W=%d, multires %d / %d): the fused "
                                           "backward exists for the built 8 x 256 / 10 / 4 architecture and the smaller ones embedded in it", l.variant, g.D, g.W, g.L, g.Lv);
    const int ppr = m.pts_per_ray > 0 ? m.pts_per_ray : 1;
    long chunk = (65536 / ppr) * (long)ppr;
    if (chunk <= 0) chunk = ppr;
    const size_t need = generic_workspace_floats(g.W, g.L, g.Lv, chunk);
    if (need > c->generic_ws_floats) {
        if (c->generic_ws) { HIP_TRY(c, hipStreamSynchronize(s)); (void)hipFree(c->generic_ws); c->generic_ws = nullptr; c->generic_ws_floats = 0; }
        HIP_TRY(c, hipMalloc((void**)&c->generic_ws, need * sizeof(float)));
        c->generic_ws_floats = need;
    }
    HIP_TRY(c, launch_generic_mlp(g, variant, m.pts, m.dirs, ppr, m.n_pts, m.out, m.out_stride, c->generic_ws, chunk, s));
    // (algorithmic FLOPs: 2 x the nn.Linear MACs of THIS architecture)
    const double ch = 3 + 6 * g.L, chv = 3 + 6 * g.Lv, W = g.W, H = g.W / 2;
    double trunk = ch * W + W;
    for (int k = 1; k < g.D; ++k) trunk += (k == 5 ? W + ch : W) * W;
    const double full = trunk + 2 * (W * H) + 3 * H + H + W + W * W + (chv + W) * W + 3 * W + 3 * (W * H + 3 * H);
    const double flop = 2.0 * (variant == 1 ? trunk : variant == 0 ? full : full - (2 * (W * H) + 3 * H + H + W));
    if (m.count_flops) c->flop_alg += (double)m.n_pts * flop;
    c->flop_exec += (double)m.n_pts * flop;
    c->slot_units += (double)m.n_pts * flop / 128.0 * 64.0;
    return IBLNERF_OK;
}

static int run_launch(iblnerf_ctx* c, hipStream_t s, const Launch& l, int which, const MlpCall& m) {
    if (l.none()) return c->fail(IBLNERF_ERR_STATE, "internal: a route table row without a kernel was executed");
    if (m.n_pts >= (1L << 31)) return c->fail(IBLNERF_ERR_INVALID, "more than 2^31 points in one MLP launch");
    if (is_generic(c, which)) return run_generic(c, s, l, which, m);
    if (l.kern == K_FP32) {
        if (which > 1 || !c->d_blob32[which] || l.variant != VAR_TRUNK || m.gen) return c->fail(IBLNERF_ERR_STATE, "internal: the fp32 trunk serves networks 0 / 1, trunk only, points from memory");
        TrunkFp32Args f;
        f.blob = c->d_blob32[which];
        for (int i = 0; i < 9; ++i) {
            size_t w, b;
            blob_offsets(i < 8 ? i : 10, &w, &b);       // positions_linears.0-7, sigma_linear
            f.w_off[i] = (long)w; f.b_off[i] = (long)b;
        }
        f.pts = m.pts; f.n = m.n_pts; f.n_dev = m.n_pts_dev; f.out_index = m.out_index; f.out = m.out; f.out_stride = m.out_stride;
        std::pair<hipEvent_t, hipEvent_t>* ev32 = nullptr;
        if (c->profiling) {
            if (c->ev_used == c->ev_pool.size()) {
                hipEvent_t e0, e1;
                HIP_TRY(c, hipEventCreate(&e0));
                HIP_TRY(c, hipEventCreate(&e1));
                c->ev_pool.emplace_back(e0, e1);
            }
            ev32 = &c->ev_pool[c->ev_used++];
            HIP_TRY(c, hipEventRecord(ev32->first, s));
        }
        HIP_TRY(c, launch_trunk_fp32(f, c->n_cu, s));
        if (ev32) HIP_TRY(c, hipEventRecord(ev32->second, s));
        if (m.trip_margin > 0.0f && m.n_pts_dev != nullptr && m.out_index == c->sel_index)
            HIP_TRY(c, launch_tripwire(c->sel_est, c->sel_index, m.n_pts_dev, m.out, m.out_stride, m.trip_margin, c->d_range_flag, m.n_pts, s, c->trip_rays, m.pts_per_ray, c->cur_R));
        if (m.n_pts_dev == nullptr) {
            c->flop_exec += (double)m.n_pts * FLOP_TRUNK;
            c->slot_units += (double)m.n_pts * FLOP_TRUNK / 128.0 * launch_slots(l);
        }
        if (m.count_flops) c->flop_alg += (double)m.n_pts * (m.flop_per_point >= 0.0 ? m.flop_per_point : FLOP_TRUNK);
        return IBLNERF_OK;
    }
    MlpArgs a;
    a.stream = l.kern == K_BF16X3 ? c->d_stream[which] : l.kern == K_F16X3 ? c->d_stream_f16[which] : c->d_stream_mx[which];
    a.range_flag = c->d_range_flag;
    a.tables = c->d_tables[which];
    a.pts = m.pts;
    a.dirs = m.dirs;
    a.out = m.out;
    a.out_stride = m.out_stride;
    a.n_pts = m.n_pts;
    a.pts_per_ray = m.pts_per_ray;
    if (m.gen) a.gen = *m.gen;
    a.n_pts_dev = m.n_pts_dev;
    a.out_index = m.out_index;
    std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
    if (c->profiling) {
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(c, hipEventCreate(&e0));
            HIP_TRY(c, hipEventCreate(&e1));
            c->ev_pool.emplace_back(e0, e1);
        }
        ev = &c->ev_pool[c->ev_used++];
        HIP_TRY(c, hipEventRecord(ev->first, s));
    }
    HIP_TRY(c, l.kern == K_MX16 ? launch_mlp_mx16(l.variant, a, c->n_cu, s) : l.kern == K_MX ? launch_mlp_mx(l.variant, a, c->n_cu, s)
               : l.kern == K_F16X3 ? launch_mlp_f16x3(l.variant, a, c->n_cu, s) : launch_mlp(l.variant, a, c->n_cu, s));
    if (ev) HIP_TRY(c, hipEventRecord(ev->second, s));
    if (m.trip_margin > 0.0f && m.n_pts_dev != nullptr && m.out_index == c->sel_index)
        HIP_TRY(c, launch_tripwire(c->sel_est, c->sel_index, m.n_pts_dev, m.out, m.out_stride * (variant_albirr(l.variant) ? RAW_CH : l.variant == VAR_REFL_LIST ? REFL_CH : 1),
                                   m.trip_margin, c->d_range_flag, m.n_pts, s, c->trip_rays, m.pts_per_ray, c->cur_R));
    // what the launch costs (whole batches here; list launches: k_count_selection adds theirs on the device, sel_count[4..7])
    const int flop_variant = l.variant == VAR_TRUNK_X ? VAR_TRUNK : l.variant;
    if (m.n_pts_dev == nullptr) {
        c->flop_exec += (double)m.n_pts * variant_flops(flop_variant);
        c->slot_units += (double)m.n_pts * variant_flops(flop_variant) / 128.0 * launch_slots(l);      // (64 MACs = 128 FLOP per slot group)
    }
    // (algorithmic FLOPs are counted once: the density column re-evaluated on the 15-slot form beside a FULL query adds time, not work)
    if (m.count_flops) c->flop_alg += (double)m.n_pts * (m.flop_per_point >= 0.0 ? m.flop_per_point : variant_flops(flop_variant));
    return IBLNERF_OK;
}

// (the entry points outside render_rays: the kernel of the query's class, whole batch)
static int run_mlp(iblnerf_ctx* c, hipStream_t s, int variant, int which, const float* pts, const float* dirs,
                   int pts_per_ray, long n_pts, float* out, int out_stride = 1, int qclass = Q_USER, const PointGen* gen = nullptr,
                   bool count_flops = true) {
    MlpCall m;
    m.pts = pts; m.dirs = dirs; m.pts_per_ray = pts_per_ray; m.n_pts = n_pts; m.out = out; m.out_stride = out_stride; m.gen = gen; m.count_flops = count_flops;
    return run_launch(c, s, pick_kernel(c, which, variant, qclass, gen != nullptr), which, m);
}

int iblnerf_network_query(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_rays, int n_samples,
                          const float* d_viewdirs, float* d_out) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (which < 0 || which > 1 || n_rays < 0 || n_samples < 1 || (n_rays > 0 && (!d_pts || !d_out)))
        return c->fail(IBLNERF_ERR_INVALID, "network_query: bad arguments");
    if (n_rays == 0) return IBLNERF_OK;
    if (!c->have_net[which]) return c->fail(IBLNERF_ERR_STATE, "network_query: weights of network %d not uploaded", which);
    HIP_TRY(c, hipSetDevice(c->opt.device));
    if (int rc = run_mlp(c, (hipStream_t)stream, d_viewdirs ? VAR_FULL : VAR_TRUNK, which, d_pts, d_viewdirs, n_samples,
                         (long)n_rays * n_samples, d_out))
        return rc;
    return arm_range_snapshot(c, (hipStream_t)stream);
}

int iblnerf_trunk_density_fp32(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_pts, float* d_out) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (which < 0 || which > 1 || n_pts < 0 || (n_pts > 0 && (!d_pts || !d_out))) return c->fail(IBLNERF_ERR_INVALID, "trunk_density_fp32: bad arguments");
    if (n_pts == 0) return IBLNERF_OK;
    if (!c->have_net[which] || !c->d_blob32[which]) return c->fail(IBLNERF_ERR_STATE, "trunk_density_fp32: weights of network %d not uploaded (or an IBLNERF_MLP_BF16X3 context)", which);
    HIP_TRY(c, hipSetDevice(c->opt.device));
    Launch l;
    l.kern = K_FP32; l.variant = VAR_TRUNK;
    MlpCall m;
    m.pts = d_pts; m.n_pts = (long)n_pts; m.out = d_out;
    return run_launch(c, (hipStream_t)stream, l, which, m);
}

int iblnerf_density_gradient(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_pts, float* d_out) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (which < 0 || which > 1 || n_pts < 0 || (n_pts > 0 && (!d_pts || !d_out)))
        return c->fail(IBLNERF_ERR_INVALID, "density_gradient: bad arguments");
    if (n_pts == 0) return IBLNERF_OK;
    if (!c->have_net[which]) return c->fail(IBLNERF_ERR_STATE, "density_gradient: weights of network %d not uploaded", which);
    HIP_TRY(c, hipSetDevice(c->opt.device));
    if (int rc = run_mlp(c, (hipStream_t)stream, VAR_TRUNK_GRAD, which, d_pts, nullptr, 1, (long)n_pts, d_out, 4)) return rc;
    return arm_range_snapshot(c, (hipStream_t)stream);
}

int iblnerf_trunk_features(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_pts, float* d_h7) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (which < 0 || which > 1 || n_pts < 0 || (n_pts > 0 && (!d_pts || !d_h7))) return c->fail(IBLNERF_ERR_INVALID, "trunk_features: bad arguments");
    if (n_pts == 0) return IBLNERF_OK;
    if (n_pts >= (1L << 31)) return c->fail(IBLNERF_ERR_INVALID, "trunk_features: more than 2^31 points");
    if (!c->have_net[which]) return c->fail(IBLNERF_ERR_STATE, "trunk_features: weights of network %d not uploaded", which);
    if (!c->d_stream_f16[which] || !c->mx_ok[which])
        return c->fail(IBLNERF_ERR_STATE, "trunk_features: needs an mlp_precision that keeps the f16x3 stream (f16x3*) and weights inside the f16 range");
    HIP_TRY(c, hipSetDevice(c->opt.device));
    MlpArgs a;
    a.stream = c->d_stream_f16[which]; a.tables = c->d_tables[which]; a.pts = d_pts; a.dirs = nullptr; a.out = d_h7; a.out_stride = 256;
    a.n_pts = n_pts; a.pts_per_ray = 1; a.range_flag = c->d_range_flag;
    HIP_TRY(c, launch_mlp_f16x3(VAR_TRUNK_FEAT, a, c->n_cu, (hipStream_t)stream));
    c->flop_alg += (double)n_pts * FLOP_TRUNK;
    return arm_range_snapshot(c, (hipStream_t)stream);
}

int iblnerf_trunk_features2(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_rays, int n_samples, const float* d_viewdirs,
                            float* d_h7, float* d_h2) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (which < 0 || which > 1 || n_rays < 0 || n_samples < 1 || (n_rays > 0 && (!d_pts || !d_viewdirs || !d_h7 || !d_h2)))
        return c->fail(IBLNERF_ERR_INVALID, "trunk_features2: bad arguments");
    if (n_rays == 0) return IBLNERF_OK;
    const long n_pts = (long)n_rays * n_samples;
    if (n_pts >= (1L << 31)) return c->fail(IBLNERF_ERR_INVALID, "trunk_features2: more than 2^31 points");
    if (c->opt.color_independent_to_direction) return c->fail(IBLNERF_ERR_STATE, "trunk_features2: a colour-independent network has no feature / view layers");
    if (!c->have_net[which]) return c->fail(IBLNERF_ERR_STATE, "trunk_features2: weights of network %d not uploaded", which);
    if (!c->d_stream_f16[which] || !c->mx_ok[which])
        return c->fail(IBLNERF_ERR_STATE, "trunk_features2: needs an mlp_precision that keeps the f16x3 stream (f16x3*) and weights inside the f16 range");
    HIP_TRY(c, hipSetDevice(c->opt.device));
    MlpArgs a;
    a.stream = c->d_stream_f16[which]; a.tables = c->d_tables[which]; a.pts = d_pts; a.dirs = d_viewdirs; a.out = d_h7; a.out2 = d_h2; a.out_stride = 256;
    a.n_pts = n_pts; a.pts_per_ray = n_samples; a.range_flag = c->d_range_flag;
    HIP_TRY(c, launch_mlp_f16x3(VAR_TRUNK_FEAT2, a, c->n_cu, (hipStream_t)stream));
    c->flop_alg += (double)n_pts * (FLOP_TRUNK + FLOP_FEAT_VIEW);
    return arm_range_snapshot(c, (hipStream_t)stream);
}

static int trunk_backward_impl(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_pts, const float* d_dsigma,
                               const float* d_dh7, float grad_scale, float* d_out, float* d_grad, const float* d_dirs = nullptr,
                               int pts_per_ray = 1, const float* d_dh2 = nullptr, const float* d_draw = nullptr);

int iblnerf_network_backward(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_rays, int n_samples,
                             const float* d_viewdirs, const float* d_draw, float grad_scale, float* d_out, float* d_grad) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || n_samples < 1 || (n_rays > 0 && (!d_viewdirs || !d_draw)))
        return c->fail(IBLNERF_ERR_INVALID, "network_backward: bad arguments");
    const bool ci = c->opt.color_independent_to_direction;
    if (which < 0 || which > 1) return c->fail(IBLNERF_ERR_INVALID, "network_backward: which must be 0 / 1");
    if (ci && !c->ci_embedded[which])
        return c->fail(IBLNERF_ERR_STATE, "network_backward: a colour-independent network's backward needs an mlp_precision that keeps the fp32 state dict (not bf16x3)");
    const int rc = trunk_backward_impl(c, stream, which, d_pts, (int64_t)n_rays * n_samples, nullptr, nullptr, grad_scale, d_out, d_grad, d_viewdirs, n_samples,
                                       nullptr, d_draw);
    if (rc || !ci) return rc;
    // the identity layers standing in for feature_linear / views_linears.0 are not parameters of such a network (ibl_nerf.py:192: unused, no gradient)
    size_t vw, vb, fw, fb;
    blob_offsets(8, &vw, &vb); blob_offsets(9, &fw, &fb);
    HIP_TRY(c, hipMemsetAsync(d_grad + vw, 0, (vb + 256 - vw) * sizeof(float), (hipStream_t)stream));
    HIP_TRY(c, hipMemsetAsync(d_grad + fw, 0, (fb + 256 - fw) * sizeof(float), (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_trunk_features2_backward(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_rays, int n_samples,
                                     const float* d_viewdirs, const float* d_dh7, const float* d_dh2, float grad_scale, float* d_out, float* d_grad) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || n_samples < 1 || (n_rays > 0 && (!d_viewdirs || !d_dh7 || !d_dh2)))
        return c->fail(IBLNERF_ERR_INVALID, "trunk_features2_backward: bad arguments");
    if (c->opt.color_independent_to_direction) return c->fail(IBLNERF_ERR_STATE, "trunk_features2_backward: a colour-independent network has no feature / view layers");
    if (which < 0 || which > 1) return c->fail(IBLNERF_ERR_INVALID, "trunk_features2_backward: which must be 0 / 1");
    return trunk_backward_impl(c, stream, which, d_pts, (int64_t)n_rays * n_samples, nullptr, d_dh7, grad_scale, d_out, d_grad, d_viewdirs, n_samples, d_dh2);
}

int iblnerf_aux_query(iblnerf_ctx* c, void* stream, int kind, const float* d_pts, int64_t n_pts, float* d_out) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (kind < 0 || kind >= N_AUX || n_pts < 0 || (n_pts > 0 && (!d_pts || !d_out))) return c->fail(IBLNERF_ERR_INVALID, "aux_query: bad arguments");
    if (!c->aux_on[kind]) return c->fail(IBLNERF_ERR_STATE, "aux_query: auxiliary network %d not uploaded (every output channel)", kind);
    if (n_pts == 0) return IBLNERF_OK;
    if (n_pts >= (1L << 31)) return c->fail(IBLNERF_ERR_INVALID, "aux_query: more than 2^31 points");
    HIP_TRY(c, hipSetDevice(c->opt.device));
    for (int ch = 0; ch < AUX_CHANNELS[kind]; ++ch)
        if (int rc = run_mlp(c, (hipStream_t)stream, VAR_TRUNK, AUX_SLOT0[kind] + ch, d_pts, nullptr, 1, (long)n_pts, d_out + ch, AUX_CHANNELS[kind], Q_AUX)) return rc;
    return arm_range_snapshot(c, (hipStream_t)stream);
}

int iblnerf_aux_backward(iblnerf_ctx* c, void* stream, int kind, int channel, const float* d_pts, int64_t n_pts, const float* d_dout,
                         float grad_scale, float* d_out, float* d_grad) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (kind < 0 || kind >= N_AUX || channel < 0 || channel >= AUX_CHANNELS[kind < 0 || kind >= N_AUX ? 0 : kind] || (n_pts > 0 && !d_dout))
        return c->fail(IBLNERF_ERR_INVALID, "aux_backward: bad arguments");
    return trunk_backward_impl(c, stream, AUX_SLOT0[kind] + channel, d_pts, n_pts, d_dout, nullptr, grad_scale, d_out, d_grad);
}

int iblnerf_trunk_backward(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_pts, const float* d_dsigma,
                           float grad_scale, float* d_out, float* d_grad) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (which < 0 || which > 1) return c->fail(IBLNERF_ERR_INVALID, "trunk_backward: which must be 0 / 1");
    if (n_pts > 0 && !d_dsigma) return c->fail(IBLNERF_ERR_INVALID, "trunk_backward: bad arguments");
    return trunk_backward_impl(c, stream, which, d_pts, n_pts, d_dsigma, nullptr, grad_scale, d_out, d_grad);
}

int iblnerf_trunk_features_backward(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_pts, const float* d_dh7,
                                    float grad_scale, float* d_out, float* d_grad) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (which < 0 || which > 1) return c->fail(IBLNERF_ERR_INVALID, "trunk_features_backward: which must be 0 / 1");
    if (n_pts > 0 && !d_dh7) return c->fail(IBLNERF_ERR_INVALID, "trunk_features_backward: bad arguments");
    return trunk_backward_impl(c, stream, which, d_pts, n_pts, nullptr, d_dh7, grad_scale, d_out, d_grad);
}

static int trunk_backward_impl(iblnerf_ctx* c, void* stream, int which, const float* d_pts, int64_t n_pts, const float* d_dsigma,
                               const float* d_dh7, float grad_scale, float* d_out, float* d_grad, const float* d_dirs, int pts_per_ray,
                               const float* d_dh2, const float* d_draw) {
    if (which < 0 || which >= N_SLOTS || n_pts < 0 || (n_pts > 0 && (!d_pts || !d_out)) || !d_grad)      // (`which`: a weight slot — network 0 / 1, or an auxiliary network's channel)
        return c->fail(IBLNERF_ERR_INVALID, "trunk_backward: bad arguments");
    int gs_exp = 0;
    if (!(grad_scale > 0.0f) || std::frexp(grad_scale, &gs_exp) != 0.5f)
        return c->fail(IBLNERF_ERR_INVALID, "trunk_backward: grad_scale must be a positive power of two");
    if (n_pts >= (1L << 31)) return c->fail(IBLNERF_ERR_INVALID, "trunk_backward: more than 2^31 points");
    if (!c->have_net[which]) return c->fail(IBLNERF_ERR_STATE, "trunk_backward: weights of network %d not uploaded", which);
    if (!c->d_stream_f16[which] || !c->mx_ok[which])
        return c->fail(IBLNERF_ERR_STATE, "trunk_backward: needs an mlp_precision that keeps the f16x3 stream (f16x3*) and weights inside the f16 range");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipMemsetAsync(d_grad, 0, blob_floats() * sizeof(float), s));
    if (n_pts == 0) return IBLNERF_OK;
    // The operand stash costs 15.2 KiB per point, so a call is walked in pieces of at most BWD_CHUNK_POINTS points (whole rays: the kernels
    // find a point's view direction by its index inside the piece); every piece adds its weight gradients into d_grad.
    // (equal-sized pieces: a short tail piece would leave most of the persistent grid idle)
    const long n_rays_total = ((long)n_pts + pts_per_ray - 1) / pts_per_ray;
    const long n_pieces = ((long)n_pts + BWD_CHUNK_POINTS - 1) / BWD_CHUNK_POINTS;
    const long rays_per_piece = std::max<long>(1, (n_rays_total + n_pieces - 1) / n_pieces);
    const long piece = std::min<long>((long)n_pts, rays_per_piece * pts_per_ray);
    const long groups = (piece + 127) / 128, wgs_max = groups * 4;
    const size_t need = (size_t)stash_bytes(wgs_max);
    const int n_split_max = (int)std::min<long>(28, std::max<long>(1, wgs_max / 8));
    if (need > c->bwd_stash_bytes) {
        HIP_TRY(c, hipStreamSynchronize(s));
        if (c->bwd_stash) (void)hipFree(c->bwd_stash);
        c->bwd_stash = nullptr; c->bwd_stash_bytes = 0;
        if (hipMalloc((void**)&c->bwd_stash, need) != hipSuccess)
            return c->fail(IBLNERF_ERR_NOMEM, "trunk_backward: hipMalloc of the %zu MiB operand stash failed (torch's caching allocator may hold the memory: "
                                              "torch.cuda.empty_cache())", need >> 20);
        c->bwd_stash_bytes = need;
    }
    const size_t pneed = (size_t)n_split_max * WGRAD_PARTIAL_FLOATS;
    if (pneed > c->bwd_partial_floats) {
        HIP_TRY(c, hipStreamSynchronize(s));
        if (c->bwd_partial) (void)hipFree(c->bwd_partial);
        c->bwd_partial = nullptr; c->bwd_partial_floats = 0;
        HIP_TRY(c, hipMalloc((void**)&c->bwd_partial, pneed * sizeof(float)));
        c->bwd_partial_floats = pneed;
    }
    const bool net = d_draw != nullptr;
    const bool feat2 = d_dh2 != nullptr || net;
    size_t wo[23], bo[23];
    for (int l = 0; l < 23; ++l) blob_offsets(l, &wo[l], &bo[l]);
    for (long p0 = 0; p0 < n_pts; p0 += piece) {
        const long np = std::min<long>(piece, (long)n_pts - p0);
        const long wgs = ((np + 127) / 128) * 4;
        WgradArgs w;
        w.n_split = (int)std::min<long>(28, std::max<long>(1, wgs / 8));
        MlpArgs a;
        a.stream = c->d_stream_f16[which]; a.tables = c->d_tables[which]; a.pts = d_pts + 3 * p0; a.out = d_out + 4 * p0; a.out_stride = 4;
        a.n_pts = np; a.pts_per_ray = pts_per_ray; a.range_flag = c->d_range_flag; a.stash = c->bwd_stash; a.grad_scale = grad_scale;
        a.dsigma = d_dsigma ? d_dsigma + p0 : nullptr;
        a.dh7 = d_dh7 ? d_dh7 + 256 * p0 : nullptr;
        a.dh2 = d_dh2 ? d_dh2 + 256 * p0 : nullptr;
        a.draw = d_draw ? d_draw + RAW_CH * p0 : nullptr;
        a.dirs = d_dirs ? d_dirs + 3 * (p0 / pts_per_ray) : nullptr;
        HIP_TRY(c, launch_mlp_f16x3(net ? VAR_NET_BWD : feat2 ? VAR_TRUNK_BWD_FEAT2 : d_dh7 ? VAR_TRUNK_BWD_FEAT : VAR_TRUNK_BWD, a, c->n_cu, s));
        c->flop_alg += (double)np * 3.0 * (net ? FLOP_FULL : FLOP_TRUNK + (feat2 ? FLOP_FEAT_VIEW : 0.0));
        w.stash = c->bwd_stash; w.partial = c->bwd_partial; w.grad = d_grad; w.wave_groups = wgs; w.partial_stride = WGRAD_PARTIAL_FLOATS;
        w.n_gemm = net ? 17 : feat2 ? 12 : 9;
        w.n_head = 0;
        w.unscale = 1.0f / grad_scale;
        long part = 0;
        const long bias_part0 = 9 * 65536L + 2 * 16384L + 8192L + 5 * 32768L;
        int n_bias = 0;
        // (k, blob layer, dZ stash, input stash, row length, first column, keeps the layer's bias, output rows)
        auto gemm_ = [&](int k, int layer, int dz_what, int x_what, int in_dim, int col_base, bool bias, int nrows = 256) {
            const int ncols = x_what == STASH_ENC ? 64 : x_what == STASH_DENC ? 32 : 256;
            w.gemm[k] = WgradGemm{dz_what, x_what, nrows, ncols, x_what == STASH_ENC ? PE_PAIRS_PER_HALF : x_what == STASH_DENC ? DE_PAIRS_PER_HALF : 0, in_dim, col_base,
                                  (long)wo[layer], part, bias ? bias_part0 + 512L * n_bias : -1L, (long)bo[layer]};
            if (bias) ++n_bias;
            part += (long)nrows * ncols;
        };
        auto gemm = [&](int k, int layer, int x_what, int in_dim, int col_base) {
            gemm_(k, layer, STASH_DZ + layer, x_what, in_dim, col_base, !(layer == 5 && x_what != STASH_ENC));   // positions_linears.5: its encoding block keeps the bias
        };
        gemm(0, 0, STASH_ENC, 63, 0);
        for (int l = 1; l <= 4; ++l) gemm(l, l, STASH_X + l - 1, 256, 0);
        gemm(5, 5, STASH_ENC, 319, 0);            // positions_linears.5: [x63 | h] (ibl_nerf.py:168)
        gemm(6, 5, STASH_X + 4, 319, 63);
        gemm(7, 6, STASH_X + 5, 256, 0);
        gemm(8, 7, STASH_X + 6, 256, 0);
        if (feat2) {   // blob layers 8 = views_linears.0 ([feature256 | dir27], ibl_nerf.py:194), 9 = feature_linear
            gemm_(9, 9, STASH_DZF, STASH_X + 7, 256, 0, true);
            gemm_(10, 8, STASH_DZV, STASH_XF, 283, 0, true);
            gemm_(11, 8, STASH_DZV, STASH_DENC, 283, 256, false);
        }
        if (net) {     // blob layers 11 / 14 = albedo / irradiance feature layers, 17..19 = additional_radiance_feature_linear.k; then the N = 1/3 heads
            for (int k = 0; k < 3; ++k) gemm_(12 + k, 17 + k, STASH_DF0 + k, STASH_XH2, 256, 0, true, 128);
            gemm_(15, 11, STASH_DFA, STASH_X + 7, 256, 0, true, 128);
            gemm_(16, 14, STASH_DFI, STASH_X + 7, 256, 0, true, 128);
            auto head = [&](int k, int layer, int x_what, int n_ksteps, int nc, int ch0) {
                w.head[k] = WgradArgs::Head{x_what, n_ksteps, nc, ch0, (long)wo[layer], (long)bo[layer]};
            };
            head(0, 10, STASH_X + 7, 16, 1, 0);            // sigma_linear
            head(1, 13, STASH_X + 7, 16, 1, 4);            // roughness_linear
            head(2, 12, STASH_FA, 8, 3, 1);                // albedo_linear
            head(3, 15, STASH_FI, 8, 1, 5);                // irradiance_linear
            head(4, 16, STASH_XH2, 16, 3, 6);              // radiance_linear
            for (int k = 0; k < 3; ++k) head(5 + k, 20 + k, STASH_F0 + k, 8, 3, 9 + 3 * k);   // additional_radiance_linear.k
            w.n_head = 8;
        }
        w.sigma_w_off = (long)wo[10]; w.sigma_b_off = (long)bo[10];
        HIP_TRY(c, launch_wgrad(w, net ? a.draw : a.dsigma, np, s));
    }
    return arm_range_snapshot(c, s);
}

int iblnerf_trim(iblnerf_ctx* c) {
    if (!c) return IBLNERF_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipDeviceSynchronize());
    if (c->bwd_stash) (void)hipFree(c->bwd_stash);
    if (c->bwd_partial) (void)hipFree(c->bwd_partial);
    c->bwd_stash = nullptr; c->bwd_stash_bytes = 0;
    c->bwd_partial = nullptr; c->bwd_partial_floats = 0;
    return IBLNERF_OK;
}

int iblnerf_composite_direct(iblnerf_ctx* c, void* stream, const float* d_raw, const float* d_z, const float* d_rays_d, int64_t n_rays, int n_samples,
                             float* d_maps, float* d_weights) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || n_samples < 1 || n_samples > 256 || (n_rays > 0 && (!d_raw || !d_z || !d_rays_d || !d_maps)))
        return c->fail(IBLNERF_ERR_INVALID, "composite_direct: bad arguments (1 <= n_samples <= 256)");
    if (n_rays == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, launch_composite_direct(d_raw, d_z, d_rays_d, (long)n_rays, n_samples, c->opt.use_radiance_linear, d_maps, d_weights, (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_composite_direct_backward(iblnerf_ctx* c, void* stream, const float* d_raw, const float* d_z, const float* d_rays_d, int64_t n_rays,
                                      int n_samples, const float* d_dmaps, const float* d_dweights, float* d_draw) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || n_samples < 1 || n_samples > 256 || (n_rays > 0 && (!d_raw || !d_z || !d_rays_d || !d_dmaps || !d_draw)))
        return c->fail(IBLNERF_ERR_INVALID, "composite_direct_backward: bad arguments (1 <= n_samples <= 256)");
    if (n_rays == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, launch_composite_direct_backward(d_raw, d_z, d_rays_d, (long)n_rays, n_samples, c->opt.use_radiance_linear, d_dmaps, d_dweights, d_draw,
                                                (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_composite_direct_backward_full(iblnerf_ctx* c, void* stream, const float* d_raw, const float* d_z, const float* d_rays_d, int64_t n_rays,
                                           int n_samples, const float* d_dmaps, const float* d_dweights, float* d_draw) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || n_samples < 1 || n_samples > 256 || (n_rays > 0 && (!d_raw || !d_z || !d_rays_d || !d_dmaps || !d_draw)))
        return c->fail(IBLNERF_ERR_INVALID, "composite_direct_backward_full: bad arguments (1 <= n_samples <= 256)");
    if (n_rays == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, launch_composite_direct_backward(d_raw, d_z, d_rays_d, (long)n_rays, n_samples, c->opt.use_radiance_linear, d_dmaps, d_dweights, d_draw,
                                                (hipStream_t)stream, 0));
    return IBLNERF_OK;
}

int iblnerf_ray_outputs_backward(iblnerf_ctx* c, void* stream, const float* d_maps, const float* d_n_dot_v, const float* d_env, float depth0,
                                 const iblnerf_maps* up, int64_t n_rays, float* d_dmaps) {
    return iblnerf_ray_outputs_backward_gt(c, stream, d_maps, d_n_dot_v, d_env, depth0, up, nullptr, n_rays, d_dmaps);
}

static int ray_outputs_backward_impl(iblnerf_ctx* c, void* stream, const float* d_maps, const float* d_n_dot_v, const float* d_env, float depth0, const float* d_depth0,
                                     const iblnerf_maps* up, const iblnerf_overrides* ovr, int64_t n_rays, float* d_dmaps, float* d_denv = nullptr);

int iblnerf_ray_outputs_backward_env(iblnerf_ctx* c, void* stream, const float* d_maps, const float* d_n_dot_v, const float* d_env, float depth0, const float* d_depth0,
                                     const iblnerf_maps* up, const iblnerf_overrides* ovr, int64_t n_rays, float* d_dmaps, float* d_denv) {
    if (c && n_rays > 0 && (!d_n_dot_v || !d_denv)) return c->fail(IBLNERF_ERR_INVALID, "ray_outputs_backward_env: needs d_n_dot_v / d_env (approximate_radiance) and d_denv");
    return ray_outputs_backward_impl(c, stream, d_maps, d_n_dot_v, d_env, d_depth0 ? 1.0f : depth0, d_depth0, up, ovr, n_rays, d_dmaps, d_denv);
}

int iblnerf_ray_outputs_backward_gt(iblnerf_ctx* c, void* stream, const float* d_maps, const float* d_n_dot_v, const float* d_env, float depth0,
                                    const iblnerf_maps* up, const iblnerf_overrides* ovr, int64_t n_rays, float* d_dmaps) {
    return ray_outputs_backward_impl(c, stream, d_maps, d_n_dot_v, d_env, depth0, nullptr, up, ovr, n_rays, d_dmaps);
}

int iblnerf_ray_outputs_backward_rays(iblnerf_ctx* c, void* stream, const float* d_maps, const float* d_n_dot_v, const float* d_env, const float* d_depth0,
                                      const iblnerf_maps* up, const iblnerf_overrides* ovr, int64_t n_rays, float* d_dmaps) {
    if (c && n_rays > 0 && d_n_dot_v && !d_depth0) return c->fail(IBLNERF_ERR_INVALID, "ray_outputs_backward_rays: d_depth0 [n_rays] is required with d_n_dot_v");
    return ray_outputs_backward_impl(c, stream, d_maps, d_n_dot_v, d_env, 1.0f, d_depth0, up, ovr, n_rays, d_dmaps);
}

static int ray_outputs_backward_impl(iblnerf_ctx* c, void* stream, const float* d_maps, const float* d_n_dot_v, const float* d_env, float depth0, const float* d_depth0,
                                     const iblnerf_maps* up, const iblnerf_overrides* ovr, int64_t n_rays, float* d_dmaps, float* d_denv) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (ovr && ovr->mode != 0)
        return c->fail(IBLNERF_ERR_STATE, "ray_outputs_backward: edit / insert overrides in a gradient-carrying render are not built (only the *_from_gt constants)");
    if (n_rays < 0 || !up || (n_rays > 0 && (!d_maps || !d_dmaps)) || ((d_n_dot_v == nullptr) != (d_env == nullptr)))
        return c->fail(IBLNERF_ERR_INVALID, "ray_outputs_backward: bad arguments (n_dot_v and env: both or neither)");
    if (d_n_dot_v && !c->have_lut) return c->fail(IBLNERF_ERR_STATE, "ray_outputs_backward: no LUT uploaded");
    if (d_n_dot_v && !(depth0 > 0.f)) return c->fail(IBLNERF_ERR_INVALID, "ray_outputs_backward: depth0 = (near + far) / 2 must be positive");
    if (n_rays == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    RayBwdArgs a{};
    a.x = d_maps; a.ndv = d_n_dot_v; a.env = d_env; a.lut = c->d_lut; a.depth0 = depth0; a.depth0_ray = d_depth0; a.denv = d_denv;
    a.out_mode = (c->opt.gamma_correct ? 1 : 0) | (c->opt.use_radiance_linear ? 2 : 0);
    a.lut_f0 = c->opt.lut_coefficient_f0; a.correct_depth = c->opt.correct_depth_for_prefiltered_radiance;
    a.g_color = up->color_map; a.g_radiance = up->radiance_map;
    for (int k = 0; k < 3; ++k) a.g_radiance_k[k] = up->radiance_map_k[k];
    a.g_irradiance = up->irradiance_map; a.g_albedo = up->albedo_map; a.g_roughness = up->roughness_map; a.g_specular = up->specular_map;
    a.g_diffuse = up->diffuse_map; a.g_prefiltered = up->prefiltered_reflected_map; a.g_disp = up->disp_map; a.g_acc = up->acc_map;
    a.g_depth = up->depth_map; a.g_target_depth = up->target_depth_map;
    if (ovr) { a.gt_albedo = ovr->d_gt_albedo; a.gt_roughness = ovr->d_gt_roughness; a.gt_irradiance = ovr->d_gt_irradiance; a.gt_depth = ovr->d_gt_depth; }
    a.dx = d_dmaps;
    HIP_TRY(c, launch_ray_outputs_backward(a, (long)n_rays, (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_coarse_z(iblnerf_ctx* c, void* stream, float near_, float far_, const float* d_t_rand, int64_t n_rays, float* d_z) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || (n_rays > 0 && !d_z)) return c->fail(IBLNERF_ERR_INVALID, "coarse_z: bad arguments");
    if (n_rays == 0) return IBLNERF_OK;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, launch_coarse_z(near_, far_, c->Sc, c->opt.lindisp, c->zc, s));
    if (d_t_rand) {
        HIP_TRY(c, launch_jitter_z(c->zc, c->Sc, d_t_rand, (long)n_rays, d_z, s));
    } else {   // the shared row, once per ray (z_vals.expand, :676)
        HIP_TRY(c, launch_broadcast_rows(c->zc, c->Sc, (long)n_rays, d_z, s));
    }
    return IBLNERF_OK;
}

int iblnerf_sample_points(iblnerf_ctx* c, void* stream, const float* d_rays_o, const float* d_rays_d, const float* d_z, int64_t n_rays, int n_samples,
                          float* d_pts) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || n_samples < 1 || (n_rays > 0 && (!d_rays_o || !d_rays_d || !d_z || !d_pts)))
        return c->fail(IBLNERF_ERR_INVALID, "sample_points: bad arguments");
    if (n_rays == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, launch_make_points(0, d_rays_o, d_rays_d, d_z, n_samples, 0.f, (long)n_rays, n_samples, d_pts, (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_coarse_z_rays(iblnerf_ctx* c, void* stream, const float* d_near, const float* d_far, const float* d_t_rand, int64_t n_rays, float* d_z) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || (n_rays > 0 && (!d_z || !d_near || !d_far))) return c->fail(IBLNERF_ERR_INVALID, "coarse_z_rays: bad arguments");
    if (n_rays == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, launch_ray_grid(d_near, d_far, c->Sc, c->opt.lindisp, d_t_rand, (long)n_rays, d_z, (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_fine_z(iblnerf_ctx* c, void* stream, const float* d_z_coarse, const float* d_weights_coarse, int64_t n_rays, const float* d_u,
                   float* d_z_fine, float* d_z_std) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || (n_rays > 0 && (!d_z_coarse || !d_weights_coarse || !d_z_fine)))
        return c->fail(IBLNERF_ERR_INVALID, "fine_z: bad arguments");
    if (c->opt.n_importance < 1) return c->fail(IBLNERF_ERR_STATE, "fine_z: the context has N_importance = 0");
    if (n_rays == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, launch_fine_z(d_z_coarse, c->Sc, c->Sc, d_weights_coarse, (long)n_rays, c->opt.n_importance, d_u, d_z_fine, d_z_std, (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_composite_sigma(iblnerf_ctx* c, void* stream, const float* d_sigma, const float* d_z, const float* d_rays_d, int64_t n_rays, int n_samples,
                            float* d_weights, float* d_depth, float* d_visibility) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || n_samples < 1 || n_samples > 256 || (n_rays > 0 && (!d_sigma || !d_z || !d_rays_d || !d_weights || !d_depth)))
        return c->fail(IBLNERF_ERR_INVALID, "composite_sigma: bad arguments (1 <= n_samples <= 256)");
    if (n_rays == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, launch_sigma_weights(d_rays_d, d_z, n_samples, d_sigma, nullptr, (long)n_rays, n_samples, d_weights, (hipStream_t)stream, d_depth, d_visibility));
    return IBLNERF_OK;
}

int iblnerf_sample_pdf(iblnerf_ctx* c, void* stream, const float* d_bins, const float* d_weights, int64_t n_rays,
                       int n_bins, int n_out, float* d_samples) {
    return iblnerf_sample_pdf_u(c, stream, d_bins, d_weights, n_rays, n_bins, n_out, nullptr, d_samples);
}

int iblnerf_sample_pdf_u(iblnerf_ctx* c, void* stream, const float* d_bins, const float* d_weights, int64_t n_rays,
                         int n_bins, int n_out, const float* d_u, float* d_samples) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || n_bins < 2 || n_bins > 257 || n_out < 1 || (n_rays > 0 && (!d_bins || !d_weights || !d_samples)))
        return c->fail(IBLNERF_ERR_INVALID, "sample_pdf: need 2 <= n_bins <= 257, n_out >= 1");
    if (n_rays == 0) return IBLNERF_OK;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, launch_sample_pdf(d_bins, n_bins, d_weights, n_bins - 1, (long)n_rays, n_bins, n_out, d_u, d_samples,
                                 (hipStream_t)stream));
    return IBLNERF_OK;
}

// Folds a snapshot of the flag words into the context: a flagged slot runs on the bf16x3 kernel from now on.
// Bits 2, 3, 4 — the estimate tripwire: a list launch refined a positive density whose estimate was half-way to being dropped (2), or overshot beyond what the conservative
// transmittance allows for (3), or — an audited sample — had BEEN dropped (4) — are REPORTED, not acted on (round 5 widened the margins here, per context and for good: results then depended on which calls a context
// had seen, and under sharding on the rank).  What follows from them is the caller's: the rays are marked in iblnerf_outputs.trip_rays and rendered once more with
// iblnerf_set_lists(ctx, 0); a PROBE that trips escalates the route it is deciding (iblnerf_escalate_route).
static int fold_flags(iblnerf_ctx* c, const unsigned* v) {
    int any = (int)(v[0] & 31u);      // bit 0: an activation / input left the f16 range; bit 1: only a backward's gradients did (mlp_kernel.hip); bits 2, 3, 4: the tripwire
    for (int slot = 0; slot < N_SLOTS; ++slot)
        if (v[1 + slot]) { c->mx_ok[slot] = false; any |= 1; }
    return any;
}

int iblnerf_range_status(iblnerf_ctx* c, int* out_of_range) {
    if (!c || !out_of_range) return IBLNERF_ERR_INVALID;
    *out_of_range = 0;
    if (!c->d_range_flag) return IBLNERF_OK;
    unsigned v[N_FLAGS] = {};
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(v, c->d_range_flag, sizeof v, hipMemcpyDeviceToHost));
    *out_of_range = fold_flags(c, v);
    if (*out_of_range) HIP_TRY(c, hipMemset(c->d_range_flag, 0, sizeof v));
    c->flag_armed = false;
    return IBLNERF_OK;
}

int iblnerf_range_peek(iblnerf_ctx* c, int* out_of_range, int* pending) {
    if (!c || !out_of_range || !pending) return IBLNERF_ERR_INVALID;
    *out_of_range = 0;
    *pending = 0;
    if (!c->d_range_flag || !c->flag_armed) return IBLNERF_OK;
    const hipError_t q = hipEventQuery(c->flag_ev);
    if (q == hipErrorNotReady) { *pending = 1; return IBLNERF_OK; }
    if (q != hipSuccess) return c->fail(IBLNERF_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
    *out_of_range = fold_flags(c, c->h_range_flag);
    return IBLNERF_OK;
}

int iblnerf_layer_ranges(iblnerf_ctx* c, void* stream, const float* d_blob, size_t n_floats, const float* d_pts, const float* d_dirs, int64_t n_pts, float* d_max) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!d_blob || !d_max || n_pts < 0 || (n_pts > 0 && (!d_pts || !d_dirs))) return c->fail(IBLNERF_ERR_INVALID, "layer_ranges: bad arguments");
    if (n_floats != blob_floats()) return c->fail(IBLNERF_ERR_INVALID, "layer_ranges: blob has %zu floats, the IBLNeRF state dict has %zu", n_floats, blob_floats());
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipMemsetAsync(d_max, 0, 15 * sizeof(float), (hipStream_t)stream));
    LayerRangeArgs a{};
    a.blob = d_blob; a.pts = d_pts; a.dirs = d_dirs; a.n = (long)n_pts; a.color_independent = c->opt.color_independent_to_direction; a.d_max = d_max;
    for (int l = 0; l < 23; ++l) {
        size_t w, b;
        blob_offsets(l, &w, &b);
        a.w_off[l] = (long)w; a.b_off[l] = (long)b;
    }
    HIP_TRY(c, launch_layer_ranges(a, (hipStream_t)stream));
    return IBLNERF_OK;
}

int iblnerf_range_flags_async(iblnerf_ctx* c, void* stream, uint32_t* d_out) {
    if (!c || !d_out) return IBLNERF_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    if (!c->d_range_flag) HIP_TRY(c, hipMemsetAsync(d_out, 0, sizeof(uint32_t), (hipStream_t)stream));
    else HIP_TRY(c, hipMemcpyAsync(d_out, c->d_range_flag, sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return IBLNERF_OK;
}

static PassOutputs slice_maps(const iblnerf_maps& m, long r0, int S, int irr_ch = 1) {
    auto off = [&](float* p, long w) { return p ? p + r0 * w : nullptr; };
    PassOutputs o;
    o.color = off(m.color_map, 3);
    o.radiance = off(m.radiance_map, 3);
    for (int k = 0; k < 3; ++k) {
        o.radiance_k[k] = off(m.radiance_map_k[k], 3);
        o.refl_coarse_k[k] = off(m.reflected_coarse_radiance_map_k[k], 3);
    }
    o.irradiance = off(m.irradiance_map, irr_ch);
    o.reflected_radiance = off(m.reflected_radiance_map, 3);
    o.prefiltered = off(m.prefiltered_reflected_map, 3);
    o.albedo = off(m.albedo_map, 3);
    o.roughness = off(m.roughness_map, 1);
    o.specular = off(m.specular_map, 3);
    o.diffuse = off(m.diffuse_map, 3);
    o.n_dot_v = off(m.n_dot_v_map, 1);
    o.normal = off(m.target_normal_map, 3);
    o.disp = off(m.disp_map, 1);
    o.acc = off(m.acc_map, 1);
    o.depth = off(m.depth_map, 1);
    o.target_depth = off(m.target_depth_map, 1);
    o.weights = off(m.weights, S);
    o.inferred_normal = off(m.inferred_normal_map, 3);
    return o;
}

// Arguments of pass A as the context's options set them (shared by the render path and the teacher-forced stage entry).
static PassAArgs pass_a_args(iblnerf_ctx* c, const float* ro, const float* rd, long R, const float* z, int z_stride, int S,
                             const float* raw, const float* sig4, const float* nrm_raw, float* weights, float near_, float far_,
                             const OverrideArgs& ov) {
    const bool tilt = c->opt.normal_mode == IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION_EPSILON;
    PassAArgs a;
    a.rays_o = ro; a.rays_d = rd; a.z = z; a.z_stride = z_stride; a.raw = raw; a.sig4 = sig4;
    a.weights = weights; a.lut = c->d_lut; a.near = near_; a.far = far_;
    a.eps = tilt ? c->opt.epsilon_direction : c->opt.epsilon;
    a.tilted_rays = tilt ? 1 : 0;
    a.grad_normal = c->opt.normal_mode == IBLNERF_NORMAL_DEPTH_GRADIENT ? 1 : c->opt.normal_mode == IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION ? 2 : 0;
    a.irradiance_sigmoid = c->aux_on[2] ? 1 : 0;
    a.nrm_raw = nrm_raw;
    a.normal_inferred = c->opt.normal_mode == IBLNERF_NORMAL_INFERRED ? 1 : 0;
    a.nrm_at_surface = c->opt.infer_normal_at_surface != 0 ? 1 : 0;
    a.lut_coefficient_F0 = c->opt.lut_coefficient_f0;
    a.correct_depth = c->opt.correct_depth_for_prefiltered_radiance;
    a.radiance_linear = c->opt.use_radiance_linear;
    a.ov = ov; a.state = c->state; a.refl_o = c->refl_o; a.refl_d = c->refl_d; a.R = R; a.S = S;
    return a;
}

// ---- the route table ------------------------------------------------------------------------------------------------------------------------------
// A render pass (one raw2outputs, ibl_nerf_renderer.py:153-527) asks the network three things: the MAIN query at the ray's samples (:201), the four epsilon-OFFSET
// copies of them for the finite-difference normal (normal_from_depth.py:139-158) and the REFLECTED ray on the coarse grid (:439-446).  "Precision where it matters"
// also decides WHERE a query is evaluated at all: a sample that is clearly empty (alpha = 0 exactly) or behind saturation carries no weight, so such a query runs
// as a density ESTIMATE (plain-f16 TRUNK form) on every sample and in its own precision only on the relevant ones (a list form of the query's kernel; the other rows
// are zero / keep their estimate).  How each of the three is evaluated is ONE row of this table — a pure function of the context's mode, routing bits and
// the checkpoint's route (iblnerf_route: decided once, by measurement, outside any render call) — which plan_* computes, iblnerf_describe_route prints and full_pass executes.
enum PassKind { PASS_COARSE = 0 /* its weights place the fine samples */, PASS_FINE = 1, PASS_SINGLE = 2 /* N_importance = 0: the coarse grid is the only pass */ };

struct QueryPlan {
    bool run = true;                 // false: the query does not exist under these options (e.g. ground-truth normals: no offset copies)
    bool list = false;               // estimate everywhere + the query's kernel on the list of relevant samples; false: `whole` on every sample
    bool open = false;               // iblnerf_decide_route's probe only: this query's list decision is still open — measured on this launch, kept iff share <= share_max
    double share_max = 1.0;
    Launch est;                      // the density estimate
    int cut0 = 0, cut1 = 0;          // ... in z-chunks [0, cut0) | [cut0, cut1) | [cut1, S), the later ones only for rays not yet saturated (0, 0: every sample at once)
    bool predicted = false;          // offset copies: the main ray's relevant range goes to the list without an estimate, estimates on the rest only (offsets_on_lists)
    bool tiers = false;              // ... and the predicted range in two tiers: the samples k_importance flags on `on_list_precise`, the others on `on_list`
    Launch on_list_precise;          // ... which also takes what the copy's OWN selection adds outside the predicted range (none: `on_list` does)
    float t_min = COARSE_SELECT_TMIN;
    Launch on_list;                  // the relevant samples
    Launch density_list;             // coarse main query: its density once more on the 15-slot form, same list
    Launch whole;                    // every sample (list == false, or the probe decided against the list)
    Launch density;                  // coarse main query, whole batch: the 15-slot density column ...
    bool density_on_list = false;    // ... on the samples its own density selects
    bool point_batch = false;        // offset copies: points through the [4][R][S][3] batch (tilted rays, IBLNERF_ROUTE_POINT_BATCH), else generated in the kernel
    bool gradient = false;           // the two autograd normal modes: density + position gradient at the main points instead of offset copies
};

static bool lists_possible(const iblnerf_ctx* c, int which, bool keep_all_rows) {
    const bool can_decide = c->deciding && !keep_all_rows;
    return sigma_p_available(c, which) && !c->p_all_points && !c->opt.color_independent_to_direction && (c->sel_decided ? c->sel_on : can_decide);
}

static bool density_fp32(const iblnerf_ctx* c, int which) { return which < 2 && c->d_blob32[which] != nullptr && !c->density_15slot; }
// the coarse density of a WHOLE batch in exact fp32 too: where it is cheap (a launch of fewer rays than a route needs) or explicit (the repeat of tripped rays)
static bool whole_batch_fp32(const iblnerf_ctx* c) { return c->lists_off || (!c->route_decided && !c->deciding && c->cur_R < SELECT_MIN_RAYS); }

static QueryPlan plan_main(const iblnerf_ctx* c, int which, int kind, int S, bool keep_all_rows) {
    QueryPlan q;
    q.t_min = c->tmin_main;
    const bool can_decide = c->deciding && !keep_all_rows;
    const bool list_ok = lists_possible(c, which, keep_all_rows);
    const int prec = c->opt.mlp_precision;
    // (the fast table's main queries on the fast kernel's list form; the safe table's — IBLNERF_ROUTE_FINE_MAIN_PRECISE, or the F16X3_MXFP6 mode — on the three-product one)
    const bool fast = prec == IBLNERF_MLP_F16X3_MXFP6X && !c->fine_main_precise;
    const bool three = (prec == IBLNERF_MLP_F16X3_MXFP6X || prec == IBLNERF_MLP_F16X3_MXFP6) && !fast && c->d_stream_f16[which] != nullptr;
    q.whole = pick_kernel(c, which, VAR_FULL, kind == PASS_COARSE ? Q_MAIN_COARSE : Q_MAIN_FINE, false);
    q.est = pick_kernel(c, which, VAR_TRUNK, Q_ESTIMATE, false);
    if (kind == PASS_COARSE && sigma_p_available(c, which)) {
        // the density column once more on the 15-slot form (three f16 + three fp6 products per block: operands to ~2^-26).  Two f16 terms
        // hold 22-23 bits of an fp32 weight / activation; through a fitted network's cancelling density sum that alone moves the fine samples of
        // some rays by more than the reference's own arithmetic does (launch scale 7).  The column overwrites raw[..., 0] as an
        // auxiliary network's output would; weights, depth and the fine samples are composited from it.  Only the RELEVANT samples need it.
        q.density.kern = K_MX; q.density.variant = VAR_TRUNK_P;
        q.density_on_list = !(c->p_all_points || (c->sel_decided ? !c->sel_on : !can_decide));
        // ... and on the LIST in exact fp32 (round 5, trunk_fp32_kernel.hip): the samples that place the fine samples, ~5 per ray, at 1/16 of the f16 rate.  (A whole
        // batch of 64 samples per ray — lists off, fog — keeps the 15-slot form: 40 ms per launch in fp32.)
        if (q.density_on_list && density_fp32(c, which)) { q.density.kern = K_FP32; q.density.variant = VAR_TRUNK; }
        // Round 6: ... and on EVERY coarse sample where that is affordable and the arithmetic must not depend on how a ray came to be evaluated whole — the repeat of a
        // tripped ray (iblnerf_set_lists 0: a handful of rays) and a call too small to hold a route (< SELECT_MIN_RAYS rays: 0.8 ms).  The 15-slot form's 1e-6 .. 1.2e-5 on a
        // coarse weight moved the fine samples of one ray in 16 384 far enough for its normal to differ by 8.6e-3 between a 1 023-ray call and a frame; in fp32 they agree.
        // (not a training step's tapped forward: 512 rays x 64 samples in fp32 would add 12 % to a 5 ms step, and its gradients are pinned on the 15-slot density)
        else if (!q.density_on_list && !keep_all_rows && density_fp32(c, which) && whole_batch_fp32(c)) { q.density.kern = K_FP32; q.density.variant = VAR_TRUNK; }
    }
    if (kind == PASS_COARSE && list_ok && !keep_all_rows && (fast || three)) {
        // (the coarse main query: its other channels on the table's kernel for weighted sums, its density on the 15-slot form either way)
        // (not in z-chunks: this query's weights place the fine samples, and sample_pdf's thresholds see the last bit of their sum — a weight of 1e-12 behind saturation
        // set to exactly zero moved z_std of one ray of a frame by 5e-5)
        q.list = true; q.open = !c->sel_decided; q.share_max = SELECT_MAX_FRACTION;
        q.on_list = pick_kernel(c, which, VAR_FULL_LIST, fast ? Q_ESTIMATE : Q_LIST3, false);
        if (fast && c->fine_tiers && !c->deciding && c->d_stream_f16[which] != nullptr && c->mx_ok[which]) {      // (the TIERED table: see the fine pass below)
            q.tiers = true;
            q.on_list_precise = pick_kernel(c, which, VAR_FULL_LIST, Q_LIST3, false);
        }
        q.density_list.kern = K_MX; q.density_list.variant = VAR_TRUNK_P;
        if (density_fp32(c, which)) { q.density_list.kern = K_FP32; q.density_list.variant = VAR_TRUNK; }
    }
    if (kind == PASS_FINE && list_ok && c->sel_decided && c->sel_on && (c->fsel_fraction < 0.0 ? can_decide : c->fsel_fraction <= FINE_SELECT_MAX_FRACTION) && !keep_all_rows &&
        (fast || three)) {
        // the FINE main query likewise: the importance samples crowd around the surface, so about 40 % of them are relevant (against 6-8 % on the coarse
        // grid) — still less than the whole network everywhere, as long as the share stays below FINE_SELECT_MAX_FRACTION.
        // The selected rows are those of the FULL form bit for bit (same kernel arithmetic); the others: the plain-f16 density estimate, zero channels.
        q.list = true; q.open = c->fsel_fraction < 0.0; q.share_max = FINE_SELECT_MAX_FRACTION;
        q.on_list = pick_kernel(c, which, VAR_FULL_LIST, fast ? Q_ESTIMATE : Q_LIST3, false);
        // the TIERED table (round 6): the fast form for every relevant sample but the handful per ray that carry a weight above the threshold — those on three f16 products
        // (what the safe table runs everywhere: its per-sample weights, at the fast table's price + 1 %)
        if (fast && c->fine_tiers && !c->deciding && c->d_stream_f16[which] != nullptr && c->mx_ok[which]) {
            q.tiers = true;
            q.on_list_precise = pick_kernel(c, which, VAR_FULL_LIST, Q_LIST3, false);
        }
        if (est_chunks(c, which)) { q.cut0 = (3 * S) / 4; q.cut1 = (7 * S) / 8; }
        if (est_chunks(c, which) && c->cuts_fine[1] > 0 && c->cuts_fine[1] < S) { q.cut0 = c->cuts_fine[0]; q.cut1 = c->cuts_fine[1]; }
    }
    return q;
}

static QueryPlan plan_offsets(const iblnerf_ctx* c, int which, int kind, int S, bool keep_all_rows, bool main_listed, bool gt_normal) {
    QueryPlan q;
    const bool coarse_grid = kind != PASS_FINE;
    const int qclass = coarse_grid ? Q_OFFSET_COARSE : Q_OFFSET_FINE;
    const bool tilt = c->opt.normal_mode == IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION_EPSILON;
    q.t_min = c->tmin_offsets;
    if (c->opt.normal_mode == IBLNERF_NORMAL_DEPTH_GRADIENT || c->opt.normal_mode == IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION) {
        q.gradient = true;
        q.whole = pick_kernel(c, which, VAR_TRUNK_GRAD, qclass, false);
        return q;
    }
    if (gt_normal || c->opt.normal_mode == IBLNERF_NORMAL_INFERRED) { q.run = false; return q; }
    if (tilt || !c->fuse_points || is_generic(c, which)) {      // (a generic architecture reads its points from memory: the [4][R][S][3] batch)
        q.point_batch = true;
        q.whole = pick_kernel(c, which, VAR_TRUNK, qclass, false);
        return q;
    }
    const bool can_decide = c->deciding && !keep_all_rows;
    const bool list_ok = lists_possible(c, which, keep_all_rows);
    const int prec = c->opt.mlp_precision;
    q.whole = pick_kernel(c, which, VAR_TRUNK, qclass, true);
    q.est = pick_kernel(c, which, VAR_TRUNK, Q_ESTIMATE, true);
    // (the fine grid's offset copies: the fast table's on the mixed trunk form's list, the safe table's on the three-product trunk form's)
    const bool fine_x_fast = prec == IBLNERF_MLP_F16X3_MXFP6X && !c->x_fine_precise;
    const bool fine_x_3 = (prec == IBLNERF_MLP_F16X3_MXFP6X || prec == IBLNERF_MLP_F16X3_MXFP6) && !fine_x_fast && c->d_stream_f16[which] != nullptr;
    const bool could_predict = main_listed && est_chunks(c, which) && !c->offsets_estimate_all;
    if (coarse_grid && S == c->Sc && sigma_p_available(c, which) && !c->p_all_points && !c->x_coarse && c->sel_decided && c->sel_on) {
        // The coarse grid's offsets need all eight layers at 2^-22 (DESIGN 4.0 ladder) — on the samples that can reach a weight: the relevant ones (neither clearly
        // empty nor behind saturation, per offset copy: ~6 % on a scene with surfaces) on the three-product f16 kernel where the mode keeps its stream — what the
        // whole-batch launch runs; 2 MB of weights against the 15-slot form's 3.9 MB, which sits at the edge of an XCD's 4 MB L2 — scattered over the estimates.
        // The others composite to the same weights bit for bit (alpha = 0) or to within 1e-10 of a weight (the saturated tail).
        q.list = true;
        if (c->d_stream_f16[which] != nullptr && c->mx_ok[which]) q.on_list = pick_kernel(c, which, VAR_TRUNK_LIST, Q_LIST3, false);
        else { q.on_list.kern = K_MX; q.on_list.variant = VAR_TRUNK_P; }
        if (est_chunks(c, which)) { q.cut0 = S / 2; q.cut1 = (3 * S) / 4; }
        q.predicted = could_predict;
    } else if (!coarse_grid && list_ok && c->sel_decided && c->sel_on && (fine_x_fast || fine_x_3)) {
        // The offsets on the fine grid (768 densities per ray, more than half of a frame): the mixed trunk form (TRUNK_X; safe table: three f16 products) on the
        // relevant ones of each offset copy — bit for bit what the whole-batch launch computes for them.  With an estimate on every sample (0.53 of a TRUNK_X
        // evaluation) this pays below a relevant share of FINE_OFFSET_SELECT_MAX_FRACTION; with the main ray's prediction (estimates only where the main ray
        // found nothing: each sample costs an estimate OR an evaluation, a few per cent both) up to OFFSET_PREDICTED_MAX_FRACTION — above that the lists' launches
        // and scattered rows cost more than they save.
        const double share_max = could_predict ? OFFSET_PREDICTED_MAX_FRACTION : (fine_x_fast ? FINE_OFFSET_SELECT_MAX_FRACTION : FINE_OFFSET_SELECT_MAX_FRACTION_3);
        if (c->xsel_fraction < 0.0 ? can_decide : c->xsel_fraction <= share_max) {
            q.list = true; q.open = c->xsel_fraction < 0.0; q.share_max = share_max;
            if (fine_x_fast) { q.on_list.kern = K_MX; q.on_list.variant = VAR_TRUNK_X_LIST; }
            else q.on_list = pick_kernel(c, which, VAR_TRUNK_LIST, Q_LIST3, false);
            if (est_chunks(c, which)) { q.cut0 = (3 * S) / 4; q.cut1 = (7 * S) / 8; }
            q.predicted = could_predict;
            // The fast table's mixed trunk form is good to 1.5e-3 in raw density, and 7 of the 8 rays of a 65 536-ray launch it left above 1e-3 on the normal (the safe
            // table: 1, the fp32 C restatement: 0) turned out to owe it to the samples OUTSIDE the predicted range — what a copy's own selection adds where its ray found
            // nothing relevant: a silhouette, the fringe of a haze; there a copy's depth hangs on one or two samples (scratch/which_query.py, scratch/tier_sweep.py).
            // They are ~2 % of the refined samples: on three f16 products (what the safe table runs everywhere) they cost 1 % of a frame and the fast table's normals
            // become the safe table's.  (Offset TIERS — the predicted range itself split by k_importance's bound T_s dist_s |depth - z_s| on the main ray, flagged
            // samples on the precise kernel — were built for the same rays and measured: they remove one ray at 4 % of a frame; off by default, tier_tau = 0.
            // The other direction was measured too (scratch/lean_sweep.py, experiment hook since removed): the predicted range on the plain f16 + 2 fp6 trunk, 6 slots
            // instead of 7.5, puts 8 normals of the 65 536-ray launch and 113 of the second checkpoint's 4 096 rays above 1e-3; the coarse grid's copies on the mixed
            // trunk form 4 coarse normals.  The table is at its floor.)
            if (q.predicted && fine_x_fast && c->d_stream_f16[which] != nullptr && c->mx_ok[which] && !c->no_offset_tiers) {
                q.on_list_precise = pick_kernel(c, which, VAR_TRUNK_LIST, Q_LIST3, false);
                q.tiers = c->tier_tau > 0.0f || c->fine_tiers;
            }
        }
    }
    return q;
}

static QueryPlan plan_reflected(const iblnerf_ctx* c, int which, bool keep_all_rows) {
    QueryPlan q;
    q.t_min = c->tmin_main;
    const int Sc = c->Sc;
    q.whole = pick_kernel(c, which, VAR_REFL, Q_REFL, false);
    q.est = pick_kernel(c, which, VAR_TRUNK, Q_ESTIMATE, false);
    if (lists_possible(c, which, keep_all_rows) && c->sel_decided && c->sel_on) {
        // the reflected ray leaves its surface into empty space and ends on the next one: a density estimate everywhere (fast TRUNK form; the same trunk
        // arithmetic the REFL form runs), the view layers and the twelve radiance channels on the relevant samples only, zero rows elsewhere (weight 0, or < 1e-8)
        q.list = true;
        q.on_list = pick_kernel(c, which, VAR_REFL_LIST, Q_ESTIMATE, false);
        if (est_chunks(c, which)) { q.cut0 = Sc / 2; q.cut1 = (3 * Sc) / 4; }
        if (est_chunks(c, which) && c->cuts_refl[1] > 0 && c->cuts_refl[1] < Sc) { q.cut0 = c->cuts_refl[0]; q.cut1 = c->cuts_refl[1]; }
    }
    return q;
}

static double list_slots(const Launch& l) { return l.none() ? 0.0 : variant_flops(l.variant) / 128.0 * launch_slots(l); }

// A density estimate of the S samples of nv = (offsets ? 4 R : R) (virtual) rays in up to THREE z-chunks: samples [0, cut0) of every ray, then [cut0, cut1) and
// [cut1, S) only of the rays whose transmittance behind the samples in front of the chunk — composited conservatively from those estimates — is not yet below t_min; the other rays'
// later samples get -1e30.  t_min is the query's own selection threshold (round 5; 1e-12 for every query before): k_select_points would not select those samples whatever their
// estimate — its transmittance only falls — so estimating them bought nothing but weights below the threshold for samples that are dropped anyway (5.5 % of a frame).  On the coarse grid half of the rays saturate in
// the first half of the grid, on the fine grid a quarter to a half of them before its last quarter (scratch/saturation_depth.py).  Rows land in c->sig4 [nv, S].
static int estimate_chunked(iblnerf_ctx* c, hipStream_t s, const Launch& est, int which, const float* ro, const float* rd, const float* z, int z_stride, int S, long R, bool offsets,
                            float eps, const float* noise, int cut0, int cut1, float t_min, double flop_alg_per_point) {
    const long nv = offsets ? 4 * R : R;
    const int cuts[4] = {0, cut0, cut1, S};          // chunks [0, cut0) of every ray, then [cut0, cut1) and [cut1, S) of the rays still alive at their start
    for (int k = 0; k < 3; ++k) {
        const int s0 = cuts[k], s1 = cuts[k + 1];
        if (s1 <= s0) continue;
        const bool first = k == 0;
        if (!first) HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
        HIP_TRY(c, launch_chunk_points(ro, rd, z, z_stride, c->sig4, noise, R, S, s0, s1, c->margin[which], t_min, c->sel_pts, c->sel_index, c->sel_count, s, offsets, eps,
                                       first, FLOP_TRUNK, list_slots(est)));
        MlpCall m;
        m.pts = c->sel_pts; m.pts_per_ray = S; m.n_pts = nv * (s1 - s0); m.out = c->sig4; m.count_flops = false; m.n_pts_dev = first ? nullptr : c->sel_count; m.out_index = c->sel_index;
        if (int rc = run_launch(c, s, est, which, m)) return rc;
    }
    c->flop_alg += (double)nv * S * flop_alg_per_point;      // (the query's algorithmic FLOPs: every sample, once)
    return IBLNERF_OK;
}

// Once per route decision: may this network's density estimates run in plain f16?  Both estimates of the probe's main-query samples (c->pts), compared by
// k_compare_estimates; one stream synchronisation.  A network whose plain-f16 trunk is ever half-way to a wrong selection keeps the f16 + 2 fp6 estimates.
// (Every later list launch repeats the comparison on the samples it refines and audits — the tripwire, k_tripwire.)
static int check_estimates(iblnerf_ctx* c, hipStream_t s, int which, long n_pts, int S) {
    c->est_checked[which] = true;
    c->est_ok[which] = false;
    int rc = run_mlp(c, s, VAR_TRUNK, which, c->pts, nullptr, S, n_pts, c->sig4, 1, Q_ESTIMATE, nullptr, false);
    if (rc) return rc;
    c->est_probe = true;
    rc = run_mlp(c, s, VAR_TRUNK, which, c->pts, nullptr, S, n_pts, c->sig4 + n_pts, 1, Q_ESTIMATE, nullptr, false);
    c->est_probe = false;
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->sel_count + 6, 0, 2 * sizeof(int), s));
    HIP_TRY(c, launch_compare_estimates(c->sig4 + n_pts, c->sig4, n_pts, COARSE_SELECT_MARGIN, MARGIN_ZONE, c->sel_count + 6, s));
    int res[2] = {0, 0};                           // [0] overshoots beyond the conservative transmittance's allowance, [1] the bits of the largest error in the zone
    HIP_TRY(c, hipMemcpyAsync(res, c->sel_count + 6, sizeof res, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    float err;
    std::memcpy(&err, &res[1], sizeof err);
    c->est_error[which] = err;
    const float margin = std::max(COARSE_SELECT_MARGIN, std::ceil((2.0f * err + 0.5f) * 2.0f) / 2.0f);
    c->est_ok[which] = res[0] == 0 && margin <= MARGIN_MAX;
    c->margin[which] = c->est_ok[which] ? margin : COARSE_SELECT_MARGIN;
    return IBLNERF_OK;
}

// (iblnerf_decide_route's probe: the length of the list k_select_points has just written)
static int read_list_length(iblnerf_ctx* c, hipStream_t s, long* n) {
    int v = 0;
    HIP_TRY(c, hipMemcpyAsync(&v, c->sel_count, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    *n = v;
    return IBLNERF_OK;
}

struct PassRays {                 // the rays of one pass of one launch
    const float* ro; const float* rd; long R;
    const float* z; int z_stride; int S;
    const float* noise;
};

// The MAIN query of a pass into c->raw [R, S, 18].  *main_listed: c->main_range holds every ray's first / last relevant sample (a selection ran and the lists are on).
static int run_main_query(iblnerf_ctx* c, hipStream_t s, int which, int kind, const PassRays& p, bool keep_all_rows, bool* main_listed) {
    const long R = p.R, n = p.R * p.S;
    const int S = p.S;
    *main_listed = false;
    QueryPlan q = plan_main(c, which, kind, S, keep_all_rows);
    bool est_counted = false;
    int rc;
    if (q.list) {
        if (q.cut1 > 0) rc = estimate_chunked(c, s, q.est, which, p.ro, p.rd, p.z, p.z_stride, S, R, false, 0.f, p.noise, q.cut0, q.cut1, c->chunk_t(q.t_min), FLOP_FULL);
        else {
            MlpCall m;
            m.pts = c->pts; m.pts_per_ray = S; m.n_pts = n; m.out = c->sig4; m.flop_per_point = FLOP_FULL;
            rc = run_launch(c, s, q.est, which, m);
        }
        if (rc) return rc;
        est_counted = true;     // (the query's algorithmic FLOPs are counted once, on its estimate)
        HIP_TRY(c, hipMemsetAsync(c->raw, 0, (size_t)n * RAW_CH * sizeof(float), s));
        HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
        c->sel_candidates += n;
        // The fine pass's tiers are decided BEFORE the lists are evaluated, on the density estimates (k_importance mode 1: own weight above the threshold — the samples of a
        // surface, whose densities are large and whose estimates are good to 11 %), so that every sample is evaluated once; what the estimates cannot tell — a hazy ray, whose
        // whole weight is a few small densities an estimate misses by its absolute error of ~1 — is found afterwards on the refined densities and evaluated once more (below).
        // (Measured: deciding all tiers afterwards, every flagged sample twice, costs 1.7 x as much.)  The coarse pass (5 relevant samples per ray) decides afterwards.
        const bool single = q.tiers && !c->tier_single && kind == PASS_FINE;
        if (single) {
            HIP_TRY(c, launch_importance(p.rd, p.z, p.z_stride, c->sig4, 1, R, S, c->tier_tau_main > 0.0f ? c->tier_tau_main : TIER_TAU_MAIN, c->tier_mask, s, 4, c->margin[which]));
            HIP_TRY(c, launch_select_points(p.ro, p.rd, p.z, p.z_stride, c->sig4, 1, p.noise, R, S, c->margin[which], q.t_min, c->sel_pts, c->sel_index, c->sel_count, s,
                                            false, 0.f, c->raw, RAW_CH, variant_flops(q.on_list_precise.variant), c->main_range, nullptr, list_slots(q.on_list_precise), c->sel_est,
                                            c->tier_mask, 1));
            MlpCall t;
            t.pts = c->sel_pts; t.dirs = p.rd; t.pts_per_ray = S; t.n_pts = n; t.out = c->raw; t.count_flops = false; t.n_pts_dev = c->sel_count; t.out_index = c->sel_index;
            t.trip_margin = c->margin[which];
            if ((rc = run_launch(c, s, q.on_list_precise, which, t))) return rc;
            HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
        }
        HIP_TRY(c, launch_select_points(p.ro, p.rd, p.z, p.z_stride, c->sig4, 1, p.noise, R, S, c->margin[which], q.t_min, c->sel_pts, c->sel_index, c->sel_count, s,
                                        false, 0.f, single ? nullptr : c->raw, RAW_CH, variant_flops(q.on_list.variant) + (q.density_list.none() ? 0.0 : FLOP_TRUNK),
                                        single ? nullptr : c->main_range, nullptr, list_slots(q.on_list) + list_slots(q.density_list), c->sel_est,
                                        single ? c->tier_mask : nullptr, 0));
        if (q.open) {     // the probe: does this network have empty space and surfaces, or is it fog?  / how many of the fine samples are relevant?
            long n_sel = 0;
            if ((rc = read_list_length(c, s, &n_sel))) return rc;
            const double share = (double)n_sel / (double)n;
            if (kind == PASS_COARSE) { c->sel_decided = true; c->sel_on = share <= q.share_max; c->coarse_share = share; }
            else c->fsel_fraction = share;
            q.list = share <= q.share_max;
        }
        if (q.list) {
            MlpCall m;
            m.pts = c->sel_pts; m.dirs = p.rd; m.pts_per_ray = S; m.n_pts = n; m.out = c->raw; m.count_flops = false; m.n_pts_dev = c->sel_count; m.out_index = c->sel_index;
            m.trip_margin = c->margin[which];
            if ((rc = run_launch(c, s, q.on_list, which, m))) return rc;
            if (q.tiers) {
                // The TIERED table's second tier (fine pass: its hazy-ray fix-up only, k_importance mode 3): of the samples just refined on the fast form, those whose own weight — now from the REFINED densities, good to 1e-3 of
                // themselves; the estimates that chose the list can be off by a unit of raw density, which misjudges a weight — exceeds the threshold are evaluated once
                // more on three f16 products and their rows overwritten (k_importance mode 1; the selection below repeats k_select_points' with that filter: same samples,
                // same estimates, no audit, no range).
                const bool density_only = q.on_list_precise.variant == VAR_TRUNK_LIST;
                // (fine pass: what the estimates' tiers MISSED — a sample whose refined weight exceeds the threshold although its estimate said empty (a haze of density 0.5
                // estimated at -0.2: 0.1 % of the second checkpoint's rays, whose per-sample weights otherwise stay at 6e-4 of SAFE's whatever the threshold), and the hazy rays —
                // into the second mask, less what the first one sent to the precise kernel already)
                unsigned long long* mask2 = single ? c->tier_mask + 4 * c->ws_rays : c->tier_mask;
                HIP_TRY(c, launch_importance(p.rd, p.z, p.z_stride, c->raw, RAW_CH, R, S, c->tier_tau_main > 0.0f ? c->tier_tau_main : TIER_TAU_MAIN, mask2, s,
                                             kind != PASS_FINE ? 1 : 2, 0.0f, single ? c->tier_mask : nullptr));
                HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
                HIP_TRY(c, launch_select_points(p.ro, p.rd, p.z, p.z_stride, c->sig4, 1, p.noise, R, S, c->margin[which], q.t_min, c->sel_pts, c->sel_index, c->sel_count, s,
                                                false, 0.f, nullptr, 0, variant_flops(q.on_list_precise.variant), nullptr, nullptr, list_slots(q.on_list_precise), c->sel_est,
                                                mask2, 1));
                MlpCall t;
                t.pts = c->sel_pts; t.dirs = density_only ? nullptr : p.rd; t.pts_per_ray = S; t.n_pts = n; t.out = c->raw; t.count_flops = false; t.n_pts_dev = c->sel_count;
                t.out_index = c->sel_index; t.out_stride = density_only ? RAW_CH : 1;
                if ((rc = run_launch(c, s, q.on_list_precise, which, t))) return rc;
                if (!q.density_list.none()) {      // (the coarse pass: the whole list once more for its exact-fp32 density, which goes over every row's density LAST)
                    HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
                    HIP_TRY(c, launch_select_points(p.ro, p.rd, p.z, p.z_stride, c->sig4, 1, p.noise, R, S, c->margin[which], q.t_min, c->sel_pts, c->sel_index, c->sel_count, s,
                                                    false, 0.f, nullptr, 0, 0.0, nullptr, nullptr, 0.0, c->sel_est));
                }
            }
            if (!q.density_list.none()) {
                m.dirs = nullptr; m.out_stride = RAW_CH; m.trip_margin = 0.0f;
                if ((rc = run_launch(c, s, q.density_list, which, m))) return rc;
            }
            *main_listed = true;
            return IBLNERF_OK;
        }
        q = plan_main(c, which, kind, S, keep_all_rows);      // (the probe decided against the list: the whole-batch row, under the decision just taken)
    }
    {
        MlpCall m;
        m.pts = c->pts; m.dirs = p.rd; m.pts_per_ray = S; m.n_pts = n; m.out = c->raw; m.count_flops = !est_counted;
        if ((rc = run_launch(c, s, q.whole, which, m))) return rc;
    }
    if (!q.density.none()) {
        MlpCall m;
        m.pts_per_ray = S; m.n_pts = n; m.out = c->raw; m.out_stride = RAW_CH; m.count_flops = false;
        bool on_list = q.density_on_list;
        if (on_list) {
            // a sample whose density (the main query's own, error < 1e-2) is below -margin has alpha = 0 exactly whatever its last bits are, and a sample behind a
            // transmittance of 1e-8 carries — with everything behind it — a weight below 1e-8.  The rest is compacted, evaluated and scattered over raw[..., 0].
            HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
            c->sel_candidates += n;
            HIP_TRY(c, launch_select_points(p.ro, p.rd, p.z, p.z_stride, c->raw, RAW_CH, p.noise, R, S, c->margin[which], c->tmin_main, c->sel_pts, c->sel_index,
                                            c->sel_count, s, false, 0.f, nullptr, 0, FLOP_TRUNK, c->main_range, nullptr, list_slots(q.density)));
            if (!c->sel_decided) {     // (the probe, in a mode whose main query takes no list)
                long n_sel = 0;
                if ((rc = read_list_length(c, s, &n_sel))) return rc;
                c->sel_decided = true;
                c->coarse_share = (double)n_sel / (double)n;
                c->sel_on = c->coarse_share <= SELECT_MAX_FRACTION;
                on_list = c->sel_on;
            }
        }
        if (on_list) { m.pts = c->sel_pts; m.n_pts_dev = c->sel_count; m.out_index = c->sel_index; *main_listed = true; }
        else m.pts = c->pts;
        if ((rc = run_launch(c, s, q.density, which, m))) return rc;
    }
    return IBLNERF_OK;
}

// The four epsilon-offset copies of a pass's samples on lists (into c->sig4 [4, R, S]): estimate + own selection + the query's kernel on the relevant ones.
// q.predicted (round 5): a copy lies 0.01 beside its ray, so the samples the MAIN ray found relevant (first to last selected, one more on either side: c->main_range)
// are almost exactly the copy's relevant ones (measured: 0.377 / 0.378 of the fine grid's samples, scratch/offset_candidates.py) — they go to the query's kernel
// WITHOUT an estimate (k_range_points mode 1); estimates run on the rest only: in front of the range for every copy (mode 2), behind it for the copies that have not
// saturated by then (mode 3; the others' get -1e30 as in estimate_chunked: 0.1-1 % of the copies, a silhouette between a ray and its copy), and the copy's OWN
// selection among those (k_select_points with skip_range) sends what the prediction missed to the kernel too.  So every sample is still either evaluated by the
// query's kernel or judged irrelevant on an estimate of ITS OWN: the main ray predicts cost, never a result (a rule that let the main ray's estimates stand in for the
// copies' was measured and rejected: it misses relevant samples with weights up to 0.9, STATE.md).  Estimates per ray: 1 024 -> ~400.
static int offsets_on_lists(iblnerf_ctx* c, hipStream_t s, int which, const QueryPlan& q, const PassRays& p, float eps, const PointGen& g, double* share) {
    const long R = p.R, n4 = 4 * p.R * p.S;
    const int S = p.S;
    int rc;
    MlpCall refine;
    refine.pts = c->sel_pts; refine.pts_per_ray = S; refine.n_pts = n4; refine.out = c->sig4; refine.count_flops = false; refine.n_pts_dev = c->sel_count; refine.out_index = c->sel_index;
    c->sel_candidates += n4;
    unsigned long long entries_before = 0;
    if (share) {      // (the probe: the call's running total of list entries so far, sel_count[2..3])
        HIP_TRY(c, hipMemcpyAsync(&entries_before, c->sel_count + 2, sizeof entries_before, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    if (q.predicted) {
        MlpCall est = refine;
        // 1: the predicted range, straight to the query's kernel (no estimate underneath: no tripwire) — in one list, or in two tiers by k_importance's flags
        if (q.tiers) HIP_TRY(c, launch_importance(p.rd, p.z, p.z_stride, c->raw, RAW_CH, R, S, c->tier_tau > 0.0f ? c->tier_tau : TIER_TAU_OFFSETS, c->tier_mask, s));
        for (int tier = q.tiers ? 1 : 0; tier >= 0; --tier) {
            const Launch& k = (q.tiers && tier == 1) ? q.on_list_precise : q.on_list;
            HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
            HIP_TRY(c, launch_range_points(p.ro, p.rd, p.z, p.z_stride, c->sig4, c->main_range, R, S, 1, c->margin[which], c->chunk_t(q.t_min), eps, c->sel_pts, c->sel_index, c->sel_count, s,
                                           FLOP_TRUNK, list_slots(k), true, q.tiers ? c->tier_mask : nullptr, tier));
            if ((rc = run_launch(c, s, k, which, refine))) return rc;
        }
        // 2, 3: estimates in front of it, and behind it where a copy is still alive
        for (int mode = 2; mode <= 3; ++mode) {
            HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
            HIP_TRY(c, launch_range_points(p.ro, p.rd, p.z, p.z_stride, c->sig4, c->main_range, R, S, mode, c->margin[which], c->chunk_t(q.t_min), eps, c->sel_pts, c->sel_index, c->sel_count,
                                           s, FLOP_TRUNK, list_slots(q.est), false));
            if ((rc = run_launch(c, s, q.est, which, est))) return rc;
        }
        c->flop_alg += (double)n4 * FLOP_TRUNK;
    } else if (q.cut1 > 0) {
        if ((rc = estimate_chunked(c, s, q.est, which, p.ro, p.rd, p.z, p.z_stride, S, R, true, eps, nullptr, q.cut0, q.cut1, c->chunk_t(q.t_min), FLOP_TRUNK))) return rc;
    } else {
        MlpCall m;
        m.pts_per_ray = S; m.n_pts = n4; m.out = c->sig4; m.gen = &g;
        if ((rc = run_launch(c, s, q.est, which, m))) return rc;
    }
    HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
    HIP_TRY(c, launch_select_points(p.ro, p.rd, p.z, p.z_stride, c->sig4, 1, nullptr, R, S, c->margin[which], q.t_min, c->sel_pts, c->sel_index, c->sel_count, s, true, eps,
                                    nullptr, 0, FLOP_TRUNK, nullptr, q.predicted ? c->main_range : nullptr, list_slots(q.on_list), c->sel_est));
    if (share) {      // the probe: (predicted +) selected share of the 4 R S samples
        unsigned long long now = 0;
        HIP_TRY(c, hipMemcpyAsync(&now, c->sel_count + 2, sizeof now, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        *share = (double)(now - entries_before) / (double)n4;
        if (*share > q.share_max) return IBLNERF_OK;      // (the caller runs the whole batch instead)
    }
    refine.trip_margin = c->margin[which];
    return run_launch(c, s, q.on_list_precise.none() ? q.on_list : q.on_list_precise, which, refine);
}

// The offset copies of a pass (or the density-gradient rows of the autograd normal modes) into c->sig4.
static int run_offset_query(iblnerf_ctx* c, hipStream_t s, int which, int kind, const PassRays& p, bool keep_all_rows, bool main_listed, bool gt_normal) {
    const long R = p.R;
    const int S = p.S;
    const bool tilt = c->opt.normal_mode == IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION_EPSILON;
    const float eps = tilt ? c->opt.epsilon_direction : c->opt.epsilon;
    QueryPlan q = plan_offsets(c, which, kind, S, keep_all_rows, main_listed, gt_normal);
    if (!q.run) return IBLNERF_OK;
    MlpCall m;
    m.pts_per_ray = S;
    if (q.gradient) {
        // the autograd normal modes (normal_from_depth.py:16-52, :102-137): density and its position gradient at the main query's own
        // points (c->pts still holds them), rows [sigma, d sigma / d x, y, z]; pass A applies the chain rule through the compositing
        m.pts = c->pts; m.n_pts = R * S; m.out = c->sig4; m.out_stride = 4;
        return run_launch(c, s, q.whole, which, m);
    }
    m.n_pts = 4 * R * S; m.out = c->sig4;
    if (q.point_batch) {
        HIP_TRY(c, launch_make_points(tilt ? 2 : 1, p.ro, p.rd, p.z, p.z_stride, eps, R, S, c->pts, s));
        m.pts = c->pts;
        return run_launch(c, s, q.whole, which, m);
    }
    // the four offset copies are generated in the MLP kernel's input stage: no [4][R][S][3] batch (9.2 KB per ray on the fine grid)
    PointGen g;
    g.rays_o = p.ro; g.rays_d = p.rd; g.z = p.z; g.z_stride = p.z_stride; g.S = S; g.RS = (unsigned)(R * S); g.eps = eps;
    m.gen = &g;
    if (q.list) {
        double share = 0.0;
        if (int rc = offsets_on_lists(c, s, which, q, p, eps, g, q.open ? &share : nullptr)) return rc;
        if (!q.open) return IBLNERF_OK;
        c->xsel_fraction = share;
        if (share <= q.share_max) return IBLNERF_OK;
        m.count_flops = false;                               // (counted with the estimates)
    }
    return run_launch(c, s, q.whole, which, m);
}

// The reflected ray of a pass on the coarse grid (:439-446) into c->refl_raw [R, Sc, 13]; c->pts holds its points.
static int run_reflected_query(iblnerf_ctx* c, hipStream_t s, int which, long R, const float* zc, int zc_stride, bool keep_all_rows) {
    const int Sc = c->Sc;
    const QueryPlan q = plan_reflected(c, which, keep_all_rows);
    MlpCall m;
    m.pts_per_ray = Sc; m.n_pts = R * Sc;
    if (!q.list) {
        m.pts = c->pts; m.dirs = c->refl_d; m.out = c->refl_raw;
        return run_launch(c, s, q.whole, which, m);
    }
    int rc;
    if (q.cut1 > 0) rc = estimate_chunked(c, s, q.est, which, c->refl_o, c->refl_d, zc, zc_stride, Sc, R, false, 0.f, nullptr, q.cut0, q.cut1, c->chunk_t(q.t_min), FLOP_REFL);
    else {
        m.pts = c->pts; m.out = c->sig4; m.flop_per_point = FLOP_REFL;
        rc = run_launch(c, s, q.est, which, m);
    }
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->refl_raw, 0, (size_t)R * Sc * REFL_CH * sizeof(float), s));
    HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
    c->sel_candidates += R * Sc;
    HIP_TRY(c, launch_select_points(c->refl_o, c->refl_d, zc, zc_stride, c->sig4, 1, nullptr, R, Sc, c->margin[which], q.t_min, c->sel_pts, c->sel_index,
                                    c->sel_count, s, false, 0.f, c->refl_raw, REFL_CH, FLOP_REFL, nullptr, nullptr, list_slots(q.on_list), c->sel_est));
    MlpCall l;
    l.pts = c->sel_pts; l.dirs = c->refl_d; l.pts_per_ray = Sc; l.n_pts = R * Sc; l.out = c->refl_raw; l.count_flops = false; l.n_pts_dev = c->sel_count; l.out_index = c->sel_index;
    l.trip_margin = c->margin[which];
    return run_launch(c, s, q.on_list, which, l);
}

// One raw2outputs pass (ibl_nerf_renderer.py:153-527) over R rays of the current launch.
// zc / zc_stride: z_vals_constant, the coarse grid the reflected ray is sampled on (one shared row, or per-ray rows under perturb).
// keep_all_rows: a tapped call whose backward reads every raw row (a training step's forward without iblnerf_set_tapped_lists) — its main queries evaluate every sample.
static int full_pass(iblnerf_ctx* c, hipStream_t s, int which, int kind, const float* ro, const float* rd, long R, const float* z,
                     int z_stride, int S, float* weights, float near_, float far_, const OverrideArgs& ov,
                     const PassOutputs& out, const float* zc, int zc_stride, const float* noise,
                     float* env_tap = nullptr, const float* near_ray = nullptr, const float* far_ray = nullptr, bool keep_all_rows = false) {
    const PassRays p{ro, rd, R, z, z_stride, S, noise};
    int rc;
    // main query: pts = o + d z, view direction = rays_d (not the normalised viewdirs, :201)
    HIP_TRY(c, launch_make_points(0, ro, rd, z, z_stride, 0.f, R, S, c->pts, s));
    // (the probe of iblnerf_decide_route: first of all, are this network's plain-f16 estimates good enough?)
    if (c->deciding && !keep_all_rows && lists_possible(c, which, keep_all_rows) && c->est_f16 && which < 2 && !c->est_checked[which])
        if ((rc = check_estimates(c, s, which, R * S, S))) return rc;
    bool main_listed = false;
    if ((rc = run_main_query(c, s, which, kind, p, keep_all_rows, &main_listed))) return rc;
    // auxiliary PositionMLPs (:291-303): same points, trunk-shaped network, out_linears row as the head; each output
    // channel overwrites its column of the raw rows, so compositing and everything after it are unchanged
    for (int k = 0; k < 3; ++k)
        for (int ch = 0; c->aux_on[k] && ch < AUX_CHANNELS[k]; ++ch)
            if ((rc = run_mlp(c, s, VAR_TRUNK, AUX_SLOT0[k] + ch, c->pts, nullptr, S, R * S, c->raw + AUX_RAW_COLUMN[k] + ch, RAW_CH, Q_AUX))) return rc;
    const bool at_surface = c->opt.infer_normal_at_surface != 0;
    if (c->aux_on[IBLNERF_AUX_NORMAL] && at_surface)      // one point per ray: x_surface (:262, :268-271); refl_o is free until pass A
        HIP_TRY(c, launch_surface_points(ro, rd, z, z_stride, c->raw, noise, R, S, ov, c->refl_o, s));
    for (int ch = 0; c->aux_on[IBLNERF_AUX_NORMAL] && ch < 3; ++ch) {   // normal_mlp at the surface point or at every sample (:273)
        rc = at_surface ? run_mlp(c, s, VAR_TRUNK, AUX_SLOT0[IBLNERF_AUX_NORMAL] + ch, c->refl_o, nullptr, 1, R, c->nrm_raw + ch, 3, Q_AUX)
                        : run_mlp(c, s, VAR_TRUNK, AUX_SLOT0[IBLNERF_AUX_NORMAL] + ch, c->pts, nullptr, S, R * S, c->nrm_raw + ch, 3, Q_AUX);
        if (rc) return rc;
    }
    // finite-difference normal: 4 offset copies of the samples (normal_from_depth.py:139-158) or 4 rays with tilted directions
    // (:55-75), trunk only; none in the ground-truth normal mode
    if ((rc = run_offset_query(c, s, which, kind, p, keep_all_rows, main_listed, ov.gt_normal != nullptr))) return rc;
    PassAArgs a = pass_a_args(c, ro, rd, R, z, z_stride, S, c->raw, c->sig4, c->aux_on[IBLNERF_AUX_NORMAL] ? c->nrm_raw : nullptr,
                              weights, near_, far_, ov);
    a.noise = noise;
    a.near_ray = near_ray; a.far_ray = far_ray;
    HIP_TRY(c, launch_pass_a(a, out, c->opt.gamma_correct, s));
    // reflected ray through the same network, always on the coarse z grid (:439-446)
    HIP_TRY(c, launch_make_points(0, c->refl_o, c->refl_d, zc, zc_stride, 0.f, R, c->Sc, c->pts, s));
    if ((rc = run_reflected_query(c, s, which, R, zc, zc_stride, keep_all_rows))) return rc;
    PassBArgs b;
    b.state = c->state; b.refl_raw = c->refl_raw; b.refl_d = c->refl_d; b.zc = zc; b.zc_stride = zc_stride; b.Sc = c->Sc;
    b.gamma_correct = c->opt.gamma_correct; b.radiance_linear = c->opt.use_radiance_linear; b.out = out; b.R = R;
    b.env_tap = env_tap;
    HIP_TRY(c, launch_pass_b(b, s));
    return IBLNERF_OK;
}

// The coarse pass reduced to its density (options.coarse_outputs = 0: all the fine sampling needs from the coarse network) into c->sig4 [R, Sc]; c->pts holds the points.
// The same route as the full coarse pass's density: a plain-f16 estimate everywhere, the 15-slot form on the samples that can carry a weight.
static int density_pass(iblnerf_ctx* c, hipStream_t s, const float* ro, const float* rd, long R, const float* zc, int zcs, const float* noise) {
    const int Sc = c->Sc;
    const long n = R * Sc;
    int rc;
    bool est_ran = false;
    if (sigma_p_available(c, 0) && !c->p_all_points && (c->sel_decided ? c->sel_on : c->deciding)) {
        if (c->deciding && c->est_f16 && !c->est_checked[0] && (rc = check_estimates(c, s, 0, n, Sc))) return rc;
        const Launch est = pick_kernel(c, 0, VAR_TRUNK, Q_ESTIMATE, false);
        Launch p15;
        p15.kern = K_MX; p15.variant = VAR_TRUNK_P;
        if (density_fp32(c, 0)) { p15.kern = K_FP32; p15.variant = VAR_TRUNK; }
        MlpCall m;
        m.pts = c->pts; m.pts_per_ray = Sc; m.n_pts = n; m.out = c->sig4;
        if ((rc = run_launch(c, s, est, 0, m))) return rc;
        est_ran = true;
        HIP_TRY(c, hipMemsetAsync(c->sel_count, 0, sizeof(int), s));
        c->sel_candidates += n;
        HIP_TRY(c, launch_select_points(ro, rd, zc, zcs, c->sig4, 1, noise, R, Sc, c->margin[0], c->tmin_main, c->sel_pts, c->sel_index, c->sel_count, s,
                                        false, 0.f, nullptr, 0, FLOP_TRUNK, nullptr, nullptr, list_slots(p15), c->sel_est));
        if (!c->sel_decided) {      // (the probe: this is also where a checkpoint's refinement decision is taken when no full coarse pass ever runs)
            long n_sel = 0;
            if ((rc = read_list_length(c, s, &n_sel))) return rc;
            c->sel_decided = true;
            c->coarse_share = (double)n_sel / (double)n;
            c->sel_on = c->coarse_share <= SELECT_MAX_FRACTION;
        }
        if (c->sel_on) {
            MlpCall l;
            l.pts = c->sel_pts; l.pts_per_ray = Sc; l.n_pts = n; l.out = c->sig4; l.count_flops = false; l.n_pts_dev = c->sel_count; l.out_index = c->sel_index;
            l.trip_margin = c->margin[0];
            return run_launch(c, s, p15, 0, l);
        }
    }
    if (density_fp32(c, 0) && sigma_p_available(c, 0) && whole_batch_fp32(c)) {      // (plan_main's rule for the whole-batch density, here for the density-only coarse pass)
        Launch f32;
        f32.kern = K_FP32; f32.variant = VAR_TRUNK;
        MlpCall m;
        m.pts = c->pts; m.pts_per_ray = Sc; m.n_pts = n; m.out = c->sig4; m.count_flops = !est_ran;
        return run_launch(c, s, f32, 0, m);
    }
    return run_mlp(c, s, VAR_TRUNK, 0, c->pts, nullptr, Sc, n, c->sig4, 1, Q_MAIN_COARSE, nullptr, !est_ran);   // (algorithmic FLOPs: once per query)
}

// Validates the caller's overrides (the reference's asserts at ibl_nerf_renderer.py:222, :232) and copies the scalar part.
static int parse_overrides(iblnerf_ctx* c, const iblnerf_overrides* ovr, OverrideArgs& ov, const float*& gt_normal) {
    std::memset(&ov, 0, sizeof ov);
    gt_normal = nullptr;
    if (ovr && ovr->mode != 0) {
        if (ovr->mode != 1 && ovr->mode != 2) return c->fail(IBLNERF_ERR_INVALID, "overrides.mode must be 0, 1 or 2");
        if (ovr->num_objects <= 0 || ovr->num_objects > 8)   // reference: assert num_*_objects > 0 (:222, :232)
            return c->fail(IBLNERF_ERR_INVALID, "num_edit_objects / num_insert_objects must be in 1..8");
        if (!ovr->d_mask) return c->fail(IBLNERF_ERR_INVALID, "overrides need the mask image rows");
        if (ovr->mode == 2 && (!ovr->d_depth || !ovr->d_normal))
            return c->fail(IBLNERF_ERR_INVALID, "insert_object needs object_insert_depth and object_insert_normal rows");
        if (ovr->mode == 1 && ovr->edit_depth && !ovr->d_depth) return c->fail(IBLNERF_ERR_INVALID, "edit_depth needs edit_depth rows");
        if (ovr->mode == 1 && ovr->edit_normal && !ovr->d_normal) return c->fail(IBLNERF_ERR_INVALID, "edit_normal needs edit_normal rows");
        if (ovr->mode == 1 && ovr->edit_albedo && ovr->edit_albedo_by_img && !ovr->d_albedo)
            return c->fail(IBLNERF_ERR_INVALID, "edit_albedo_by_img needs edit_albedo rows");
        if (ovr->n_roughness_list < 0 || ovr->n_roughness_list > 8) return c->fail(IBLNERF_ERR_INVALID, "roughness list too long");
        if (ovr->edit_roughness_by_img && (ovr->mode != 1 || !ovr->edit_roughness || !ovr->d_roughness))
            return c->fail(IBLNERF_ERR_INVALID, "edit_roughness_by_img needs edit_intrinsic, edit_roughness and the per-ray d_roughness rows");
        ov.mode = ovr->mode; ov.num_objects = ovr->num_objects;
        ov.edit_depth = ovr->edit_depth; ov.edit_normal = ovr->edit_normal; ov.edit_albedo = ovr->edit_albedo;
        ov.edit_albedo_by_img = ovr->edit_albedo_by_img; ov.edit_roughness = ovr->edit_roughness;
        ov.n_rough_list = ovr->n_roughness_list;
        ov.mask_stride = 3; ov.depth_stride = 1;
        std::memcpy(ov.rough_list, ovr->roughness_list, sizeof ov.rough_list);
        std::memcpy(ov.albedo_list, ovr->albedo_list, sizeof ov.albedo_list);
        std::memcpy(ov.irr_list, ovr->irradiance_list, sizeof ov.irr_list);
    }
    if (c->opt.normal_mode == IBLNERF_NORMAL_GROUND_TRUTH) {
        if (!ovr || !ovr->d_gt_normal)
            return c->fail(IBLNERF_ERR_INVALID, "normal_mode ground_truth needs overrides.d_gt_normal (gt_values[\"normal\"] rows)");
        gt_normal = ovr->d_gt_normal;
    }
    if (c->opt.normal_mode == IBLNERF_NORMAL_INFERRED && !c->aux_on[IBLNERF_AUX_NORMAL])
        return c->fail(IBLNERF_ERR_STATE, "normal_mode inferred needs a normal_mlp (iblnerf_upload_aux_weights, IBLNERF_AUX_NORMAL)");
    return IBLNERF_OK;
}

// The image-row pointers of the overrides, advanced to ray r0 of the call.
static OverrideArgs rows_from(const OverrideArgs& ov, const iblnerf_overrides* ovr, const float* gt_normal, long r0) {
    OverrideArgs o = ov;
    o.gt_normal = gt_normal ? gt_normal + 3 * r0 : nullptr;
    if (ovr) {
        o.gt_albedo = ovr->d_gt_albedo ? ovr->d_gt_albedo + 3 * r0 : nullptr;
        o.gt_roughness = ovr->d_gt_roughness ? ovr->d_gt_roughness + r0 : nullptr;
        o.gt_irradiance = ovr->d_gt_irradiance ? ovr->d_gt_irradiance + 3 * r0 : nullptr;
        o.gt_depth = ovr->d_gt_depth ? ovr->d_gt_depth + r0 : nullptr;
    }
    if (o.mode) {
        o.mask = ovr->d_mask + 3 * r0;
        o.depth_img = ovr->d_depth ? ovr->d_depth + r0 : nullptr;
        o.normal_img = ovr->d_normal ? ovr->d_normal + 3 * r0 : nullptr;
        o.albedo_img = ovr->d_albedo ? ovr->d_albedo + 3 * r0 : nullptr;
        o.rough_img = (ovr->edit_roughness_by_img && ovr->d_roughness) ? ovr->d_roughness + r0 : nullptr;
    }
    return o;
}

int iblnerf_render_rays(iblnerf_ctx* c, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays,
                        float near_, float far_, const iblnerf_overrides* ovr, const iblnerf_outputs* outs) {
    return iblnerf_render_rays_sampled(c, stream, d_rays_o, d_rays_d, n_rays, near_, far_, ovr, nullptr, outs);
}

int iblnerf_render_rays_sampled(iblnerf_ctx* c, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays,
                                float near_, float far_, const iblnerf_overrides* ovr, const iblnerf_sampling* smp,
                                const iblnerf_outputs* outs) {
    return iblnerf_render_rays_tapped(c, stream, d_rays_o, d_rays_d, n_rays, near_, far_, ovr, smp, outs, nullptr);
}

int iblnerf_render_rays_tapped(iblnerf_ctx* c, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays,
                               float near_, float far_, const iblnerf_overrides* ovr, const iblnerf_sampling* smp,
                               const iblnerf_outputs* outs, const iblnerf_taps* taps) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (taps && !c->opt.coarse_outputs && c->opt.n_importance > 0)
        return c->fail(IBLNERF_ERR_STATE, "render_rays_tapped: the coarse pass's raw rows exist only with options.coarse_outputs");
    if (n_rays < 0 || !outs) return c->fail(IBLNERF_ERR_INVALID, "render_rays: negative ray count / null outputs");
    if (n_rays == 0) return IBLNERF_OK;   // empty batch: torch hands out null data pointers for zero-row tensors
    if (!d_rays_o || !d_rays_d) return c->fail(IBLNERF_ERR_INVALID, "render_rays: null rays");
    const bool fine = c->opt.n_importance > 0;
    const float* t_rand = smp ? smp->d_t_rand : nullptr;
    const float* u_rand = smp ? smp->d_u : nullptr;
    const float* noise_c = smp ? smp->d_noise_coarse : nullptr;      // raw_noise_std > 0: per-sample density noise of each pass
    const float* noise_f = smp ? smp->d_noise_fine : nullptr;
    if ((t_rand == nullptr) != (u_rand == nullptr) && fine)   // perturb > 0 switches both (det = (perturb == 0), :703)
        return c->fail(IBLNERF_ERR_INVALID, "render_rays: d_t_rand and d_u go together (perturb > 0 jitters the grid and draws the fine samples)");
    const float* near_ray = smp ? smp->d_near : nullptr;             // per-ray planes (:802-805 with [n, 1] tensors)
    const float* far_ray = smp ? smp->d_far : nullptr;
    if ((near_ray == nullptr) != (far_ray == nullptr)) return c->fail(IBLNERF_ERR_INVALID, "render_rays: d_near and d_far go together");
    if (!c->have_net[0]) return c->fail(IBLNERF_ERR_STATE, "render_rays: network_fn weights not uploaded");
    if (!c->have_lut) return c->fail(IBLNERF_ERR_STATE, "render_rays: brdf_lut not uploaded");
    const int fine_net = c->have_net[1] ? 1 : 0;   // run_fn = network_fn if network_fine is None (:705)
    OverrideArgs ov;
    const float* gt_normal = nullptr;
    if (int rc = parse_overrides(c, ovr, ov, gt_normal)) return rc;
    const int irr_ch = (ovr && ovr->d_gt_irradiance) ? 3 : 1;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    c->ev_used = 0;
    c->flop_alg = 0.0;
    c->sel_candidates = 0;
    c->flop_exec = 0.0;
    c->slot_units = 0.0;
    HIP_TRY(c, hipMemsetAsync(c->sel_count + 2, 0, 8 * sizeof(int), s));
    const int Sc = c->Sc, Sf = c->Sf;
    // a tapped call's backward reads every main raw row — unless the caller said its backward needs the live rows only (iblnerf_set_tapped_lists) and a route is decided
    const bool keep_rows = taps != nullptr && !(c->tapped_lists && c->route_decided);
    // iblnerf_set_lists(ctx, 0): this call runs as a context whose lists are off does (every query evaluates all of its samples), whatever route is decided
    struct ListsOff {
        iblnerf_ctx* c; bool on, decided, sel;
        explicit ListsOff(iblnerf_ctx* c_) : c(c_), on(c_->lists_off && !c_->deciding), decided(c_->sel_decided), sel(c_->sel_on) { if (on) { c->sel_decided = true; c->sel_on = false; } }
        ~ListsOff() { if (on) { c->sel_decided = decided; c->sel_on = sel; } c->trip_rays = nullptr; }
    } lists_guard(c);
    HIP_TRY(c, launch_coarse_z(near_, far_, Sc, c->opt.lindisp, c->zc, s));
    if ((t_rand || near_ray) && !c->zc_ray) HIP_TRY(c, hipMalloc((void**)&c->zc_ray, (size_t)c->ws_rays * Sc * sizeof(float)));
    // equal-sized launches (a short tail launch would leave most of the persistent grid idle)
    const long n_launch = (n_rays + c->ws_rays - 1) / c->ws_rays;
    const long per_launch = n_launch ? (n_rays + n_launch - 1) / n_launch : 0;
    for (long r0 = 0; r0 < n_rays; r0 += per_launch) {
        const long R = (n_rays - r0 < per_launch) ? n_rays - r0 : per_launch;
        const float* ro = d_rays_o + 3 * r0;
        const float* rd = d_rays_d + 3 * r0;
        const OverrideArgs o = rows_from(ov, ovr, gt_normal, r0);
        // the coarse grid of this launch: one shared row, or per-ray rows after the stratified jitter (:678-692)
        const float* zc = c->zc;
        int zcs = 0;
        if (near_ray) {   // every ray its own grid (and its jitter)
            HIP_TRY(c, launch_ray_grid(near_ray + r0, far_ray + r0, Sc, c->opt.lindisp, t_rand ? t_rand + r0 * Sc : nullptr, R, c->zc_ray, s));
            zc = c->zc_ray;
            zcs = Sc;
        } else if (t_rand) {
            HIP_TRY(c, launch_jitter_z(c->zc, Sc, t_rand + r0 * Sc, R, c->zc_ray, s));
            zc = c->zc_ray;
            zcs = Sc;
        }
        const float* nr = near_ray ? near_ray + r0 : nullptr;
        const float* fr = far_ray ? far_ray + r0 : nullptr;
        c->trip_rays = outs->trip_rays ? outs->trip_rays + r0 : nullptr;      // (k_tripwire marks the rays of this launch whose estimates were thin)
        c->cur_R = R;
        int rc;
        // taps (iblnerf_render_rays_tapped): this launch's z rows and main raw rows, out of the workspace before the next pass reuses it
        auto tap_z = [&](float* dst, const float* z, int zstride, int S) -> int {
            if (!dst) return IBLNERF_OK;
            if (zstride == 0) HIP_TRY(c, launch_broadcast_rows(z, S, R, dst + r0 * S, s));
            else HIP_TRY(c, hipMemcpyAsync(dst + r0 * S, z, (size_t)R * S * sizeof(float), hipMemcpyDeviceToDevice, s));   // (per-ray rows are contiguous: stride = S)
            return IBLNERF_OK;
        };
        auto tap_raw = [&](float* dst, int S) -> int {
            if (!dst) return IBLNERF_OK;
            HIP_TRY(c, hipMemcpyAsync(dst + r0 * S * RAW_CH, c->raw, (size_t)R * S * RAW_CH * sizeof(float), hipMemcpyDeviceToDevice, s));
            return IBLNERF_OK;
        };
        if (!fine) {
            rc = full_pass(c, s, 0, PASS_SINGLE, ro, rd, R, zc, zcs, Sc, c->w_c, near_, far_, o, slice_maps(outs->fine, r0, Sc, irr_ch), zc, zcs,
                           noise_c ? noise_c + r0 * Sc : nullptr, taps && taps->d_env_coarse ? taps->d_env_coarse + r0 * 12 : nullptr, nr, fr, keep_rows);
            if (rc) return rc;
            if (taps && ((rc = tap_z(taps->d_z_coarse, zc, zcs, Sc)) || (rc = tap_raw(taps->d_raw_coarse, Sc)))) return rc;
            continue;
        }
        if (c->opt.coarse_outputs) {
            rc = full_pass(c, s, 0, PASS_COARSE, ro, rd, R, zc, zcs, Sc, c->w_c, near_, far_, o, slice_maps(outs->coarse, r0, Sc, irr_ch), zc, zcs,
                           noise_c ? noise_c + r0 * Sc : nullptr, taps && taps->d_env_coarse ? taps->d_env_coarse + r0 * 12 : nullptr, nr, fr, keep_rows);
            if (rc) return rc;
            if (taps && ((rc = tap_z(taps->d_z_coarse, zc, zcs, Sc)) || (rc = tap_raw(taps->d_raw_coarse, Sc)))) return rc;
        } else {   // density only: all the fine sampling needs from the coarse network
            HIP_TRY(c, launch_make_points(0, ro, rd, zc, zcs, 0.f, R, Sc, c->pts, s));
            const float* nz = noise_c ? noise_c + r0 * Sc : nullptr;
            if ((rc = density_pass(c, s, ro, rd, R, zc, zcs, nz))) return rc;
            HIP_TRY(c, launch_sigma_weights(rd, zc, zcs, c->sig4, nz, R, Sc, c->w_c, s));
        }
        HIP_TRY(c, launch_fine_z(zc, zcs, Sc, c->w_c, R, c->opt.n_importance, u_rand ? u_rand + r0 * c->opt.n_importance : nullptr, c->z_fine,
                                 outs->z_std ? outs->z_std + r0 : nullptr, s));
        rc = full_pass(c, s, fine_net, PASS_FINE, ro, rd, R, c->z_fine, Sf, Sf, c->w_f, near_, far_, o, slice_maps(outs->fine, r0, Sf, irr_ch), zc, zcs,
                       noise_f ? noise_f + r0 * Sf : nullptr, taps && taps->d_env_fine ? taps->d_env_fine + r0 * 12 : nullptr, nr, fr, keep_rows);
        if (rc) return rc;
        if (taps && ((rc = tap_z(taps->d_z_fine, c->z_fine, Sf, Sf)) || (rc = tap_raw(taps->d_raw_fine, Sf)))) return rc;
    }
    if (c->posdir_out_ch && outs->inferred_depth_map) {   // infer_depth (:722-726): depth_mlp(rays_o, viewdirs), relu of output 0
        if (c->posdir_out_ch != 1) return c->fail(IBLNERF_ERR_STATE, "render_rays: the depth_mlp must have one output (ibl_nerf.py:294-296)");
        PosDirArgs a{c->d_posdir, d_rays_o, d_rays_d, outs->inferred_depth_map, (long)n_rays, 1, 1, 1};
        HIP_TRY(c, launch_posdir_mlp(a, s));
    }
    return arm_range_snapshot(c, s);
}

int iblnerf_composite_pass(iblnerf_ctx* c, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_,
                           float far_, const iblnerf_overrides* ovr, const iblnerf_stage_inputs* in, const iblnerf_maps* maps) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (n_rays < 0 || !in || !maps) return c->fail(IBLNERF_ERR_INVALID, "composite_pass: negative ray count / null inputs / null maps");
    if (n_rays == 0) return IBLNERF_OK;
    if (n_rays > c->ws_rays) return c->fail(IBLNERF_ERR_INVALID, "composite_pass: at most max_rays_per_launch (%ld) rays per call", c->ws_rays);
    const int S = in->n_samples;
    if (S < 1 || S > c->Smax) return c->fail(IBLNERF_ERR_INVALID, "composite_pass: n_samples must be in 1..%d", c->Smax);
    if (!d_rays_o || !d_rays_d || !in->d_z || !in->d_raw || !in->d_refl_raw)
        return c->fail(IBLNERF_ERR_INVALID, "composite_pass: rays, d_z, d_raw and d_refl_raw are required");
    if (!c->have_lut) return c->fail(IBLNERF_ERR_STATE, "composite_pass: brdf_lut not uploaded");
    OverrideArgs ov;
    const float* gt_normal = nullptr;
    if (int rc = parse_overrides(c, ovr, ov, gt_normal)) return rc;
    if (c->opt.normal_mode == IBLNERF_NORMAL_DEPTH_GRADIENT || c->opt.normal_mode == IBLNERF_NORMAL_DEPTH_GRADIENT_DIRECTION)
        return c->fail(IBLNERF_ERR_STATE, "composite_pass: not built for the two depth-gradient normal modes (their pass reads [n, S, 4] density-gradient "
                                          "rows, not the [4, n, S] offset densities this entry takes)");
    const bool offsets = !gt_normal && c->opt.normal_mode != IBLNERF_NORMAL_INFERRED;
    if (offsets && !in->d_sigma_offsets) return c->fail(IBLNERF_ERR_INVALID, "composite_pass: this normal mode needs d_sigma_offsets");
    if (c->opt.normal_mode == IBLNERF_NORMAL_INFERRED && !in->d_normal_raw)
        return c->fail(IBLNERF_ERR_INVALID, "composite_pass: normal_mode inferred needs d_normal_raw");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, launch_coarse_z(near_, far_, c->Sc, c->opt.lindisp, c->zc, s));
    const OverrideArgs o = rows_from(ov, ovr, gt_normal, 0);
    const int irr_ch = (ovr && ovr->d_gt_irradiance) ? 3 : 1;
    const PassOutputs out = slice_maps(*maps, 0, S, irr_ch);
    PassAArgs a = pass_a_args(c, d_rays_o, d_rays_d, n_rays, in->d_z, S, S, in->d_raw, in->d_sigma_offsets, in->d_normal_raw, c->w_f,
                              near_, far_, o);
    a.stage = in->d_stage;
    HIP_TRY(c, launch_pass_a(a, out, c->opt.gamma_correct, s));
    if (in->d_refl_o) HIP_TRY(c, hipMemcpyAsync(in->d_refl_o, c->refl_o, (size_t)n_rays * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (in->d_refl_d) HIP_TRY(c, hipMemcpyAsync(in->d_refl_d, c->refl_d, (size_t)n_rays * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    PassBArgs b;
    b.state = c->state; b.refl_raw = in->d_refl_raw; b.refl_d = c->refl_d; b.zc = c->zc; b.Sc = c->Sc;
    b.gamma_correct = c->opt.gamma_correct; b.radiance_linear = c->opt.use_radiance_linear; b.out = out; b.R = n_rays;
    HIP_TRY(c, launch_pass_b(b, s));
    return IBLNERF_OK;
}

int iblnerf_last_selection(iblnerf_ctx* c, int64_t* n_selected, int64_t* n_candidates) {
    if (!c || !n_selected || !n_candidates) return IBLNERF_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipDeviceSynchronize());
    unsigned long long tot = 0;
    HIP_TRY(c, hipMemcpy(&tot, c->sel_count + 2, sizeof tot, hipMemcpyDeviceToHost));
    *n_selected = (int64_t)tot;
    *n_candidates = (int64_t)c->sel_candidates;
    return IBLNERF_OK;
}

int iblnerf_estimate_policy(iblnerf_ctx* c, int which, int* checked, int* plain_f16) {
    if (!c || which < 0 || which > 1 || !checked || !plain_f16) return IBLNERF_ERR_INVALID;
    *checked = c->est_checked[which] ? 1 : 0;
    *plain_f16 = (c->est_f16 && c->est_checked[which] && c->est_ok[which]) ? 1 : 0;
    return IBLNERF_OK;
}

int iblnerf_last_executed_flops(iblnerf_ctx* c, double* flop_executed) {
    if (!c || !flop_executed) return IBLNERF_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipDeviceSynchronize());
    double on_lists = 0.0;
    HIP_TRY(c, hipMemcpy(&on_lists, c->sel_count + 4, sizeof on_lists, hipMemcpyDeviceToHost));
    *flop_executed = c->flop_exec + on_lists;
    return IBLNERF_OK;
}

int iblnerf_set_chunk_cuts(iblnerf_ctx* c, int fine_cut0, int fine_cut1, int refl_cut0, int refl_cut1) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (fine_cut0 < 0 || fine_cut1 < fine_cut0 || refl_cut0 < 0 || refl_cut1 < refl_cut0 || (fine_cut1 > 0 && fine_cut0 < 1) || (refl_cut1 > 0 && refl_cut0 < 1))
        return c->fail(IBLNERF_ERR_INVALID, "set_chunk_cuts: 1 <= cut0 <= cut1 (0, 0: the built-in cuts)");
    c->cuts_fine[0] = fine_cut0; c->cuts_fine[1] = fine_cut1; c->cuts_refl[0] = refl_cut0; c->cuts_refl[1] = refl_cut1;
    return IBLNERF_OK;
}

int iblnerf_set_select_tmin(iblnerf_ctx* c, float t_main, float t_offsets, float t_chunk) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!(t_main > 0.f && t_main < 1.f && t_offsets > 0.f && t_offsets < 1.f && t_chunk >= 0.f && t_chunk < 1.f))
        return c->fail(IBLNERF_ERR_INVALID, "set_select_tmin: thresholds must lie in (0, 1) (t_chunk 0: each query's own)");
    if (!(t_chunk <= t_offsets && t_offsets <= t_main))      // (a chunk threshold above the copies' own leaves samples without an estimate that the selection still audits)
        return c->fail(IBLNERF_ERR_INVALID, "set_select_tmin: thresholds must be ordered t_chunk <= t_offsets <= t_main");
    c->tmin_main = t_main; c->tmin_offsets = t_offsets; c->tmin_chunk = t_chunk;
    return IBLNERF_OK;
}

int iblnerf_set_tier_thresholds(iblnerf_ctx* c, float tau_offsets, float tau_main) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!(tau_offsets >= 0.0f) || !(tau_main >= 0.0f)) return c->fail(IBLNERF_ERR_INVALID, "set_tier_thresholds: thresholds >= 0 (0 = the built-in ones)");
    c->tier_tau = tau_offsets;
    c->tier_tau_main = tau_main;
    return IBLNERF_OK;
}

int iblnerf_set_offset_tier_threshold(iblnerf_ctx* c, float tau) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!(tau >= 0.0f)) return c->fail(IBLNERF_ERR_INVALID, "set_offset_tier_threshold: tau >= 0 (0 = no tiers)");
    c->tier_tau = tau;
    return IBLNERF_OK;
}


int iblnerf_last_slot_units(iblnerf_ctx* c, double* slot_units) {
    if (!c || !slot_units) return IBLNERF_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->opt.device));
    HIP_TRY(c, hipDeviceSynchronize());
    double on_lists = 0.0;
    HIP_TRY(c, hipMemcpy(&on_lists, c->sel_count + 8, sizeof on_lists, hipMemcpyDeviceToHost));
    *slot_units = c->slot_units + on_lists;
    return IBLNERF_OK;
}

static void fill_route(const iblnerf_ctx* c, iblnerf_route* r) {
    std::memset(r, 0, sizeof *r);
    r->decided = c->route_decided ? 1 : 0;
    for (int w = 0; w < 2; ++w) r->estimates_plain_f16[w] = (c->est_f16 && c->est_checked[w] && c->est_ok[w]) ? 1 : 0;
    r->tripped = c->tripped;
    r->coarse_share = c->coarse_share;
    r->fine_main_share = c->fsel_fraction;
    r->fine_offsets_share = c->xsel_fraction;
    for (int w = 0; w < 2; ++w) { r->select_margin[w] = c->margin[w]; r->estimate_error[w] = c->est_error[w]; }
}

int iblnerf_get_route(iblnerf_ctx* c, iblnerf_route* out) {
    if (!c || !out) return IBLNERF_ERR_INVALID;
    fill_route(c, out);
    return IBLNERF_OK;
}

int iblnerf_copy_route(iblnerf_ctx* dst, const iblnerf_ctx* src) {
    if (!dst || !src) return IBLNERF_ERR_INVALID;
    // the route exactly as `src` holds it, ladder steps included (iblnerf_set_route of iblnerf_get_route's struct would lose them: a margin doubled by
    // iblnerf_escalate_route keeps its plain-f16 estimates, a route imposed with tripped = 1 does not)
    dst->route_decided = src->route_decided;
    dst->tripped = src->tripped;
    dst->sel_decided = src->sel_decided; dst->sel_on = src->sel_on; dst->coarse_share = src->coarse_share;
    dst->fsel_fraction = src->fsel_fraction; dst->xsel_fraction = src->xsel_fraction;
    for (int w = 0; w < 2; ++w) {
        dst->est_checked[w] = src->est_checked[w]; dst->est_ok[w] = src->est_ok[w];
        dst->margin[w] = src->margin[w]; dst->est_error[w] = src->est_error[w];
    }
    return IBLNERF_OK;
}

int iblnerf_set_route(iblnerf_ctx* c, const iblnerf_route* r) {
    if (!c || !r) return IBLNERF_ERR_INVALID;
    reset_route(c, 0);
    if (!r->decided) return IBLNERF_OK;
    c->route_decided = true;
    c->tripped = r->tripped;
    for (int w = 0; w < 2; ++w) { c->est_checked[w] = true; c->est_ok[w] = r->estimates_plain_f16[w] != 0 && !c->tripped; }
    c->sel_decided = true;
    c->coarse_share = r->coarse_share;
    c->sel_on = r->coarse_share >= 0.0 && r->coarse_share <= SELECT_MAX_FRACTION && c->tripped < 2;
    // (a share that was not measured switches its list off: 2.0 is above every threshold)
    c->fsel_fraction = r->fine_main_share >= 0.0 ? r->fine_main_share : 2.0;
    c->xsel_fraction = r->fine_offsets_share >= 0.0 ? r->fine_offsets_share : 2.0;
    for (int w = 0; w < 2; ++w) {
        c->margin[w] = (r->select_margin[w] >= COARSE_SELECT_MARGIN && r->select_margin[w] <= MARGIN_MAX) ? r->select_margin[w] : COARSE_SELECT_MARGIN;
        c->est_error[w] = r->estimate_error[w];
    }
    return IBLNERF_OK;
}

// The ladder a tripped PROBE climbs, in order of what a step costs: (1) an UNDERESTIMATE on plain-f16 estimates (bit 2 alone) doubles both selection margins, up to
// MARGIN_MAX — a few more samples refined; (2) otherwise the plain-f16 estimates go: f16 + 2 fp6 from now on (six matrix slots instead of four per estimate, nothing
// else changes); (3) on f16 + 2 fp6 estimates already (a network whose density cancels beyond THAT form's 2^-16: tests/test_gpu_fitted.py builds one) no estimate of
// this checkpoint can be trusted — the lists go off, every query evaluates all of its samples.
int iblnerf_escalate_route(iblnerf_ctx* c, int trip_bits) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!(trip_bits & 28)) return c->fail(IBLNERF_ERR_INVALID, "escalate_route: trip_bits carries none of bits 2, 3, 4 of the range flags");
    if (!c->route_decided) return c->fail(IBLNERF_ERR_STATE, "escalate_route: no route is decided on this context");
    // (a network the probe refused plain-f16 estimates for counts as "on six slots already" only if the other one is there too, or never takes a list: ADVICE r5)
    const bool plain0 = est_plain(c, 0), plain1 = est_plain(c, c->have_net[1] ? 1 : 0);
    if (!plain0 && !plain1) {
        c->est_ok[0] = c->est_ok[1] = false;
        c->sel_decided = true; c->sel_on = false;
        c->tripped = 2;
    } else if (!(trip_bits & 8) && std::min(c->margin[0], c->margin[1]) < MARGIN_MAX) {
        for (int w = 0; w < 2; ++w) c->margin[w] = std::min(MARGIN_MAX, 2.0f * c->margin[w]);
        c->tripped = std::max(c->tripped, 1);
    } else {
        c->est_ok[0] = c->est_ok[1] = false;
        c->margin[0] = c->margin[1] = COARSE_SELECT_MARGIN;
        c->tripped = std::max(c->tripped, 1);
    }
    return IBLNERF_OK;
}

int iblnerf_set_lists(iblnerf_ctx* c, int enabled) {
    if (!c) return IBLNERF_ERR_INVALID;
    c->lists_off = enabled == 0;
    return IBLNERF_OK;
}

int iblnerf_set_tapped_lists(iblnerf_ctx* c, int enabled) {
    if (!c) return IBLNERF_ERR_INVALID;
    c->tapped_lists = enabled != 0;
    return IBLNERF_OK;
}

int iblnerf_decide_route(iblnerf_ctx* c, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_, float far_, iblnerf_route* out) {
    return iblnerf_decide_route_outputs(c, stream, d_rays_o, d_rays_d, n_rays, near_, far_, nullptr, nullptr, out);
}

int iblnerf_decide_route_outputs(iblnerf_ctx* c, void* stream, const float* d_rays_o, const float* d_rays_d, int64_t n_rays, float near_, float far_,
                                 const iblnerf_overrides* ovr, const iblnerf_outputs* outs_or_null, iblnerf_route* out) {
    if (!c) return IBLNERF_ERR_INVALID;
    if (!d_rays_o || !d_rays_d || n_rays < SELECT_MIN_RAYS || n_rays > c->ws_rays)
        return c->fail(IBLNERF_ERR_INVALID, "decide_route: needs %ld <= n_rays <= max_rays_per_launch (%ld) probe rays", SELECT_MIN_RAYS, c->ws_rays);
    reset_route(c, 0);
    iblnerf_outputs outs;                          // every map null: the probe's results are discarded ...
    std::memset(&outs, 0, sizeof outs);
    if (outs_or_null) outs = *outs_or_null;        // ... or kept: the probe's render under the table in effect, for a caller that compares tables on the same rays
    outs.trip_rays = nullptr;                      // (a probe that trips escalates its route: iblnerf_escalate_route)
    c->deciding = true;
    const int rc = iblnerf_render_rays_tapped(c, stream, d_rays_o, d_rays_d, n_rays, near_, far_, outs_or_null ? ovr : nullptr, nullptr, &outs, nullptr);
    c->deciding = false;
    if (rc) { reset_route(c, 0); return rc; }
    // whatever the probe did not reach stays off: from here on nothing is decided inside a render call
    c->route_decided = true;
    if (!c->sel_decided) { c->sel_decided = true; c->sel_on = false; }
    for (int w = 0; w < 2; ++w)
        if (!c->est_checked[w]) { c->est_checked[w] = true; c->est_ok[w] = false; }
    if (c->fsel_fraction < 0.0) c->fsel_fraction = 2.0;
    if (c->xsel_fraction < 0.0) c->xsel_fraction = 2.0;
    if (out) fill_route(c, out);
    return IBLNERF_OK;
}

int iblnerf_describe_route(iblnerf_ctx* c, char* buf, size_t n) {
    if (!c || (!buf && n)) return -1;
    std::string t;
    char line[512];
    auto add = [&](const char* fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(line, sizeof line, fmt, ap);
        va_end(ap);
        t += line;
    };
    add("route: %s; estimates plain f16: coarse net %d (margin %.1f), fine net %d (margin %.1f)%s; relevant shares on the probe: coarse grid %.3f, fine main %.3f, fine offsets %.3f\n",
        c->route_decided ? "decided" : "NOT decided (every query evaluates all of its samples)", (int)est_plain(c, 0), (double)c->margin[0], (int)est_plain(c, 1), (double)c->margin[1], c->tripped == 2 ? " (tripwire fired twice: lists off)" : c->tripped ? " (tripwire fired)" : "",
        c->coarse_share, c->fsel_fraction, c->xsel_fraction);
    const bool fine = c->opt.n_importance > 0;
    const int fine_net = c->have_net[1] ? 1 : 0;
    auto row = [&](const char* pass, const char* query, const QueryPlan& q, int S) {
        if (!q.run) { add("%-7s %-10s none\n", pass, query); return; }
        if (!q.list) {
            add("%-7s %-10s whole batch: %s", pass, query, launch_name(q.whole));
            if (!q.density.none()) add(" + density on %s (%s)", launch_name(q.density), q.density_on_list ? "the samples it selects" : "every sample");
            add("%s\n", q.gradient ? " [density gradient]" : q.point_batch ? " [point batch]" : "");
            return;
        }
        add("%-7s %-10s estimate: %s", pass, query, launch_name(q.est));
        if (q.predicted) add(" on the samples outside the main ray's relevant range only (predicted range -> list directly%s)", q.tiers ? ", in two tiers" : "");
        else if (q.cut1 > 0) add(" in z-chunks [0,%d) [%d,%d) [%d,%d)", q.cut0, q.cut0, q.cut1, q.cut1, S);
        else add(" on every sample");
        add("; select T > %.0e; list: %s", (double)q.t_min, launch_name(q.on_list));
        if (q.tiers && q.predicted) add(" | flagged samples (T dist |depth - z| > %.0e): %s", (double)(c->tier_tau > 0.0f ? c->tier_tau : TIER_TAU_OFFSETS), launch_name(q.on_list_precise));
        if (q.tiers && !q.predicted) add(" | samples of weight > %.0e: %s", (double)(c->tier_tau_main > 0.0f ? c->tier_tau_main : TIER_TAU_MAIN), launch_name(q.on_list_precise));
        if (!q.on_list_precise.none()) add(" | own selection outside the predicted range: %s", launch_name(q.on_list_precise));
        if (!q.density_list.none()) add(" + %s", launch_name(q.density_list));
        add("\n");
    };
    struct { const char* name; int which, kind, S; bool on; } passes[] = {
        {"coarse", 0, fine ? PASS_COARSE : PASS_SINGLE, c->Sc, !fine || c->opt.coarse_outputs != 0}, {"fine", fine_net, PASS_FINE, c->Sf, fine}};
    for (auto& p : passes) {
        if (!p.on) { if (fine) add("coarse  density    %s\n", (sigma_p_available(c, 0) && !c->p_all_points && c->sel_decided && c->sel_on) ? "estimate + 15-slot list" : "whole batch"); continue; }
        const QueryPlan m = plan_main(c, p.which, p.kind, p.S, false);
        row(p.name, "main", m, p.S);
        row(p.name, "offsets", plan_offsets(c, p.which, p.kind, p.S, false, m.list || m.density_on_list, c->opt.normal_mode == IBLNERF_NORMAL_GROUND_TRUTH), p.S);
        row(p.name, "reflected", plan_reflected(c, p.which, false), c->Sc);
    }
    if (buf && n) { std::strncpy(buf, t.c_str(), n - 1); buf[n - 1] = 0; }
    return (int)t.size();
}

int iblnerf_set_profiling(iblnerf_ctx* c, int enabled) {
    if (!c) return IBLNERF_ERR_INVALID;
    c->profiling = enabled != 0;
    return IBLNERF_OK;
}

int iblnerf_last_mlp_time(iblnerf_ctx* c, float* ms_total, int* n_launches, double* flop_algorithmic) {
    if (!c) return IBLNERF_ERR_INVALID;
    float tot = 0.f;
    for (size_t i = 0; i < c->ev_used; ++i) {
        HIP_TRY(c, hipEventSynchronize(c->ev_pool[i].second));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second));
        tot += ms;
    }
    if (ms_total) *ms_total = tot;
    if (n_launches) *n_launches = (int)c->ev_used;
    if (flop_algorithmic) *flop_algorithmic = c->flop_alg;
    return IBLNERF_OK;
}

}  // extern "C"
