// Device weight-stream layout shared by the host packer (pack.cpp) and the fused MLP kernel
// (mlp_kernel.hip).
//
// The reference evaluates IBLNeRF.forward (src/nerf_models/ibl_nerf.py:154-210) as 23 separate
// nn.Linear calls.  Here one wavefront owns 32 sample points and keeps their 256-wide activation
// in registers for the whole network; the weights are the MFMA *A* operand and arrive as a stream
// of 32 KiB "chunks" through an LDS ring.  Every fp32 weight w is stored as a bf16 pair
// (hi = rne(w), lo = rne(w - hi)) and every GEMM is three MFMA products
// (Wh*Xh + Wh*Xl + Wl*Xh, fp32 accumulate): ~17 significand bits per operand, which is what the
// 1e-3 parity bar needs.
//
// MFMA shape: v_mfma_f32_32x32x16_bf16.  Lane l = (i = l & 31, h = l >> 5).
//   A (weights):      lane holds 8 k-slots (8h .. 8h+7) of output row i of the 32-row tile
//   B (activations):  lane holds the same 8 k-slots of point column i
//   D (result):       lane holds, for point column i, rows (r&3) + 8*(r>>2) + 4h, r = 0..15
// The result layout of one layer IS the B layout of the next if the next layer's k-slots are
// numbered accordingly, so the packer permutes the K order of every weight matrix and no
// activation ever leaves the register file:
//   k-step j = 2t + s (t = input tile 0..7, s = 0..1), slot (h, e)  <->  input feature
//   32t + (e&3) + 8*(2s + (e>>2)) + 4h            (= accumulator register r = 8s + e of tile t)
//
// One k-step of one 32-row tile is 2 KiB: [64 lanes x 8 bf16 hi][64 lanes x 8 bf16 lo], so a
// lane's fragment is one conflict-free ds_read_b128 at lane*16.
#pragma once
#include <stdint.h>

namespace ibl {

constexpr int KSTEP_BYTES = 2048;
constexpr int CHUNK_KSTEPS = 16;
constexpr int CHUNK_BYTES = KSTEP_BYTES * CHUNK_KSTEPS;  // 32 KiB
constexpr int RING_SLOTS = 3;

// The stream is a flat sequence of k-steps; a chunk is 16 consecutive k-steps regardless of tile
// boundaries.  Per layer, tile after tile; inside a tile first the encoding k-steps (if the layer
// has a concatenated encoding input), then the 16 k-steps over the 256-feature activation:
//   layer                          k-steps/tile   tiles   chunks   first chunk
constexpr int CH_L0 = 0;     // positions_linears.0        4 (PE)          8       2
constexpr int CH_L1 = 2;     // positions_linears.1..4     16              8       8 each (2..33)
constexpr int CH_L5 = 34;    // positions_linears.5        4 (PE) + 16     8       10
constexpr int CH_L6 = 44;    // positions_linears.6        16              8       8
constexpr int CH_L7 = 52;    // positions_linears.7        16              8       8   (trunk-only eval ends at 60)
constexpr int CH_FEAT = 60;  // feature_linear             16              8       8
constexpr int CH_ALB = 68;   // albedo_feature_linear      16              4       4
constexpr int CH_IRR = 72;   // irradiance_feature_linear  16              4       4
constexpr int CH_VIEW = 76;  // views_linears.0            2 (DE) + 16     8       9
constexpr int CH_AR = 85;    // additional_radiance_feature_linear.{0,1,2}  16   4 each   12
constexpr int N_CHUNKS_NET = 97;   // the network's forward layers
constexpr int N_CHUNKS_TRUNK = 60;
// Backward stream of the trunk (the density-gradient query, VAR_TRUNK_GRAD: d sigma / d position): the same tile format
// with TRANSPOSED matrices, layers in reverse order.  Rows of a tile = 32 INPUT features of the layer, K = its 256 output
// features in accumulator order, so the backward chain dZ(l-1) = (W(l)^T dZ(l)) * [Z(l-1) > 0] is the forward loop run on
// these chunks.  positions_linears.5 has 10 row tiles (8 hidden + 2 for its 63 encoding columns), .0 has the 2 encoding
// tiles only; an encoding tile's row (r&3) + 8*(r>>2) + 4h is encoding slot 16*tile + r of lane half h (enc_ref_index), so
// that the gradient of a slot lands in the lane half that computed the slot's sine / cosine.
//   layer                          row tiles   chunks   first chunk
constexpr int CH_G7 = 97;    // positions_linears.7^T      8           8
constexpr int CH_G6 = 105;   // positions_linears.6^T      8           8
constexpr int CH_G5 = 113;   // positions_linears.5^T      8 + 2       10
constexpr int CH_G4 = 123;   // positions_linears.4..1^T   8           8 each (123..154)
constexpr int CH_G0 = 155;   // positions_linears.0^T      2           2
constexpr int N_CHUNKS_GRAD = 60;
// ... and of the two 256-wide head layers between the trunk and the view-dependent heads (VAR_TRUNK_BWD_FEAT2):
constexpr int CH_GV = 157;   // views_linears.0^T, its 256 feature columns (the direction columns carry no gradient)   8   8
constexpr int CH_GF = 165;   // feature_linear^T                                                                     8   8
constexpr int N_CHUNKS_GRAD2 = 16;
// ... and of the whole network (VAR_NET_BWD): the three additional_radiance_feature layers meet in dL/dh2 and feature_linear meets the
// albedo / irradiance feature layers in dL/dh7, so their transposes are packed K-concatenated, one tile = all k-steps of its 32 rows:
constexpr int CH_GA = 173;   // [ARF.2^T (8 k-steps) | ARF.0^T (8) | ARF.1^T (8)] per tile: 8 tiles x 24 k-steps   12 chunks
constexpr int CH_GH = 185;   // [albedo_feature^T (8) | irradiance_feature^T (8) | feature_linear^T (16)] per tile: 8 x 32   16 chunks
constexpr int N_CHUNKS_GRAD3 = 28;
constexpr int N_CHUNKS = N_CHUNKS_NET + N_CHUNKS_GRAD + N_CHUNKS_GRAD2 + N_CHUNKS_GRAD3;   // 201
constexpr int PE_KSTEPS = 4;   // 64 slots, 63 used
constexpr int DE_KSTEPS = 2;   // 32 slots, 27 used
constexpr long STREAM_BYTES = (long)N_CHUNKS * CHUNK_BYTES;  // 6.3 MiB per network

// fp32 side tables (biases in accumulator-lane layout + the tiny N=1/3 heads that run on the VALU)
// Lane-layout entry [tile][h][r] holds the value for feature 32*tile_local + (r&3) + 8*(r>>2) + 4h.
constexpr int BT_L0 = 0;      // bias tiles: layer l tile t -> 8*l + t   (0..63)
constexpr int BT_FEAT = 64;   // 64..71
constexpr int BT_ALB = 72;    // 72..75
constexpr int BT_IRR = 76;    // 76..79
constexpr int BT_VIEW = 80;   // 80..87
constexpr int BT_AR = 88;     // 88..99
constexpr int N_BIAS_TILES = 100;
constexpr int TAB_BIAS = 0;                         // [100][2][16]
constexpr int TAB_SIG = TAB_BIAS + N_BIAS_TILES * 32;   // sigma_linear.weight      [8][2][16]
constexpr int TAB_ROUGH = TAB_SIG + 256;            // roughness_linear.weight  [8][2][16]
constexpr int TAB_ALB = TAB_ROUGH + 256;            // albedo_linear.weight     [3][4][2][16]
constexpr int TAB_IRR = TAB_ALB + 384;              // irradiance_linear.weight [4][2][16]
constexpr int TAB_RAD = TAB_IRR + 128;              // radiance_linear.weight   [3][8][2][16]
constexpr int TAB_AR = TAB_RAD + 768;               // additional_radiance_linear.k.weight [3][3][4][2][16]
constexpr int TAB_SCALAR = TAB_AR + 1152;           // 18 output biases in raw-channel order
constexpr int TAB_FLOATS = TAB_SCALAR + 32;
constexpr int TAB_BYTES = TAB_FLOATS * 4;           // 24 704 B

constexpr int LDS_RING_BYTES = RING_SLOTS * CHUNK_BYTES;   // 96 KiB
constexpr int LDS_BYTES = LDS_RING_BYTES + TAB_BYTES;      // 123 008 B

// Operand stash of VAR_TRUNK_BWD for the weight-gradient kernel (wgrad_kernel.hip), per WAVE GROUP of 32 points (wave w of point
// group g: 4 g + w), in the kernel's own fragment layout so that every store is one coalesced 16-byte piece per lane: the f16 `hi`
// fragment of k-step j (8 features 32(j>>1) + acc_feature(8(j&1) + e, h) of point lane&31) at  j * 1024 + lane * 16.
//   [STASH_X  + layer l][wave group]  post-ReLU output of positions_linears.l   (16 KiB each)
//   [STASH_DZ + layer l][wave group]  dL / d pre-activation of positions_linears.l
//   [STASH_ENC][wave group]           the encoding's 4 k-steps (4 KiB each; slot order of enc_ref_index)
//   VAR_TRUNK_BWD_FEAT2 adds  [STASH_XF] feature_linear's output (views_linears.0's input), [STASH_DZV] / [STASH_DZF] dL / d pre-activation
//   of views_linears.0 / feature_linear, and [STASH_DENC] the direction encoding's 2 k-steps (2 KiB each)
//   VAR_NET_BWD adds  [STASH_XH2] relu(views_linears.0), [STASH_F0 + k] relu(additional_radiance_feature_linear.k), [STASH_FA] / [STASH_FI] relu of the
//   albedo / irradiance feature layers, and [STASH_DF0 + k], [STASH_DFA], [STASH_DFI] their dL / d pre-activation (128 wide: k-steps 0..7 of an entry)
constexpr int STASH_X = 0, STASH_DZ = 8, STASH_XF = 16, STASH_DZV = 17, STASH_DZF = 18, STASH_XH2 = 19, STASH_F0 = 20, STASH_FA = 23, STASH_FI = 24,
              STASH_DF0 = 25, STASH_DFA = 28, STASH_DFI = 29, STASH_N_ACT = 30, STASH_ENC = 30, STASH_DENC = 31;
constexpr long STASH_ACT_BYTES = 16 * 1024;   // per wave group and activation
constexpr long STASH_ENC_BYTES = 4 * 1024;
constexpr long STASH_DENC_BYTES = 2 * 1024;
__host__ __device__ constexpr long stash_bytes(long wave_groups) { return wave_groups * (STASH_N_ACT * STASH_ACT_BYTES + STASH_ENC_BYTES + STASH_DENC_BYTES); }
__host__ __device__ constexpr long stash_offset(int what, long wave_groups, long wg) {
    return what < STASH_N_ACT ? (what * wave_groups + wg) * STASH_ACT_BYTES
           : what == STASH_ENC ? STASH_N_ACT * wave_groups * STASH_ACT_BYTES + wg * STASH_ENC_BYTES
                               : wave_groups * (STASH_N_ACT * STASH_ACT_BYTES + STASH_ENC_BYTES) + wg * STASH_DENC_BYTES;
}

// feature held by accumulator register r of lane-half h (tile-local, 0..31)
__host__ __device__ constexpr int acc_feature(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Encoding slot maps.  A half-wave h owns slots sl = 8*jj + e (jj = encoding k-step, e = 0..7).
// Returns the reference embedding index (positional_embedder.py:33-34 order
// [x, sin(2^0 x), cos(2^0 x), ...]) or -1 for a zero pad slot.
//   positions (L=10): 30 (freq k, coord c) pairs m = 3k + c; half h owns m = 15h .. 15h+14;
//                     slot 2u = sin(pair 15h+u), 2u+1 = cos; slots 30,31 = (x,y) | (z,pad)
//   directions (L=4): 12 pairs; half h owns m = 6h .. 6h+5; slots 12,13 = (x,y) | (z,pad); 14,15 pad
__host__ __device__ constexpr int enc_ref_index(int sl, int h, int pairs_per_half) {
    const int n = 2 * pairs_per_half;
    if (sl < n) {
        const int m = pairs_per_half * h + (sl >> 1);
        return 3 + 6 * (m / 3) + ((sl & 1) ? 3 : 0) + (m % 3);
    }
    if (sl == n) return h == 0 ? 0 : 2;
    if (sl == n + 1) return h == 0 ? 1 : -1;
    return -1;
}
constexpr int PE_PAIRS_PER_HALF = 15;
constexpr int DE_PAIRS_PER_HALF = 6;

// Raw output channel order of IBLNeRF.forward (ibl_nerf.py:200-208):
// [sigma, albedo3, roughness, irradiance, radiance3, radiance_1 x3, radiance_2 x3, radiance_3 x3]
constexpr int RAW_CH = 18;
constexpr int REFL_CH = 13;   // sigma + channels 6..17 (what raw2outputs_simple reads, ibl_nerf_renderer.py:38-68)

// FULL: all 18 channels; TRUNK: sigma only (ibl_nerf.py:175-176); REFL: sigma + the 12 radiance channels (what
// raw2outputs_simple reads).  *_CI: the same for a network built with is_color_independent_to_direction (ibl_nerf.py:192):
// the radiance heads read the trunk output, feature_linear and views_linears are not evaluated.
enum Variant { VAR_FULL = 0, VAR_TRUNK = 1, VAR_REFL = 2, VAR_FULL_CI = 3, VAR_REFL_CI = 4,
               VAR_TRUNK_X = 5,     // fast kernel only: TRUNK with its first two layers as three f16 products (layout_mx.h)
               VAR_TRUNK_GRAD = 6,    // three-product kernels only: TRUNK forward + its backward chain, out = [sigma, d sigma / d x, y, z]
               VAR_TRUNK_BWD = 7,     // ... with an upstream gradient per point, and the operands of the weight gradient stashed (STASH_*)
               VAR_TRUNK_FEAT = 8,    // f16x3 kernel only: TRUNK forward whose output is the 256 trunk features h7 (fp32 rows), not sigma
               VAR_TRUNK_BWD_FEAT = 9,    // VAR_TRUNK_BWD with the upstream gradient given on those features (dL/dh7 rows) instead of on sigma
               VAR_TRUNK_FEAT2 = 10,      // ... one layer pair further: outputs h7 AND h2 = relu(views_linears.0([feature_linear(h7), dir27])) rows
               VAR_TRUNK_BWD_FEAT2 = 11,  // its backward: dL/dh7 and dL/dh2 rows in; also stashes for feature_linear's and views_linears.0's weight gradients
               VAR_NET_BWD = 12,          // the whole network's backward: dL/d raw rows [n, 18] in, every layer differentiated
               VAR_TRUNK_P = 13,          // fast kernel only: TRUNK with every layer as three f16 products + three block-scaled fp6 products for the 2^-22 terms
                                          // (15 matrix slots per 64 MACs; operands to ~2^-26): the coarse pass's density, which places the fine samples
               VAR_REFL_LIST = 15,        // fast kernel only: REFL / FULL on a compact list of points (length in device memory, a flat index r * S + s per point: the
               VAR_FULL_LIST = 16,        // ray's direction and the output row are found through it) — the relevant samples of a query (k_select_points)
               VAR_TRUNK_X_LIST = 17,     // ... and TRUNK_X on such a list (the fine grid's offset copies)
               VAR_TRUNK_LIST = 18 };     // three-product f16 kernel only (which also serves VAR_FULL_LIST): TRUNK on such a list — the safe table's fine pass
__host__ __device__ constexpr bool variant_trunk_x(int v) { return v == VAR_TRUNK_X || v == VAR_TRUNK_X_LIST; }
__host__ __device__ constexpr bool variant_list(int v) { return v == VAR_REFL_LIST || v == VAR_FULL_LIST || v == VAR_TRUNK_X_LIST || v == VAR_TRUNK_LIST; }
__host__ __device__ constexpr int variant_base(int v) {     // the form a list variant evaluates
    return v == VAR_REFL_LIST ? VAR_REFL : v == VAR_FULL_LIST ? VAR_FULL : v == VAR_TRUNK_X_LIST ? VAR_TRUNK_X : v == VAR_TRUNK_LIST ? VAR_TRUNK : v;
}
__host__ __device__ constexpr bool variant_ci(int v) { return v == VAR_FULL_CI || v == VAR_REFL_CI; }
__host__ __device__ constexpr bool variant_albirr(int v) { return v == VAR_FULL || v == VAR_FULL_CI || v == VAR_FULL_LIST; }   // albedo / roughness / irradiance heads

}  // namespace ibl
