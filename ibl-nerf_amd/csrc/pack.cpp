// Host-side packer: reference state-dict blob (fp32, [out,in] row-major weight then bias per
// layer, in the registration order of src/nerf_models/ibl_nerf.py:45-72) -> the device chunk
// stream + side tables described in layout.h.  Pure CPU code, no HIP calls.
#include "pack.h"

#include <cmath>
#include <cstring>

namespace ibl {

namespace {

struct LayerDesc { int out, in; };
// SCHEMA of ibl-nerf_amd/checkpoint.py (== IBLNeRF.state_dict() order)
enum LayerId {
    L_POS0 = 0, L_POS1, L_POS2, L_POS3, L_POS4, L_POS5, L_POS6, L_POS7,
    L_VIEWS, L_FEATURE, L_SIGMA, L_ALB_F, L_ALB, L_ROUGH, L_IRR_F, L_IRR, L_RAD,
    L_AR_F0, L_AR_F1, L_AR_F2, L_AR0, L_AR1, L_AR2, N_LAYERS
};
const LayerDesc kLayers[N_LAYERS] = {
    {256, 63}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 319}, {256, 256}, {256, 256},
    {256, 283}, {256, 256}, {1, 256}, {128, 256}, {3, 128}, {1, 256}, {128, 256}, {1, 128}, {3, 256},
    {128, 256}, {128, 256}, {128, 256}, {3, 128}, {3, 128}, {3, 128}};

struct Net {
    const float* w[N_LAYERS];
    const float* b[N_LAYERS];
    explicit Net(const float* blob) {
        size_t off = 0;
        for (int l = 0; l < N_LAYERS; ++l) {
            w[l] = blob + off; off += (size_t)kLayers[l].out * kLayers[l].in;
            b[l] = blob + off; off += (size_t)kLayers[l].out;
        }
    }
    float W(int l, int o, int i) const { return w[l][(size_t)o * kLayers[l].in + i]; }
};

// round-to-nearest-even fp32 -> bf16 (same as v_cvt_pk_bf16_f32 for finite inputs)
inline uint16_t bf16_rne(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

inline void put_split(uint16_t* kstep_base, int lane, int e, float w) {
    const uint16_t hi = bf16_rne(w);
    const uint16_t lo = bf16_rne(w - bf16_to_f32(hi));
    kstep_base[lane * 8 + e] = hi;                         // first KiB: hi fragments
    kstep_base[512 + lane * 8 + e] = lo;                   // second KiB: lo fragments
}

// 16 k-steps of rows [row0, row0+32) of layer l over a 256-feature activation whose reference
// columns start at col_base (K order permuted to the accumulator layout, see layout.h)
void pack_h(uint16_t* ks0, const Net& n, int l, int row0, int col_base) {
    for (int j = 0; j < 16; ++j) {
        uint16_t* ks = ks0 + (size_t)j * (KSTEP_BYTES / 2);
        const int t = j >> 1, s = j & 1;
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31, h = lane >> 5;
            for (int e = 0; e < 8; ++e) {
                const int feat = 32 * t + acc_feature(8 * s + e, h);
                put_split(ks, lane, e, n.W(l, row0 + i, col_base + feat));
            }
        }
    }
}

// nk k-steps of rows [row0, row0+32) over an encoding whose reference columns start at col_base
void pack_enc(uint16_t* ks0, const Net& n, int l, int row0, int col_base, int pairs_per_half, int nk) {
    for (int jj = 0; jj < nk; ++jj) {
        uint16_t* ks = ks0 + (size_t)jj * (KSTEP_BYTES / 2);
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31, h = lane >> 5;
            for (int e = 0; e < 8; ++e) {
                const int ref = enc_ref_index(8 * jj + e, h, pairs_per_half);
                put_split(ks, lane, e, ref < 0 ? 0.0f : n.W(l, row0 + i, col_base + ref));
            }
        }
    }
}

// lane-layout table of a length-(32*ntiles) vector
void lane_table(float* dst, const float* src, int ntiles) {
    for (int t = 0; t < ntiles; ++t)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) dst[(t * 2 + h) * 16 + r] = src[32 * t + acc_feature(r, h)];
}

}  // namespace

size_t blob_floats() {
    size_t n = 0;
    for (int l = 0; l < N_LAYERS; ++l) n += (size_t)kLayers[l].out * kLayers[l].in + kLayers[l].out;
    return n;  // 798 994
}

void pack_network(const float* blob, void* stream_out, float* tab) {
    const Net n(blob);
    uint16_t* s = reinterpret_cast<uint16_t*>(stream_out);
    constexpr size_t KS = KSTEP_BYTES / 2;   // uint16 per k-step
    auto at = [&](int chunk) { return s + (size_t)chunk * (CHUNK_BYTES / 2); };
    std::memset(stream_out, 0, STREAM_BYTES);
    std::memset(tab, 0, TAB_BYTES);

    // positions_linears.0: K = 63 positional-encoding columns
    for (int t = 0; t < 8; ++t) pack_enc(at(CH_L0) + t * PE_KSTEPS * KS, n, L_POS0, 32 * t, 0, PE_PAIRS_PER_HALF, PE_KSTEPS);
    // positions_linears.1..4
    for (int l = 1; l <= 4; ++l)
        for (int t = 0; t < 8; ++t) pack_h(at(CH_L1 + 8 * (l - 1) + t), n, L_POS0 + l, 32 * t, 0);
    // positions_linears.5: input = cat([x63, h]) (ibl_nerf.py:168): columns 0..62 encoding, 63..318 hidden
    for (int t = 0; t < 8; ++t) {
        uint16_t* tile = at(CH_L5) + (size_t)t * (PE_KSTEPS + 16) * KS;
        pack_enc(tile, n, L_POS5, 32 * t, 0, PE_PAIRS_PER_HALF, PE_KSTEPS);
        pack_h(tile + PE_KSTEPS * KS, n, L_POS5, 32 * t, 63);
    }
    for (int t = 0; t < 8; ++t) pack_h(at(CH_L6 + t), n, L_POS6, 32 * t, 0);
    for (int t = 0; t < 8; ++t) pack_h(at(CH_L7 + t), n, L_POS7, 32 * t, 0);
    for (int t = 0; t < 8; ++t) pack_h(at(CH_FEAT + t), n, L_FEATURE, 32 * t, 0);
    for (int t = 0; t < 4; ++t) pack_h(at(CH_ALB + t), n, L_ALB_F, 32 * t, 0);
    for (int t = 0; t < 4; ++t) pack_h(at(CH_IRR + t), n, L_IRR_F, 32 * t, 0);
    // views_linears.0: input = cat([feature256, dir27]) (ibl_nerf.py:194): columns 0..255 feature, 256..282 dir
    for (int t = 0; t < 8; ++t) {
        uint16_t* tile = at(CH_VIEW) + (size_t)t * (DE_KSTEPS + 16) * KS;
        pack_enc(tile, n, L_VIEWS, 32 * t, 256, DE_PAIRS_PER_HALF, DE_KSTEPS);
        pack_h(tile + DE_KSTEPS * KS, n, L_VIEWS, 32 * t, 0);
    }
    for (int k = 0; k < 3; ++k)
        for (int t = 0; t < 4; ++t) pack_h(at(CH_AR + 4 * k + t), n, L_AR_F0 + k, 32 * t, 0);

    // biases
    for (int l = 0; l < 8; ++l) lane_table(tab + TAB_BIAS + (BT_L0 + 8 * l) * 32, n.b[L_POS0 + l], 8);
    lane_table(tab + TAB_BIAS + BT_FEAT * 32, n.b[L_FEATURE], 8);
    lane_table(tab + TAB_BIAS + BT_ALB * 32, n.b[L_ALB_F], 4);
    lane_table(tab + TAB_BIAS + BT_IRR * 32, n.b[L_IRR_F], 4);
    lane_table(tab + TAB_BIAS + BT_VIEW * 32, n.b[L_VIEWS], 8);
    for (int k = 0; k < 3; ++k) lane_table(tab + TAB_BIAS + (BT_AR + 4 * k) * 32, n.b[L_AR_F0 + k], 4);
    // VALU heads
    lane_table(tab + TAB_SIG, n.w[L_SIGMA], 8);
    lane_table(tab + TAB_ROUGH, n.w[L_ROUGH], 8);
    for (int c = 0; c < 3; ++c) lane_table(tab + TAB_ALB + c * 128, n.w[L_ALB] + c * 128, 4);
    lane_table(tab + TAB_IRR, n.w[L_IRR], 4);
    for (int c = 0; c < 3; ++c) lane_table(tab + TAB_RAD + c * 256, n.w[L_RAD] + c * 256, 8);
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) lane_table(tab + TAB_AR + (k * 3 + c) * 128, n.w[L_AR0 + k] + c * 128, 4);
    float* sc = tab + TAB_SCALAR;
    sc[0] = n.b[L_SIGMA][0];
    for (int c = 0; c < 3; ++c) sc[1 + c] = n.b[L_ALB][c];
    sc[4] = n.b[L_ROUGH][0];
    sc[5] = n.b[L_IRR][0];
    for (int c = 0; c < 3; ++c) sc[6 + c] = n.b[L_RAD][c];
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) sc[9 + 3 * k + c] = n.b[L_AR0 + k][c];
}

}  // namespace ibl
