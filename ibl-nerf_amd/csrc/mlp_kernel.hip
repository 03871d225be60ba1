// Fused IBLNeRF forward (positional encoding -> 8x256 trunk with skip -> multi-head outputs) for
// gfx950.  Replaces `run_network` + `IBLNeRF.forward_not_freezed`
// (src/nerf_models/ibl_nerf.py:236-252, :154-210; encoder src/nerf_models/positional_embedder.py:4-52).
//
// One workgroup = 4 wavefronts (one per SIMD, the kernel owns the whole 512-register file);
// each wavefront owns 32 sample points and keeps their 256-wide activation in registers across
// all layers (see layout.h for why the MFMA result layout can be fed straight back as the next
// B operand).  Weights are streamed once per 128 points through a 3-slot LDS ring with
// global_load_lds (LDS-DMA), one 32 KiB chunk = 48 MFMAs per wave.  Every GEMM runs as three
// bf16 MFMA products on (hi, lo) splits with fp32 accumulation; the N=1/N=3 heads run on the
// VALU in fp32 straight from the fp32 accumulators.
//
// Two flavours compile from this file:
//   (default)     bf16 hi/lo splits: ~2^-17 per operand, the full fp32 exponent range — the wide-range fallback;
//   -DIBL_F16X3   f16 hi/lo splits (namespace f16x3k): the same three products at the same matrix-core rate with ~2^-22 per
//                 operand (11 + 11 significand bits; f16 denormals are honoured by the MFMA and by v_fma_mix, probed in
//                 scratch/probe_f16_denorm.hip) — the precise mode.  On a checkpoint with surfaces the density head's
//                 cancellation amplifies operand round-off ~100x: bf16x3 leaves 2e-3 on sigma (abs) and 2e-3 relative on the
//                 depth of grazing rays, this flavour 1e-4 / 9e-5 (scratch/prec_probe_fitted.py).  Inputs, weights and
//                 activations must stay below 65504: the kernel tracks the largest magnitude it converts and raises
//                 MlpArgs::range_flag (the caller repeats the call on the bf16 flavour).
#include <hip/hip_runtime.h>

#include <cstddef>

#include <cstdio>
#include <type_traits>

#include "layout.h"
#include "mlp_args.h"
#include "sincos_enc.h"

namespace ibl {
#ifdef IBL_F16X3
namespace f16x3k {
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));   // (the fragment type keeps its name: 8 x 16-bit operands)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#else
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Act { bf16x8 hi[16]; bf16x8 lo[16]; };  // 256 features = 16 k-steps of B fragments
struct Enc { bf16x8 hi[4]; bf16x8 lo[4]; };    // up to 64 encoding slots = 4 k-steps
struct Half { bf16x8 hi[8]; bf16x8 lo[8]; };   // 128 features = 8 k-steps (the gradient of one 128-wide feature layer)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#ifdef IBL_ABLATE_NO_MFMA    // timing ablation only: keep the operands live, skip the matrix pipe
__device__ __forceinline__ f32x16 mfma_stub(bf16x8 a, bf16x8 b, f32x16 c) {
    asm volatile("" :: "v"(a), "v"(b));
    return c;
}
#define MFMA(a, b, c) mfma_stub((a), (b), (c))
#elif defined(IBL_F16X3)
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#else
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#endif

#ifdef IBL_TRACE   // debug build only: cycle stamps of workgroup 0 / wave 0, one layer of the first point group
__device__ long long g_trace[256];
#define TRACE(P, slot) do { if ((P).trace) (P).tr[(slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define TRACE(P, slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// weight-stream pipeline.  The program's chunks are walked cyclically and never drained: while
// chunk c is being consumed, chunk c+1 has landed (or is landing) and the loads of chunk c+2 are
// issued; after the last chunk of a point group the stream wraps to chunk 0 of the next group.
// Ring slot of a chunk = (running chunk count) % 3.  No branches: begin()/end() are straight-line
// so a whole tile (48 MFMAs + the previous tile's epilogue) is ONE basic block for the scheduler.
// ---------------------------------------------------------------------------------------------
template <int VARIANT>
struct Pipe {
    static constexpr bool CI = variant_ci(VARIANT), ALBIRR = variant_albirr(VARIANT);
    static constexpr int N_PROG = (VARIANT == VAR_TRUNK || VARIANT == VAR_TRUNK_FEAT) ? N_CHUNKS_TRUNK : (VARIANT == VAR_TRUNK_GRAD || VARIANT == VAR_TRUNK_BWD || VARIANT == VAR_TRUNK_BWD_FEAT) ? N_CHUNKS_TRUNK + N_CHUNKS_GRAD
                                  : VARIANT == VAR_TRUNK_FEAT2 ? N_CHUNKS_TRUNK + 8 + 9 : VARIANT == VAR_TRUNK_BWD_FEAT2 ? N_CHUNKS_TRUNK + 8 + 9 + N_CHUNKS_GRAD2 + N_CHUNKS_GRAD
                                  : VARIANT == VAR_NET_BWD ? N_CHUNKS_NET + 12 + 8 + 16 + N_CHUNKS_GRAD
                                  : N_CHUNKS_TRUNK + (CI ? 0 : 8 + 9) + (ALBIRR ? 8 : 0) + 12;
    const char* stream;
    char* ring;          // generic pointer to the ring (for ds_read)
    unsigned lds_ring;   // LDS byte address of the ring (for M0)
    unsigned voff;       // lane*16 + wave*8192: this lane's byte offset inside a chunk for the DMA
    int lane, wave;
    int slot, slot2;     // ring slot of the chunk being consumed / of the chunk two ahead
    int prog2;           // program position (0..N_PROG-1) of the chunk two ahead
#ifdef IBL_TRACE
    bool trace = false;
    long long* tr = nullptr;
#endif

    // program position -> stream chunk: a variant's program is the trunk followed by the head layers it evaluates
    __device__ __forceinline__ static int stream_chunk(int p) {
        if (p < N_CHUNKS_TRUNK) return p;
        int q = p - N_CHUNKS_TRUNK;
        if (VARIANT == VAR_TRUNK_GRAD || VARIANT == VAR_TRUNK_BWD || VARIANT == VAR_TRUNK_BWD_FEAT) return CH_G7 + q;   // the backward stream follows the trunk
        if (VARIANT == VAR_NET_BWD) {   // the whole forward (FULL's chunks), then dL/dh2, dFeat, dL/dh7, the trunk
            if (q < N_CHUNKS_NET - N_CHUNKS_TRUNK) return p;
            q -= N_CHUNKS_NET - N_CHUNKS_TRUNK;
            if (q < 12) return CH_GA + q;
            if (q < 20) return CH_GV + (q - 12);
            if (q < 36) return CH_GH + (q - 20);
            return CH_G7 + (q - 36);
        }
        if (VARIANT == VAR_TRUNK_FEAT2 || VARIANT == VAR_TRUNK_BWD_FEAT2) {   // trunk, feature_linear, views_linears.0 [, their transposes, the trunk's]
            if (q < 8) return CH_FEAT + q;
            if (q < 17) return CH_VIEW + (q - 8);
            q -= 17;
            return q < N_CHUNKS_GRAD2 ? CH_GV + q : CH_G7 + (q - N_CHUNKS_GRAD2);
        }
        if (!CI) { if (q < 8) return CH_FEAT + q; q -= 8; }
        if (ALBIRR) { if (q < 8) return CH_ALB + q; q -= 8; }
        if (!CI) { if (q < 9) return CH_VIEW + q; q -= 9; }
        return CH_AR + q;
    }
    // One 32 KiB chunk = 8 LDS-DMA instructions per wave (wave w copies bytes [8192w, 8192w+8192),
    // piece i = 1 KiB).  Scalar base + one VGPR offset (saddr form) so no 64-bit VGPR address exists;
    // the loads are invisible to hipcc's waitcnt bookkeeping and are counted by hand (end()).  M0
    // carries the wave-uniform LDS destination; the DMA adds lane*16 itself.
    // One LDS-DMA instruction costs the issuing wave ~60 cycles (measured with s_memtime: issuing
    // all eight back to back after the chunk barrier stalled the wave's MFMA stream for ~500
    // cycles per chunk), so the pieces are issued ONE PER K-STEP, each in the shadow of that
    // k-step's MFMAs.
    __device__ __forceinline__ void issue_piece(int prog, int slt, int i) const {
#ifdef IBL_ABLATE_NO_LOADS   // timing ablation only (results are garbage): no weight traffic
        return;
#endif
        const char* src = stream + (size_t)stream_chunk(prog) * CHUNK_BYTES;                        // uniform (SGPR pair)
        const unsigned dst = lds_ring + (unsigned)slt * CHUNK_BYTES + wave * 8192 + i * 1024;       // uniform
        const unsigned v = voff + i * 1024;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %3\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(v), "s"(dst), "s"(src)
            : "memory");
    }
    __device__ __forceinline__ void issue(int prog, int slt) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) issue_piece(prog, slt, i);
    }
    __device__ __forceinline__ void start() {   // once per kernel
        issue(0, 0);
        issue(1, 1);
        slot = 0;
        slot2 = 2;
        prog2 = 2;
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");   // chunk 0 landed everywhere
    }
    // this lane's fragment base inside the chunk being consumed
    __device__ __forceinline__ const char* frag() const { return ring + slot * CHUNK_BYTES + lane * 16; }
    // piece i of the chunk two ahead (one per k-step, chunk-relative k-steps DMA_K0 .. DMA_K0+7)
    __device__ __forceinline__ void prefetch_piece(int i) const { issue_piece(prog2, slot2, i); }
    // done with the current chunk: the next one must have landed (mine: vmcnt, everyone's: barrier).
    // The barrier is also the WAR fence for the slot the loads issued by the next begin() overwrite.
    // Wait and barrier are ONE asm statement with a memory clobber: the s_barrier builtin alone is
    // not a compiler memory fence, and a ds_read of the next chunk hoisted above it would read a
    // slot other waves are still filling.
    __device__ __forceinline__ void end() {
#ifdef IBL_ABLATE_NO_BARRIER   // timing ablation only
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
#endif
        slot = slot == RING_SLOTS - 1 ? 0 : slot + 1;
        slot2 = slot2 == RING_SLOTS - 1 ? 0 : slot2 + 1;
        prog2 = prog2 == N_PROG - 1 ? 0 : prog2 + 1;
    }
    __device__ __forceinline__ void drain() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// ReLU as a signed-integer max on the float bits: negative floats (sign bit set) are negative
// ints, so max(bits, 0) is exactly max(x, +0) for every non-NaN x, in ONE v_max_i32 (fmaxf costs a
// canonicalising v_max first).
__device__ __forceinline__ float relu_bits(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

#ifdef IBL_F16X3
// (x0, x1) -> packed f16 pairs hi = rne(x), lo = rne(x - hi): v_cvt_pk_f16_f32, then the residuals in one v_fma_mix{lo,hi}_f16
// each (they read the f16 half of hi directly and round x - hi, exact in fp32, to f16 — a denormal when |x| < 0.25)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 xv = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, f16x2));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x0), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "v"(hi));
}
// Range guard of this flavour: none in the hot loop (the kernel sits on the register cliff: one more long-lived value costs
// hundreds of spills).  An activation or input at or beyond 65520 converts to hi = +inf, lo = rne(x - inf) = -inf; the next layer
// accumulates Wh*(+inf) + Wh*(-inf) = NaN in EVERY output row (0 * inf for a zero weight) — the NEGATIVE quiet NaN 0xffc00000
// (scratch/probe_mfma_nan.hip), which the integer-max ReLU of the bf16 flavour would turn into 0.  This flavour's ReLU is gfx950's
// v_maximum3_f32 (IEEE 754-2019 maximum: a NaN operand of either sign propagates; one instruction, like the integer max), so the
// NaN reaches at least one output channel of that point, and the kernel's tail checks the outputs.
__device__ __forceinline__ float relu_keepnan(float x) { return __builtin_elementwise_maximum(x, 0.0f); }
#else
// (x0, x1) -> packed bf16 pairs hi = rne(x), lo = rne(x - hi): cvt_pk, shift, and, packed sub, cvt_pk
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 xv = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, bf16x2));
    const f32x2 hv = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(xv - hv, bf16x2));
}
#endif

// Keeps a value materialised where it is computed (an empty asm the optimiser cannot see through):
// the consumers of an epilogue's results are a whole layer (or the whole kernel) away, and
// without the pins whole epilogues are deferred out of the MFMA shadow they were placed in.
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }

// Accumulator of one tile.  IBL_DUAL_ACC (experiment): two independent MFMA chains (main = Wh*Xh +
// Wl*Xh, cross = Wh*Xl) summed in the epilogue; a pure-MFMA micro-benchmark sustains 14 % more with two
// chains than with one dependent chain under the power cap (scratch/mfma_peak.hip).
struct Acc {
    f32x16 main;
#ifdef IBL_DUAL_ACC
    f32x16 cross;
#endif
};

// Epilogue of one finished 32-feature tile, cut into 8 slices of two accumulator registers so the
// GEMM loop of the NEXT tile can interleave one slice per k-step between its MFMAs:
//   STORE: v = [ReLU](acc) -> (hi, lo) bf16 B fragments of k-steps 2T, 2T+1 of `dst`
//   NCH  : running fp32 dot products of v with N=1/3 head weight rows (sigma, roughness, albedo,
//          irradiance, radiance heads) held in lane layout in LDS.
// The layer bias is already in the accumulator (it is the MFMA chain's initial C).
//   MASK : (density-gradient variant) the ReLU's pass bits of this tile, 16 per lane, go to the wave's mask area in LDS:
//          u16 at mrow + 128 * T (mrow = this lane's slot in the layer's row, see MASK_* below)
//          MASK = 2 (VAR_TRUNK_BWD): the hi fragments also go to the operand stash (layout.h: STASH_X), srow + 1024 * k-step;
//          MASK = 3: the stash only (a layer without ReLU); MASK = 4: the stash only and no `dst` (a 128-wide feature layer: its pass bits are
//          read back from the stashed values)
//   FOUT : (VAR_TRUNK_FEAT) the fp32 values themselves go to the caller: frow = this point's 256-float row + 4h, four floats per two slices
template <bool STORE, bool RELU, int NCH, int MASK = 0, bool FOUT = false>
struct Epi {
    Act* dst;
    float* part[NCH > 0 ? NCH : 1];
    const float* tab[NCH > 0 ? NCH : 1];   // this lane-half's row of tile 0 of each head table ([tile][2][16])
    char* mrow;
    char* srow;
    float* frow;
    u32x4 h, l;
    unsigned mb;
    f32x2 keep;

    template <int T, int I>
    __device__ __forceinline__ void slice(const Acc& a) {
#ifdef IBL_DUAL_ACC
        float x0 = a.main[2 * I] + a.cross[2 * I], x1 = a.main[2 * I + 1] + a.cross[2 * I + 1];
#else
        const f32x16& acc = a.main;
        float x0 = acc[2 * I], x1 = acc[2 * I + 1];
#endif
        if constexpr (RELU) {
#if defined(IBL_F16X3) && !defined(IBL_ABLATE_RELU_BITS)   // (timing ablation: the integer-max ReLU, which loses the range guard)
            x0 = relu_keepnan(x0);
            x1 = relu_keepnan(x1);
#else
            x0 = relu_bits(x0);
            x1 = relu_bits(x1);
#endif
        }
        if constexpr (FOUT) {
            if constexpr ((I & 1) == 0) keep = f32x2{x0, x1};
            else if (frow != nullptr) *reinterpret_cast<f32x4*>(frow + 32 * T + 8 * (I >> 1)) = f32x4{keep[0], keep[1], x0, x1};
        }
        if constexpr (MASK == 1 || MASK == 2) {
            const unsigned bits = (x0 > 0.0f ? 1u << (2 * I) : 0u) | (x1 > 0.0f ? 2u << (2 * I) : 0u);
            mb = I == 0 ? bits : (mb | bits);
            if constexpr (I == 7) *reinterpret_cast<unsigned short*>(mrow + 128 * T) = (unsigned short)mb;
        }
        if constexpr (STORE || MASK == 4) {
            unsigned hh, ll;
            split_pair(x0, x1, hh, ll);
            h[I & 3] = hh;
            l[I & 3] = ll;
            if constexpr ((I & 3) == 3) {
                asm volatile("" : "+v"(h), "+v"(l));
                if constexpr (STORE) {
                    dst->hi[2 * T + (I >> 2)] = __builtin_bit_cast(bf16x8, h);
                    dst->lo[2 * T + (I >> 2)] = __builtin_bit_cast(bf16x8, l);
                }
                if constexpr (MASK >= 2) *reinterpret_cast<u32x4*>(srow + 1024 * (2 * T + (I >> 2))) = h;
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const f32x2 w = *reinterpret_cast<const f32x2*>(tab[c] + T * 32 + 2 * I);
            *part[c] = fmaf(x1, w[1], fmaf(x0, w[0], *part[c]));
            if constexpr (I == 7) pin(*part[c]);
        }
    }
};

// Epilogue of a tile of the BACKWARD chain (density-gradient variant).  Tiles 0..7 are the gradient with respect to the previous
// layer's 256 post-ReLU features: times the recorded pass bits, then (hi, lo) fragments of `dst` as in the forward.  Tiles 8, 9
// (positions_linears.5^T) / all tiles with ENC_ACC (positions_linears.0^T) are the gradient with respect to the 64 encoding
// slots: kept in fp32, genc[16 * tile + r].
// MODE 1: no pass bits (the layer in front has no ReLU: feature_linear).  MODE 3: see rtab.  MODE 2: the caller's gradient rows are added first
// (dL/dh7 arrives both through feature_linear and directly from the heads the caller keeps): addrow = this point's row + 4h, or null.
template <bool ENC_ACC, bool STASH = false, int MODE = 0>
struct EpiG {
    Act* dst;
    const char* mrow;
    float* genc;
    char* srow;        // STASH: this layer's dZ stash row of the wave group (+ lane * 16)
    const float* addrow;
    float addscale;
    const float* rtab[3];   // MODE 3: the N = 1/3 heads that read this activation directly add their rank-1 terms: head-weight tables in lane
    float rch[3];           // layout (as the forward's), times the point's upstream gradient of that channel
    u32x4 h, l;
    unsigned mw;

    template <int T, int I>
    __device__ __forceinline__ void slice(const Acc& a) {
        float x0 = a.main[2 * I], x1 = a.main[2 * I + 1];
        if constexpr (ENC_ACC) {
            genc[16 * T + 2 * I] += x0;
            genc[16 * T + 2 * I + 1] += x1;
            if constexpr (I == 7) { pin(genc[16 * T]); }
        } else if constexpr (T >= 8) {
            genc[16 * (T - 8) + 2 * I] = x0;
            genc[16 * (T - 8) + 2 * I + 1] = x1;
        } else {
            if constexpr (MODE == 2) {
                if (addrow != nullptr) {   // accumulator registers 2I, 2I+1 = features 32T + (2I & 3) + 8 (I >> 1) + 4h, +1
                    const f32x2 r = *reinterpret_cast<const f32x2*>(addrow + 32 * T + ((2 * I) & 3) + 8 * (I >> 1));
                    x0 = fmaf(r[0], addscale, x0);
                    x1 = fmaf(r[1], addscale, x1);
                }
            }
            if constexpr (MODE == 3) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x2 w = *reinterpret_cast<const f32x2*>(rtab[c] + T * 32 + 2 * I);
                    x0 = fmaf(w[0], rch[c], x0);
                    x1 = fmaf(w[1], rch[c], x1);
                }
            }
            if constexpr (MODE != 1) {
                if constexpr (I == 0) mw = *reinterpret_cast<const unsigned short*>(mrow + 128 * T);
                x0 = (mw & (1u << (2 * I))) ? x0 : 0.0f;
                x1 = (mw & (2u << (2 * I))) ? x1 : 0.0f;
            }
            unsigned hh, ll;
            split_pair(x0, x1, hh, ll);
            h[I & 3] = hh;
            l[I & 3] = ll;
            if constexpr ((I & 3) == 3) {
                asm volatile("" : "+v"(h), "+v"(l));
                dst->hi[2 * T + (I >> 2)] = __builtin_bit_cast(bf16x8, h);
                dst->lo[2 * T + (I >> 2)] = __builtin_bit_cast(bf16x8, l);
                if constexpr (STASH) *reinterpret_cast<u32x4*>(srow + 1024 * (2 * T + (I >> 2))) = h;
            }
        }
    }
};
// LDS of the density-gradient variant behind the side tables: 16 zero floats per lane half (the backward layers' "bias"), then per
// wave 8 KiB of pass bits [layer 8][tile 8][lane 64] u16
constexpr int MASK_ZERO_OFF = LDS_BYTES;
constexpr int MASK_OFF = LDS_BYTES + 128;
constexpr int LDS_BYTES_GRAD = MASK_OFF + 4 * 8192;   // 155 904 of 163 840
constexpr int LDS_BYTES_GRAD2 = MASK_OFF + 4 * 9216;  // VAR_TRUNK_BWD_FEAT2, VAR_NET_BWD: a ninth row of pass bits (views_linears.0): 160 000

// One layer of the k-step stream: NT output tiles; per tile NKE encoding k-steps (B = enc) then NKH
// (16, or 0 for the first layer) k-steps over the 256-feature activation `in`; three MFMA products
// per k-step.  Chunk boundaries (every 16 k-steps of the flat stream) are compile-time positions.
//
// Software pipeline inside a tile (one wave per SIMD: nothing else hides latency):
//   * the A fragments of k-step j+1 are read from LDS before the MFMAs of k-step j are issued;
//   * the epilogue of the PREVIOUS tile runs one slice per k-step between this tile's MFMAs
//     (pend(I) = slice I of the previous layer's last tile for tile 0, epi.slice<t-1, I> after);
//   * sched_barrier(0) fences keep that order at k-step granularity (inside a k-step the compiler
//     still interleaves the three MFMAs with the slice's VALU work).
// The last tile's accumulator is returned for the next layer's `pend`.
constexpr int DMA_K0 = 2;   // chunk-relative k-step after which the first DMA piece of the chunk two ahead is issued
template <int NT, int NKE, int NKH, int VARIANT, class PEND, class EPI, class ENC>
__device__ __forceinline__ Acc run_layer(Pipe<VARIANT>& P, const Act& in, const ENC& enc, const float* bias_tab,
                                         PEND&& pend, EPI& epi, int bias_stride = 32) {
    constexpr int N = NKE + NKH;                       // k-steps per tile
    Acc prev;
    prev.main = f32x16{0};
#ifdef IBL_DUAL_ACC
    prev.cross = f32x16{0};
#endif
    const char* frag = P.frag();
    bf16x8 ah = *reinterpret_cast<const bf16x8*>(frag);
    bf16x8 al = *reinterpret_cast<const bf16x8*>(frag + 1024);
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        // the accumulator starts at the layer bias (lane layout [tile][h][16]): no add in the epilogue
        Acc acc;
        acc.main = *reinterpret_cast<const f32x16*>(bias_tab + t * bias_stride);   // (0 for the backward layers: one row of zeros)
#ifdef IBL_DUAL_ACC
        acc.cross = f32x16{0};
#define ACC_X acc.cross
#else
#define ACC_X acc.main
#endif
        static_for<0, N>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int ks = t * N + j;              // k-step index inside the layer
            constexpr bool last = (t == NT - 1 && j == N - 1);
            constexpr bool next_new_chunk = ((ks + 1) % CHUNK_KSTEPS == 0);
            bf16x8 ah_n = ah, al_n = al;
            if constexpr (j == 0) TRACE(P, 4 * t + 0);
            if constexpr (j == 8) TRACE(P, 4 * t + 1);
#ifndef IBL_ABLATE_NO_FRAG    // (timing ablation: reuse the first fragment, no LDS reads in the loop)
            if constexpr (!last && !next_new_chunk) {   // prefetch the next k-step's fragments
                constexpr int off = ((ks + 1) % CHUNK_KSTEPS) * KSTEP_BYTES;
                ah_n = *reinterpret_cast<const bf16x8*>(frag + off);
                al_n = *reinterpret_cast<const bf16x8*>(frag + off + 1024);
            }
#endif
            // hipcc otherwise sinks these reads back to just before their first use (one k-step
            // later), exposing the full LDS latency to a wave that has nothing else to run
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j < NKE) {
                acc.main = MFMA(ah, enc.hi[j], acc.main);
                ACC_X = MFMA(ah, enc.lo[j], ACC_X);
                acc.main = MFMA(al, enc.hi[j], acc.main);
            } else {
                acc.main = MFMA(ah, in.hi[j - NKE], acc.main);
                ACC_X = MFMA(ah, in.lo[j - NKE], ACC_X);
                acc.main = MFMA(al, in.hi[j - NKE], acc.main);
            }
            // one LDS-DMA piece of the chunk two ahead per k-step, behind this k-step's MFMAs
            if constexpr (ks % CHUNK_KSTEPS >= DMA_K0 && ks % CHUNK_KSTEPS < DMA_K0 + 8)
                P.prefetch_piece(ks % CHUNK_KSTEPS - DMA_K0);
            // previous tile's epilogue: slices spread over k-steps 1 .. N-1
#ifndef IBL_ABLATE_NO_EPI     // (timing ablation: no epilogue work at all)
            static_for<0, 8>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (j == 1 + (i * (N - 1)) / 8) {
                    if constexpr (t == 0) pend(I);
                    else epi.template slice<(t > 0 ? t - 1 : 0), i>(prev);
                }
            });
#endif
            __builtin_amdgcn_sched_barrier(0);          // [MFMA x3 + one epilogue slice] stays one k-step wide
            if constexpr (next_new_chunk) TRACE(P, 4 * t + 2);
            if constexpr (!last && next_new_chunk) {    // the stream continues in the next ring slot
                P.end();
                TRACE(P, 4 * t + 3);
                frag = P.frag();
                ah_n = *reinterpret_cast<const bf16x8*>(frag);
                al_n = *reinterpret_cast<const bf16x8*>(frag + 1024);
            }
            ah = ah_n;
            al = al_n;
        });
        // Pin the finished accumulator here: its only consumer is the deferred epilogue, and without
        // this the optimiser sinks the whole MFMA chain across the chunk barrier to that use.
        // "a": keep it in the accumulator file.
#ifdef IBL_DUAL_ACC
        asm volatile("" : "+a"(acc.main), "+a"(acc.cross));
#else
        asm volatile("" : "+a"(acc.main));
#endif
        prev = acc;
    });
    P.end();
    return prev;
}

// [x, sin(2^k x), cos(2^k x)] in the slot order of layout.h::enc_ref_index (sincos_enc.h: one
// extended-precision range reduction per coordinate, exact 2^k scaling per frequency).
template <int PAIRS, int NK>
__device__ __forceinline__ void enc_values(float x, float y, float z, int h, float (&vals)[8 * NK]) {
    const float mul = h ? (float)(1 << (PAIRS / 3)) : 1.0f;   // half h starts at frequency index h * PAIRS/3
    const TurnPair tx = to_turns(x), ty = to_turns(y), tz = to_turns(z);
#pragma unroll
    for (int u = 0; u < PAIRS; ++u) {
        const TurnPair tc = (u % 3 == 0) ? tx : ((u % 3 == 1) ? ty : tz);
        sincos_turns(tc, (float)(1 << (u / 3)) * mul, &vals[2 * u], &vals[2 * u + 1]);
    }
    vals[2 * PAIRS] = h ? z : x;
    vals[2 * PAIRS + 1] = h ? 0.0f : y;
#pragma unroll
    for (int i = 2 * PAIRS + 2; i < 8 * NK; ++i) vals[i] = 0.0f;
}
template <int PAIRS, int NK>
__device__ __forceinline__ void encode(float x, float y, float z, int h, Enc& enc) {
    float vals[8 * NK];
    enc_values<PAIRS, NK>(x, y, z, h, vals);
#pragma unroll
    for (int jj = 0; jj < NK; ++jj) {
        u32x4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned hh, ll;
            split_pair(vals[8 * jj + 2 * e], vals[8 * jj + 2 * e + 1], hh, ll);
            hv[e] = hh;
            lv[e] = ll;
        }
        enc.hi[jj] = __builtin_bit_cast(bf16x8, hv);
        enc.lo[jj] = __builtin_bit_cast(bf16x8, lv);
    }
}

template <int VARIANT_IN>
__global__ __launch_bounds__(256, 1) void mlp_kernel(MlpArgs a) {
    // a list variant (VAR_FULL_LIST / VAR_TRUNK_LIST: a compact list of points with a flat index each, its length in device memory — the relevant samples of a
    // query, k_select_points) evaluates its base form; only where points come from, whose direction they take and where the rows go differs
    constexpr int VARIANT = variant_base(VARIANT_IN);
    constexpr bool LIST = variant_list(VARIANT_IN);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    float* tabs = reinterpret_cast<float*>(smem + LDS_RING_BYTES);

    // side tables -> LDS once per workgroup
    for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += 256)
        reinterpret_cast<f32x4*>(tabs)[i] = reinterpret_cast<const f32x4*>(a.tables)[i];
    if constexpr (VARIANT == VAR_TRUNK_GRAD || VARIANT == VAR_TRUNK_BWD || VARIANT == VAR_TRUNK_BWD_FEAT || VARIANT == VAR_TRUNK_BWD_FEAT2 || VARIANT == VAR_NET_BWD)
        if (threadIdx.x < 32) reinterpret_cast<float*>(smem + MASK_ZERO_OFF)[threadIdx.x] = 0.0f;
    __syncthreads();
    const float* ltab = tabs + h * 16;   // this lane-half's 16-float row inside every [2][16] entry

#ifdef IBL_TRACE
    if (blockIdx.x == 0 && threadIdx.x == 0) g_trace[41] = 777;
#endif
    Pipe<VARIANT> P;
    P.stream = a.stream;
    P.ring = smem;
    P.lane = lane;
    P.wave = wave;
    P.lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    P.voff = lane * 16 + wave * 8192;
    P.start();

    long n_total = a.n_pts;
    if constexpr (LIST) n_total = *a.n_pts_dev;
    const long n_groups = (n_total + 127) / 128;
    for (long g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const long p = g * 128 + wave * 32 + (lane & 31);
        const bool valid = p < n_total;
        float px = 0.f, py = 0.f, pz = 0.f;
        PointGenK gen = nullptr;
#ifndef IBL_NO_POINT_GEN   // (-DIBL_NO_POINT_GEN: A/B build of scratch/trunk_ab.sh, the input stage without the generation branch)
        if constexpr (VARIANT == VAR_TRUNK && !LIST) gen = kernarg_point_gen((unsigned)offsetof(MlpArgs, gen));
#endif
        if (gen != nullptr && gen->rays_o != nullptr) {   // the offset copies of the epsilon-normal, generated here (gen_points.h) instead of read from a batch
            if (valid) gen_offset_point(load_point_gen(gen), (unsigned)p, px, py, pz);
        } else if (valid) {
            px = a.pts[3 * p + 0];
            py = a.pts[3 * p + 1];
            pz = a.pts[3 * p + 2];
        }
        Enc pe, de;
        encode<PE_PAIRS_PER_HALF, PE_KSTEPS>(px, py, pz, h, pe);
        if constexpr (VARIANT != VAR_TRUNK && VARIANT != VAR_TRUNK_FEAT && VARIANT != VAR_TRUNK_GRAD && VARIANT != VAR_TRUNK_BWD &&
                      VARIANT != VAR_TRUNK_BWD_FEAT && !variant_ci(VARIANT)) {
            // direction encoding of this point's ray (run_network expands viewdirs over the samples,
            // ibl_nerf.py:244-247)
            float dx = 0.f, dy = 0.f, dz = 0.f;
            if (valid) {
                const unsigned r = (LIST ? (unsigned)a.out_index[p] : (unsigned)p) / (unsigned)a.pts_per_ray;   // n_pts < 2^31 per launch
                dx = a.dirs[3 * (size_t)r + 0];
                dy = a.dirs[3 * (size_t)r + 1];
                dz = a.dirs[3 * (size_t)r + 2];
            }
            encode<DE_PAIRS_PER_HALF, DE_KSTEPS>(dx, dy, dz, h, de);
        }

        Act A, B;
        float part[RAW_CH];
#pragma unroll
        for (int c = 0; c < RAW_CH; ++c) part[c] = 0.0f;
        const float* bias = ltab + TAB_BIAS;   // + tile*32: this lane-half's 16 biases of a tile
        auto none = [](auto) {};
        if constexpr (VARIANT == VAR_TRUNK_GRAD || VARIANT == VAR_TRUNK_BWD || VARIANT == VAR_TRUNK_BWD_FEAT || VARIANT == VAR_TRUNK_BWD_FEAT2 || VARIANT == VAR_NET_BWD) {
            // ---- density and its gradient with respect to the position: the trunk forward with every ReLU's pass bits
            // recorded, then the backward chain dZ(l-1) = (W(l)^T dZ(l)) * bits(l-1) on the transposed stream (what autograd
            // does for normal_from_depth.py:16-52, :102-137 through run_network, restricted to d raw[..., 0] / d pts).
            // VAR_TRUNK_BWD: dZ(7) carries the caller's dL / d sigma, and every layer's input and dZ fragments go to the stash
            // the weight-gradient kernel reads (train.py:479-481's backward through the trunk) ----
            constexpr bool NET = VARIANT == VAR_NET_BWD;             // the whole network: upstream gradient = dL/d raw rows
            constexpr bool FEAT2 = VARIANT == VAR_TRUNK_BWD_FEAT2;   // ... and dL/dh2 rows: feature_linear and views_linears.0 are differentiated here too
            constexpr bool ROWS = VARIANT == VAR_TRUNK_BWD_FEAT;   // upstream gradient = dL/dh7 rows
            constexpr bool BWD = VARIANT == VAR_TRUNK_BWD || ROWS || FEAT2 || NET;
            constexpr int MK = BWD ? 2 : 1;
            char* mbase = smem + MASK_OFF + wave * ((FEAT2 || NET) ? 9216 : 8192) + lane * 2;   // + 1024 * layer + 128 * tile
            const float* zero = reinterpret_cast<const float*>(smem + MASK_ZERO_OFF) + h * 16;
            const long wgs = n_groups * 4, wgi = g * 4 + wave;          // wave groups of 32 points (layout.h: STASH_*)
            auto srow = [&](int what) -> char* { return BWD ? a.stash + stash_offset(what, wgs, wgi) + lane * 16 : nullptr; };
            if constexpr (BWD) {
#pragma unroll
                for (int jj = 0; jj < PE_KSTEPS; ++jj) *reinterpret_cast<bf16x8*>(srow(STASH_ENC) + 1024 * jj) = pe.hi[jj];
            }
            using EM = Epi<true, true, 0, MK>;
            EM eA{&A, {nullptr}, {nullptr}, mbase, srow(STASH_X + 0)}, eB{&B, {nullptr}, {nullptr}, mbase + 1024, srow(STASH_X + 1)};
#define IBL_PEND(e, T, acc) [&](auto I) { (e).template slice<T, decltype(I)::value>(acc); }
            Acc pacc = run_layer<8, PE_KSTEPS, 0>(P, A, pe, bias + BT_L0 * 32, none, eA);                    // layer 0 -> A
            pacc = run_layer<8, 0, 16>(P, A, pe, bias + (BT_L0 + 8) * 32, IBL_PEND(eA, 7, pacc), eB);         // 1 -> B
            eA.mrow = mbase + 2 * 1024; eA.srow = srow(STASH_X + 2);
            pacc = run_layer<8, 0, 16>(P, B, pe, bias + (BT_L0 + 16) * 32, IBL_PEND(eB, 7, pacc), eA);        // 2 -> A
            eB.mrow = mbase + 3 * 1024; eB.srow = srow(STASH_X + 3);
            pacc = run_layer<8, 0, 16>(P, A, pe, bias + (BT_L0 + 24) * 32, IBL_PEND(eA, 7, pacc), eB);        // 3 -> B
            eA.mrow = mbase + 4 * 1024; eA.srow = srow(STASH_X + 4);
            pacc = run_layer<8, 0, 16>(P, B, pe, bias + (BT_L0 + 32) * 32, IBL_PEND(eB, 7, pacc), eA);        // 4 -> A
            eB.mrow = mbase + 5 * 1024; eB.srow = srow(STASH_X + 5);
            pacc = run_layer<8, PE_KSTEPS, 16>(P, A, pe, bias + (BT_L0 + 40) * 32, IBL_PEND(eA, 7, pacc), eB);  // 5 (skip) -> B
            eA.mrow = mbase + 6 * 1024; eA.srow = srow(STASH_X + 6);
            pacc = run_layer<8, 0, 16>(P, B, pe, bias + (BT_L0 + 48) * 32, IBL_PEND(eB, 7, pacc), eA);        // 6 -> A
            // layer 7: sigma head + bits (its fragments are formed only for the stash: the head's weight gradient reads them)
            Epi<BWD, true, 1, MK> e7{&B, {&part[0]}, {ltab + TAB_SIG}, mbase + 7 * 1024, srow(STASH_X + 7)};
            pacc = run_layer<8, 0, 16>(P, A, pe, bias + (BT_L0 + 56) * 32, IBL_PEND(eA, 7, pacc), e7);
            static_for<0, 8>([&](auto I) { e7.template slice<7, decltype(I)::value>(pacc); });
            const float sigma = part[0] + __shfl_xor(part[0], 32) + tabs[TAB_SCALAR];
            float up = 1.0f;                                   // dL / d sigma of this point (invalid points: 0, they add nothing to any gradient)
            if constexpr (BWD && !ROWS && !FEAT2 && !NET) up = valid ? a.dsigma[p] * a.grad_scale : 0.0f;
            if constexpr (FEAT2) {
                // feature_linear (no activation: B = h7 -> A) and views_linears.0 ([feature, dir27] -> ReLU: only its pass bits are kept),
                // ibl_nerf.py:193-197; then their backward: dZv = dL/dh2 * bits -> A, dFeat = Wv^T dZv -> B, dZ(7) = (Wf^T dFeat + dL/dh7) * bits(7) -> A
#pragma unroll
                for (int jj = 0; jj < DE_KSTEPS; ++jj) *reinterpret_cast<bf16x8*>(srow(STASH_DENC) + 1024 * jj) = de.hi[jj];
                Epi<true, false, 0, 3> eF{&A, {nullptr}, {nullptr}, nullptr, srow(STASH_XF)};
                pacc = run_layer<8, 0, 16>(P, B, pe, bias + BT_FEAT * 32, none, eF);
                Epi<false, true, 0, 1> eV{nullptr, {nullptr}, {nullptr}, mbase + 8 * 1024, nullptr};
                pacc = run_layer<8, DE_KSTEPS, 16>(P, A, de, bias + BT_VIEW * 32, IBL_PEND(eF, 7, pacc), eV);
                static_for<0, 8>([&](auto I) { eV.template slice<7, decltype(I)::value>(pacc); });
                const float sc_ = valid ? a.grad_scale : 0.0f;
                static_for<0, 8>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    const unsigned mw = *reinterpret_cast<const unsigned short*>(mbase + 8 * 1024 + 128 * t);
                    static_for<0, 2>([&](auto Q) {
                        constexpr int q = decltype(Q)::value;
                        const float* row = a.dh2 + (size_t)(valid ? p : 0) * 256 + 32 * t + 16 * q + 4 * h;
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(row), r1 = *reinterpret_cast<const f32x4*>(row + 8);
                        const float wv[8] = {r0[0] * sc_, r0[1] * sc_, r0[2] * sc_, r0[3] * sc_, r1[0] * sc_, r1[1] * sc_, r1[2] * sc_, r1[3] * sc_};
                        u32x4 hv, lv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = 4 * q + e;
                            unsigned hh, ll;
                            split_pair((mw & (1u << (2 * i))) ? wv[2 * e] : 0.0f, (mw & (2u << (2 * i))) ? wv[2 * e + 1] : 0.0f, hh, ll);
                            hv[e] = hh;
                            lv[e] = ll;
                        }
                        A.hi[2 * t + q] = __builtin_bit_cast(bf16x8, hv);
                        A.lo[2 * t + q] = __builtin_bit_cast(bf16x8, lv);
                        *reinterpret_cast<u32x4*>(srow(STASH_DZV) + 1024 * (2 * t + q)) = hv;
                    });
                });
            }

            // (declared here: the FEAT2 / NET forms' last head layer hands its pending tile to the trunk's first backward layer)
            float dr[RAW_CH];   // NET: this point's dL / d raw, times the gradient scale
            if constexpr (NET) {
#pragma unroll
                for (int c = 0; c < RAW_CH; ++c) dr[c] = valid ? a.draw[(size_t)p * RAW_CH + c] * a.grad_scale : 0.0f;
            }
            EpiG<false, true, 1> gF{&B, nullptr, nullptr, (FEAT2 || NET) ? srow(STASH_DZF) : nullptr, nullptr, 0.0f, {nullptr, nullptr, nullptr}, {0.f, 0.f, 0.f}};
            EpiG<false, true, NET ? 3 : 2> g7{&A, mbase + 7 * 1024, nullptr, (FEAT2 || NET) ? srow(STASH_DZ + 7) : nullptr,
                                             (FEAT2 && valid) ? a.dh7 + (size_t)p * 256 + 4 * h : nullptr, a.grad_scale,
                                             {ltab + TAB_SIG, ltab + TAB_ROUGH, ltab + TAB_ROUGH}, {NET ? dr[0] : 0.f, NET ? dr[4] : 0.f, 0.f}};
            if constexpr (NET) {
                // ---- the head layers, forward in FULL's order (chunks 60..96), keeping what the backward needs: feature_linear's output and
                // h2 as fragments + stash, the five 128-wide feature layers' outputs in the stash only ----
#pragma unroll
                for (int jj = 0; jj < DE_KSTEPS; ++jj) *reinterpret_cast<bf16x8*>(srow(STASH_DENC) + 1024 * jj) = de.hi[jj];
                Epi<true, false, 0, 3> eF{&A, {nullptr}, {nullptr}, nullptr, srow(STASH_XF)};
                pacc = run_layer<8, 0, 16>(P, B, pe, bias + BT_FEAT * 32, none, eF);                                   // feature_linear: h7 (B) -> A
                Epi<false, true, 0, 4> eAl{nullptr, {nullptr}, {nullptr}, nullptr, srow(STASH_FA)};
                pacc = run_layer<4, 0, 16>(P, B, pe, bias + BT_ALB * 32, IBL_PEND(eF, 7, pacc), eAl);                  // albedo_feature_linear
                Epi<false, true, 0, 4> eIr{nullptr, {nullptr}, {nullptr}, nullptr, srow(STASH_FI)};
                pacc = run_layer<4, 0, 16>(P, B, pe, bias + BT_IRR * 32, IBL_PEND(eAl, 3, pacc), eIr);                 // irradiance_feature_linear
                Epi<true, true, 0, 2> eV{&B, {nullptr}, {nullptr}, mbase + 8 * 1024, srow(STASH_XH2)};
                pacc = run_layer<8, DE_KSTEPS, 16>(P, A, de, bias + BT_VIEW * 32, IBL_PEND(eIr, 3, pacc), eV);         // views_linears.0 -> h2 (B)
                Epi<false, true, 0, 4> eA0{nullptr, {nullptr}, {nullptr}, nullptr, srow(STASH_F0)}, eA1{nullptr, {nullptr}, {nullptr}, nullptr, srow(STASH_F0 + 1)},
                    eA2{nullptr, {nullptr}, {nullptr}, nullptr, srow(STASH_F0 + 2)};
                pacc = run_layer<4, 0, 16>(P, B, pe, bias + BT_AR * 32, IBL_PEND(eV, 7, pacc), eA0);                   // additional_radiance_feature_linear.0-2
                pacc = run_layer<4, 0, 16>(P, B, pe, bias + (BT_AR + 4) * 32, IBL_PEND(eA0, 3, pacc), eA1);
                pacc = run_layer<4, 0, 16>(P, B, pe, bias + (BT_AR + 8) * 32, IBL_PEND(eA1, 3, pacc), eA2);
                static_for<0, 8>([&](auto I) { eA2.template slice<3, decltype(I)::value>(pacc); });
                // dL/d pre-activation of a 128-wide feature layer = [its stashed output > 0] * (its N = 1/3 head's weights . upstream channels):
                // 8 k-steps of fragments, formed on the VALU from the head tables (lane layout, channel c at tab + 128 c)
                auto form = [&](auto NC, const char* fstash, const float* tab, const float* dch, bf16x8* hi, bf16x8* lo, char* dstash) {
                    constexpr int nc = decltype(NC)::value;
                    static_for<0, 8>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        const bf16x8 f = *reinterpret_cast<const bf16x8*>(fstash + 1024 * j);
                        u32x4 hv, lv;
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            float v[2];
#pragma unroll
                            for (int o = 0; o < 2; ++o) {
                                const int e = 2 * e2 + o;
                                float acc = 0.0f;
#pragma unroll
                                for (int c = 0; c < nc; ++c) acc = fmaf(tab[128 * c + (j >> 1) * 32 + 8 * (j & 1) + e], dch[c], acc);
                                v[o] = (float)f[e] != 0.0f ? acc : 0.0f;
                            }
                            unsigned hh, ll;
                            split_pair(v[0], v[1], hh, ll);
                            hv[e2] = hh;
                            lv[e2] = ll;
                        }
                        hi[j] = __builtin_bit_cast(bf16x8, hv);
                        lo[j] = __builtin_bit_cast(bf16x8, lv);
                        *reinterpret_cast<u32x4*>(dstash + 1024 * j) = hv;
                    });
                };
                using N1 = std::integral_constant<int, 1>;
                using N3 = std::integral_constant<int, 3>;
                // the feature layers' outputs are read back from the stash this lane wrote them to: all its stores must have completed first
                // (once per 128 points; also drains the two prefetched weight chunks, which have long landed)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                {
                    Act DF01;      // dF.0 | dF.1
                    Half DF2;
                    form(N3{}, srow(STASH_F0), ltab + TAB_AR, &dr[9], &DF01.hi[0], &DF01.lo[0], srow(STASH_DF0));
                    form(N3{}, srow(STASH_F0 + 1), ltab + TAB_AR + 384, &dr[12], &DF01.hi[8], &DF01.lo[8], srow(STASH_DF0 + 1));
                    form(N3{}, srow(STASH_F0 + 2), ltab + TAB_AR + 768, &dr[15], &DF2.hi[0], &DF2.lo[0], srow(STASH_DF0 + 2));
                    // dL/dh2 = sum_k ARF.k^T dF.k + radiance_linear^T dRad;  dZv = that * bits(views) -> A
                    EpiG<false, true, 3> gV{&A, mbase + 8 * 1024, nullptr, srow(STASH_DZV), nullptr, 0.0f,
                                            {ltab + TAB_RAD, ltab + TAB_RAD + 256, ltab + TAB_RAD + 512}, {dr[6], dr[7], dr[8]}};
                    pacc = run_layer<8, 8, 16>(P, DF01, DF2, zero, none, gV, 0);
                    pacc = run_layer<8, 0, 16>(P, A, pe, zero, IBL_PEND(gV, 7, pacc), gF, 0);                          // Wv^T (feature columns): dZv (A) -> dFeat (B)
                }
                {
                    Act DFAI;      // dFa | dFi
                    form(N3{}, srow(STASH_FA), ltab + TAB_ALB, &dr[1], &DFAI.hi[0], &DFAI.lo[0], srow(STASH_DFA));
                    form(N1{}, srow(STASH_FI), ltab + TAB_IRR, &dr[5], &DFAI.hi[8], &DFAI.lo[8], srow(STASH_DFI));
                    // dL/dh7 = Wf^T dFeat + ALBF^T dFa + IRRF^T dFi + sigma_linear^T dsigma + roughness_linear^T drough;  dZ7 = that * bits(7) -> A
                    // (dFeat is the SECOND operand: its last k-steps are completed by the pending epilogue during this layer's first tile, whose
                    // slices are spread over all 32 k-steps — as the first operand it would be read at k-steps 14, 15, before slice 7 at k-step 28)
                    pacc = run_layer<8, 16, 16>(P, B, DFAI, zero, IBL_PEND(gF, 7, pacc), g7, 0);
                }
            } else if constexpr (FEAT2) {
                pacc = run_layer<8, 0, 16>(P, A, pe, zero, none, gF, 0);                                       // Wv^T (feature columns): dZv (A) -> dFeat (B)
                pacc = run_layer<8, 0, 16>(P, B, pe, zero, IBL_PEND(gF, 7, pacc), g7, 0);                      // Wf^T: dFeat (B) [+ dL/dh7] -> dZ7 (A)
            } else {
                // dZ(7) = dL/dsigma * sigma_linear.weight * bits(7) -> A
                static_for<0, 8>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    const unsigned mw = *reinterpret_cast<const unsigned short*>(mbase + 7 * 1024 + 128 * t);
                    static_for<0, 2>([&](auto Q) {
                        constexpr int q = decltype(Q)::value;
                        u32x4 hv, lv;
                        float wv[8];   // dL / d h7 of accumulator registers 8q .. 8q+7 = features 32t + 16q + 4h + {0..3, 8..11}
                        if constexpr (ROWS) {
                            const float* row = a.dh7 + (size_t)(valid ? p : 0) * 256 + 32 * t + 16 * q + 4 * h;
                            const f32x4 r0 = *reinterpret_cast<const f32x4*>(row), r1 = *reinterpret_cast<const f32x4*>(row + 8);
                            const float sc_ = valid ? a.grad_scale : 0.0f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { wv[e] = r0[e] * sc_; wv[4 + e] = r1[e] * sc_; }
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) wv[e] = ltab[TAB_SIG + t * 32 + 8 * q + e] * up;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = 4 * q + e;
                            unsigned hh, ll;
                            split_pair((mw & (1u << (2 * i))) ? wv[2 * e] : 0.0f, (mw & (2u << (2 * i))) ? wv[2 * e + 1] : 0.0f, hh, ll);
                            hv[e] = hh;
                            lv[e] = ll;
                        }
                        A.hi[2 * t + q] = __builtin_bit_cast(bf16x8, hv);
                        A.lo[2 * t + q] = __builtin_bit_cast(bf16x8, lv);
                        if constexpr (BWD) *reinterpret_cast<u32x4*>(srow(STASH_DZ + 7) + 1024 * (2 * t + q)) = hv;
                    });
                });
            }
            float genc[32];
            EpiG<false, BWD> gA{&A, mbase, genc, nullptr, nullptr, 0.0f, {nullptr, nullptr, nullptr}, {0.f, 0.f, 0.f}}, gB{&B, mbase + 6 * 1024, genc, srow(STASH_DZ + 6), nullptr, 0.0f, {nullptr, nullptr, nullptr}, {0.f, 0.f, 0.f}};
            if constexpr (FEAT2 || NET) pacc = run_layer<8, 0, 16>(P, A, pe, zero, IBL_PEND(g7, 7, pacc), gB, 0);     // W7^T: dZ7 (A) -> dZ6 (B)
            else pacc = run_layer<8, 0, 16>(P, A, pe, zero, none, gB, 0);
            gA.mrow = mbase + 5 * 1024; gA.srow = srow(STASH_DZ + 5);
            pacc = run_layer<8, 0, 16>(P, B, pe, zero, IBL_PEND(gB, 7, pacc), gA, 0);                          // W6^T -> dZ5 (A)
            gB.mrow = mbase + 4 * 1024; gB.srow = srow(STASH_DZ + 4);
            pacc = run_layer<10, 0, 16>(P, A, pe, zero, IBL_PEND(gA, 7, pacc), gB, 0);                         // W5^T -> dZ4 (B), encoding gradient
            gA.mrow = mbase + 3 * 1024; gA.srow = srow(STASH_DZ + 3);
            pacc = run_layer<8, 0, 16>(P, B, pe, zero, IBL_PEND(gB, 9, pacc), gA, 0);                          // W4^T -> dZ3 (A)
            gB.mrow = mbase + 2 * 1024; gB.srow = srow(STASH_DZ + 2);
            pacc = run_layer<8, 0, 16>(P, A, pe, zero, IBL_PEND(gA, 7, pacc), gB, 0);                          // W3^T -> dZ2 (B)
            gA.mrow = mbase + 1 * 1024; gA.srow = srow(STASH_DZ + 1);
            pacc = run_layer<8, 0, 16>(P, B, pe, zero, IBL_PEND(gB, 7, pacc), gA, 0);                          // W2^T -> dZ1 (A)
            gB.mrow = mbase; gB.srow = srow(STASH_DZ + 0);
            pacc = run_layer<8, 0, 16>(P, A, pe, zero, IBL_PEND(gA, 7, pacc), gB, 0);                          // W1^T -> dZ0 (B)
            EpiG<true> g0{nullptr, nullptr, genc, nullptr, nullptr, 0.0f, {nullptr, nullptr, nullptr}, {0.f, 0.f, 0.f}};
            pacc = run_layer<2, 0, 16>(P, B, pe, zero, IBL_PEND(gB, 7, pacc), g0, 0);                          // W0^T: + encoding gradient
            static_for<0, 8>([&](auto I) { g0.template slice<1, decltype(I)::value>(pacc); });
#undef IBL_PEND
            // chain rule through the encoding: slot 2u = sin(f x_c), 2u+1 = cos(f x_c) of pair u (c = u % 3, f = 2^(u/3) [* 32 in half 1]),
            // slots 30, 31 = (x, y) | (z, pad)
            float vals[8 * PE_KSTEPS];
            if (valid) {   // (re-read: three registers less across the sixteen layers)
                px = a.pts[3 * p + 0];
                py = a.pts[3 * p + 1];
                pz = a.pts[3 * p + 2];
            }
            enc_values<PE_PAIRS_PER_HALF, PE_KSTEPS>(px, py, pz, h, vals);
            float g3[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int u = 0; u < PE_PAIRS_PER_HALF; ++u) {
                const float f = (float)(1 << (u / 3)) * (h ? (float)(1 << (PE_PAIRS_PER_HALF / 3)) : 1.0f);
                g3[u % 3] += f * (genc[2 * u] * vals[2 * u + 1] - genc[2 * u + 1] * vals[2 * u]);
            }
            g3[h ? 2 : 0] += genc[2 * PE_PAIRS_PER_HALF];
            if (!h) g3[1] += genc[2 * PE_PAIRS_PER_HALF + 1];
#pragma unroll
            for (int c = 0; c < 3; ++c) g3[c] += __shfl_xor(g3[c], 32);
            if constexpr (BWD) {
                const float inv = 1.0f / a.grad_scale;
#pragma unroll
                for (int c = 0; c < 3; ++c) g3[c] *= inv;
            }
            if (valid && h == 0) {
                f32x4 o = {sigma, g3[0], g3[1], g3[2]};
                *reinterpret_cast<f32x4*>(a.out + 4 * p) = o;
            }
#ifdef IBL_F16X3
            {   // range guard (see split_pair).  Bit 0: the forward left the f16 range (sigma is not finite); bit 1: only the backward did —
                // a loss scale too large for this batch, which the caller answers with a smaller scale, not with the bf16 kernels
                const float chk = fmaf(g3[0], 0.0f, fmaf(g3[1], 0.0f, g3[2] * 0.0f));
                const bool fwd_bad = !(fabsf(sigma) < __builtin_inff());
                if (valid && (fwd_bad || chk != chk) && a.range_flag != nullptr) atomicOr(a.range_flag, fwd_bad ? 1u : 2u);
            }
#endif
            continue;
        }
        // whole epilogue of a layer's last tile, run before anything else may read its results
        auto flush = [&](auto& e, auto T, const Acc& acc) {
            static_for<0, 8>([&](auto I) { e.template slice<decltype(T)::value, decltype(I)::value>(acc); });
        };
        using T7 = std::integral_constant<int, 7>;
        using T3 = std::integral_constant<int, 3>;
        // ReLU -> fragments of A / B
        Epi<true, true, 0> eA{&A, {nullptr}, {nullptr}}, eB{&B, {nullptr}, {nullptr}};

        // ---- positions_linears.0 : 63 -> 256, ReLU  (-> A) -------------------------------------
        Acc pacc = run_layer<8, PE_KSTEPS, 0>(P, A /*unused*/, pe, bias + BT_L0 * 32, none, eA);
        // ---- positions_linears.1..4 : 256 -> 256, ReLU, two layers per trip (A -> B -> A) -------
#ifdef IBL_ABLATE_LOOPONLY    // timing ablation: 58 layers through the SAME 2-layer loop body (I-cache resident)
        for (int l = 1; l <= 57; l += 2) {
#else
        for (int l = 1; l <= 3; l += 2) {
#endif
#ifdef IBL_TRACE
            P.trace = (blockIdx.x == 0 && wave == 0 && g == (long)blockIdx.x + 2 * (long)gridDim.x && l == 1);
            P.tr = reinterpret_cast<long long*>(smem + LDS_BYTES);
#endif
            pacc = run_layer<8, 0, 16>(P, A, pe, bias + (BT_L0 + 8 * (l & 3)) * 32,
                                       [&](auto I) { eA.template slice<7, decltype(I)::value>(pacc); }, eB);
#ifdef IBL_TRACE
            if (P.trace) { for (int i = lane; i < 32; i += 64) g_trace[i] = P.tr[i]; g_trace[40] = 12345; }
            P.trace = false;
#endif
            pacc = run_layer<8, 0, 16>(P, B, pe, bias + (BT_L0 + 8 * (l & 3) + 8) * 32,
                                       [&](auto I) { eB.template slice<7, decltype(I)::value>(pacc); }, eA);
        }
        // ---- positions_linears.5 : cat([x63, h]) -> 256, ReLU (ibl_nerf.py:167-168)  (A -> B) ---
        pacc = run_layer<8, PE_KSTEPS, 16>(P, A, pe, bias + (BT_L0 + 40) * 32,
                                           [&](auto I) { eA.template slice<7, decltype(I)::value>(pacc); }, eB);
        // ---- positions_linears.6  (B -> A) -------------------------------------------------------
        pacc = run_layer<8, 0, 16>(P, B, pe, bias + (BT_L0 + 48) * 32,
                                   [&](auto I) { eB.template slice<7, decltype(I)::value>(pacc); }, eA);
        // ---- positions_linears.7  (A -> B); sigma_linear / roughness_linear on its fp32 activations
        constexpr bool CI = variant_ci(VARIANT), ALBIRR = variant_albirr(VARIANT);
        const float* rad[3] = {ltab + TAB_RAD, ltab + TAB_RAD + 256, ltab + TAB_RAD + 512};
        // with is_color_independent_to_direction the radiance_linear rows are dotted with h7 itself (ibl_nerf.py:192, :199)
        auto e7 = [&] {
            if constexpr (VARIANT == VAR_FULL)
                return Epi<true, true, 2>{&B, {&part[0], &part[4]}, {ltab + TAB_SIG, ltab + TAB_ROUGH}};
            else if constexpr (VARIANT == VAR_FULL_CI)
                return Epi<true, true, 5>{&B, {&part[0], &part[4], &part[6], &part[7], &part[8]},
                                          {ltab + TAB_SIG, ltab + TAB_ROUGH, rad[0], rad[1], rad[2]}};
            else if constexpr (VARIANT == VAR_REFL_CI)
                return Epi<true, true, 4>{&B, {&part[0], &part[6], &part[7], &part[8]}, {ltab + TAB_SIG, rad[0], rad[1], rad[2]}};
            else if constexpr (VARIANT == VAR_TRUNK_FEAT)   // h7 itself is the output (sigma is still formed: it carries the range guard's NaN)
                return Epi<false, true, 1, 0, true>{&B, {&part[0]}, {ltab + TAB_SIG}, nullptr, nullptr, valid ? a.out + (size_t)p * 256 + 4 * h : nullptr};
            else if constexpr (VARIANT == VAR_TRUNK_FEAT2)  // ... and the operand of feature_linear
                return Epi<true, true, 1, 0, true>{&B, {&part[0]}, {ltab + TAB_SIG}, nullptr, nullptr, valid ? a.out + (size_t)p * 256 + 4 * h : nullptr};
            else
                return Epi<VARIANT != VAR_TRUNK, true, 1>{&B, {&part[0]}, {ltab + TAB_SIG}};
        }();
        pacc = run_layer<8, 0, 16>(P, A, pe, bias + (BT_L0 + 56) * 32,
                                   [&](auto I) { eA.template slice<7, decltype(I)::value>(pacc); }, e7);

        if constexpr (VARIANT == VAR_TRUNK || VARIANT == VAR_TRUNK_FEAT) {
            flush(e7, T7{}, pacc);
        } else if constexpr (VARIANT == VAR_TRUNK_FEAT2) {
            // feature_linear (no activation) and views_linears.0 ([feature, dir27], ReLU) — ibl_nerf.py:193-197; its output rows go to out2
            Epi<true, false, 0> eFeat{&A, {nullptr}, {nullptr}};
            pacc = run_layer<8, 0, 16>(P, B, pe, bias + BT_FEAT * 32, [&](auto I) { e7.template slice<7, decltype(I)::value>(pacc); }, eFeat);
            Epi<false, true, 0, 0, true> eV{nullptr, {nullptr}, {nullptr}, nullptr, nullptr, valid ? a.out2 + (size_t)p * 256 + 4 * h : nullptr};
            pacc = run_layer<8, DE_KSTEPS, 16>(P, A, de, bias + BT_VIEW * 32, [&](auto I) { eFeat.template slice<7, decltype(I)::value>(pacc); }, eV);
            flush(eV, T7{}, pacc);
        } else {
            Epi<true, false, 0> eFeat{&A, {nullptr}, {nullptr}};
            Epi<false, true, 3> eAlb{nullptr, {&part[1], &part[2], &part[3]},
                                     {ltab + TAB_ALB, ltab + TAB_ALB + 128, ltab + TAB_ALB + 256}};
            Epi<false, true, 1> eIrr{nullptr, {&part[5]}, {ltab + TAB_IRR}};
            Epi<true, true, 3> eView{&B, {&part[6], &part[7], &part[8]}, {rad[0], rad[1], rad[2]}};
            Epi<false, true, 3> eAr0{nullptr, {&part[9], &part[10], &part[11]},
                                     {ltab + TAB_AR, ltab + TAB_AR + 128, ltab + TAB_AR + 256}};
            Epi<false, true, 3> eAr1{nullptr, {&part[12], &part[13], &part[14]},
                                     {ltab + TAB_AR + 384, ltab + TAB_AR + 512, ltab + TAB_AR + 640}};
            Epi<false, true, 3> eAr2{nullptr, {&part[15], &part[16], &part[17]},
                                     {ltab + TAB_AR + 768, ltab + TAB_AR + 896, ltab + TAB_AR + 1024}};
            // the epilogue still owed by the previous layer, as the `pend` of the next one
#define IBL_PEND(e, T, acc) [&](auto I) { (e).template slice<T, decltype(I)::value>(acc); }
            // feature_linear : 256 -> 256, no activation (B = h7 -> A = feature)
            if constexpr (!CI) pacc = run_layer<8, 0, 16>(P, B, pe, bias + BT_FEAT * 32, IBL_PEND(e7, 7, pacc), eFeat);
            // albedo_feature_linear (ReLU) -> albedo_linear ; irradiance_feature_linear -> irradiance_linear (both read h7 = B)
            if constexpr (ALBIRR) {
                Acc qacc;
                if constexpr (CI) qacc = run_layer<4, 0, 16>(P, B, pe, bias + BT_ALB * 32, IBL_PEND(e7, 7, pacc), eAlb);
                else qacc = run_layer<4, 0, 16>(P, B, pe, bias + BT_ALB * 32, IBL_PEND(eFeat, 7, pacc), eAlb);
                pacc = run_layer<4, 0, 16>(P, B, pe, bias + BT_IRR * 32, IBL_PEND(eAlb, 3, qacc), eIrr);
            }
            // views_linears.0 : cat([feature, dir27]) -> 256, ReLU (A -> B) ; radiance_linear
            if constexpr (!CI) {
                if constexpr (ALBIRR) pacc = run_layer<8, DE_KSTEPS, 16>(P, A, de, bias + BT_VIEW * 32, IBL_PEND(eIrr, 3, pacc), eView);
                else pacc = run_layer<8, DE_KSTEPS, 16>(P, A, de, bias + BT_VIEW * 32, IBL_PEND(eFeat, 7, pacc), eView);
            }
            // additional_radiance_feature_linear.k (ReLU) -> additional_radiance_linear.k, on B = views output, or h7 when colour-independent
            if constexpr (!CI) pacc = run_layer<4, 0, 16>(P, B, pe, bias + BT_AR * 32, IBL_PEND(eView, 7, pacc), eAr0);
            else if constexpr (ALBIRR) pacc = run_layer<4, 0, 16>(P, B, pe, bias + BT_AR * 32, IBL_PEND(eIrr, 3, pacc), eAr0);
            else pacc = run_layer<4, 0, 16>(P, B, pe, bias + BT_AR * 32, IBL_PEND(e7, 7, pacc), eAr0);
            pacc = run_layer<4, 0, 16>(P, B, pe, bias + (BT_AR + 4) * 32, IBL_PEND(eAr0, 3, pacc), eAr1);
            pacc = run_layer<4, 0, 16>(P, B, pe, bias + (BT_AR + 8) * 32, IBL_PEND(eAr1, 3, pacc), eAr2);
#undef IBL_PEND
            flush(eAr2, T3{}, pacc);
        }

        // ---- combine the two lane halves, add head biases, store ------------------------------
        const float* sc = tabs + TAB_SCALAR;
        if constexpr (VARIANT == VAR_TRUNK || VARIANT == VAR_TRUNK_FEAT || VARIANT == VAR_TRUNK_FEAT2) {
            const float s = part[0] + __shfl_xor(part[0], 32) + sc[0];
            if (VARIANT == VAR_TRUNK && valid && h == 0) a.out[(LIST ? (long)a.out_index[p] : (long)p) * a.out_stride] = s;
#ifdef IBL_F16X3
            if (valid && !(fabsf(s) < __builtin_inff()) && a.range_flag != nullptr) atomicOr(a.range_flag, 1u);   // range guard (see split_pair)
#endif
        } else {
            float tot[RAW_CH];
#pragma unroll
            for (int c = 0; c < RAW_CH; ++c) tot[c] = part[c] + __shfl_xor(part[c], 32) + sc[c];
#ifdef IBL_F16X3
            {   // range guard (see split_pair): 0 * x is NaN exactly when x is inf or NaN
                float chk = 0.0f;
#pragma unroll
                for (int c = 0; c < RAW_CH; ++c) chk = fmaf(tot[c], 0.0f, chk);
                if (valid && chk != chk && a.range_flag != nullptr) atomicOr(a.range_flag, 1u);
            }
#endif
            if (valid) {
                const long row = LIST ? (long)a.out_index[p] : p;        // (a list: the sample's own row of the query's output)
                if constexpr (variant_albirr(VARIANT)) {
                    float* o = a.out + row * RAW_CH;
                    if (h == 0) {
#pragma unroll
                        for (int c = 0; c < 9; ++c) o[c] = tot[c];
                    } else {
#pragma unroll
                        for (int c = 9; c < 18; ++c) o[c] = tot[c];
                    }
                } else {
                    float* o = a.out + row * REFL_CH;
                    if (h == 0) {
                        o[0] = tot[0];
#pragma unroll
                        for (int c = 1; c < 7; ++c) o[c] = tot[5 + c];
                    } else {
#pragma unroll
                        for (int c = 7; c < 13; ++c) o[c] = tot[5 + c];
                    }
                }
            }
        }
    }
    P.drain();   // the two chunks prefetched past the end are never consumed
}

// The five instantiations compile as five objects (build.py passes -DIBL_VARIANT=0..4) so that they build side by
// side; the object of VAR_FULL also carries the dispatcher.  Without the macro everything lands in one object
// (scratch/build_ablate.sh, trace builds).
#ifdef IBL_TRACE
constexpr int LDS_LAUNCH = LDS_BYTES + 2048;
#else
constexpr int LDS_LAUNCH = LDS_BYTES;
#endif
#ifdef IBL_F16X3
}  // namespace f16x3k
#define IBL_KNS f16x3k::
#define IBL_LAUNCH_NAME(V) launch_mlp_f16x3_v##V
#define IBL_DISPATCH launch_mlp_f16x3
#else
#define IBL_KNS
#define IBL_LAUNCH_NAME(V) launch_mlp_v##V
#define IBL_DISPATCH launch_mlp
#endif
template <int VARIANT>
static hipError_t launch_variant(const MlpArgs& a, int grid, hipStream_t stream) {
    static bool attr_set[64] = {};   // per device: one process may hold contexts on several (iblnerf_options.device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)IBL_KNS mlp_kernel<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (VARIANT == VAR_TRUNK_BWD_FEAT2 || VARIANT == VAR_NET_BWD) ? IBL_KNS LDS_BYTES_GRAD2 : (VARIANT == VAR_TRUNK_GRAD || VARIANT == VAR_TRUNK_BWD || VARIANT == VAR_TRUNK_BWD_FEAT) ? IBL_KNS LDS_BYTES_GRAD : LDS_BYTES + 2048);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(IBL_KNS mlp_kernel<VARIANT>, dim3(grid), dim3(256), (VARIANT == VAR_TRUNK_BWD_FEAT2 || VARIANT == VAR_NET_BWD) ? IBL_KNS LDS_BYTES_GRAD2 : (VARIANT == VAR_TRUNK_GRAD || VARIANT == VAR_TRUNK_BWD || VARIANT == VAR_TRUNK_BWD_FEAT) ? IBL_KNS LDS_BYTES_GRAD : IBL_KNS LDS_LAUNCH, stream, a);
    return hipGetLastError();
}
#define IBL_DEFINE_LAUNCH(V) hipError_t IBL_LAUNCH_NAME(V)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<V>(a, grid, s); }
#if defined(IBL_VARIANT)
#if IBL_VARIANT == 0
IBL_DEFINE_LAUNCH(0)
#elif IBL_VARIANT == 1
IBL_DEFINE_LAUNCH(1)
#elif IBL_VARIANT == 2
IBL_DEFINE_LAUNCH(2)
#elif IBL_VARIANT == 3
IBL_DEFINE_LAUNCH(3)
#elif IBL_VARIANT == 6
IBL_DEFINE_LAUNCH(6)
#elif IBL_VARIANT == 7
IBL_DEFINE_LAUNCH(7)
#elif IBL_VARIANT == 8
IBL_DEFINE_LAUNCH(8)
#elif IBL_VARIANT == 9
IBL_DEFINE_LAUNCH(9)
#elif IBL_VARIANT == 10
IBL_DEFINE_LAUNCH(10)
#elif IBL_VARIANT == 11
IBL_DEFINE_LAUNCH(11)
#elif IBL_VARIANT == 12
IBL_DEFINE_LAUNCH(12)
#elif IBL_VARIANT == 16
IBL_DEFINE_LAUNCH(16)
#elif IBL_VARIANT == 18
IBL_DEFINE_LAUNCH(18)
#else
IBL_DEFINE_LAUNCH(4)
#endif
#else
IBL_DEFINE_LAUNCH(0) IBL_DEFINE_LAUNCH(1) IBL_DEFINE_LAUNCH(2) IBL_DEFINE_LAUNCH(3) IBL_DEFINE_LAUNCH(4) IBL_DEFINE_LAUNCH(6) IBL_DEFINE_LAUNCH(7) IBL_DEFINE_LAUNCH(8) IBL_DEFINE_LAUNCH(9) IBL_DEFINE_LAUNCH(10) IBL_DEFINE_LAUNCH(11) IBL_DEFINE_LAUNCH(12)
#endif
#undef IBL_DEFINE_LAUNCH

#if !defined(IBL_VARIANT) || IBL_VARIANT == 0
hipError_t IBL_LAUNCH_NAME(0)(const MlpArgs&, int, hipStream_t);
hipError_t IBL_LAUNCH_NAME(1)(const MlpArgs&, int, hipStream_t);
hipError_t IBL_LAUNCH_NAME(2)(const MlpArgs&, int, hipStream_t);
hipError_t IBL_LAUNCH_NAME(3)(const MlpArgs&, int, hipStream_t);
hipError_t IBL_LAUNCH_NAME(4)(const MlpArgs&, int, hipStream_t);
hipError_t IBL_LAUNCH_NAME(6)(const MlpArgs&, int, hipStream_t);
#ifdef IBL_F16X3
hipError_t IBL_LAUNCH_NAME(7)(const MlpArgs&, int, hipStream_t);   // (the stash is f16: this flavour only)
hipError_t IBL_LAUNCH_NAME(8)(const MlpArgs&, int, hipStream_t);
hipError_t IBL_LAUNCH_NAME(9)(const MlpArgs&, int, hipStream_t);
hipError_t IBL_LAUNCH_NAME(10)(const MlpArgs&, int, hipStream_t);
hipError_t IBL_LAUNCH_NAME(11)(const MlpArgs&, int, hipStream_t);
hipError_t IBL_LAUNCH_NAME(12)(const MlpArgs&, int, hipStream_t);
hipError_t IBL_LAUNCH_NAME(16)(const MlpArgs&, int, hipStream_t);   // (the list variants: built for this flavour only)
hipError_t IBL_LAUNCH_NAME(18)(const MlpArgs&, int, hipStream_t);
#endif
hipError_t IBL_DISPATCH(int variant, const MlpArgs& a, int n_cu, hipStream_t stream) {
    if (a.n_pts <= 0) return hipSuccess;
    const long n_groups = (a.n_pts + 127) / 128;
    const int grid = (int)(n_groups < n_cu ? n_groups : n_cu);
    hipError_t rc;
    switch (variant) {
        case VAR_FULL: rc = IBL_LAUNCH_NAME(0)(a, grid, stream); break;
        case VAR_TRUNK: rc = IBL_LAUNCH_NAME(1)(a, grid, stream); break;
        case VAR_REFL: rc = IBL_LAUNCH_NAME(2)(a, grid, stream); break;
        case VAR_FULL_CI: rc = IBL_LAUNCH_NAME(3)(a, grid, stream); break;
        case VAR_REFL_CI: rc = IBL_LAUNCH_NAME(4)(a, grid, stream); break;
        case VAR_TRUNK_GRAD: rc = IBL_LAUNCH_NAME(6)(a, grid, stream); break;
#ifdef IBL_F16X3
        case VAR_TRUNK_BWD: rc = IBL_LAUNCH_NAME(7)(a, grid, stream); break;
        case VAR_TRUNK_FEAT: rc = IBL_LAUNCH_NAME(8)(a, grid, stream); break;
        case VAR_TRUNK_BWD_FEAT: rc = IBL_LAUNCH_NAME(9)(a, grid, stream); break;
        case VAR_TRUNK_FEAT2: rc = IBL_LAUNCH_NAME(10)(a, grid, stream); break;
        case VAR_TRUNK_BWD_FEAT2: rc = IBL_LAUNCH_NAME(11)(a, grid, stream); break;
        case VAR_NET_BWD: rc = IBL_LAUNCH_NAME(12)(a, grid, stream); break;
        case VAR_FULL_LIST: rc = IBL_LAUNCH_NAME(16)(a, grid, stream); break;
        case VAR_TRUNK_LIST: rc = IBL_LAUNCH_NAME(18)(a, grid, stream); break;
#endif
        default: return hipErrorInvalidValue;
    }
    if (rc != hipSuccess) return rc;
#if defined(IBL_TRACE) && !defined(IBL_F16X3)
    if (variant == VAR_TRUNK) {
        (void)hipDeviceSynchronize();
        long long h[32];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trace), sizeof h);
        long long chk = 0, chk2 = 0; hipError_t e2 = hipMemcpyFromSymbol(&chk, HIP_SYMBOL(g_trace), 8, 40 * 8);
        (void)hipMemcpyFromSymbol(&chk2, HIP_SYMBOL(g_trace), 8, 41 * 8);
        printf("[trace] s41=%lld sentinel=%lld err=%s  tile: start->mid  mid->end  end->barrier+prefetch  | tile period\n", chk2, chk, hipGetErrorString(e2));
        for (int t = 0; t < 8; ++t)
            printf("[trace] t%d: %6lld %6lld %6lld | %6lld\n", t, h[4 * t + 1] - h[4 * t], h[4 * t + 2] - h[4 * t + 1],
                   t < 7 ? h[4 * t + 3] - h[4 * t + 2] : 0LL, t < 7 ? h[4 * t + 4] - h[4 * t] : 0LL);
    }
#endif
    return hipGetLastError();
}
#endif

}  // namespace ibl
