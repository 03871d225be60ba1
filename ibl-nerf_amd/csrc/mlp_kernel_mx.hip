// Fused IBLNeRF forward, f16 + MX-fp6 product scheme (layout_mx.h) — the faster of the two variants of
// the fused MLP kernel; mlp_kernel.hip (three bf16 products) is the wide-range variant it falls back to.
// Replaces `run_network` + `IBLNeRF.forward_not_freezed` (src/nerf_models/ibl_nerf.py:236-252, :154-210;
// encoder src/nerf_models/positional_embedder.py:4-52).
//
// Same skeleton as mlp_kernel.hip: one workgroup = 4 wavefronts (one per SIMD), 32 sample points per
// wavefront whose 256-wide activation never leaves the register file, weights streamed once per 128
// points through a 3-slot LDS ring by LDS-DMA, N=1/N=3 heads on the VALU from the fp32 accumulators.
// Per K=64 block of a 32-row tile: 4 f16 MFMAs (main term) + 2 block-scaled fp6 MFMAs (the two
// residual terms) into one fp32 accumulator = 6 "slots" of ~32 cycles.
#include <hip/hip_runtime.h>

#include <cstddef>

#include <type_traits>

#include "mlp_args.h"
#include "layout_mx.h"
#include "sincos_enc.h"

namespace ibl {
// -DIBL_MX_F16ONLY builds the plain-f16 flavour of this kernel (namespace mxk16, launch_mlp_mx16): per block only the four f16
// MFMAs, no residual forms in the epilogue, only the f16 half of the weight stream.  2^-11 per operand: enough for every query
// that neither places samples nor feeds the finite-difference normal (the fine pass's main query, the reflected-ray queries;
// scratch/prec_probe_f16f8.py, mode "f16+fp6|f16_only|keepcoarse"), not for the others.
//
// VAR_TRUNK_X (this kernel only): the trunk-only form with positions_linears.0 and .1 evaluated as THREE f16 products on hi/lo splits
// (Wh Xh + Wh Xl + Wl Xh, 2^-22 per operand) inside the same pipeline; the other six layers as above.  On a network with surfaces the
// density's error is set by the first layers (an error injected early is amplified by every later layer; scratch/prec_probe_layers.py,
// prec_probe_mixed.py): with these two at 2^-22 the finite-difference normal of the offset queries is that of the f16x3 kernel
// (1.9e-4 against 2.6e-3 on the worst of 1 024 rays, emulated) for +17 % matrix instructions instead of +100 %.  A logical block of
// those layers is a PAIR of stream blocks: the network's block (its f16 area = Wh) and a residual block (its f16 area = f16(W - Wh),
// layout_mx.h CH_RES); a chunk of four blocks is assembled by the four waves' LDS-DMA from the two places.  Per pair: slots 0-3 of
// the first block issue Wh Xh and Wh Xl (one operand read, two MFMAs), slots 0-3 of the second Wl Xh; their fp6 slots are idle.
// -DIBL_MX_F16ONLY -DIBL_MX_VARIANT=1 (the plain-f16 TRUNK form = the density-ESTIMATE kernel of api.cpp's Q_ESTIMATE, most of a frame's matrix time since round 4)
// is built as its own flavour, IBL_MX_EST: two workgroups per CU (<= 256 registers, an LDS ring of the blocks' f16 halves only: 3 x 16 KB + tables = 72 KB), a plain
// loop per layer (operands read two slots ahead, each tile's epilogue right behind its MFMAs) instead of the hand-scheduled slot program — the second wave of a SIMD
// hides what the program's software pipeline hides in the one-wave kernels, and the body no longer needs 500 registers.
#if defined(IBL_MX_F16ONLY) && defined(IBL_MX_VARIANT) && IBL_MX_VARIANT == 1 && !defined(IBL_MX_NO_EST)
#define IBL_MX_EST
#ifndef IBL_MX_EST_TILES
#define IBL_MX_EST_TILES 1      // 32-point tiles per wave: 1 = two workgroups per CU (11.8 ms per estimate launch of the bench frame); 2 = one workgroup per CU whose
                                // weight fragments each feed two MFMAs (A/B build: 14.3 ms — 143 spilled registers, 447 accumulator-file moves per evaluation)
#endif
#endif
#ifdef IBL_MX_F16ONLY
#define IBL_MXK mxk16
constexpr bool F16O = true;
#else
#define IBL_MXK mxk
constexpr bool F16O = false;
#endif
namespace IBL_MXK {

using namespace ibl::mx;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// B operands of one K=64 block of this wave's 32 points: 4 f16 k-steps, fp6 of the f16 values, fp6 of
// the f16 residuals, and the two e8m0 block scales (byte 0: x6, byte 1: l6).  29 registers.
struct Blk {
    u32x16 hv;        // the 4 f16 k-steps (4 dwords each) as ONE vector: the fp6 conversion reads all 16 registers, so they
                      // are kept consecutive from the start; k-step S is quarter<S>(hv) (sub-vectors are cut in SSA)
    u32x4 x6a, l6a;   // fp6 bits 0..127   (6-wide vectors are kept out of the structs: they defeat SROA)
    u32x2 x6b, l6b;   // fp6 bits 128..191
    unsigned sc;
};
struct Act { Blk b[4]; };   // 256 features

#ifdef IBL_MX_EST
constexpr int PF = 2;
#elif defined(IBL_MX_F16ONLY)
constexpr int PF = 8;    // (plain f16: no fp6 operand forms to hold, so eight entries = two whole blocks ahead are affordable; LDS latency under four waves' reads is not
                         // covered by four MFMAs)
#else
constexpr int PF = 4;                                          // A operands are read from LDS this many slots ahead (PF rotating entries:
                                                               // the read for slot G+PF is issued right after slot G's MFMA has consumed its entry)
#endif
// A operands read ahead from the LDS ring, 4 rotating entries: q for the four f16 slots of a block; q+d (+sc at
// slot 4) for its two fp6 slots
struct Pre {
    u32x4 q[PF];
    u32x2 d[PF];
    unsigned sc[PF];
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#ifdef IBL_MX_F16ONLY
constexpr int SLOTS_PER_BLOCK = 4;    // plain f16: no residual slots (with two empty ones the operand read-ahead of PF slots shrank to one or two MFMAs at every block end)
#else
constexpr int SLOTS_PER_BLOCK = 6;
#endif
constexpr int CHUNK_SLOTS = CHUNK_BLOCKS * SLOTS_PER_BLOCK;   // 24 (16)
constexpr int SYNC_SLOT = CHUNK_SLOTS - PF - 1;                // after this slot the next chunk must be readable
// One DMA piece of the chunk two ahead is issued in each of 8 chunk-relative slots.  A piece costs its wave
// 60+ cycles of issue, more when it meets LDS reads, so the slots chosen are those whose operand prefetch is a
// single ds_read (block slots 2..5; slots 0 and 1 fetch the three- and two-read fp6 operands), in the first
// half of the chunk so the data has a full chunk and a half to land before sync_next needs it.
__host__ __device__ constexpr int dma_piece(int cr) {
    const int b = cr / SLOTS_PER_BLOCK, s = cr % SLOTS_PER_BLOCK;
#ifdef IBL_MX_F16ONLY
    return (b < 2 && s >= 2) ? 2 * b + (s - 2) : -1;     // (the four pieces of the f16 half)
#else
    return (b < 2 && s >= 2) ? 4 * b + (s - 2) : -1;
#endif
}
// Tile-local slot after whose MFMA slice i of the previous tile's epilogue runs.  When the previous tile is
// the last one of the PREVIOUS layer, its slices still write block 3 of this layer's input (f16 k-step 2 at
// slice 3, k-step 3 + the fp6 forms at slice 7); that block's slots are the tile's last six, so slice 7 must
// be done before the fourth-last slot issues.
__host__ __device__ constexpr int slice_slot(int i, int ns) { return ns >= CHUNK_SLOTS ? 1 + (i * (ns - 5)) / 7 : 1 + (i * (ns - 1)) / 8; }
// A slice is one dependency chain of ~11 VALU instructions; beside a single MFMA only about six are free and each
// further one costs its full latency (a slot-level s_memtime trace showed slice slots 50-100 cycles longer than the
// others).  So every slice runs as two stages in two consecutive slots: A = read the accumulators, block max, ReLU;
// B = f16 pair, residuals, store, head dot products.  Stage q = 2*slice + stage of the previous tile runs after the
// MFMA of slot stage_slot(q): 16 stages over the tile's first slots, the last one (slice 7, B) before slot ns - 5.
constexpr int N_STAGES = 2;   // (three stages — reads+max / ReLU / rest — measured 3 % slower than two)
__host__ __device__ constexpr int stage_slot(int q, int ns) { return ns >= CHUNK_SLOTS ? 1 + (q * (ns - 6)) / (8 * N_STAGES) : 1 + (q * (ns - 1)) / (8 * N_STAGES); }
// ... in a three-product layer (run_layer_x3) the last logical block of the input is the tile's last TWELVE slots (a pair of stream
// blocks), so the pending epilogue that completes it must be through before slot ns - 12
__host__ __device__ constexpr int stage_slot_x3(int q, int ns) { return ns >= 2 * CHUNK_SLOTS ? 1 + (q * (ns - 14)) / (8 * N_STAGES) : 1 + (q * (ns - 1)) / (8 * N_STAGES); }

// ---------------------------------------------------------------------------------------------
// weight-stream pipeline (cf. Pipe in mlp_kernel.hip): cyclic, never drained.  While chunk c is
// consumed, chunk c+1 has landed or is landing and chunk c+2 is being issued.  Because operands are
// read PF slots ahead, the "next chunk has landed" wait + barrier sits PF+1 slots BEFORE the chunk
// boundary (sync_next), and the ring indices rotate at the boundary itself (advance).
// ---------------------------------------------------------------------------------------------
#ifdef IBL_MX_EST
constexpr int R_CHUNK = CHUNK_BYTES / 2, R_BLOCK = BLOCK_BYTES / 2;     // the ring holds the f16 half of each block only
#else
constexpr int R_CHUNK = CHUNK_BYTES, R_BLOCK = BLOCK_BYTES;
#endif
constexpr int R_RING_BYTES = RING_SLOTS * R_CHUNK;
constexpr int R_LDS_BYTES = R_RING_BYTES + TAB_BYTES;

template <int VARIANT>
struct Pipe {
    static constexpr bool CI = variant_ci(VARIANT), ALBIRR = variant_albirr(VARIANT);
    static constexpr bool PP = VARIANT == VAR_TRUNK_P;                  // every trunk block as a (network, residual) pair
    static constexpr bool X = variant_trunk_x(VARIANT) || PP;             // a wave's block comes from its own place in the stream
    static constexpr int N_PROG = PP ? mx::N_CHUNKS_TRUNK_P : X ? mx::N_CHUNKS_TRUNK_X : VARIANT == VAR_TRUNK ? mx::N_CHUNKS_TRUNK
                                                        : mx::N_CHUNKS_TRUNK + (CI ? 0 : 8 + 10) + (ALBIRR ? 8 : 0) + 12;
    const char* stream;
    char* ring;
    unsigned lds_ring;
    unsigned voff;
    int lane, wave;
    int slot, slot1, slot2;   // ring slots of the chunk being consumed, the next one, the one two ahead
    int prog2;                // program position of the chunk two ahead

    // VAR_TRUNK_X: stream block that wave w copies for program position p (layers 0 and 1: network block / residual block pairs)
    __device__ __forceinline__ static int x_block(int p, int w) {
        if constexpr (PP) {                            // chunk p = logical blocks 2p, 2p+1 of the trunk, each as [network block, residual block]
            const int lb = 2 * p + (w >> 1);
            return (w & 1) ? mx::CH_RES * 4 + lb : lb;
        }
        if (p < 4) {                                   // L0: chunk p = tiles 2p, 2p+1, each as [network, residual]
            const int tile = 2 * p + (w >> 1);
            return (w & 1) ? mx::CH_RES * 4 + tile : mx::CH_L0 * 4 + tile;
        }
        if (p < 20) {                                  // L1: chunk c = tile c/2, logical blocks 2(c&1), 2(c&1)+1
            const int c = p - 4, lb = 4 * (c >> 1) + 2 * (c & 1) + (w >> 1);
            return (w & 1) ? mx::CH_RES * 4 + 8 + lb : mx::CH_L1 * 4 + lb;
        }
        return (mx::CH_L1 + 8 + (p - 20)) * 4 + w;     // positions_linears.2 .. 7 as they are
    }
    __device__ __forceinline__ static int stream_chunk(int p) {
        // program position -> stream chunk: a variant's program is the trunk followed by the head layers it evaluates
        if (p < mx::N_CHUNKS_TRUNK) return p;
        int q = p - mx::N_CHUNKS_TRUNK;
        if (!CI) { if (q < 8) return mx::CH_FEAT + q; q -= 8; }
        if (ALBIRR) { if (q < 8) return mx::CH_ALB + q; q -= 8; }
        if (!CI) { if (q < 10) return mx::CH_VIEW + q; q -= 10; }
        return mx::CH_AR + q;
    }
    // Piece i of a chunk (wave w copies bytes [8192 w, 8192 w + 8192) in 8 pieces of 1 KiB).  The instruction's
    // immediate offset moves BOTH the global and the LDS address, so pieces 0..3 (and 4..7) share one M0 value
    // and one address register; M0 is rewritten only at pieces 0 and 4.
    template <int I>
    __device__ __forceinline__ void issue_piece(int prog, int slt) const {
#ifdef IBL_MX_ABLATE_NO_LOADS   // timing ablation only (results are garbage): no weight traffic
        return;
#endif
#ifdef IBL_MX_ABLATE_HALF_LOADS  // timing ablation only (results are garbage): every other piece of the weight stream
        if constexpr (I % 2 == 1) return;
#endif
        if constexpr (F16O && I >= 4) return;   // pieces 4..7 of a wave are the fp6 half of its block
        // (VAR_TRUNK_X: the wave's block comes from its own place in the stream, so the wave term moves from the offset register to the base)
        const char* src = X ? stream + (size_t)x_block(prog, wave) * BLOCK_BYTES : stream + (size_t)stream_chunk(prog) * CHUNK_BYTES;
        const unsigned dst = lds_ring + (unsigned)slt * R_CHUNK + wave * R_BLOCK + (I / 4) * 4096;
        const unsigned v = (X ? voff - wave * 8192 : voff) + (I / 4) * 4096;
        if constexpr (I % 4 != 0) {
            asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(v), "s"(src), "n"((I % 4) * 1024) : "memory");
#ifdef IBL_MX_DOUBLE_DMA   // measurement only: every piece twice (same bytes to the same place) prices one LDS-DMA instruction
            asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(v), "s"(src), "n"((I % 4) * 1024) : "memory");
#endif
            return;
        }
        // M0 (the DMA's LDS base) is left holding `dst`: nothing else in this kernel reads M0 (gfx9+ DS and
        // lane instructions do not), which scratch/mxdev.sh checks in the generated ISA; saving and restoring
        // it cost two more SALU issues per piece in a loop that is issue-bound.
        asm volatile(
            "s_mov_b32 m0, %1\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %2"
#ifdef IBL_MX_DOUBLE_DMA
            "\n\tglobal_load_lds_dwordx4 %0, %2"
#endif
            :
            : "v"(v), "s"(dst), "s"(src)
            : "memory");
    }
    __device__ __forceinline__ void start() {
        static_for<0, 8>([&](auto I) { issue_piece<decltype(I)::value>(0, 0); });
        static_for<0, 8>([&](auto I) { issue_piece<decltype(I)::value>(1, 1); });
        slot = 0;
        slot1 = 1;
        slot2 = 2;
        prog2 = 2;
#ifdef IBL_MX_DOUBLE_DMA
        asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#elif defined(IBL_MX_ABLATE_HALF_LOADS) || defined(IBL_MX_F16ONLY)
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
#endif
    }
    // byte address of block `blk` of the current / next chunk, this lane's 16-byte column
    __device__ __forceinline__ const char* block(bool next, int blk) const {
        return ring + (next ? slot1 : slot) * R_CHUNK + blk * R_BLOCK;
    }
    template <int I>
    __device__ __forceinline__ void prefetch_piece() const { issue_piece<I>(prog2, slot2); }
    // The next chunk's 8 pieces have landed (mine: vmcnt; everyone's: barrier).  All reads of the CURRENT chunk
    // have been issued by this point (they run PF slots ahead) and lgkmcnt(0) completes them, so the barrier is
    // also the WAR fence for the DMA that overwrites this chunk's ring slot two chunks from now.  (Dropping the
    // lgkmcnt(0) measured no faster: 12.78 vs 12.77 ms on the TRUNK benchmark.)
#ifdef IBL_MX_DOUBLE_DMA
    __device__ __forceinline__ void sync_next() const { asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#elif defined(IBL_MX_ABLATE_HALF_LOADS) || defined(IBL_MX_F16ONLY)
    __device__ __forceinline__ void sync_next() const { asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#elif defined(IBL_MX_ABLATE_NO_BARRIER)   // timing ablation only (racy)
    __device__ __forceinline__ void sync_next() const { asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); }
#else
    __device__ __forceinline__ void sync_next() const { asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
    __device__ __forceinline__ void advance() {
        slot = slot1;
        slot1 = slot2;
        slot2 = slot2 == RING_SLOTS - 1 ? 0 : slot2 + 1;
        prog2 = prog2 == N_PROG - 1 ? 0 : prog2 + 1;
    }
    __device__ __forceinline__ void drain() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

template <int KIND, int E>
__device__ __forceinline__ void load_frag(Pre& pf, const char* blk, int lane) {
    if constexpr (KIND < 4) {
        pf.q[E] = *reinterpret_cast<const u32x4*>(blk + OFF_F16 + KIND * 1024 + lane * 16);
    } else if constexpr (F16O) {
    } else if constexpr (KIND == 4) {
        pf.q[E] = *reinterpret_cast<const u32x4*>(blk + OFF_W6A + lane * 16);
        pf.d[E] = *reinterpret_cast<const u32x2*>(blk + OFF_W6B + lane * 8);
        pf.sc[E] = *reinterpret_cast<const unsigned*>(blk + OFF_SC + lane * 4);
    } else {
        pf.q[E] = *reinterpret_cast<const u32x4*>(blk + OFF_R6A + lane * 16);
        pf.d[E] = *reinterpret_cast<const u32x2*>(blk + OFF_R6B + lane * 8);
    }
}

// 192 bits of fp6 in the low six dwords of the MFMA's 8-dword operand (the upper two are not read for fp6)
// The two pieces go through an empty asm first: otherwise the optimiser turns "load 4 dwords, widen to 8 with
// undefined tail" into ONE 8-dword load that runs over the neighbouring struct members, and objects accessed
// with such overlapping loads are no longer promoted to registers (everything lands in scratch memory).
template <bool PIN>
__device__ __forceinline__ i32x8 fp6_operand(const u32x4& q, const u32x2& d) {
    u32x4 qq = q;
    u32x2 dd = d;
    if constexpr (PIN) asm("" : "+v"(qq), "+v"(dd));   // needed for every operand that lives in a struct before SROA (Blk AND Pre)
    i32x8 v;
    v[0] = (int)qq[0];
    v[1] = (int)qq[1];
    v[2] = (int)qq[2];
    v[3] = (int)qq[3];
    v[4] = (int)dd[0];
    v[5] = (int)dd[1];
    return v;
}

template <int S>
__device__ __forceinline__ u32x4 quarter(const u32x16& v) { return __builtin_shufflevector(v, v, 4 * S, 4 * S + 1, 4 * S + 2, 4 * S + 3); }
template <int S>
__device__ __forceinline__ u32x16 with_quarter(const u32x16& v, const u32x4& q) {
    const u32x16 w = __builtin_shufflevector(q, q, 0, 1, 2, 3, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    if constexpr (S == 0) return __builtin_shufflevector(v, w, 16, 17, 18, 19, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    else if constexpr (S == 1) return __builtin_shufflevector(v, w, 0, 1, 2, 3, 16, 17, 18, 19, 8, 9, 10, 11, 12, 13, 14, 15);
    else if constexpr (S == 2) return __builtin_shufflevector(v, w, 0, 1, 2, 3, 4, 5, 6, 7, 16, 17, 18, 19, 12, 13, 14, 15);
    else return __builtin_shufflevector(v, w, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19);
}

// slot s of a block: accumulate one of the six products
template <int S>
__device__ __forceinline__ f32x16 slot_mfma(const u32x4& aq, const u32x2& ad, unsigned wsc, const Blk& b, f32x16 c) {
    if constexpr (S < 4) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aq), __builtin_bit_cast(f16x8, quarter<S>(b.hv)), c, 0, 0, 0);
    } else if constexpr (F16O) {     // plain-f16 flavour: the two residual slots are empty
        return c;
    } else if constexpr (S == 4) {   // fp6(W) [scale byte 0] x fp6(X - f16 X) [scale byte 1]
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp6_operand<true>(aq, ad), fp6_operand<true>(b.l6a, b.l6b), c, 2, 2, 0, (int)wsc, 1, (int)b.sc);
    } else {                         // fp6(W - f16 W) [scale byte 1] x fp6(f16 X) [scale byte 0]
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp6_operand<true>(aq, ad), fp6_operand<true>(b.x6a, b.x6b), c, 2, 2, 1, (int)wsc, 0, (int)b.sc);
    }
}

__device__ __forceinline__ float relu_bits(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }

// (x0, x1) -> packed f16 pair h = rne(x) and packed f16 pair of the residuals x - h
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hb, unsigned& lb) {
    const f32x2 xv = {x0, x1};
    hb = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, f16x2));
    if constexpr (F16O) {
        lb = 0;
        return;
    }
    // residuals x - (float)h, rounded to f16 and written straight into the low / high half of one register:
    // v_fma_mix{lo,hi}_f16 read the f16 half of h directly (plain C++ costs cvt + sub + cvt_pk per pair)
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(x0), "v"(hb));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(x1), "v"(hb));
}

// Completes a block once its 32 f16 values (b.h) and their residuals (lres, same element order) are in
// registers: block scale from the running max, fp6 forms of both, scale bytes.  `peak` keeps the
// largest magnitude seen (for the f16 range check at the end of the kernel).
__device__ __forceinline__ void finish_block(Blk& b, const u32x16& lres, int& mxv, unsigned& peak) {
    unsigned mb = (unsigned)mxv;
    peak = mb > peak ? mb : peak;
    if constexpr (F16O) {       // plain-f16 flavour: only the range guard needs the block max
        mxv = 0;
        return;
    }
    mb = mb > 0x0d800000u ? mb : 0x0d800000u;          // >= 2^-100: an all-zero block gets a harmless tiny scale
    const unsigned e = mb >> 23;                        // biased exponent of the block max
    const float sh = __builtin_bit_cast(float, (e - 2) << 23);    // max / sh in [4, 8)
    const float sl = __builtin_bit_cast(float, (e - 14) << 23);   // |residual| <= 2^(e-11) -> / sl <= 8
    const u32x16 hv = b.hv;
    const auto x6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, hv), sh);
    const auto l6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, lres), sl);
    b.x6a = u32x4{(unsigned)x6[0], (unsigned)x6[1], (unsigned)x6[2], (unsigned)x6[3]};
    b.x6b = u32x2{(unsigned)x6[4], (unsigned)x6[5]};
    b.l6a = u32x4{(unsigned)l6[0], (unsigned)l6[1], (unsigned)l6[2], (unsigned)l6[3]};
    b.l6b = u32x2{(unsigned)l6[4], (unsigned)l6[5]};
    b.sc = (e - 2) | ((e - 14) << 8);
    mxv = 0;
}

// Epilogue of one finished 32-feature tile in 8 slices of two accumulator registers (cf. Epi in
// mlp_kernel.hip).  STORE: v = [ReLU](acc) -> f16 k-steps 2(T&1), 2(T&1)+1 of block T>>1 of `dst`, residuals
// staged in `lres`; after the odd tile of a pair the block is finished (scales + fp6 forms).
// NCH: running fp32 dot products with N=1/3 head rows.
template <bool STORE, bool RELU, int NCH, bool KEEP_LO = false>
struct Epi {
    Act* dst;
    f32x2* part[NCH > 0 ? NCH : 1];   // per head channel: two interleaved partial sums (even / odd element of each pair), two v_fma_f32 per pair
    const float* tab[NCH > 0 ? NCH : 1];
    unsigned* peak;
    u32x16* lo_dst = nullptr;   // KEEP_LO (VAR_TRUNK_X, layer 0): where the f16 residuals of each finished block go (input of a three-product
                                // layer); set once at construction — a pointer that changes at run time would keep the array out of registers
    u32x16 lres;
    u32x4 hq;
    int mxv;   // running block max as the int image of a non-negative float (ordering is the same; one v_max3_i32 per pair)

    float sx0, sx1;   // the slice in flight between its two stages
    f32x2 hw[NCH > 0 ? NCH : 1];   // ... and its head weights: read from LDS in stage A so that stage B never waits for them

    template <int T, int I>
    __device__ __forceinline__ void stage_a(const f32x16& acc) {
        float x0 = acc[2 * I], x1 = acc[2 * I + 1];
        if constexpr (STORE) {
            // block max on the raw accumulator bits: with ReLU a negative value (negative int) never wins and the
            // max of the survivors is the max after ReLU; without it the sign bit is cleared first
            const int b0 = __builtin_bit_cast(int, x0), b1 = __builtin_bit_cast(int, x1);
            if constexpr (RELU) mxv = max(max(mxv, b0), b1);   // one v_max3_i32
            else mxv = max(mxv, max(b0 & 0x7fffffff, b1 & 0x7fffffff));
        }
        if constexpr (RELU) {
            x0 = relu_bits(x0);
            x1 = relu_bits(x1);
        }
        sx0 = x0;
        sx1 = x1;
        pin(sx0);
        pin(sx1);
#pragma unroll
        for (int c = 0; c < NCH; ++c) hw[c] = *reinterpret_cast<const f32x2*>(tab[c] + T * 32 + 2 * I);
    }
    template <int T, int I>
    __device__ __forceinline__ void stage_b() {
        const float x0 = sx0, x1 = sx1;
        if constexpr (STORE) {
            constexpr int j = 2 * (T & 1) + (I >> 2);   // f16 k-step of the block
            unsigned hb, lb;
            split_pair(x0, x1, hb, lb);
            hq[I & 3] = hb;
            lres[4 * j + (I & 3)] = lb;
            if constexpr ((I & 3) == 3) {
                asm volatile("" : "+v"(hq));
                dst->b[T >> 1].hv = with_quarter<j>(dst->b[T >> 1].hv, hq);
            }
            if constexpr (I == 7 && (T & 1) == 1) {
                if constexpr (KEEP_LO) lo_dst[T >> 1] = lres;
                finish_block(dst->b[T >> 1], lres, mxv, *peak);
            }
        }
#pragma unroll
#ifdef IBL_MX_ABLATE_NO_HEADS   // timing ablation only (results are garbage): no head dot products
        for (int c = 0; c < 0; ++c) {
#else
        for (int c = 0; c < NCH; ++c) {
#endif
#ifdef IBL_MX_HEADS_PK_FMA   // A/B build (scratch/mx_heads_ab.sh): one v_pk_fma_f32 per pair — round 1's form, 0.8 % slower (FULL, 65 536 x 128 points:
                             // 21.79 against 21.60 ms, same box, head-weight prefetch kept in both): packed f32 VALU beside MFMAs is an anti-lever
            *part[c] = __builtin_elementwise_fma(f32x2{x0, x1}, hw[c], *part[c]);
#else
            (*part[c])[0] = fmaf(x0, hw[c][0], (*part[c])[0]);
            (*part[c])[1] = fmaf(x1, hw[c][1], (*part[c])[1]);
#endif
            if constexpr (I == 7) asm volatile("" : "+v"(*part[c]));
        }
    }
    template <int T, int I, int K>
    __device__ __forceinline__ void stage(const f32x16& acc) {
        if constexpr (K == 0) stage_a<T, I>(acc);
        else stage_b<T, I>();
    }
    template <int T, int I>
    __device__ __forceinline__ void slice(const f32x16& acc) {
        stage_a<T, I>(acc);
        stage_b<T, I>();
    }
};

// One layer: NT output tiles; per tile an optional encoding block then NH (4 or 0) blocks over the
// 256-feature activation `in`; 6 slots per block.  Software pipeline (one wave per SIMD):
//   * the A operand of slot G+PF is read from LDS at slot G (pf[] is carried across tiles and layers);
//   * the previous tile's epilogue runs in 8 slices spread over this tile's slots;
//   * one LDS-DMA piece of the chunk two ahead in each of the 8 slots dma_piece() names;
//   * sched_barrier(0) fences keep that order at slot granularity.
template <int NT, bool HAS_ENC, int NH, int VARIANT, class PEND, class EPI>
__device__ __forceinline__ f32x16 run_layer(Pipe<VARIANT>& P, Pre& pf, unsigned& wsc, const Act& in, const Blk& enc,
                                            const float* bias_tab, PEND&& pend, EPI& epi) {
    constexpr int NB = (HAS_ENC ? 1 : 0) + NH;   // blocks per tile
    constexpr int NS = NB * SLOTS_PER_BLOCK;     // slots per tile
    static_assert((NT * NS) % CHUNK_SLOTS == 0, "a layer is a whole number of chunks");
    f32x16 prev = {0};
    f32x16 bias_next = *reinterpret_cast<const f32x16*>(bias_tab);
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc = bias_next;   // the chain starts at the layer bias (read from LDS during the previous tile)
        static_for<0, NS>([&](auto GS) {
            constexpr int g = decltype(GS)::value;   // slot inside the tile
            constexpr int G = t * NS + g;            // slot inside the layer
            constexpr int cr = G % CHUNK_SLOTS;      // slot inside the chunk
            constexpr int bb = g / SLOTS_PER_BLOCK, s = g % SLOTS_PER_BLOCK;
            if constexpr (s == 4 && !F16O) wsc = pf.sc[G % PF];
            if constexpr (HAS_ENC && bb == 0) acc = slot_mfma<s>(pf.q[G % PF], pf.d[G % PF], wsc, enc, acc);
            else acc = slot_mfma<s>(pf.q[G % PF], pf.d[G % PF], wsc, in.b[bb - (HAS_ENC ? 1 : 0)], acc);
#ifdef IBL_MX_DOUBLE_MFMA   // measurement only (results are garbage): every matrix instruction twice
            if constexpr (HAS_ENC && bb == 0) acc = slot_mfma<s>(pf.q[G % PF], pf.d[G % PF], wsc, enc, acc);
            else acc = slot_mfma<s>(pf.q[G % PF], pf.d[G % PF], wsc, in.b[bb - (HAS_ENC ? 1 : 0)], acc);
#endif
            // A operand of slot G + PF, into the entry this slot's MFMA has just consumed
            {
                constexpr int Gp = G + PF;
                constexpr bool next = (Gp / CHUNK_SLOTS) != (G / CHUNK_SLOTS);
                constexpr int blk = (Gp / SLOTS_PER_BLOCK) % CHUNK_BLOCKS;
                load_frag<Gp % SLOTS_PER_BLOCK, Gp % PF>(pf, P.block(next, blk), P.lane);
#ifdef IBL_MX_DOUBLE_LDS    // measurement only: every operand read twice
                asm volatile("" : "+v"(pf.q[Gp % PF]));
                load_frag<Gp % SLOTS_PER_BLOCK, Gp % PF>(pf, P.block(next, blk), P.lane);
#endif
            }
            if constexpr (g == NS / 2 && t + 1 < NT) bias_next = *reinterpret_cast<const f32x16*>(bias_tab + (t + 1) * 32);
            if constexpr (dma_piece(cr) >= 0) P.template prefetch_piece<(dma_piece(cr) >= 0 ? dma_piece(cr) : 0)>();
#ifndef IBL_MX_ABLATE_NO_EPI     // timing ablation only: no epilogue work
            static_for<0, 8 * N_STAGES>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                if constexpr (g == stage_slot(q, NS)) {
                    if constexpr (t == 0) pend(std::integral_constant<int, q / N_STAGES>{}, std::integral_constant<int, q % N_STAGES>{});
                    else epi.template stage<(t > 0 ? t - 1 : 0), q / N_STAGES, q % N_STAGES>(prev);
                }
            });
#endif
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (cr == SYNC_SLOT) P.sync_next();
            if constexpr (cr == CHUNK_SLOTS - 1) P.advance();
        });
        // Keep the finished chain where it is; its consumer is the deferred epilogue.  This file is built with
        // -mllvm -amdgpu-mfma-vgpr-form (build.py): the accumulator chains live in arch VGPRs, so the epilogue reads them
        // without v_accvgpr_read, and the register allocator parks finished fp6 operand forms in AGPRs (one
        // v_accvgpr_write each; the MFMAs read them from there directly) - 575 accumulator-file moves per TRUNK evaluation
        // instead of 896 with the chains pinned in AGPRs ("+a", -DIBL_MX_AGPR_CHAIN: the earlier form, 2 % slower).
#ifdef IBL_MX_AGPR_CHAIN
        asm volatile("" : "+a"(acc));
#else
        asm volatile("" : "+v"(acc));
#endif
        prev = acc;
    });
    return prev;
}


#ifdef IBL_MX_EST
// The estimate kernel keeps an activation as sixteen 4-register k-steps (the MFMA's B operand), not as four 16-register blocks: nothing here reads a whole block
// (no fp6 conversion), and 16-wide register tuples do not pack into a 256-register budget (the same body on `Act`: 200-280 spilled registers).
struct ActE { u32x4 k[16]; };    // k[4 b + s]: f16 k-step s of block b

// Epilogue of one finished 32-feature tile T: ReLU, f16 pairs into k-steps 2(T&1), 2(T&1)+1 of block T>>1 of `dst` (the element order of Epi::stage_b), the
// running maximum for the f16 range guard — or, for the last trunk layer, the density head's dot product on the fp32 activations.
template <int T>
__device__ __forceinline__ void est_store(const f32x16& acc, ActE& dst, f16x2& peak16) {
#pragma unroll
    for (int J = 0; J < 2; ++J) {
        u32x4 hq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 xv = {acc[8 * J + 2 * e], acc[8 * J + 2 * e + 1]};
            const f16x2 hv = __builtin_elementwise_max(__builtin_convertvector(xv, f16x2), f16x2{(_Float16)0.0f, (_Float16)0.0f});     // ReLU on the packed pair
            peak16 = __builtin_elementwise_max(peak16, hv);       // (an overflowed conversion is +inf here: the range guard reads it at the end of the kernel)
            hq[e] = __builtin_bit_cast(unsigned, hv);
        }
        dst.k[4 * (T >> 1) + 2 * (T & 1) + J] = hq;
    }
}
template <int T>
__device__ __forceinline__ void est_head(const f32x16& acc, const float* tab, f32x2& sig) {
#pragma unroll
    for (int I = 0; I < 8; ++I) {
        const f32x2 w = *reinterpret_cast<const f32x2*>(tab + T * 32 + 2 * I);
        sig[0] = fmaf(relu_bits(acc[2 * I]), w[0], sig[0]);
        sig[1] = fmaf(relu_bits(acc[2 * I + 1]), w[1], sig[1]);
    }
}

// One layer of the estimate kernel: per output tile the MFMAs of its blocks (operand of slot G + PF read at slot G), then the tile's epilogue at once.
// NP point tiles per wave share every weight fragment.  enc: the encoding's four k-steps (HAS_ENC: the tile's first block).  HEAD: the output is not stored but
// dotted with the density head's row.
template <int NT, bool HAS_ENC, int NH, bool HEAD, int NP, int VARIANT>
__device__ __forceinline__ void run_layer_simple(Pipe<VARIANT>& P, Pre& pf, const ActE (&in)[NP], const u32x4 (&enc)[NP][4], const float* bias_tab, ActE (&dst)[NP],
                                                 const float* head_tab, f32x2 (&sig)[NP], f16x2& peak16) {
    constexpr int NB = (HAS_ENC ? 1 : 0) + NH;
    constexpr int NS = NB * SLOTS_PER_BLOCK;
    static_assert((NT * NS) % CHUNK_SLOTS == 0, "a layer is a whole number of chunks");
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc[NP];
        acc[0] = *reinterpret_cast<const f32x16*>(bias_tab + t * 32);
#pragma unroll
        for (int q = 1; q < NP; ++q) acc[q] = acc[0];
        static_for<0, NS>([&](auto GS) {
            constexpr int g = decltype(GS)::value;
            constexpr int G = t * NS + g;
            constexpr int cr = G % CHUNK_SLOTS;
            constexpr int bb = g / SLOTS_PER_BLOCK, s = g % SLOTS_PER_BLOCK;
            const f16x8 aw = __builtin_bit_cast(f16x8, pf.q[G % PF]);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if constexpr (HAS_ENC && bb == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, enc[q][s]), acc[q], 0, 0, 0);
                else acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, in[q].k[4 * (bb - (HAS_ENC ? 1 : 0)) + s]), acc[q], 0, 0, 0);
            }
            {
                constexpr int Gp = G + PF;
                constexpr bool next = (Gp / CHUNK_SLOTS) != (G / CHUNK_SLOTS);
                constexpr int blk = (Gp / SLOTS_PER_BLOCK) % CHUNK_BLOCKS;
                load_frag<Gp % SLOTS_PER_BLOCK, Gp % PF>(pf, P.block(next, blk), P.lane);
            }
            if constexpr (dma_piece(cr) >= 0) P.template prefetch_piece<(dma_piece(cr) >= 0 ? dma_piece(cr) : 0)>();
            if constexpr (cr == SYNC_SLOT) P.sync_next();
            if constexpr (cr == CHUNK_SLOTS - 1) P.advance();
        });
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if constexpr (HEAD) est_head<t>(acc[q], head_tab, sig[q]);
            else est_store<t>(acc[q], dst[q], peak16);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}
#endif

// A layer as THREE f16 products (VAR_TRUNK_X, layers 0 and 1): every logical block is a pair of stream blocks, the network's (its
// f16 area = Wh) then a residual block (f16 area = f16(W - Wh)).  The slot machinery (operand prefetch PF slots ahead, DMA pieces,
// chunk synchronisation, epilogue stages) is run_layer's; what a slot issues differs:
//   first block of a pair,  slots 0-3:  acc += Wh_k Xh_k ; acc += Wh_k Xl_k     (one operand read, two MFMAs)
//   second block of a pair, slots 0-3:  acc += Wl_k Xh_k
//   slots 4, 5 of both: nothing (the prefetched fp6 operands are not used)
// `lo`: the f16 residuals of the input blocks (encode / Epi::lo_dst), same element order as Blk::hv.
template <int NT, bool HAS_ENC, int NH, int VARIANT, class PEND, class EPI>
__device__ __forceinline__ f32x16 run_layer_x3(Pipe<VARIANT>& P, Pre& pf, unsigned& wsc, const Act& in, const u32x16* lo, const Blk& enc,
                                               const u32x16& enc_lo, const float* bias_tab, PEND&& pend, EPI& epi) {
    constexpr int NL = (HAS_ENC ? 1 : 0) + NH;   // logical blocks per tile
    constexpr int NB = 2 * NL;                   // stream blocks per tile
    constexpr int NS = NB * SLOTS_PER_BLOCK;     // slots per tile
    static_assert((NT * NS) % CHUNK_SLOTS == 0, "a layer is a whole number of chunks");
    f32x16 prev = {0};
    f32x16 bias_next = *reinterpret_cast<const f32x16*>(bias_tab);
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc = bias_next;
        static_for<0, NS>([&](auto GS) {
            constexpr int g = decltype(GS)::value;
            constexpr int G = t * NS + g;
            constexpr int cr = G % CHUNK_SLOTS;
            constexpr int bb = g / SLOTS_PER_BLOCK, s = g % SLOTS_PER_BLOCK;
            constexpr int lb = bb >> 1;          // logical block, and which of its two stream blocks
            constexpr bool resid = (bb & 1) != 0;
            if constexpr (s < 4) {
                const f16x8 aw = __builtin_bit_cast(f16x8, pf.q[G % 4]);
                if constexpr (HAS_ENC && lb == 0) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, quarter<s>(enc.hv)), acc, 0, 0, 0);
                    if constexpr (!resid) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, quarter<s>(enc_lo)), acc, 0, 0, 0);
                } else {
                    constexpr int ib = lb - (HAS_ENC ? 1 : 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, quarter<s>(in.b[ib].hv)), acc, 0, 0, 0);
                    if constexpr (!resid) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, quarter<s>(lo[ib])), acc, 0, 0, 0);
                }
            }
            {
                constexpr int Gp = G + PF;
                constexpr bool next = (Gp / CHUNK_SLOTS) != (G / CHUNK_SLOTS);
                constexpr int blk = (Gp / SLOTS_PER_BLOCK) % CHUNK_BLOCKS;
                load_frag<Gp % SLOTS_PER_BLOCK, Gp % 4>(pf, P.block(next, blk), P.lane);
            }
            if constexpr (g == NS / 2 && t + 1 < NT) bias_next = *reinterpret_cast<const f32x16*>(bias_tab + (t + 1) * 32);
            if constexpr (dma_piece(cr) >= 0) P.template prefetch_piece<(dma_piece(cr) >= 0 ? dma_piece(cr) : 0)>();
            static_for<0, 8 * N_STAGES>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                if constexpr (g == stage_slot_x3(q, NS)) {
                    if constexpr (t == 0) pend(std::integral_constant<int, q / N_STAGES>{}, std::integral_constant<int, q % N_STAGES>{});
                    else epi.template stage<(t > 0 ? t - 1 : 0), q / N_STAGES, q % N_STAGES>(prev);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (cr == SYNC_SLOT) P.sync_next();
            if constexpr (cr == CHUNK_SLOTS - 1) P.advance();
        });
        asm volatile("" : "+v"(acc));
        prev = acc;
    });
    return prev;
}

// ---------------------------------------------------------------------------------------------
// VAR_TRUNK_P: the 15-slot form.  Every trunk layer as three f16 products on (hi, lo) splits PLUS three block-scaled fp6 products
// for everything at 2^-22 of the result:
//     y = Wh Xh + Wh Xl + Wl Xh                      12 slots (f16; Wl from the residual block, Xl kept in registers)
//       + fp6(Wh) fp6(X3) + fp6(Wl) fp6(Xl) + fp6(W3) fp6(Xh)     3 slots (W3 / X3 = what two f16 terms leave of the fp32 value)
// Two f16 terms hold 22-23 bits of an fp32 operand (fewer where the lo term falls into the f16 denormals): "the reference run on a
// checkpoint stored with 22-bit mantissas" (launch scale 7), which moves the fine samples of rays whose density
// is a heavily cancelling sum.  With the three correction products every operand is represented to ~2^-26, below fp32's own 2^-24.
// Used for ONE query: the coarse pass's density, which places the fine samples (api.cpp: Q_MAIN_COARSE).
// Registers: per K = 64 block of an activation hv (16) + lo (16) + fp6(X3) (6) + scales (1); fp6(Xh) and fp6(Xl) are NOT kept — they are
// converted from hv / lo where a tile uses them (two v_cvt_scalef32_pk32_fp6_f16 per logical block and tile, in the shadow of the
// block's two-MFMA slots) — 312 registers for the in / out pair instead of 408.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x32 __attribute__((ext_vector_type(32)));
struct BlkP {
    u32x16 hv;        // the 4 f16 k-steps of Xh
    u32x16 lo;        // ... of Xl = f16(X - Xh), same element order
    u32x4 t6a;        // fp6(X3), X3 = X - Xh - Xl: bits 0..127
    u32x2 t6b;        // bits 128..191
    unsigned sc;      // e8m0 scales: byte 0 fp6(Xh), byte 1 fp6(Xl), byte 2 fp6(X3)
};
struct ActP { BlkP b[4]; };

// (x0, x1) -> packed f16 pairs h = rne(x), l = rne(x - h) and the packed bf16 pair of what is left, x - h - l (exact in fp32; bf16 keeps its
// exponent where f16 would underflow, and the fp6 conversion reads 4 of its 8 significant bits)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& hb, unsigned& lb, unsigned& tb) {
    const f32x2 xv = {x0, x1};
    hb = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, f16x2));
    float r0, r1, y0, y1;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(hb));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(hb));
    const f32x2 rv = {r0, r1};
    lb = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, f16x2));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(y0) : "v"(r0), "v"(lb));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(y1) : "v"(r1), "v"(lb));
    const f32x2 yv = {y0, y1};
    tb = __builtin_bit_cast(unsigned, __builtin_convertvector(yv, bf16x2));
}

// Completes a block of the 15-slot form: scales from the running max, the fp6 form of X3.  With 2^E <= max < 2^(E+1): |Xh| < 2^(E+1),
// |Xl| <= 2^(E-11), |X3| <= 2^(E-22) — or <= 2^-25 where Xl sits in the f16 denormals (any E), hence the floor on the third scale.
__device__ __forceinline__ void finish_block_p(BlkP& b, const u32x16& lres, const u32x16& tres, int& mxv, unsigned& peak) {
    unsigned mb = (unsigned)mxv;
    peak = mb > peak ? mb : peak;
    mb = mb > 0x0d800000u ? mb : 0x0d800000u;
    const unsigned e = mb >> 23;
    const unsigned et = e - 25 > 99u ? e - 25 : 99u;                  // scale of fp6(X3): 2^(E-25), at least 2^-28 (|X3| <= 2^-25 -> <= 8)
    const float st = __builtin_bit_cast(float, et << 23);
    const auto t6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_bf16(__builtin_bit_cast(bf16x32, tres), st);
    b.lo = lres;
    b.t6a = u32x4{(unsigned)t6[0], (unsigned)t6[1], (unsigned)t6[2], (unsigned)t6[3]};
    b.t6b = u32x2{(unsigned)t6[4], (unsigned)t6[5]};
    b.sc = (e - 2) | ((e - 14) << 8) | (et << 16);
    mxv = 0;
}

// Epilogue of a tile of the 15-slot form (cf. Epi): STORE: v = ReLU(acc) -> the three terms of block T>>1 of `dst`
template <bool STORE, int NCH>
struct EpiP {
    ActP* dst;
    f32x2* part[NCH > 0 ? NCH : 1];
    const float* tab[NCH > 0 ? NCH : 1];
    unsigned* peak;
    u32x16 lres, tres;
    u32x4 hq;
    int mxv;
    float sx0, sx1;
    f32x2 hw[NCH > 0 ? NCH : 1];

    template <int T, int I>
    __device__ __forceinline__ void stage_a(const f32x16& acc) {
        float x0 = acc[2 * I], x1 = acc[2 * I + 1];
        if constexpr (STORE) mxv = max(max(mxv, __builtin_bit_cast(int, x0)), __builtin_bit_cast(int, x1));   // (ReLU layers only: a negative never wins)
        x0 = relu_bits(x0);
        x1 = relu_bits(x1);
        sx0 = x0;
        sx1 = x1;
        pin(sx0);
        pin(sx1);
#pragma unroll
        for (int c = 0; c < NCH; ++c) hw[c] = *reinterpret_cast<const f32x2*>(tab[c] + T * 32 + 2 * I);
    }
    template <int T, int I>
    __device__ __forceinline__ void stage_b() {
        const float x0 = sx0, x1 = sx1;
        if constexpr (STORE) {
            constexpr int j = 2 * (T & 1) + (I >> 2);
            unsigned hb, lb, tb;
            split3_pair(x0, x1, hb, lb, tb);
            hq[I & 3] = hb;
            lres[4 * j + (I & 3)] = lb;
            tres[4 * j + (I & 3)] = tb;
            if constexpr ((I & 3) == 3) {
                asm volatile("" : "+v"(hq));
                dst->b[T >> 1].hv = with_quarter<j>(dst->b[T >> 1].hv, hq);
            }
            if constexpr (I == 7 && (T & 1) == 1) finish_block_p(dst->b[T >> 1], lres, tres, mxv, *peak);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            (*part[c])[0] = fmaf(x0, hw[c][0], (*part[c])[0]);
            (*part[c])[1] = fmaf(x1, hw[c][1], (*part[c])[1]);
            if constexpr (I == 7) asm volatile("" : "+v"(*part[c]));
        }
    }
    template <int T, int I, int K>
    __device__ __forceinline__ void stage(const f32x16& acc) {
        if constexpr (K == 0) stage_a<T, I>(acc);
        else stage_b<T, I>();
    }
    template <int T, int I>
    __device__ __forceinline__ void slice(const f32x16& acc) {
        stage_a<T, I>(acc);
        stage_b<T, I>();
    }
};

// A layer of the 15-slot form.  The slot machinery is run_layer_x3's (a logical block = a network block then its residual block); per pair:
//   network block   slots 0-3:  acc += Wh_k Xh_k ; acc += Wh_k Xl_k        (one operand read, two MFMAs)
//                   slot 4:     acc += fp6(Wh) fp6(X3)                      (the block's fp6(W) area, scale byte 0 | scale byte 2 of the activation block)
//                   slot 5:     acc += fp6(Wl) fp6(Xl)                      (its fp6(W - f16 W) area, byte 1 | byte 1)
//   residual block  slots 0-3:  acc += Wl_k Xh_k
//                   slot 4:     acc += fp6(W3) fp6(Xh)                      (the residual block's fp6 area, byte 0 | byte 0)
//                   slot 5:     nothing
// fp6(Xl) and fp6(Xh) of the input block are converted in slots 0 and 2 of the network block, behind its pairs of MFMAs.
template <int NT, bool HAS_ENC, int NH, int VARIANT, class PEND, class EPI>
__device__ __forceinline__ f32x16 run_layer_p(Pipe<VARIANT>& P, Pre& pf, unsigned& wsc, const ActP& in, const BlkP& enc, const float* bias_tab,
                                              PEND&& pend, EPI& epi) {
    constexpr int NL = (HAS_ENC ? 1 : 0) + NH;
    constexpr int NB = 2 * NL;
    constexpr int NS = NB * SLOTS_PER_BLOCK;
    static_assert((NT * NS) % CHUNK_SLOTS == 0, "a layer is a whole number of chunks");
    f32x16 prev = {0};
    u32x4 x6a, l6a;       // fp6 forms of the logical block in flight
    u32x2 x6b, l6b;
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc = *reinterpret_cast<const f32x16*>(bias_tab + t * 32);
        static_for<0, NS>([&](auto GS) {
            constexpr int g = decltype(GS)::value;
            constexpr int G = t * NS + g;
            constexpr int cr = G % CHUNK_SLOTS;
            constexpr int bb = g / SLOTS_PER_BLOCK, s = g % SLOTS_PER_BLOCK;
            constexpr int lb = bb >> 1;
            constexpr bool resid = (bb & 1) != 0;
            constexpr bool is_enc = HAS_ENC && lb == 0;
            constexpr int ib = is_enc ? 0 : lb - (HAS_ENC ? 1 : 0);
            const BlkP& xb = is_enc ? enc : in.b[ib];
            if constexpr (s == 4) wsc = pf.sc[G % 4];
            if constexpr (s < 4) {
                const f16x8 aw = __builtin_bit_cast(f16x8, pf.q[G % 4]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, quarter<s>(xb.hv)), acc, 0, 0, 0);
                if constexpr (!resid) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw, __builtin_bit_cast(f16x8, quarter<s>(xb.lo)), acc, 0, 0, 0);
            } else if constexpr (s == 4 && !resid) {
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp6_operand<true>(pf.q[G % 4], pf.d[G % 4]), fp6_operand<true>(xb.t6a, xb.t6b), acc, 2, 2, 0,
                                                                      (int)wsc, 2, (int)xb.sc);
            } else if constexpr (s == 5 && !resid) {
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp6_operand<true>(pf.q[G % 4], pf.d[G % 4]), fp6_operand<false>(l6a, l6b), acc, 2, 2, 1,
                                                                      (int)wsc, 1, (int)xb.sc);
            } else if constexpr (s == 4 && resid) {
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp6_operand<true>(pf.q[G % 4], pf.d[G % 4]), fp6_operand<false>(x6a, x6b), acc, 2, 2, 0,
                                                                      (int)wsc, 0, (int)xb.sc);
            }
            if constexpr (!resid && s == 0) {       // fp6(Xl) of this logical block: used by slot 5
                const float sl = __builtin_bit_cast(float, ((xb.sc >> 8) & 0xffu) << 23);
                const auto l6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, xb.lo), sl);
                l6a = u32x4{(unsigned)l6[0], (unsigned)l6[1], (unsigned)l6[2], (unsigned)l6[3]};
                l6b = u32x2{(unsigned)l6[4], (unsigned)l6[5]};
            }
            if constexpr (!resid && s == 2) {       // fp6(Xh): used by slot 4 of the residual block
                const float sh = __builtin_bit_cast(float, (xb.sc & 0xffu) << 23);
                const auto x6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, xb.hv), sh);
                x6a = u32x4{(unsigned)x6[0], (unsigned)x6[1], (unsigned)x6[2], (unsigned)x6[3]};
                x6b = u32x2{(unsigned)x6[4], (unsigned)x6[5]};
            }
            {
                constexpr int Gp = G + PF;
                constexpr bool next = (Gp / CHUNK_SLOTS) != (G / CHUNK_SLOTS);
                constexpr int blk = (Gp / SLOTS_PER_BLOCK) % CHUNK_BLOCKS;
                // (slot 5 of a residual block issues nothing: its operand is not fetched)
                if constexpr (!(Gp % SLOTS_PER_BLOCK == 5 && ((Gp / SLOTS_PER_BLOCK) & 1) == 1))
                    load_frag<Gp % SLOTS_PER_BLOCK, Gp % 4>(pf, P.block(next, blk), P.lane);
            }
            if constexpr (dma_piece(cr) >= 0) P.template prefetch_piece<(dma_piece(cr) >= 0 ? dma_piece(cr) : 0)>();
            static_for<0, 8 * N_STAGES>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                if constexpr (g == stage_slot_x3(q, NS)) {
                    if constexpr (t == 0) pend(std::integral_constant<int, q / N_STAGES>{}, std::integral_constant<int, q % N_STAGES>{});
                    else epi.template stage<(t > 0 ? t - 1 : 0), q / N_STAGES, q % N_STAGES>(prev);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (cr == SYNC_SLOT) P.sync_next();
            if constexpr (cr == CHUNK_SLOTS - 1) P.advance();
        });
        asm volatile("" : "+v"(acc));
        prev = acc;
    });
    return prev;
}

// the encoding as a block of the 15-slot form
template <int PAIRS>
__device__ __forceinline__ void encode_p(float x, float y, float z, int h, BlkP& enc, unsigned& peak) {
    float vals[32];
    const float mul = h ? (float)(1 << (PAIRS / 3)) : 1.0f;
    const TurnPair tx = to_turns(x), ty = to_turns(y), tz = to_turns(z);
#pragma unroll
    for (int u = 0; u < PAIRS; ++u) {
        const TurnPair tc = (u % 3 == 0) ? tx : ((u % 3 == 1) ? ty : tz);
        sincos_turns(tc, (float)(1 << (u / 3)) * mul, &vals[2 * u], &vals[2 * u + 1]);
    }
    vals[2 * PAIRS] = h ? z : x;
    vals[2 * PAIRS + 1] = h ? 0.0f : y;
#pragma unroll
    for (int i = 2 * PAIRS + 2; i < 32; ++i) vals[i] = 0.0f;
    u32x16 lres, tres;
    int mxv = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u32x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned hb, lb, tb;
            split3_pair(vals[8 * j + 2 * e], vals[8 * j + 2 * e + 1], hb, lb, tb);
            hv[e] = hb;
            lres[4 * j + e] = lb;
            tres[4 * j + e] = tb;
            mxv = max(mxv, max(__builtin_bit_cast(int, vals[8 * j + 2 * e]) & 0x7fffffff, __builtin_bit_cast(int, vals[8 * j + 2 * e + 1]) & 0x7fffffff));
        }
        if (j == 0) enc.hv = with_quarter<0>(enc.hv, hv);
        else if (j == 1) enc.hv = with_quarter<1>(enc.hv, hv);
        else if (j == 2) enc.hv = with_quarter<2>(enc.hv, hv);
        else enc.hv = with_quarter<3>(enc.hv, hv);
    }
    finish_block_p(enc, lres, tres, mxv, peak);
}

// [x, sin(2^k x), cos(2^k x)] in the slot order of layout.h::enc_ref_index -> one block
template <int PAIRS>
__device__ __forceinline__ void encode(float x, float y, float z, int h, Blk& enc, unsigned& peak, u32x16* lo_out = nullptr) {
    float vals[32];
    const float mul = h ? (float)(1 << (PAIRS / 3)) : 1.0f;
    const TurnPair tx = to_turns(x), ty = to_turns(y), tz = to_turns(z);
#pragma unroll
    for (int u = 0; u < PAIRS; ++u) {
        const TurnPair tc = (u % 3 == 0) ? tx : ((u % 3 == 1) ? ty : tz);
        sincos_turns(tc, (float)(1 << (u / 3)) * mul, &vals[2 * u], &vals[2 * u + 1]);
    }
    vals[2 * PAIRS] = h ? z : x;
    vals[2 * PAIRS + 1] = h ? 0.0f : y;
#pragma unroll
    for (int i = 2 * PAIRS + 2; i < 32; ++i) vals[i] = 0.0f;
    u32x16 lres;
    int mxv = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u32x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned hb, lb;
            split_pair(vals[8 * j + 2 * e], vals[8 * j + 2 * e + 1], hb, lb);
            hv[e] = hb;
            lres[4 * j + e] = lb;
            mxv = max(mxv, max(__builtin_bit_cast(int, vals[8 * j + 2 * e]) & 0x7fffffff, __builtin_bit_cast(int, vals[8 * j + 2 * e + 1]) & 0x7fffffff));
        }
        if (j == 0) enc.hv = with_quarter<0>(enc.hv, hv);
        else if (j == 1) enc.hv = with_quarter<1>(enc.hv, hv);
        else if (j == 2) enc.hv = with_quarter<2>(enc.hv, hv);
        else enc.hv = with_quarter<3>(enc.hv, hv);
    }
    if (lo_out != nullptr) *lo_out = lres;   // the f16 residuals themselves (three-product layers)
    finish_block(enc, lres, mxv, peak);
}

#if defined(IBL_MX_EST) && IBL_MX_EST_TILES == 1
#define IBL_MX_WGS_PER_CU 2
#else
#define IBL_MX_WGS_PER_CU 1
#endif
template <int VARIANT>
__global__ __launch_bounds__(256, IBL_MX_WGS_PER_CU) void mlp_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    float* tabs = reinterpret_cast<float*>(smem + R_RING_BYTES);
    for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += 256)
        reinterpret_cast<f32x4*>(tabs)[i] = reinterpret_cast<const f32x4*>(a.tables)[i];
    __syncthreads();
    const float* ltab = tabs + h * 16;

#if defined(IBL_MX_ABLATE_NO_LOADS) || defined(IBL_MX_ABLATE_HALF_LOADS)
    for (int i = threadIdx.x; i < LDS_RING_BYTES / 16; i += 256) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
#endif
    Pipe<VARIANT> P;
    P.stream = a.stream;
    P.ring = smem;
    P.lane = lane;
    P.wave = wave;
    P.lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    P.voff = lane * 16 + wave * 8192;
    P.start();
    Pre pf;
    static_for<0, PF>([&](auto I) {     // the operands of the first PF slots
        constexpr int i = decltype(I)::value;
        load_frag<i % SLOTS_PER_BLOCK, i>(pf, P.block(false, i / SLOTS_PER_BLOCK), lane);
    });
    unsigned wsc = 0, peak = 0;

    constexpr bool LIST = VARIANT == VAR_REFL_LIST || VARIANT == VAR_FULL_LIST || VARIANT == VAR_TRUNK_X_LIST;      // a compact list of points with a flat index each (MlpArgs::out_index)
    long n_total = a.n_pts;
#ifdef IBL_MX_EST
    constexpr bool EST_FLAVOUR = true;      // (the estimate kernel also takes a list: estimates in z-chunks, api.cpp estimate_chunked)
#else
    constexpr bool EST_FLAVOUR = false;
#endif
    if constexpr (VARIANT == VAR_TRUNK_P || VARIANT == VAR_TRUNK || LIST || EST_FLAVOUR) {
        if (a.n_pts_dev != nullptr) n_total = *a.n_pts_dev;      // a compact list: its length is known on the device only (k_select_points)
    }
#if defined(IBL_MX_EST) && IBL_MX_EST_TILES > 1
    const long n_groups = (n_total + 128 * IBL_MX_EST_TILES - 1) / (128 * IBL_MX_EST_TILES);
#else
    const long n_groups = (n_total + 127) / 128;
#endif
#ifdef IBL_MX_ABLATE_PROLOGUE   // timing ablation only (results are garbage): the input stage (points + encoding) runs in the first iteration only
    Blk pe, de;
    u32x16 pe_lo, loA[4];
#endif
    for (long g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const long p = g * 128 + wave * 32 + (lane & 31);
        const bool valid = p < n_total;
        constexpr bool TRUNKV = VARIANT == VAR_TRUNK || variant_trunk_x(VARIANT) || VARIANT == VAR_TRUNK_P;
#ifdef IBL_MX_ABLATE_PROLOGUE
        if (g == blockIdx.x) {
#endif
        float px = 0.f, py = 0.f, pz = 0.f;
        PointGenK gen = nullptr;
#ifndef IBL_NO_POINT_GEN   // (-DIBL_NO_POINT_GEN: A/B build of scratch/trunk_ab.sh, the input stage without the generation branch)
        if constexpr (TRUNKV) gen = kernarg_point_gen((unsigned)offsetof(MlpArgs, gen));
#endif
        if (gen != nullptr && gen->rays_o != nullptr) {   // the offset copies of the epsilon-normal, generated here (gen_points.h) instead of read from a batch
            if (valid) gen_offset_point(load_point_gen(gen), (unsigned)p, px, py, pz);
        } else if (valid) {
            px = a.pts[3 * p + 0];
            py = a.pts[3 * p + 1];
            pz = a.pts[3 * p + 2];
        }
#ifdef IBL_MX_EST
        {
            constexpr int NP = IBL_MX_EST_TILES;
            u32x4 pe[NP][4];
            long pq[NP];
            bool okq[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                pq[q] = NP == 1 ? p : g * (128L * NP) + wave * (32L * NP) + 32 * q + (lane & 31);
                okq[q] = pq[q] < n_total;
                float x = px, y = py, z = pz;
                if (NP > 1) {
                    x = y = z = 0.f;
                    if (gen != nullptr && gen->rays_o != nullptr) {
                        if (okq[q]) gen_offset_point(load_point_gen(gen), (unsigned)pq[q], x, y, z);
                    } else if (okq[q]) {
                        x = a.pts[3 * pq[q] + 0]; y = a.pts[3 * pq[q] + 1]; z = a.pts[3 * pq[q] + 2];
                    }
                }
                Blk enc_blk;
                encode<PE_PAIRS_PER_HALF>(x, y, z, h, enc_blk, peak, nullptr);
                pe[q][0] = quarter<0>(enc_blk.hv); pe[q][1] = quarter<1>(enc_blk.hv); pe[q][2] = quarter<2>(enc_blk.hv); pe[q][3] = quarter<3>(enc_blk.hv);
            }
            ActE A[NP], B[NP];
            f32x2 sig[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) sig[q] = f32x2{0.0f, 0.0f};
            f16x2 peak16 = {(_Float16)0.0f, (_Float16)0.0f};
            const float* bias = ltab + TAB_BIAS;
            run_layer_simple<8, true, 0, false>(P, pf, A /*unused*/, pe, bias + BT_L0 * 32, A, nullptr, sig, peak16);                       // 0 -> A
            for (int l = 1; l <= 3; l += 2) {                                                                                                 // 1..4: A -> B -> A
                run_layer_simple<8, false, 4, false>(P, pf, A, pe, bias + (BT_L0 + 8 * (l & 3)) * 32, B, nullptr, sig, peak16);
                run_layer_simple<8, false, 4, false>(P, pf, B, pe, bias + (BT_L0 + 8 * (l & 3) + 8) * 32, A, nullptr, sig, peak16);
            }
            run_layer_simple<8, true, 4, false>(P, pf, A, pe, bias + (BT_L0 + 40) * 32, B, nullptr, sig, peak16);                            // 5 (skip): A -> B
            run_layer_simple<8, false, 4, false>(P, pf, B, pe, bias + (BT_L0 + 48) * 32, A, nullptr, sig, peak16);                           // 6: B -> A
            run_layer_simple<8, false, 4, true>(P, pf, A, pe, bias + (BT_L0 + 56) * 32, B /*unused*/, ltab + TAB_SIG, sig, peak16);          // 7: A -> sigma head on its fp32 activations
            {   // the largest activation of the group, as the bits of a non-negative float (what the range guard at the end of the kernel compares)
                const float pk = fmaxf((float)peak16[0], (float)peak16[1]);
                const unsigned pb = __builtin_bit_cast(unsigned, pk);
                peak = pb > peak ? pb : peak;
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const float p0 = sig[q][0] + sig[q][1];
                const float sg = p0 + __shfl_xor(p0, 32) + tabs[TAB_SCALAR];
                if (okq[q] && h == 0) a.out[(a.out_index != nullptr ? (long)a.out_index[pq[q]] : pq[q]) * a.out_stride] = sg;
            }
            continue;
        }
#endif
        if constexpr (VARIANT == VAR_TRUNK_P) {
            // ---- the 15-slot form: every trunk layer through run_layer_p, activations as (hi, lo, fp6 third term) ----
            BlkP pe;
            encode_p<PE_PAIRS_PER_HALF>(px, py, pz, h, pe, peak);
            asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));     // (the point itself stays live: three registers; its encoding does not, see layer 5)
            ActP A, B;
            f32x2 sig = {0.0f, 0.0f};
            const float* bias = ltab + TAB_BIAS;
            auto none = [](auto, auto) {};
            EpiP<true, 0> eA{&A, {nullptr}, {nullptr}, &peak}, eB{&B, {nullptr}, {nullptr}, &peak};
            eA.mxv = 0;
            eB.mxv = 0;
            f32x16 pacc = run_layer_p<8, true, 0>(P, pf, wsc, A /*unused*/, pe, bias + BT_L0 * 32, none, eA);              // 0 -> A
            for (int l = 1; l <= 3; l += 2) {                                                                                 // 1..4: A -> B -> A
                pacc = run_layer_p<8, false, 4>(P, pf, wsc, A, pe, bias + (BT_L0 + 8 * (l & 3)) * 32,
                                                [&](auto I, auto K) { eA.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, eB);
                pacc = run_layer_p<8, false, 4>(P, pf, wsc, B, pe, bias + (BT_L0 + 8 * (l & 3) + 8) * 32,
                                                [&](auto I, auto K) { eB.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, eA);
            }
            // the skip layer reads the encoding again: encoded again here (39 registers not held across layers 1-4; the input stage is < 1 % of a group's time)
            encode_p<PE_PAIRS_PER_HALF>(px, py, pz, h, pe, peak);
            pacc = run_layer_p<8, true, 4>(P, pf, wsc, A, pe, bias + (BT_L0 + 40) * 32,                                       // 5 (skip): A -> B
                                           [&](auto I, auto K) { eA.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, eB);
            pacc = run_layer_p<8, false, 4>(P, pf, wsc, B, pe, bias + (BT_L0 + 48) * 32,                                      // 6: B -> A
                                            [&](auto I, auto K) { eB.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, eA);
            EpiP<false, 1> e7{nullptr, {&sig}, {ltab + TAB_SIG}, &peak};                                                      // 7: A -> sigma head on its fp32 activations
            pacc = run_layer_p<8, false, 4>(P, pf, wsc, A, pe, bias + (BT_L0 + 56) * 32,
                                            [&](auto I, auto K) { eA.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, e7);
            static_for<0, 8>([&](auto I) { e7.template slice<7, decltype(I)::value>(pacc); });
            const float p0 = sig[0] + sig[1];
            const float sg = p0 + __shfl_xor(p0, 32) + tabs[TAB_SCALAR];
            if (valid && h == 0) a.out[(a.out_index != nullptr ? (long)a.out_index[p] : (long)p) * a.out_stride] = sg;
            continue;
        }
#ifndef IBL_MX_ABLATE_PROLOGUE
        Blk pe, de;
        u32x16 pe_lo, loA[4];   // VAR_TRUNK_X: f16 residuals of the encoding and of layer 0's output
#endif
        encode<PE_PAIRS_PER_HALF>(px, py, pz, h, pe, peak, variant_trunk_x(VARIANT) ? &pe_lo : nullptr);
        if constexpr (!TRUNKV && !variant_ci(VARIANT)) {
            float dx = 0.f, dy = 0.f, dz = 0.f;
            if (valid) {
                const unsigned r = (LIST ? (unsigned)a.out_index[p] : (unsigned)p) / (unsigned)a.pts_per_ray;
                dx = a.dirs[3 * (size_t)r + 0];
                dy = a.dirs[3 * (size_t)r + 1];
                dz = a.dirs[3 * (size_t)r + 2];
            }
            encode<DE_PAIRS_PER_HALF>(dx, dy, dz, h, de, peak);
        }
#ifdef IBL_MX_ABLATE_PROLOGUE
        }
#endif

        Act A, B;
        f32x2 part[RAW_CH];
#pragma unroll
        for (int c = 0; c < RAW_CH; ++c) part[c] = f32x2{0.0f, 0.0f};
        const float* bias = ltab + TAB_BIAS;
        auto none = [](auto, auto) {};
        auto flush = [&](auto& e, auto T, const f32x16& acc) {
            static_for<0, 8>([&](auto I) { e.template slice<decltype(T)::value, decltype(I)::value>(acc); });
        };
        using T7 = std::integral_constant<int, 7>;
        using T3 = std::integral_constant<int, 3>;
        Epi<true, true, 0> eA{&A, {nullptr}, {nullptr}, &peak}, eB{&B, {nullptr}, {nullptr}, &peak};

        f32x16 pacc;
        if constexpr (variant_trunk_x(VARIANT)) {
            // positions_linears.0 and .1 as three f16 products (-> A with its f16 residuals in loA, -> B), then .2 (B -> A)
            Epi<true, true, 0, true> eA0{&A, {nullptr}, {nullptr}, &peak, loA};
            pacc = run_layer_x3<8, true, 0>(P, pf, wsc, A /*unused*/, loA /*unused*/, pe, pe_lo, bias + BT_L0 * 32, none, eA0);
            pacc = run_layer_x3<8, false, 4>(P, pf, wsc, A, loA, pe, pe_lo, bias + (BT_L0 + 8) * 32,
                                             [&](auto I, auto K) { eA0.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, eB);
            pacc = run_layer<8, false, 4>(P, pf, wsc, B, pe, bias + (BT_L0 + 16) * 32,
                                          [&](auto I, auto K) { eB.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, eA);
        } else {
            // positions_linears.0 : 63 -> 256, ReLU (-> A)
            pacc = run_layer<8, true, 0>(P, pf, wsc, A /*unused*/, pe, bias + BT_L0 * 32, none, eA);
        }
        // positions_linears.1..4, two layers per trip (A -> B -> A)   (VAR_TRUNK_X: .3 and .4 only)
        for (int l = (variant_trunk_x(VARIANT) ? 3 : 1); l <= 3; l += 2) {
            pacc = run_layer<8, false, 4>(P, pf, wsc, A, pe, bias + (BT_L0 + 8 * (l & 3)) * 32,
                                          [&](auto I, auto K) { eA.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, eB);
            pacc = run_layer<8, false, 4>(P, pf, wsc, B, pe, bias + (BT_L0 + 8 * (l & 3) + 8) * 32,
                                          [&](auto I, auto K) { eB.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, eA);
        }
        // positions_linears.5 : cat([x63, h]) (ibl_nerf.py:167-168) (A -> B)
        pacc = run_layer<8, true, 4>(P, pf, wsc, A, pe, bias + (BT_L0 + 40) * 32,
                                     [&](auto I, auto K) { eA.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, eB);
        // positions_linears.6 (B -> A)
        pacc = run_layer<8, false, 4>(P, pf, wsc, B, pe, bias + (BT_L0 + 48) * 32,
                                      [&](auto I, auto K) { eB.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, eA);
        // positions_linears.7 (A -> B); sigma_linear / roughness_linear on its fp32 activations
        constexpr bool CI = variant_ci(VARIANT), ALBIRR = variant_albirr(VARIANT);
        const float* rad[3] = {ltab + TAB_RAD, ltab + TAB_RAD + 256, ltab + TAB_RAD + 512};
        // with is_color_independent_to_direction the radiance_linear rows are dotted with h7 itself (ibl_nerf.py:192, :199)
        auto e7 = [&] {
            if constexpr (VARIANT == VAR_FULL || VARIANT == VAR_FULL_LIST)
                return Epi<true, true, 2>{&B, {&part[0], &part[4]}, {ltab + TAB_SIG, ltab + TAB_ROUGH}, &peak};
            else if constexpr (VARIANT == VAR_FULL_CI)
                return Epi<true, true, 5>{&B, {&part[0], &part[4], &part[6], &part[7], &part[8]},
                                          {ltab + TAB_SIG, ltab + TAB_ROUGH, rad[0], rad[1], rad[2]}, &peak};
            else if constexpr (VARIANT == VAR_REFL_CI)
                return Epi<true, true, 4>{&B, {&part[0], &part[6], &part[7], &part[8]}, {ltab + TAB_SIG, rad[0], rad[1], rad[2]}, &peak};
            else
                return Epi<!TRUNKV, true, 1>{&B, {&part[0]}, {ltab + TAB_SIG}, &peak};
        }();
        pacc = run_layer<8, false, 4>(P, pf, wsc, A, pe, bias + (BT_L0 + 56) * 32,
                                      [&](auto I, auto K) { eA.template stage<7, decltype(I)::value, decltype(K)::value>(pacc); }, e7);

        if constexpr (TRUNKV) {
            flush(e7, T7{}, pacc);
        } else {
            Epi<true, false, 0> eFeat{&A, {nullptr}, {nullptr}, &peak};
            Epi<false, true, 3> eAlb{nullptr, {&part[1], &part[2], &part[3]},
                                     {ltab + TAB_ALB, ltab + TAB_ALB + 128, ltab + TAB_ALB + 256}, &peak};
            Epi<false, true, 1> eIrr{nullptr, {&part[5]}, {ltab + TAB_IRR}, &peak};
            Epi<true, true, 3> eView{&B, {&part[6], &part[7], &part[8]}, {rad[0], rad[1], rad[2]}, &peak};
            Epi<false, true, 3> eAr0{nullptr, {&part[9], &part[10], &part[11]},
                                     {ltab + TAB_AR, ltab + TAB_AR + 128, ltab + TAB_AR + 256}, &peak};
            Epi<false, true, 3> eAr1{nullptr, {&part[12], &part[13], &part[14]},
                                     {ltab + TAB_AR + 384, ltab + TAB_AR + 512, ltab + TAB_AR + 640}, &peak};
            Epi<false, true, 3> eAr2{nullptr, {&part[15], &part[16], &part[17]},
                                     {ltab + TAB_AR + 768, ltab + TAB_AR + 896, ltab + TAB_AR + 1024}, &peak};
            // the epilogue still owed by the previous layer, as the `pend` of the next one
#define IBL_PEND(e, T, acc) [&](auto I, auto K) { (e).template stage<T, decltype(I)::value, decltype(K)::value>(acc); }
            // feature_linear : no activation (B = h7 -> A = feature)
            if constexpr (!CI) pacc = run_layer<8, false, 4>(P, pf, wsc, B, pe, bias + BT_FEAT * 32, IBL_PEND(e7, 7, pacc), eFeat);
            // albedo_feature_linear -> albedo_linear ; irradiance_feature_linear -> irradiance_linear (both read h7 = B)
            if constexpr (ALBIRR) {
                f32x16 qacc;
                if constexpr (CI) qacc = run_layer<4, false, 4>(P, pf, wsc, B, pe, bias + BT_ALB * 32, IBL_PEND(e7, 7, pacc), eAlb);
                else qacc = run_layer<4, false, 4>(P, pf, wsc, B, pe, bias + BT_ALB * 32, IBL_PEND(eFeat, 7, pacc), eAlb);
                pacc = run_layer<4, false, 4>(P, pf, wsc, B, pe, bias + BT_IRR * 32, IBL_PEND(eAlb, 3, qacc), eIrr);
            }
            // views_linears.0 : cat([feature, dir27]) (A -> B); radiance_linear
            if constexpr (!CI) {
                if constexpr (ALBIRR) pacc = run_layer<8, true, 4>(P, pf, wsc, A, de, bias + BT_VIEW * 32, IBL_PEND(eIrr, 3, pacc), eView);
                else pacc = run_layer<8, true, 4>(P, pf, wsc, A, de, bias + BT_VIEW * 32, IBL_PEND(eFeat, 7, pacc), eView);
            }
            // additional_radiance_feature_linear.k -> additional_radiance_linear.k, on B = views output, or h7 when colour-independent
            if constexpr (!CI) pacc = run_layer<4, false, 4>(P, pf, wsc, B, pe, bias + BT_AR * 32, IBL_PEND(eView, 7, pacc), eAr0);
            else if constexpr (ALBIRR) pacc = run_layer<4, false, 4>(P, pf, wsc, B, pe, bias + BT_AR * 32, IBL_PEND(eIrr, 3, pacc), eAr0);
            else pacc = run_layer<4, false, 4>(P, pf, wsc, B, pe, bias + BT_AR * 32, IBL_PEND(e7, 7, pacc), eAr0);
            pacc = run_layer<4, false, 4>(P, pf, wsc, B, pe, bias + (BT_AR + 4) * 32, IBL_PEND(eAr0, 3, pacc), eAr1);
            pacc = run_layer<4, false, 4>(P, pf, wsc, B, pe, bias + (BT_AR + 8) * 32, IBL_PEND(eAr1, 3, pacc), eAr2);
#undef IBL_PEND
            flush(eAr2, T3{}, pacc);
        }

        const float* sc = tabs + TAB_SCALAR;
        if constexpr (TRUNKV) {
            const float p0 = part[0][0] + part[0][1];
            const float s = p0 + __shfl_xor(p0, 32) + sc[0];
            if (valid && h == 0) {
                // (the plain TRUNK form also takes a list when it serves as the density ESTIMATE of a network whose plain-f16 estimates were refused: z-chunks and the
                // offset copies' front / behind ranges, api.cpp estimate_chunked / offsets_on_lists — a run-time question there, like VAR_TRUNK_P's)
                const bool scattered = LIST || (VARIANT == VAR_TRUNK && a.out_index != nullptr);
                a.out[(scattered ? (long)a.out_index[p] : (long)p) * a.out_stride] = s;
            }
        } else {
            float tot[RAW_CH];
#pragma unroll
            for (int c = 0; c < RAW_CH; ++c) {
                const float pc = part[c][0] + part[c][1];
                tot[c] = pc + __shfl_xor(pc, 32) + sc[c];
            }
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
    P.drain();
    // f16 range check: any encoded input or activation at or beyond 65520 overflowed its f16 form
    if (a.range_flag != nullptr && peak >= 0x477ff000u) atomicOr(a.range_flag, 1u);
}

}  // namespace mxk / mxk16

// The three instantiations are compiled as three objects (build.py passes -DIBL_MX_VARIANT=0|1|2) so that they
// build side by side; the object for VAR_FULL also carries the dispatcher.  Without the macro (scratch/mxdev.sh
// with -DIBL_MX_DEV_TRUNK_ONLY, or a plain compile) everything lands in one object.
template <int VARIANT>
static hipError_t launch_variant(const MlpArgs& a, int grid, hipStream_t stream) {
    static bool attr_set[64] = {};   // per device: one process may hold contexts on several (iblnerf_options.device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)IBL_MXK::mlp_kernel<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, IBL_MXK::R_LDS_BYTES);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(IBL_MXK::mlp_kernel<VARIANT>, dim3(grid), dim3(256), IBL_MXK::R_LDS_BYTES, stream, a);
    return hipGetLastError();
}

#ifdef IBL_MX_F16ONLY
#define IBL_L(x) launch_mlp_mx16_##x
#define IBL_DISPATCH launch_mlp_mx16
#else
#define IBL_L(x) launch_mlp_mx_##x
#define IBL_DISPATCH launch_mlp_mx
#endif
#if defined(IBL_MX_VARIANT)
#if IBL_MX_VARIANT == 0
hipError_t IBL_L(full)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_FULL>(a, grid, s); }
#elif IBL_MX_VARIANT == 1
#ifdef IBL_MX_EST
hipError_t IBL_L(trunk)(const MlpArgs& a, int grid, hipStream_t s) {      // (two workgroups per CU: the dispatcher's grid is one per CU)
    const long n_groups = (a.n_pts + 128 * IBL_MX_EST_TILES - 1) / (128 * IBL_MX_EST_TILES);
    const long wgs = (IBL_MX_EST_TILES == 1 ? 2L : 1L) * grid;
    return launch_variant<VAR_TRUNK>(a, (int)(n_groups < wgs ? n_groups : wgs), s);
}
#else
hipError_t IBL_L(trunk)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_TRUNK>(a, grid, s); }
#endif
#elif IBL_MX_VARIANT == 2
hipError_t IBL_L(refl)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_REFL>(a, grid, s); }
#elif IBL_MX_VARIANT == 5
hipError_t IBL_L(trunk_x)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_TRUNK_X>(a, grid, s); }
#elif IBL_MX_VARIANT == 13
hipError_t IBL_L(trunk_p)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_TRUNK_P>(a, grid, s); }
#elif IBL_MX_VARIANT == 15
hipError_t IBL_L(refl_list)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_REFL_LIST>(a, grid, s); }
#elif IBL_MX_VARIANT == 16
hipError_t IBL_L(full_list)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_FULL_LIST>(a, grid, s); }
#elif IBL_MX_VARIANT == 17
hipError_t IBL_L(trunk_x_list)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_TRUNK_X_LIST>(a, grid, s); }
#elif IBL_MX_VARIANT == 3
hipError_t IBL_L(full_ci)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_FULL_CI>(a, grid, s); }
#else
hipError_t IBL_L(refl_ci)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_REFL_CI>(a, grid, s); }
#endif
#else
hipError_t IBL_L(trunk)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_TRUNK>(a, grid, s); }
#ifndef IBL_MX_F16ONLY
hipError_t IBL_L(trunk_x)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_TRUNK_X>(a, grid, s); }
hipError_t IBL_L(trunk_p)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_TRUNK_P>(a, grid, s); }
hipError_t IBL_L(trunk_x_list)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_TRUNK_X_LIST>(a, grid, s); }
#endif
#ifndef IBL_MX_DEV_TRUNK_ONLY
hipError_t IBL_L(full)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_FULL>(a, grid, s); }
hipError_t IBL_L(refl)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_REFL>(a, grid, s); }
#ifndef IBL_MX_F16ONLY
hipError_t IBL_L(refl_list)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_REFL_LIST>(a, grid, s); }
hipError_t IBL_L(full_list)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_FULL_LIST>(a, grid, s); }
#endif
hipError_t IBL_L(full_ci)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_FULL_CI>(a, grid, s); }
hipError_t IBL_L(refl_ci)(const MlpArgs& a, int grid, hipStream_t s) { return launch_variant<VAR_REFL_CI>(a, grid, s); }
#endif
#endif

#if !defined(IBL_MX_VARIANT) || IBL_MX_VARIANT == 0
hipError_t IBL_L(full)(const MlpArgs& a, int grid, hipStream_t s);
hipError_t IBL_L(trunk)(const MlpArgs& a, int grid, hipStream_t s);
hipError_t IBL_L(refl)(const MlpArgs& a, int grid, hipStream_t s);
hipError_t IBL_L(full_ci)(const MlpArgs& a, int grid, hipStream_t s);
hipError_t IBL_L(refl_ci)(const MlpArgs& a, int grid, hipStream_t s);
hipError_t IBL_L(trunk_x)(const MlpArgs& a, int grid, hipStream_t s);
hipError_t IBL_L(trunk_p)(const MlpArgs& a, int grid, hipStream_t s);
hipError_t IBL_L(refl_list)(const MlpArgs& a, int grid, hipStream_t s);
hipError_t IBL_L(full_list)(const MlpArgs& a, int grid, hipStream_t s);
hipError_t IBL_L(trunk_x_list)(const MlpArgs& a, int grid, hipStream_t s);
hipError_t IBL_DISPATCH(int variant, const MlpArgs& a, int n_cu, hipStream_t stream) {
    if (a.n_pts <= 0) return hipSuccess;
    const long n_groups = (a.n_pts + 127) / 128;
    const int grid = (int)(n_groups < n_cu ? n_groups : n_cu);
    switch (variant) {
#ifndef IBL_MX_DEV_TRUNK_ONLY
        case VAR_FULL: return IBL_L(full)(a, grid, stream);
        case VAR_REFL: return IBL_L(refl)(a, grid, stream);
        case VAR_FULL_CI: return IBL_L(full_ci)(a, grid, stream);
        case VAR_REFL_CI: return IBL_L(refl_ci)(a, grid, stream);
#endif
        case VAR_TRUNK: return IBL_L(trunk)(a, grid, stream);   // (in plain f16: a density ESTIMATE only, api.cpp Q_ESTIMATE — never what feeds the finite-difference normal)
#ifndef IBL_MX_F16ONLY
        case VAR_TRUNK_X: return IBL_L(trunk_x)(a, grid, stream);
        case VAR_TRUNK_P: return IBL_L(trunk_p)(a, grid, stream);
        case VAR_TRUNK_X_LIST: return IBL_L(trunk_x_list)(a, grid, stream);
#ifndef IBL_MX_DEV_TRUNK_ONLY
        case VAR_REFL_LIST: return IBL_L(refl_list)(a, grid, stream);
        case VAR_FULL_LIST: return IBL_L(full_list)(a, grid, stream);
#endif
#endif
        default: return hipErrorInvalidValue;
    }
}
#endif

}  // namespace ibl
