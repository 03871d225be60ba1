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
#include <hip/hip_runtime.h>

#include <type_traits>

#include "layout.h"
#include "kernels.h"
#include "sincos_enc.h"

namespace ibl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Act { bf16x8 hi[16]; bf16x8 lo[16]; };  // 256 features = 16 k-steps of B fragments
struct Enc { bf16x8 hi[4]; bf16x8 lo[4]; };    // up to 64 encoding slots = 4 k-steps

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------------
// weight-stream pipeline: chunk at program position c lives in ring slot c % 3; the loads for
// position c+2 are issued when position c starts computing.
// ---------------------------------------------------------------------------------------------
template <int VARIANT>
struct Pipe {
    static constexpr int N_PROG = VARIANT == VAR_FULL ? N_CHUNKS : (VARIANT == VAR_TRUNK ? N_CHUNKS_TRUNK : N_CHUNKS - 8);
    // program position -> stream chunk: the reflected-ray variant skips the 8 albedo / irradiance feature chunks
    const char* stream;
    char* ring;          // generic pointer to the ring (for ds_read)
    unsigned lds_ring;   // LDS byte address of the ring (for M0)
    unsigned voff;       // lane*16 + wave*8192: this lane's byte offset inside a chunk for the DMA
    int lane, wave;
    int pos;

    __device__ __forceinline__ static int stream_chunk(int p) {
        if (VARIANT == VAR_REFL) return p < CH_ALB ? p : p + 8;   // skip albedo / irradiance feature chunks
        return p;
    }
    // One 32 KiB chunk = 8 LDS-DMA instructions per wave (wave w copies bytes [8192w, 8192w+8192)).
    // Scalar base + one VGPR offset (saddr form) so no per-piece 64-bit VGPR address exists; the
    // loads are invisible to hipcc's waitcnt bookkeeping and are counted by hand (begin()/end()).
    // M0 carries the wave-uniform LDS destination; the DMA adds lane*16 itself.
    __device__ __forceinline__ void issue(int p) const {
        const char* src = stream + (size_t)stream_chunk(p) * CHUNK_BYTES;             // uniform (SGPR pair)
        const unsigned dst = lds_ring + (unsigned)(p % RING_SLOTS) * CHUNK_BYTES + wave * 8192;  // uniform
        unsigned keep, t;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "v_mov_b32 %1, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %4\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "v_add_u32 %1, 0x400, %1\n\t"
            "global_load_lds_dwordx4 %1, %4\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "v_add_u32 %1, 0x400, %1\n\t"
            "global_load_lds_dwordx4 %1, %4\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "v_add_u32 %1, 0x400, %1\n\t"
            "global_load_lds_dwordx4 %1, %4\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "v_add_u32 %1, 0x400, %1\n\t"
            "global_load_lds_dwordx4 %1, %4\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "v_add_u32 %1, 0x400, %1\n\t"
            "global_load_lds_dwordx4 %1, %4\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "v_add_u32 %1, 0x400, %1\n\t"
            "global_load_lds_dwordx4 %1, %4\n\t"
            "s_add_u32 m0, m0, 0x400\n\t"
            "v_add_u32 %1, 0x400, %1\n\t"
            "global_load_lds_dwordx4 %1, %4\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep), "=&v"(t)
            : "v"(voff), "s"(dst), "s"(src)
            : "memory", "scc");
    }
    __device__ __forceinline__ void start() {   // per point-group prologue
        pos = 0;
        issue(0);
        issue(1);
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    }
    // begin computing chunk `pos`: returns this lane's fragment base inside the slot
    __device__ __forceinline__ const char* begin() {
        if (pos + 2 < N_PROG) issue(pos + 2);
        return ring + (pos % RING_SLOTS) * CHUNK_BYTES + lane * 16;
    }
    __device__ __forceinline__ void end() {
        // chunk pos+1 must have landed (mine), then everyone's; the barrier is also the WAR fence
        // for the slot that position pos+3 will overwrite.
        // (wait and barrier in ONE asm statement with a memory clobber: the s_barrier builtin alone
        // is not a compiler memory fence, and a ds_read of the next chunk hoisted above it would
        // read a slot other waves are still filling)
        if (pos + 2 < N_PROG) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        ++pos;
    }
};

// One layer of the k-step stream: NT output tiles; per tile NKE encoding k-steps (B = enc) then NKH
// (16, or 0 for the first layer) k-steps over the 256-feature activation `in`; three MFMA products
// per k-step.  Chunk boundaries
// (every 16 k-steps of the flat stream) are compile-time positions.  epi(T, acc) consumes a
// finished tile.
template <int NT, int NKE, int NKH = 16, int VARIANT, class EPI>
__device__ __forceinline__ void run_layer(Pipe<VARIANT>& P, const Act& in, const Enc& enc, EPI&& epi) {
    const char* frag = P.begin();
    static_for<0, NT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        f32x16 acc = f32x16{0};
        static_for<0, NKE + NKH>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int ks = t * (NKE + NKH) + j;
            if constexpr (ks % CHUNK_KSTEPS == 0 && ks != 0) {
                P.end();
                frag = P.begin();
            }
            constexpr int off = (ks % CHUNK_KSTEPS) * KSTEP_BYTES;
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(frag + off);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(frag + off + 1024);
            if constexpr (j < NKE) {
                acc = MFMA(ah, enc.hi[j], acc);
                acc = MFMA(ah, enc.lo[j], acc);
                acc = MFMA(al, enc.hi[j], acc);
            } else {
                acc = MFMA(ah, in.hi[j - NKE], acc);
                acc = MFMA(ah, in.lo[j - NKE], acc);
                acc = MFMA(al, in.hi[j - NKE], acc);
            }
        });
        epi(T, acc);
    });
    P.end();
}

template <bool RELU>
__device__ __forceinline__ f32x16 bias_act(const f32x16& acc, const float* tab) {
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 b = reinterpret_cast<const f32x4*>(tab)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = acc[4 * i + j] + b[j];
            v[4 * i + j] = RELU ? fmaxf(x, 0.0f) : x;
        }
    }
    return v;
}

// fp32 tile result -> (hi, lo) bf16 B fragments of k-steps 2T, 2T+1 of the next layer
template <int T>
__device__ __forceinline__ void split_store(const f32x16& v, Act& out) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bf16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = v[8 * s + e];
            const __bf16 hh = (__bf16)x;
            h[e] = hh;
            l[e] = (__bf16)(x - (float)hh);
        }
        out.hi[2 * T + s] = h;
        out.lo[2 * T + s] = l;
    }
}

// Keeps a running head sum materialised where it is computed: without it the optimiser defers
// whole tile epilogues to the end of the kernel (their results are only needed there) and the
// accumulators they read get spilled to scratch.
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }

__device__ __forceinline__ float dot16(const f32x16& v, const float* tab) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 w = reinterpret_cast<const f32x4*>(tab)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) s = fmaf(v[4 * i + j], w[j], s);
    }
    return s;
}

// [x, sin(2^k x), cos(2^k x)] in the slot order of layout.h::enc_ref_index (sincos_enc.h: one
// extended-precision range reduction per coordinate, exact 2^k scaling per frequency).
template <int PAIRS, int NK>
__device__ __forceinline__ void encode(float x, float y, float z, int h, Enc& enc) {
    float vals[8 * NK];
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
#pragma unroll
    for (int jj = 0; jj < NK; ++jj) {
        bf16x8 hv, lv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = vals[8 * jj + e];
            const __bf16 hh = (__bf16)v;
            hv[e] = hh;
            lv[e] = (__bf16)(v - (float)hh);
        }
        enc.hi[jj] = hv;
        enc.lo[jj] = lv;
    }
}

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void mlp_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    float* tabs = reinterpret_cast<float*>(smem + LDS_RING_BYTES);

    // side tables -> LDS once per workgroup
    for (int i = threadIdx.x; i < TAB_FLOATS / 4; i += 256)
        reinterpret_cast<f32x4*>(tabs)[i] = reinterpret_cast<const f32x4*>(a.tables)[i];
    __syncthreads();
    const float* ltab = tabs + h * 16;   // this lane-half's 16-float row inside every [2][16] entry

    Pipe<VARIANT> P;
    P.stream = a.stream;
    P.ring = smem;
    P.lane = lane;
    P.wave = wave;
    P.lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    P.voff = lane * 16 + wave * 8192;

    const long n_groups = (a.n_pts + 127) / 128;
    for (long g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const long p = g * 128 + wave * 32 + (lane & 31);
        const bool valid = p < a.n_pts;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (valid) {
            px = a.pts[3 * p + 0];
            py = a.pts[3 * p + 1];
            pz = a.pts[3 * p + 2];
        }
        Enc pe;
        encode<PE_PAIRS_PER_HALF, PE_KSTEPS>(px, py, pz, h, pe);

        P.start();

        Act A, B;
        float part[RAW_CH];

        // ---- positions_linears.0 : 63 -> 256, ReLU ---------------------------------------------
        run_layer<8, PE_KSTEPS, 0>(P, A /*unused*/, pe, [&](auto T, const f32x16& acc) {
            constexpr int t = decltype(T)::value;
            split_store<t>(bias_act<true>(acc, ltab + TAB_BIAS + (BT_L0 + t) * 32), A);
        });
        // ---- positions_linears.1..4 : 256 -> 256, ReLU -----------------------------------------
        for (int l = 1; l <= 4; ++l) {
            run_layer<8, 0>(P, A, pe, [&](auto T, const f32x16& acc) {
                constexpr int t = decltype(T)::value;
                split_store<t>(bias_act<true>(acc, ltab + TAB_BIAS + (BT_L0 + 8 * l + t) * 32), B);
            });
            A = B;
        }
        // ---- positions_linears.5 : cat([x63, h]) -> 256, ReLU (ibl_nerf.py:167-168) ------------
        run_layer<8, PE_KSTEPS>(P, A, pe, [&](auto T, const f32x16& acc) {
            constexpr int t = decltype(T)::value;
            split_store<t>(bias_act<true>(acc, ltab + TAB_BIAS + (BT_L0 + 40 + t) * 32), B);
        });
        // ---- positions_linears.6 ----------------------------------------------------------------
        run_layer<8, 0>(P, B, pe, [&](auto T, const f32x16& acc) {
            constexpr int t = decltype(T)::value;
            split_store<t>(bias_act<true>(acc, ltab + TAB_BIAS + (BT_L0 + 48 + t) * 32), A);
        });
        // ---- positions_linears.7 ; sigma_linear / roughness_linear on its fp32 activations -----
        part[0] = 0.0f;
        part[4] = 0.0f;
        run_layer<8, 0>(P, A, pe, [&](auto T, const f32x16& acc) {
            constexpr int t = decltype(T)::value;
            const f32x16 v = bias_act<true>(acc, ltab + TAB_BIAS + (BT_L0 + 56 + t) * 32);
            if constexpr (VARIANT != VAR_TRUNK) split_store<t>(v, B);
            part[0] += dot16(v, ltab + TAB_SIG + t * 32);
            pin(part[0]);
            if constexpr (VARIANT == VAR_FULL) {
                part[4] += dot16(v, ltab + TAB_ROUGH + t * 32);
                pin(part[4]);
            }
        });

        if constexpr (VARIANT != VAR_TRUNK) {
#pragma unroll
            for (int c = 1; c < RAW_CH; ++c)
                if (c != 4) part[c] = 0.0f;
            // ---- feature_linear : 256 -> 256, no activation (B = h7 -> A = feature) ------------
            run_layer<8, 0>(P, B, pe, [&](auto T, const f32x16& acc) {
                constexpr int t = decltype(T)::value;
                split_store<t>(bias_act<false>(acc, ltab + TAB_BIAS + (BT_FEAT + t) * 32), A);
            });
            if constexpr (VARIANT == VAR_FULL) {
                // ---- albedo_feature_linear (ReLU) -> albedo_linear -----------------------------
                run_layer<4, 0>(P, B, pe, [&](auto T, const f32x16& acc) {
                    constexpr int t = decltype(T)::value;
                    const f32x16 v = bias_act<true>(acc, ltab + TAB_BIAS + (BT_ALB + t) * 32);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        part[1 + c] += dot16(v, ltab + TAB_ALB + c * 128 + t * 32);
                        pin(part[1 + c]);
                    }
                });
                // ---- irradiance_feature_linear (ReLU) -> irradiance_linear ---------------------
                run_layer<4, 0>(P, B, pe, [&](auto T, const f32x16& acc) {
                    constexpr int t = decltype(T)::value;
                    const f32x16 v = bias_act<true>(acc, ltab + TAB_BIAS + (BT_IRR + t) * 32);
                    part[5] += dot16(v, ltab + TAB_IRR + t * 32);
                    pin(part[5]);
                });
            }
            // direction encoding of this point's ray (run_network expands viewdirs over the samples,
            // ibl_nerf.py:244-247)
            Enc de;
            {
                float dx = 0.f, dy = 0.f, dz = 0.f;
                if (valid) {
                    const unsigned r = (unsigned)p / (unsigned)a.pts_per_ray;   // n_pts < 2^31 per launch
                    dx = a.dirs[3 * (size_t)r + 0];
                    dy = a.dirs[3 * (size_t)r + 1];
                    dz = a.dirs[3 * (size_t)r + 2];
                }
                encode<DE_PAIRS_PER_HALF, DE_KSTEPS>(dx, dy, dz, h, de);
            }
            // ---- views_linears.0 : cat([feature, dir27]) -> 256, ReLU (A -> B) ; radiance_linear -
            run_layer<8, DE_KSTEPS>(P, A, de, [&](auto T, const f32x16& acc) {
                constexpr int t = decltype(T)::value;
                const f32x16 v = bias_act<true>(acc, ltab + TAB_BIAS + (BT_VIEW + t) * 32);
                split_store<t>(v, B);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    part[6 + c] += dot16(v, ltab + TAB_RAD + c * 256 + t * 32);
                    pin(part[6 + c]);
                }
            });
            // ---- additional_radiance_feature_linear.k (ReLU) -> additional_radiance_linear.k ----
            static_for<0, 3>([&](auto Kk) {
                constexpr int k = decltype(Kk)::value;
                run_layer<4, 0>(P, B, pe, [&](auto T, const f32x16& acc) {
                    constexpr int t = decltype(T)::value;
                    const f32x16 v = bias_act<true>(acc, ltab + TAB_BIAS + (BT_AR + 4 * k + t) * 32);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        part[9 + 3 * k + c] += dot16(v, ltab + TAB_AR + (3 * k + c) * 128 + t * 32);
                        pin(part[9 + 3 * k + c]);
                    }
                });
            });
        }

        // ---- combine the two lane halves, add head biases, store ------------------------------
        const float* sc = tabs + TAB_SCALAR;
        if constexpr (VARIANT == VAR_TRUNK) {
            const float s = part[0] + __shfl_xor(part[0], 32) + sc[0];
            if (valid && h == 0) a.out[p] = s;
        } else {
            float tot[RAW_CH];
#pragma unroll
            for (int c = 0; c < RAW_CH; ++c) tot[c] = part[c] + __shfl_xor(part[c], 32) + sc[c];
            if (valid) {
                if constexpr (VARIANT == VAR_FULL) {
                    float* o = a.out + p * RAW_CH;
                    if (h == 0) {
#pragma unroll
                        for (int c = 0; c < 9; ++c) o[c] = tot[c];
                    } else {
#pragma unroll
                        for (int c = 9; c < 18; ++c) o[c] = tot[c];
                    }
                } else {
                    float* o = a.out + p * REFL_CH;
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
        // all lanes' stores/loads retire before the next group's pipeline restarts counting
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

hipError_t launch_mlp(int variant, const MlpArgs& a, int n_cu, hipStream_t stream) {
    if (a.n_pts <= 0) return hipSuccess;
    const long n_groups = (a.n_pts + 127) / 128;
    const int grid = (int)(n_groups < n_cu ? n_groups : n_cu);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mlp_kernel<VAR_FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)mlp_kernel<VAR_TRUNK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)mlp_kernel<VAR_REFL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    switch (variant) {
        case VAR_FULL: hipLaunchKernelGGL(mlp_kernel<VAR_FULL>, dim3(grid), dim3(256), LDS_BYTES, stream, a); break;
        case VAR_TRUNK: hipLaunchKernelGGL(mlp_kernel<VAR_TRUNK>, dim3(grid), dim3(256), LDS_BYTES, stream, a); break;
        case VAR_REFL: hipLaunchKernelGGL(mlp_kernel<VAR_REFL>, dim3(grid), dim3(256), LDS_BYTES, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ibl
