// Weight-stream layout of the f16 + MX-fp6 variant of the fused MLP kernel (mlp_kernel_mx.hip); the
// bf16x3 variant's layout is layout.h.  Same network (src/nerf_models/ibl_nerf.py:154-210), same side
// tables, same 32 KiB chunks through the same 3-slot LDS ring; what changes is how one fp32 GEMM
// y = W x is split into matrix-core products.
//
// Product scheme.  Per K = 64 "block" of a 32-row output tile
//     y += f16(W) * f16(X)                       4 x v_mfma_f32_32x32x16_f16          (main term)
//        + fp6(W) * fp6(X - f16(X))              1 x v_mfma_scale_f32_32x32x64_f8f6f4  (activation residual)
//        + fp6(W - f16(W)) * fp6(f16(X))         1 x v_mfma_scale_f32_32x32x64_f8f6f4  (weight residual)
// all accumulated in the same fp32 registers.  fp6 = e2m3 with one power-of-two (e8m0) scale per 32
// consecutive k of a row / a point, which the scaled MFMA applies in hardware.  The residuals are
// 2^-12 of their operand, so 4 significant bits on them leave ~2^-16 relative error per product:
// the same parity class as the three-bf16-product scheme (scratch/prec_probe_f16f8.py: normals at the
// fp32-vs-fp32 floor, direct channels < 1e-6) at half the matrix-core cycles (6 MFMAs of 32 cycles
// per block instead of 12; scratch/mfma_mix.hip measures 1.9-2.0x on random data under the power cap).
// Range: activations and weights must stay below 65504 (f16); the kernel raises a flag otherwise and
// the host re-renders with the bf16x3 kernel (renderer.py).
//
// MFMA operand layouts (verified on the device, scratch/probe_mx.hip, probe_mx2.hip):
//   32x32x16 f16      : as the bf16 form (layout.h): lane (i = l&31, h = l>>5) holds k-slots 8h..8h+7
//   32x32x64 f8f6f4   : lane (i, h) holds k = 32h .. 32h+31 of row / column i, fp6 slot j at bits
//                       [6j, 6j+6) of 6 VGPRs; its e8m0 scale is byte `op_sel` of the lane's scale VGPR
//   result            : row (r&3) + 8*(r>>2) + 4h, column i  (shape-determined, same for both)
// Block b of a 256-feature activation = the previous layer's output tiles 2b and 2b+1.  Its f16
// k-step j (0..3), slot e  <->  accumulator register 8*(j&1) + e of tile 2b + (j>>1)    (as layout.h),
// and its fp6 slot jj = 8j + e is the same element (v_cvt_scalef32_pk32_fp6_f16 converts the block's
// own 32 packed f16 values in order), so one K permutation serves all three weight forms.
//
// One block in the stream = 8 KiB = 4 units of KSTEP_BYTES:
//   [0,    4096)  f16(W): 4 k-steps x [64 lanes x 8 f16]                      ds_read_b128 at lane*16
//   [4096, 5120)  fp6(W)        bits   0..127 of each lane's 192             ds_read_b128 at lane*16
//   [5120, 6144)  fp6(W - f16W) bits   0..127                                 ds_read_b128 at lane*16
//   [6144, 6656)  fp6(W)        bits 128..191                                 ds_read_b64  at lane*8
//   [6656, 7168)  fp6(W - f16W) bits 128..191                                 ds_read_b64  at lane*8
//   [7168, 7424)  per lane u32: byte 0 = e8m0 scale of fp6(W), byte 1 = of fp6(W - f16W)   ds_read_b32
//   [7424, 8192)  zero pad
// A chunk = 4 blocks = the 256-feature part of one 32-row tile.  Encodings are one block each (the 63
// positional slots fill it; the 27 directional slots use its first 14 k of each half), so every layer is a
// whole number of chunks and a block never straddles a chunk.
#pragma once
#include "layout.h"

namespace ibl {
namespace mx {

constexpr int BLOCK_BYTES = 8192;
constexpr int CHUNK_BLOCKS = CHUNK_BYTES / BLOCK_BYTES;   // 4
constexpr int OFF_F16 = 0, OFF_W6A = 4096, OFF_R6A = 5120, OFF_W6B = 6144, OFF_R6B = 6656, OFF_SC = 7168;

//   layer                          blocks/tile   tiles   chunks   first chunk
constexpr int CH_L0 = 0;     // positions_linears.0        1 (PE)         8       2
constexpr int CH_L1 = 2;     // positions_linears.1..4     4              8       8 each (2..33)
constexpr int CH_L5 = 34;    // positions_linears.5        1 (PE) + 4     8       10
constexpr int CH_L6 = 44;    // positions_linears.6        4              8       8
constexpr int CH_L7 = 52;    // positions_linears.7        4              8       8   (trunk-only eval ends at 60)
constexpr int CH_FEAT = 60;  // feature_linear             4              8       8
constexpr int CH_ALB = 68;   // albedo_feature_linear      4              4       4
constexpr int CH_IRR = 72;   // irradiance_feature_linear  4              4       4
constexpr int CH_VIEW = 76;  // views_linears.0            1 (DE) + 4     8       10
constexpr int CH_AR = 86;    // additional_radiance_feature_linear.{0,1,2}  4   4 each   12
constexpr int N_CHUNKS_NET = 98;   // the network's blocks
// Residual blocks for the three-f16-product layers of this kernel: residual block r belongs to network block r of the TRUNK (blocks 0 .. 239 of the
// stream, in stream order: L0 tile t = block t; L1..L4 tile t, block b = 8 + 32 (l - 1) + 4 t + b; L5 tile t = 136 + 5 t (encoding) .. + 4;
// L6, L7 = 176 + 32 (l - 6) + 4 t + b) and holds, slot for slot,
//     f16 area           f16(W - f16 W)                         = Wl, the second f16 term of the weight
//     fp6(W) area + SC   fp6(W - f16 W - Wl), scale in byte 0   = W3, what two f16 terms leave of the fp32 weight (one float32 ulp at most)
// Users: VAR_TRUNK_X (the mixed TRUNK form: positions_linears.0 and .1 as THREE f16 products Wh Xh + Wh Xl + Wl Xh; errors injected in the first layers
// dominate the density's error on a fitted network) reads residual blocks 0 .. 39, their f16 area only; VAR_TRUNK_P (the 15-slot form:
// every trunk layer as three f16 products + three block-scaled fp6 products for the terms at 2^-22 of the result: Wh X3 + Wl Xl + W3 Xh, X3 = what the
// two f16 terms leave of the activation) reads all 240 and their fp6 area too.
constexpr int CH_RES = 98;
constexpr int N_RES_BLOCKS = 240;
constexpr int N_CHUNKS = CH_RES + N_RES_BLOCKS / 4;   // 158
constexpr int N_CHUNKS_TRUNK = 60;
constexpr int N_CHUNKS_TRUNK_X = 4 + 16 + 50;         // program of VAR_TRUNK_X: L0 and L1 as (network block, residual block) pairs
constexpr int N_CHUNKS_TRUNK_P = 120;                 // program of VAR_TRUNK_P: every trunk block as such a pair (240 logical blocks, two per chunk)
constexpr long STREAM_BYTES = (long)N_CHUNKS * CHUNK_BYTES;

// e2m3 (bias 1): value of a 6-bit code, and round-to-nearest-even encoding with saturation at 7.5
__host__ __device__ inline float fp6_value(int c) {
    const int e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m * 0.125f : (1.0f + m * 0.125f) * (float)(1 << (e - 1));
    return (c & 32) ? -v : v;
}

}  // namespace mx
}  // namespace ibl
