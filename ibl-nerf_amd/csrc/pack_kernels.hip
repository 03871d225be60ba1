// Device-side weight packer: fp32 state-dict blob in HBM -> the two MLP weight streams + side tables, by gather
// maps built once on the host (pack.cpp: build_pack_maps).  Arithmetic identical to the host packer (pack.cpp), so
// a network uploaded from device memory renders bit-identically to one uploaded from the host.  This is what makes
// per-optimizer-step weight refresh cheap (src/train.py's test renders and the no-grad queries of a training step):
// no device->host copy, no host packing, no synchronisation.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "layout_mx.h"

namespace ibl {

namespace {

__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ int fp6_encode(float x) {
    const float a = fabsf(x);
    int c;
    if (a < 1.0f) c = (int)nearbyintf(a * 8.0f);
    else if (a < 2.0f) c = 8 + (int)nearbyintf((a - 1.0f) * 8.0f);
    else if (a < 4.0f) c = 16 + (int)nearbyintf((a - 2.0f) * 4.0f);
    else c = 24 + (int)nearbyintf((a - 4.0f) * 2.0f);
    if (c > 31) c = 31;
    return c | (x < 0.0f ? 32 : 0);
}

__device__ __forceinline__ unsigned fp6_block(const float* v, unsigned out[6]) {
    float mx = 0.0f;
    for (int j = 0; j < 32; ++j) mx = fmaxf(mx, fabsf(v[j]));
    for (int q = 0; q < 6; ++q) out[q] = 0;
    if (!(mx > 0.0f)) return 127u;
    int ex;
    (void)frexpf(mx, &ex);
    int se = ex - 1 - 2;
    se = se < -126 ? -126 : (se > 127 ? 127 : se);
    const float inv = ldexpf(1.0f, -se);
    for (int j = 0; j < 32; ++j) {
        const unsigned code = (unsigned)fp6_encode(v[j] * inv);
        const int bit = 6 * j;
        out[bit >> 5] |= code << (bit & 31);
        if ((bit & 31) > 26) out[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
    }
    return (unsigned)(se + 127);
}

// one thread per bf16x3 stream element (k-step, lane, slot); the f16x3 stream has the same layout with f16 pairs
__global__ void k_pack_bf16(const float* blob, const unsigned short* id_stream, unsigned short* stream, unsigned short* stream_f16,
                            long n_elems, unsigned* range_flag) {
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (t >= n_elems) return;
    const long ks = t >> 9, r = t & 511;                       // 512 (lane, slot) pairs per k-step
    const long at = ks * (KSTEP_BYTES / 2) + r;
    const unsigned idx = (unsigned)id_stream[at] | ((unsigned)id_stream[at + 512] << 16);
    const float w = idx ? blob[idx - 1] : 0.0f;
    const unsigned short hi = bf16_rne(w);
    const float hf = __builtin_bit_cast(float, (unsigned)hi << 16);
    stream[at] = hi;
    stream[at + 512] = bf16_rne(w - hf);
    if (stream_f16 != nullptr) {
        const _Float16 h = (_Float16)w;
        const _Float16 l = (_Float16)(w - (float)h);
        stream_f16[at] = __builtin_bit_cast(unsigned short, h);
        stream_f16[at + 512] = __builtin_bit_cast(unsigned short, l);
        if (!(fabsf(w) < 65504.0f) && range_flag) atomicOr(range_flag, 1u);
    }
}

// one thread per (MX block, lane): 32 weights -> f16 fragments, fp6 forms of the weights and of their f16 residuals, scales
__global__ void k_pack_mx(const float* blob, const int* map, char* stream, long n_lanes, unsigned* range_flag) {
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    const long blk_i = t >> 6;
    const int lane = (int)(t & 63);
    char* blk = stream + blk_i * mx::BLOCK_BYTES;
    const int* m = map + t * 32;
    if (blk_i >= (long)mx::CH_RES * mx::CHUNK_BLOCKS) {   // residual block of a trunk block (layout_mx.h): Wl = f16(w - f16 w) in the f16 area, fp6(w - f16 w - Wl) + scale in the fp6(W) area
        float res3[32];
        for (int jj = 0; jj < 32; ++jj) {
            const int idx = m[jj];
            const float x = idx ? blob[idx - 1] : 0.0f;
            const float r = x - (float)(_Float16)x;
            const _Float16 rl = (_Float16)r;
            *reinterpret_cast<_Float16*>(blk + mx::OFF_F16 + (jj >> 3) * 1024 + lane * 16 + (jj & 7) * 2) = rl;
            res3[jj] = r - (float)rl;
        }
        unsigned t6[6];
        const unsigned st = fp6_block(res3, t6);
        unsigned* wa = reinterpret_cast<unsigned*>(blk + mx::OFF_W6A + lane * 16);
        for (int q = 0; q < 4; ++q) wa[q] = t6[q];
        unsigned* wb = reinterpret_cast<unsigned*>(blk + mx::OFF_W6B + lane * 8);
        wb[0] = t6[4]; wb[1] = t6[5];
        *reinterpret_cast<unsigned*>(blk + mx::OFF_SC + lane * 4) = st;
        return;
    }
    float full[32], res[32];
    bool bad = false;
    for (int jj = 0; jj < 32; ++jj) {
        const int idx = m[jj];
        const float x = idx ? blob[idx - 1] : 0.0f;
        const _Float16 h = (_Float16)x;
        *reinterpret_cast<_Float16*>(blk + mx::OFF_F16 + (jj >> 3) * 1024 + lane * 16 + (jj & 7) * 2) = h;
        full[jj] = x;
        res[jj] = x - (float)h;
        bad |= !(fabsf(x) < 65504.0f);
    }
    unsigned c6[6], r6[6];
    const unsigned sw = fp6_block(full, c6), sr = fp6_block(res, r6);
    unsigned* wa = reinterpret_cast<unsigned*>(blk + mx::OFF_W6A + lane * 16);
    unsigned* ra = reinterpret_cast<unsigned*>(blk + mx::OFF_R6A + lane * 16);
    for (int q = 0; q < 4; ++q) { wa[q] = c6[q]; ra[q] = r6[q]; }
    unsigned* wb = reinterpret_cast<unsigned*>(blk + mx::OFF_W6B + lane * 8);
    unsigned* rb = reinterpret_cast<unsigned*>(blk + mx::OFF_R6B + lane * 8);
    wb[0] = c6[4]; wb[1] = c6[5]; rb[0] = r6[4]; rb[1] = r6[5];
    *reinterpret_cast<unsigned*>(blk + mx::OFF_SC + lane * 4) = sw | (sr << 8);
    if (bad && range_flag) atomicOr(range_flag, 1u);            // f16(W) is not finite: the renders must be redone on bf16x3
}

__global__ void k_pack_tab(const float* blob, const int* map, float* tab, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) tab[t] = map[t] ? blob[map[t] - 1] : 0.0f;
}

}  // namespace

namespace {
// views_linears.0 [256 out][283 in] <- [I | 0], feature_linear [256][256] <- I, both biases <- 0
__global__ void k_identity_embed(float* views_w, float* views_b, float* feat_w, float* feat_b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 256 * 283) views_w[i] = (i / 283 == i % 283) ? 1.0f : 0.0f;
    if (i < 256 * 256) feat_w[i] = (i / 256 == i % 256) ? 1.0f : 0.0f;
    if (i < 256) { views_b[i] = 0.0f; feat_b[i] = 0.0f; }
}
}  // namespace

hipError_t launch_identity_embed(float* d_blob, size_t views_w, size_t views_b, size_t feat_w, size_t feat_b, hipStream_t s) {
    hipLaunchKernelGGL(k_identity_embed, dim3((256 * 283 + 255) / 256), dim3(256), 0, s, d_blob + views_w, d_blob + views_b, d_blob + feat_w, d_blob + feat_b);
    return hipGetLastError();
}

hipError_t launch_pack_weights(const float* d_blob, const PackMaps& maps, char* d_stream_bf16, char* d_stream_mx, char* d_stream_f16,
                               float* d_tab, unsigned* d_range_flag, hipStream_t s) {
    const long n16 = (long)N_CHUNKS * CHUNK_KSTEPS * 512;
    hipLaunchKernelGGL(k_pack_bf16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, d_blob, maps.id_stream,
                       reinterpret_cast<unsigned short*>(d_stream_bf16), reinterpret_cast<unsigned short*>(d_stream_f16), n16, d_range_flag);
    hipLaunchKernelGGL(k_pack_tab, dim3((TAB_FLOATS + 255) / 256), dim3(256), 0, s, d_blob, maps.tab, d_tab, TAB_FLOATS);
    if (d_stream_mx != nullptr) {
        const long nl = (long)mx::N_CHUNKS * mx::CHUNK_BLOCKS * 64;
        hipLaunchKernelGGL(k_pack_mx, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, s, d_blob, maps.mx, d_stream_mx, nl, d_range_flag);
    }
    return hipGetLastError();
}

}  // namespace ibl
