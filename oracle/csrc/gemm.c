/* CPU ORACLE (test infrastructure, NOT the product) — the dense layer of oracle/csrc/render.c, compiled once per instruction set
 * (oracle/build_cpu.py: -DGEMM_ISA=avx512 -mavx512f, -DGEMM_ISA=avx2 -mavx2 -mfma, -DGEMM_ISA=base) and picked at run time.
 *
 *   Y[p][0..N) = act(X[p][0..K) . Wt[0..K)[0..N) + bias)         nn.Linear (ibl_nerf.py:154-210 calls it 23 times per point)
 *
 * Wt is the TRANSPOSED weight ([in][out], out contiguous), so the vector lanes run over output features and every output element is
 * one accumulator that takes its K products in order k = 0, 1, 2, ... (one fused multiply-add each where the build has FMA) and the
 * bias last: the result does not depend on the vector width, the row blocking or the thread count — avx512 and avx2 builds agree
 * bit for bit; only the base build (no FMA) rounds differently.  (The reference's own GEMM is MKL / oneDNN sgemm, whose summation
 * order is not specified: any order is inside the tolerance the golden tests allow an fp32 implementation.)
 *
 * Register tile: MR = 6 rows of X times 2 vectors of outputs = 12 accumulators, 2 loads of Wt and 6 broadcasts per k. */
#include <stddef.h>

#ifndef GEMM_VB
#define GEMM_VB 16
#endif
#define VL (GEMM_VB / 4)
#define MR 6
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(CAT(name, _), GEMM_ISA)

typedef float vf __attribute__((vector_size(GEMM_VB), aligned(4), may_alias));

/* rows: a multiple of MR is computed (the caller's buffers are padded to it); N a multiple of 2 * VL (128 and 256 are) */
void FN(ibl_cpu_linear)(const float* X, int ldx, const float* Wt, const float* bias, float* Y, int ldy, int rows, int K, int N, int relu) {
    for (int r0 = 0; r0 < rows; r0 += MR) {
        const float* x0 = X + (size_t)r0 * ldx;
        for (int n0 = 0; n0 < N; n0 += 2 * VL) {
            vf a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0}, a20 = {0}, a21 = {0}, a30 = {0}, a31 = {0}, a40 = {0}, a41 = {0}, a50 = {0}, a51 = {0};
            const float* w = Wt + n0;
            for (int k = 0; k < K; ++k, w += N) {
                const vf w0 = *(const vf*)w, w1 = *(const vf*)(w + VL);
                float x;
                x = x0[k];            a00 += w0 * x; a01 += w1 * x;
                x = x0[k + ldx];      a10 += w0 * x; a11 += w1 * x;
                x = x0[k + 2 * ldx];  a20 += w0 * x; a21 += w1 * x;
                x = x0[k + 3 * ldx];  a30 += w0 * x; a31 += w1 * x;
                x = x0[k + 4 * ldx];  a40 += w0 * x; a41 += w1 * x;
                x = x0[k + 5 * ldx];  a50 += w0 * x; a51 += w1 * x;
            }
            const vf b0 = *(const vf*)(bias + n0), b1 = *(const vf*)(bias + n0 + VL);
            vf acc[MR][2] = {{a00 + b0, a01 + b1}, {a10 + b0, a11 + b1}, {a20 + b0, a21 + b1}, {a30 + b0, a31 + b1}, {a40 + b0, a41 + b1}, {a50 + b0, a51 + b1}};
            for (int r = 0; r < MR; ++r) {
                float* y = Y + (size_t)(r0 + r) * ldy + n0;
                for (int h = 0; h < 2; ++h) {
                    vf v = acc[r][h];
                    if (relu) {
                        for (int l = 0; l < VL; ++l) v[l] = v[l] < 0.f ? 0.f : v[l];       /* a NaN stays a NaN, as in torch.relu and np.maximum */
                    }
                    *(vf*)(y + h * VL) = v;
                }
            }
        }
    }
}
