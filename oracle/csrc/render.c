/* CPU ORACLE (test infrastructure, NOT the product) — see oracle/iblnerf_cpu.h.  C restatement of the reference's forward path;
 * every function cites the reference lines it follows (paths relative to /root/reference/src).  Compiled with -ffp-contract=off:
 * each product and sum below is rounded on its own, in the order written. */
#include "../iblnerf_cpu.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------------------------------
 * the network: nerf_models/ibl_nerf.py:17-86 (registration order = blob order), :154-210 (forward_not_freezed)
 * ---------------------------------------------------------------------------------------------------------------------- */
#define NL 23
enum { L_VIEWS = 8, L_FEAT = 9, L_SIGMA = 10, L_ALB_F = 11, L_ALB = 12, L_ROUGH = 13, L_IRR_F = 14, L_IRR = 15, L_RAD = 16, L_ADD_F = 17, L_ADD = 20 };
static const int L_OUT[NL] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 128, 3, 1, 128, 1, 3, 128, 128, 128, 3, 3, 3};
static const int L_IN[NL] = {63, 256, 256, 256, 256, 319, 256, 256, 283, 256, 256, 256, 128, 256, 256, 128, 256, 256, 256, 256, 128, 128, 128};
#define N_PARAMS 798994
#define PB 192              /* points per dense-layer block: a multiple of the 6-row register tile; the activations of a block stay in L2 */
#define LD_E 320            /* [x63 | h256] of the skip connection (:168), padded */
#define LD_V 288            /* [feature256 | e_dirs27] (:194-197), padded */

typedef void (*linear_fn)(const float*, int, const float*, const float*, float*, int, int, int, int, int);
void ibl_cpu_linear_avx512(const float*, int, const float*, const float*, float*, int, int, int, int, int);
void ibl_cpu_linear_avx2(const float*, int, const float*, const float*, float*, int, int, int, int, int);
void ibl_cpu_linear_base(const float*, int, const float*, const float*, float*, int, int, int, int, int);

static linear_fn g_linear;
static const char* g_isa = "";
static void pick_isa(void) {
    if (g_linear) return;
    __builtin_cpu_init();
    const char* force = getenv("IBLNERF_CPU_ISA");     /* tests: run another build on the same machine */
    int want512 = __builtin_cpu_supports("avx512f"), want2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (force && !strcmp(force, "avx2")) want512 = 0;
    if (force && !strcmp(force, "base")) want512 = want2 = 0;
    if (want512) { g_linear = ibl_cpu_linear_avx512; g_isa = "avx512"; }
    else if (want2) { g_linear = ibl_cpu_linear_avx2; g_isa = "avx2"; }
    else { g_linear = ibl_cpu_linear_base; g_isa = "base"; }
}
const char* iblnerf_cpu_isa(void) { pick_isa(); return g_isa; }

static char g_error[512];
const char* iblnerf_cpu_last_error(void) { return g_error; }
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

typedef struct {
    const float* W[NL];     /* [out][in] rows of the blob */
    const float* B[NL];
    float* Wt[NL];          /* [in][out] copies of the wide layers (out >= 128), 64-byte aligned */
    int color_independent;
} Net;

static void* xalloc(size_t bytes) {
    void* p = NULL;
    if (posix_memalign(&p, 64, (bytes + 63) & ~(size_t)63)) return NULL;
    return p;
}

static int net_init(Net* n, const float* blob, int color_independent) {
    memset(n, 0, sizeof *n);
    n->color_independent = color_independent;
    const float* p = blob;
    for (int l = 0; l < NL; ++l) {
        n->W[l] = p; p += (size_t)L_OUT[l] * L_IN[l];
        n->B[l] = p; p += L_OUT[l];
        if (L_OUT[l] >= 128) {
            n->Wt[l] = (float*)xalloc(sizeof(float) * L_OUT[l] * L_IN[l]);
            if (!n->Wt[l]) return -1;
            for (int o = 0; o < L_OUT[l]; ++o)
                for (int i = 0; i < L_IN[l]; ++i) n->Wt[l][(size_t)i * L_OUT[l] + o] = n->W[l][(size_t)o * L_IN[l] + i];
        }
    }
    return (p - blob) == N_PARAMS ? 0 : -1;
}
static void net_free(Net* n) { for (int l = 0; l < NL; ++l) free(n->Wt[l]); }

typedef struct { float *E, *H0, *H1, *V, *F; } Scratch;
static int scratch_init(Scratch* s) {
    s->E = (float*)xalloc(sizeof(float) * PB * LD_E);
    s->H0 = (float*)xalloc(sizeof(float) * PB * 256);
    s->H1 = (float*)xalloc(sizeof(float) * PB * 256);
    s->V = (float*)xalloc(sizeof(float) * PB * LD_V);
    s->F = (float*)xalloc(sizeof(float) * PB * 128);
    return (s->E && s->H0 && s->H1 && s->V && s->F) ? 0 : -1;
}
static void scratch_free(Scratch* s) { free(s->E); free(s->H0); free(s->H1); free(s->V); free(s->F); }

/* nerf_models/positional_embedder.py:4-52: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]; 2.**linspace(0, L-1, L) are exact powers of two */
static void embed(const float* x, int n_freq, float* out) {
    out[0] = x[0]; out[1] = x[1]; out[2] = x[2];
    float f = 1.f;
    for (int k = 0; k < n_freq; ++k, f *= 2.f) {
        float* o = out + 3 + 6 * k;
        for (int c = 0; c < 3; ++c) {
            const float a = x[c] * f;
            o[c] = sinf(a);
            o[3 + c] = cosf(a);
        }
    }
}

/* the N = 1 / 3 heads: one row of W against one activation vector, products exact in double, one rounding of the sum */
static float head(const Net* n, int l, int row, const float* h) {
    const float* w = n->W[l] + (size_t)row * L_IN[l];
    double acc = 0.0;
    for (int k = 0; k < L_IN[l]; ++k) acc += (double)w[k] * (double)h[k];
    return (float)acc + n->B[l][row];
}

static void linear(const Net* n, int l, const float* X, int ldx, float* Y, int ldy, int rows, int relu) {
    g_linear(X, ldx, n->Wt[l], n->B[l], Y, ldy, rows, L_IN[l], L_OUT[l], relu);
}

/* ibl_nerf.py:154-210 over a list of points.  pts [P][3]; dirs: NULL = the density alone (:175-176), else row (p / pts_per_ray) of dirs
 * [.][3] is point p's direction.  out: [P][18] (sigma, albedo 3, roughness, irradiance, radiance 3, three coarse radiances 3 each) or [P]. */
static void mlp_eval(const Net* n, Scratch* s, const float* pts, long P_total, const float* dirs, int pts_per_ray, float* out) {
    for (long p0 = 0; p0 < P_total; p0 += PB) {
        const int P = (int)((P_total - p0) < PB ? (P_total - p0) : PB);
        const int rows = (P + 5) / 6 * 6;
        for (int p = 0; p < P; ++p) embed(pts + 3 * (p0 + p), 10, s->E + (size_t)p * LD_E);
        for (int p = P; p < rows; ++p) memset(s->E + (size_t)p * LD_E, 0, sizeof(float) * LD_E);
        linear(n, 0, s->E, LD_E, s->H0, 256, rows, 1);
        linear(n, 1, s->H0, 256, s->H1, 256, rows, 1);
        linear(n, 2, s->H1, 256, s->H0, 256, rows, 1);
        linear(n, 3, s->H0, 256, s->H1, 256, rows, 1);
        linear(n, 4, s->H1, 256, s->E + 63, LD_E, rows, 1);       /* :168 h = cat([input_pts, h]) */
        linear(n, 5, s->E, LD_E, s->H0, 256, rows, 1);
        linear(n, 6, s->H0, 256, s->H1, 256, rows, 1);
        linear(n, 7, s->H1, 256, s->H0, 256, rows, 1);
        const float* h = s->H0;
        if (!dirs) {
            for (int p = 0; p < P; ++p) out[p0 + p] = head(n, L_SIGMA, 0, h + (size_t)p * 256);
            continue;
        }
        float* o = out + 18 * p0;
        for (int p = 0; p < P; ++p) {
            o[18 * p + 0] = head(n, L_SIGMA, 0, h + (size_t)p * 256);
            o[18 * p + 4] = head(n, L_ROUGH, 0, h + (size_t)p * 256);
        }
        linear(n, L_ALB_F, h, 256, s->F, 128, rows, 1);
        for (int p = 0; p < P; ++p)
            for (int c = 0; c < 3; ++c) o[18 * p + 1 + c] = head(n, L_ALB, c, s->F + (size_t)p * 128);
        linear(n, L_IRR_F, h, 256, s->F, 128, rows, 1);
        for (int p = 0; p < P; ++p) o[18 * p + 5] = head(n, L_IRR, 0, s->F + (size_t)p * 128);
        const float* h2 = h;                                        /* :192 is_color_independent_to_direction: the radiance heads read h */
        if (!n->color_independent) {
            linear(n, L_FEAT, h, 256, s->V, LD_V, rows, 0);        /* :193 no activation */
            long last_ray = -1;
            for (int p = 0; p < rows; ++p) {
                float* e = s->V + (size_t)p * LD_V + 256;
                const long ray = p < P ? (p0 + p) / pts_per_ray : -2;
                if (p >= P) memset(e, 0, sizeof(float) * 27);
                else if (ray == last_ray) memcpy(e, e - LD_V, sizeof(float) * 27);
                else embed(dirs + 3 * ray, 4, e);
                last_ray = ray;
            }
            linear(n, L_VIEWS, s->V, LD_V, s->H1, 256, rows, 1);   /* :194-197 */
            h2 = s->H1;
        }
        for (int p = 0; p < P; ++p)
            for (int c = 0; c < 3; ++c) o[18 * p + 6 + c] = head(n, L_RAD, c, h2 + (size_t)p * 256);
        for (int k = 0; k < 3; ++k) {                               /* :202-206 */
            linear(n, L_ADD_F + k, h2, 256, s->F, 128, rows, 1);
            for (int p = 0; p < P; ++p)
                for (int c = 0; c < 3; ++c) o[18 * p + 9 + 3 * k + c] = head(n, L_ADD + k, c, s->F + (size_t)p * 128);
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * small pieces
 * ---------------------------------------------------------------------------------------------------------------------- */
/* float32 torch.linspace as ATen's CPU kernel fills it: from both ends, each element ONE fused multiply-add (emulated in double,
 * which holds the 24-bit x small-int product and the sum exactly before the single rounding) */
static void torch_linspace(float start, float end, int steps, float* out) {
    if (steps == 1) { out[0] = start; return; }
    const double step = (double)((end - start) / (float)(steps - 1));
    for (int i = 0; i < steps; ++i)
        out[i] = i < steps / 2 ? (float)((double)start + step * i) : (float)((double)end - step * (steps - 1 - i));
}

/* torch.sum(x, -1) of one contiguous float32 row, n < 512: ATen's vectorized inner sum for 8-float vectors — 4 interleaved partial
 * vectors over the first 4 * (n / 32) vectors, later vectors into partial 0, ((p0 + p1) + p2) + p3, the scalar tail summed from 0,
 * then the 8 lanes added to it in order (aten/src/ATen/native/cpu/SumKernel.cpp; pinned against torch.sum by the golden tests) */
static float aten_row_sum(const float* x, int n) {
    const int vs = n / 8, g = vs / 4;
    float p[4][8];
    memset(p, 0, sizeof p);
    for (int i = 0; i < g; ++i)
        for (int j = 0; j < 4; ++j)
            for (int c = 0; c < 8; ++c) p[j][c] = p[j][c] + x[8 * (4 * i + j) + c];
    for (int j = 4 * g; j < vs; ++j)
        for (int c = 0; c < 8; ++c) p[0][c] = p[0][c] + x[8 * j + c];
    float fin = 0.f;
    for (int t = 8 * vs; t < n; ++t) fin = fin + x[t];
    for (int c = 0; c < 8; ++c) fin = fin + (((p[0][c] + p[1][c]) + p[2][c]) + p[3][c]);
    return fin;
}

static float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
static float relu1(float x) { return x < 0.f ? 0.f : x; }
static float clip01(float x) { return x != x ? x : (x < 0.f ? 0.f : (x > 1.f ? 1.f : x)); }
static void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static float norm3(const float* v) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
/* torch.nn.functional.normalize(dim=-1, eps=1e-12) */
static void normalize3(const float* v, float* o) {
    float n = norm3(v);
    if (!(n > 1e-12f) && n == n) n = 1e-12f;
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}

/* ibl_nerf_renderer.py:203-206: dists = cat(z[1:] - z[:-1], 1e10) * |d| */
static void ray_dists(const float* z, int S, const float* d, float* dists) {
    const float nrm = norm3(d);
    for (int s = 0; s + 1 < S; ++s) dists[s] = (z[s + 1] - z[s]) * nrm;
    dists[S - 1] = 1e10f * nrm;
}

/* :241-245 (and :44-52, normal_from_depth.py:160-170): alpha = 1 - exp(-relu(sigma) dist); weights = alpha * cumprod(1 - alpha + 1e-10)
 * shifted by one — ATen's CPU cumprod accumulates float tensors in double and rounds each prefix to float */
static void alpha_weights(const float* sigma, int stride, const float* dists, int S, float* w) {
    double T = 1.0;
    for (int s = 0; s < S; ++s) {
        const float alpha = 1.f - expf(-relu1(sigma[(size_t)s * stride]) * dists[s]);
        w[s] = alpha * (float)T;
        T *= (double)((1.f - alpha) + 1e-10f);
    }
}

/* nerf_models/nerf_renderer_helper.py:91-134, det=True: bins [nb], weights [nb - 1] -> ns samples.  The `denom < 1e-5` test (:128-129)
 * sits one float32 ulp from an empty bin's cdf step, so the row sum takes torch.sum's own order */
static void sample_pdf_row(const float* bins, const float* weights, int nb, int ns, const float* u, float* out, float* tmp /* 2 * nb */) {
    float* w = tmp;
    float* cdf = tmp + nb;
    const int nw = nb - 1;
    for (int i = 0; i < nw; ++i) w[i] = weights[i] + 1e-5f;
    const float sum = aten_row_sum(w, nw);
    double acc = 0.0;                                   /* torch.cumsum: double accumulate, each prefix rounded */
    cdf[0] = 0.f;
    for (int i = 0; i < nw; ++i) {
        acc += (double)(w[i] / sum);
        cdf[i + 1] = (float)acc;
    }
    for (int j = 0; j < ns; ++j) {
        int lo = 0, hi = nb;                            /* torch.searchsorted(cdf, u, right=True): first index with cdf > u */
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (cdf[mid] > u[j]) hi = mid; else lo = mid + 1;     /* a NaN u compares false: lands at nb, as torch */
        }
        const int below = lo - 1 > 0 ? lo - 1 : 0, above = lo < nb - 1 ? lo : nb - 1;
        float den = cdf[above] - cdf[below];
        if (den < 1e-5f) den = 1.f;
        const float t = (u[j] - cdf[below]) / den;
        out[j] = bins[below] + t * (bins[above] - bins[below]);
    }
}

/* F.grid_sample(bilinear, zeros padding, align_corners=True) of the [3,512,512] LUT at (n.v, roughness): ibl_nerf_renderer.py:418-421 */
static void lut_fetch(const float* lut, float ndv, float rough, float* env) {
    const int Hh = 512, Ww = 512;
    const float gx = 2.f * ndv - 1.f, gy = 2.f * rough - 1.f;
    const float x = ((gx + 1.f) / 2.f) * (float)(Ww - 1), y = ((gy + 1.f) / 2.f) * (float)(Hh - 1);
    const float x0 = floorf(x), y0 = floorf(y);
    env[0] = env[1] = env[2] = 0.f;
    for (int k = 0; k < 4; ++k) {
        const int dx = k & 1, dy = k >> 1;
        const float xi = x0 + dx, yi = y0 + dy;
        const float wx = dx ? (x - x0) : (x0 + 1.f - x), wy = dy ? (y - y0) : (y0 + 1.f - y);
        if (!(xi >= 0 && xi <= Ww - 1 && yi >= 0 && yi <= Hh - 1)) continue;
        const float wgt = wx * wy;
        const size_t at = (size_t)(int)yi * Ww + (int)xi;
        for (int c = 0; c < 3; ++c) env[c] = env[c] + lut[(size_t)c * Hh * Ww + at] * wgt;
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * one pass over a block of rays: raw2outputs, ibl_nerf_renderer.py:153-527 (approximate_radiance=True)
 * ---------------------------------------------------------------------------------------------------------------------- */
typedef struct {
    const iblnerf_options* opt;
    const float *lut, *rays_o, *rays_d;
    const iblnerf_overrides* ov;
    float near_, far_;
} Job;

typedef struct {
    Scratch sc;
    float *pts, *opts, *raw, *sig4, *w, *dists, *rpts, *rraw, *tmp;
} Work;

static int work_init(Work* k, int RB, int Smax, int Sc) {
    memset(k, 0, sizeof *k);
    if (scratch_init(&k->sc)) return -1;
    const size_t P = (size_t)RB * Smax;
    k->pts = (float*)xalloc(sizeof(float) * P * 3);
    k->opts = (float*)xalloc(sizeof(float) * P * 3);
    k->raw = (float*)xalloc(sizeof(float) * P * 18);
    k->sig4 = (float*)xalloc(sizeof(float) * P * 4);
    k->w = (float*)xalloc(sizeof(float) * P);
    k->dists = (float*)xalloc(sizeof(float) * P);
    k->rpts = (float*)xalloc(sizeof(float) * (size_t)RB * Sc * 3);
    k->rraw = (float*)xalloc(sizeof(float) * (size_t)RB * Sc * 18);
    k->tmp = (float*)xalloc(sizeof(float) * (4 * (size_t)Smax + 64));
    return (k->pts && k->opts && k->raw && k->sig4 && k->w && k->dists && k->rpts && k->rraw && k->tmp) ? 0 : -1;
}
static void work_free(Work* k) {
    scratch_free(&k->sc);
    free(k->pts); free(k->opts); free(k->raw); free(k->sig4); free(k->w); free(k->dists); free(k->rpts); free(k->rraw); free(k->tmp);
}

static float srgb(const Job* j, float x) {              /* output_f (:487): tonemap (:30-31) then rgb_to_srgb (:26-27) */
    if (j->opt->use_radiance_linear) x = x / (x + 1.f);
    return j->opt->gamma_correct ? powf(x + 1e-12f, (float)(1.0 / 2.2)) : x;
}
static float srgb_only(const Job* j, float x) { return j->opt->gamma_correct ? powf(x + 1e-12f, (float)(1.0 / 2.2)) : x; }
static float radiance_f(const Job* j, float x) { return j->opt->use_radiance_linear ? relu1(x) : sigmoidf(x); }   /* :192-197 */

/* object q <=> 9 (q + 1) / 255 < m < 11 (q + 1) / 255 (:223-228 / :233-238) */
static int in_mask(float m, int q) { return (float)(11 * (q + 1) / 255.) > m && m > (float)(9 * (q + 1) / 255.); }

#define PUT3(ptr, ray, v) do { if (ptr) { (ptr)[3 * (ray)] = (v)[0]; (ptr)[3 * (ray) + 1] = (v)[1]; (ptr)[3 * (ray) + 2] = (v)[2]; } } while (0)
#define PUT1(ptr, ray, v) do { if (ptr) (ptr)[ray] = (v); } while (0)

/* z [nr][S]; zc [nr][Sc] (z_vals_constant: the reflected ray's samples, :440); weights_out [nr][S] (may be NULL); maps NULL = only
 * the density of the main query (a coarse pass that only places the fine samples) */
static void pass(const Job* j, const Net* net, Work* k, long r0, int nr, const float* z, int S, const float* zc, int Sc,
                 const iblnerf_maps* m, float* weights_out) {
    const iblnerf_options* opt = j->opt;
    const float* RO = j->rays_o + 3 * r0;
    const float* RD = j->rays_d + 3 * r0;
    for (int i = 0; i < nr; ++i)
        for (int s = 0; s < S; ++s)
            for (int c = 0; c < 3; ++c) k->pts[3 * ((size_t)i * S + s) + c] = RO[3 * i + c] + RD[3 * i + c] * z[(size_t)i * S + s];     /* :200 */
    const int density_only = m == NULL;
    if (density_only) mlp_eval(net, &k->sc, k->pts, (long)nr * S, NULL, S, k->raw);
    else mlp_eval(net, &k->sc, k->pts, (long)nr * S, RD, S, k->raw);                                  /* :201 (rays_d, not viewdirs) */
    const int rs = density_only ? 1 : 18;
    for (int i = 0; i < nr; ++i) {
        ray_dists(z + (size_t)i * S, S, RD + 3 * i, k->dists + (size_t)i * S);
        alpha_weights(k->raw + (size_t)i * S * rs, rs, k->dists + (size_t)i * S, S, k->w + (size_t)i * S);
        if (weights_out) memcpy(weights_out + (size_t)i * S, k->w + (size_t)i * S, sizeof(float) * S);
    }
    if (density_only) return;

    /* epsilon normal: normal_from_depth.py:139-183 — four density queries at pts +- eps right, +- eps up */
    const float eps = opt->epsilon;
    const int eps_normal = opt->normal_mode == IBLNERF_NORMAL_DEPTH_GRADIENT_EPSILON;
    if (eps_normal) {
        for (int v = 0; v < 4; ++v) {
            for (int i = 0; i < nr; ++i) {
                const float up0[3] = {0.f, 1.f, 0.f};
                float right[3], up[3], off[3];
                cross3(RD + 3 * i, up0, right);
                cross3(right, RD + 3 * i, up);
                const float* dir = v < 2 ? right : up;
                for (int c = 0; c < 3; ++c) off[c] = (v & 1) ? -(eps * dir[c]) : eps * dir[c];                       /* :151-154 */
                for (int s = 0; s < S; ++s)
                    for (int c = 0; c < 3; ++c) k->opts[3 * ((size_t)i * S + s) + c] = k->pts[3 * ((size_t)i * S + s) + c] + off[c];
            }
            mlp_eval(net, &k->sc, k->opts, (long)nr * S, NULL, S, k->sig4 + (size_t)v * nr * S);
        }
    }

    const float depth_0 = (j->far_ + j->near_) * 0.5f;                                                             /* :456 */
    const iblnerf_overrides* ov = j->ov;
    const int mode = ov ? ov->mode : 0;
    float* rowbuf = k->tmp;                                                                                         /* [S] */
    float* wtmp = k->tmp + S;                                                                                       /* [S] */
    /* first the per-ray quantities up to the reflected direction, then ONE reflected-ray query for the block, then the rest */
    float (*keep)[24] = (float (*)[24])malloc(sizeof(float[24]) * nr);
    for (int i = 0; i < nr; ++i) {
        const long ray = r0 + i;
        const float* zz = z + (size_t)i * S;
        const float* w = k->w + (size_t)i * S;
        const float* raw = k->raw + (size_t)i * S * 18;
        const float* d = RD + 3 * i;
        const float* o = RO + 3 * i;
        for (int s = 0; s < S; ++s) rowbuf[s] = w[s] * zz[s];
        float depth = aten_row_sum(rowbuf, S);                                                                      /* :249 */
        const float acc = aten_row_sum(w, S);
        float m_val = 0.f;
        int mask_all = 0;
        if (mode) { m_val = ov->d_mask[3 * ray]; mask_all = m_val > 0.f; }
        if (mode == 1 && ov->edit_depth && mask_all) depth = ov->d_depth[ray];                                      /* :253-254 (target_depth_map aliases depth_map) */
        if (mode == 2 && mask_all) depth = ov->d_depth[ray];                                                        /* :255-256 */
        const float ratio = depth / acc;
        const float disp = 1.f / (ratio != ratio ? ratio : (ratio > 1e-10f ? ratio : 1e-10f));                      /* :258 */
        float xs[3];
        for (int c = 0; c < 3; ++c) xs[c] = o[c] + d[c] * depth;                                                    /* :262 */
        double a3[3] = {0, 0, 0}, r4[4][3] = {{0}};
        for (int s = 0; s < S; ++s) {
            for (int c = 0; c < 3; ++c) a3[c] += (double)(w[s] * sigmoidf(raw[18 * s + 1 + c]));                    /* :281-282 */
            for (int q = 0; q < 4; ++q)
                for (int c = 0; c < 3; ++c) r4[q][c] += (double)(w[s] * radiance_f(j, raw[18 * s + 6 + 3 * q + c]));
        }
        float albedo[3] = {(float)a3[0], (float)a3[1], (float)a3[2]};
        for (int s = 0; s < S; ++s) rowbuf[s] = w[s] * sigmoidf(raw[18 * s + 4]);
        float rough = aten_row_sum(rowbuf, S);                                                                      /* :284-285 */
        for (int s = 0; s < S; ++s) rowbuf[s] = w[s] * radiance_f(j, raw[18 * s + 5]);
        float irr = aten_row_sum(rowbuf, S);                                                                        /* :287-288 */

        float normal[3];
        if (eps_normal) {
            const float up0[3] = {0.f, 1.f, 0.f};
            float right[3], up[3], D[4], dx[3], dy[3], cr[3];
            cross3(d, up0, right);
            cross3(right, d, up);
            for (int v = 0; v < 4; ++v) {
                alpha_weights(k->sig4 + ((size_t)v * nr + i) * S, 1, k->dists + (size_t)i * S, S, wtmp);
                for (int s = 0; s < S; ++s) wtmp[s] = wtmp[s] * zz[s];
                D[v] = aten_row_sum(wtmp, S);
            }
            const float e2 = 2.f * eps;
            for (int c = 0; c < 3; ++c) {                                                                           /* :172-173 */
                dx[c] = e2 * right[c] + (D[0] - D[1]) * d[c];
                dy[c] = e2 * up[c] + (D[2] - D[3]) * d[c];
            }
            cross3(dx, dy, cr);
            normalize3(cr, normal);
        } else {                                                                                                    /* ground_truth, :370-371 */
            float g[3];
            for (int c = 0; c < 3; ++c) g[c] = 2.f * ov->d_gt_normal[3 * ray + c] - 1.f;
            normalize3(g, normal);
        }
        if (mode == 1) {                                                                                            /* :378-398 */
            if (ov->edit_normal && mask_all) {
                float g[3];
                for (int c = 0; c < 3; ++c) g[c] = 2.f * ov->d_normal[3 * ray + c] - 1.f;
                normalize3(g, normal);
            }
            if (ov->edit_albedo) {
                if (ov->edit_albedo_by_img) { if (mask_all) for (int c = 0; c < 3; ++c) albedo[c] = ov->d_albedo[3 * ray + c]; }
                else for (int q = 0; q < ov->num_objects; ++q) if (in_mask(m_val, q)) for (int c = 0; c < 3; ++c) albedo[c] = ov->albedo_list[3 * q + c];
            }
            if (ov->edit_roughness) {
                if (ov->edit_roughness_by_img) { if (mask_all) rough = ov->d_roughness[ray]; }                          /* :394-395, resolved per chunk by the caller (iblnerf.h) */
                else for (int q = 0; q < ov->n_roughness_list; ++q) if (in_mask(m_val, q)) rough = ov->roughness_list[q];
            }
        } else if (mode == 2) {                                                                                     /* :400-410 */
            if (mask_all) {
                float g[3];
                for (int c = 0; c < 3; ++c) g[c] = 2.f * ov->d_normal[3 * ray + c] - 1.f;
                normalize3(g, normal);
            }
            for (int q = 0; q < ov->num_objects; ++q) if (in_mask(m_val, q)) {
                rough = ov->roughness_list[q];
                if (ov->irradiance_list[q] > 0.f) irr = ov->irradiance_list[q];
                for (int c = 0; c < 3; ++c) albedo[c] = ov->albedo_list[3 * q + c];
            }
        }
        const float ndv = clip01(((-d[0]) * normal[0] + (-d[1]) * normal[1]) + (-d[2]) * normal[2]);                /* :412-413 */
        const float ndot = (normal[0] * d[0] + normal[1] * d[1]) + normal[2] * d[2];
        float refl[3];
        for (int c = 0; c < 3; ++c) refl[c] = d[c] - (2.f * ndot) * normal[c];                                      /* :440 reflected direction */
        for (int s = 0; s < Sc; ++s)
            for (int c = 0; c < 3; ++c) k->rpts[3 * ((size_t)i * Sc + s) + c] = xs[c] + refl[c] * zc[(size_t)i * Sc + s];
        float* kp = keep[i];
        kp[0] = depth; kp[1] = acc; kp[2] = disp; kp[3] = rough; kp[4] = irr; kp[5] = ndv;
        for (int c = 0; c < 3; ++c) { kp[6 + c] = albedo[c]; kp[9 + c] = normal[c]; kp[12 + c] = refl[c]; }
        for (int q = 0; q < 3; ++q) kp[15 + q] = 0.f;
        if (m) {
            PUT3(m->radiance_map, ray, ((float[3]){srgb(j, (float)r4[0][0]), srgb(j, (float)r4[0][1]), srgb(j, (float)r4[0][2])}));
            for (int q = 0; q < 3; ++q)
                PUT3(m->radiance_map_k[q], ray, ((float[3]){srgb(j, (float)r4[q + 1][0]), srgb(j, (float)r4[q + 1][1]), srgb(j, (float)r4[q + 1][2])}));
            if (m->weights) memcpy(m->weights + (size_t)ray * S, w, sizeof(float) * S);
        }
    }

    /* the reflected rays: :442-448, raw2outputs_simple :38-68 — the network is queried with the reflected direction as view direction */
    float* refl_dirs = (float*)calloc((size_t)3 * (nr > 0 ? nr : 1), sizeof(float));
    for (int i = 0; i < nr; ++i) for (int c = 0; c < 3; ++c) refl_dirs[3 * i + c] = keep[i][12 + c];
    mlp_eval(net, &k->sc, k->rpts, (long)nr * Sc, refl_dirs, Sc, k->rraw);
    free(refl_dirs);

    for (int i = 0; i < nr; ++i) {
        const long ray = r0 + i;
        const float* kp = keep[i];
        const float depth = kp[0], acc = kp[1], disp = kp[2], rough = kp[3], irr = kp[4], ndv = kp[5];
        const float *albedo = kp + 6, *normal = kp + 9, *refl = kp + 12;
        const float* rraw = k->rraw + (size_t)i * Sc * 18;
        float* rd_ = k->tmp;
        float* rw = k->tmp + Sc;
        ray_dists(zc + (size_t)i * Sc, Sc, refl, rd_);
        alpha_weights(rraw, 18, rd_, Sc, rw);
        double pm[4][3] = {{0}};
        for (int s = 0; s < Sc; ++s)
            for (int q = 0; q < 4; ++q)
                for (int c = 0; c < 3; ++c) pm[q][c] += (double)(rw[s] * radiance_f(j, rraw[18 * s + 6 + 3 * q + c]));
        float pref_maps[4][3];
        for (int q = 0; q < 4; ++q) for (int c = 0; c < 3; ++c) pref_maps[q][c] = (float)pm[q][c];

        float env[3];
        lut_fetch(j->lut, ndv, rough, env);                                                                         /* :418-421 */
        const float metal = 1.f - rough;                                                                            /* :424 */
        float F0[3], fres[3], spec_coef[3], pref[3], diffuse[3], specular[3], color[3];
        const float p5 = powf(clip01(1.f - ndv), 5.f);
        for (int c = 0; c < 3; ++c) {
            F0[c] = 0.04f * (1.f - metal) + albedo[c] * metal;                                                      /* :425-427 */
            const float mx = (1.f - rough) > F0[c] ? (1.f - rough) : F0[c];                                         /* microfacet.py:8-12 */
            fres[c] = F0[c] + (mx - F0[c]) * p5;
            spec_coef[c] = (opt->lut_coefficient_f0 ? F0[c] : fres[c]) * env[0] + env[1];                           /* :433-436 */
        }
        float level = opt->correct_depth_for_prefiltered_radiance ? clip01(rough * depth / depth_0) : rough;        /* :455-461 */
        int i1 = level != level ? 0 : (int)(level * 3.f);
        i1 = i1 < 0 ? 0 : (i1 > 3 ? 3 : i1);                                                                        /* :464-465 */
        const int i2 = i1 + 1 > 3 ? 3 : i1 + 1;
        const float rem = level * 3.f - (float)i1;
        for (int c = 0; c < 3; ++c) {
            pref[c] = (1.f - rem) * pref_maps[i1][c] + rem * pref_maps[i2][c];                                      /* :468-470 */
            diffuse[c] = (1.f - fres[c]) * (1.f - metal) * albedo[c] * irr;                                         /* :472 */
            specular[c] = spec_coef[c] * pref[c];
            color[c] = diffuse[c] + specular[c];
        }
        if (!m) continue;
        PUT3(m->color_map, ray, ((float[3]){srgb(j, color[0]), srgb(j, color[1]), srgb(j, color[2])}));
        for (int q = 0; q < 3; ++q)
            PUT3(m->reflected_coarse_radiance_map_k[q], ray, ((float[3]){srgb(j, pref_maps[q + 1][0]), srgb(j, pref_maps[q + 1][1]), srgb(j, pref_maps[q + 1][2])}));
        PUT1(m->irradiance_map, ray, srgb(j, irr));
        PUT3(m->reflected_radiance_map, ray, ((float[3]){srgb(j, pref_maps[0][0]), srgb(j, pref_maps[0][1]), srgb(j, pref_maps[0][2])}));
        PUT3(m->prefiltered_reflected_map, ray, ((float[3]){srgb(j, pref[0]), srgb(j, pref[1]), srgb(j, pref[2])}));
        PUT3(m->albedo_map, ray, ((float[3]){srgb_only(j, albedo[0]), srgb_only(j, albedo[1]), srgb_only(j, albedo[2])}));
        PUT1(m->roughness_map, ray, rough);
        PUT3(m->specular_map, ray, ((float[3]){srgb(j, specular[0]), srgb(j, specular[1]), srgb(j, specular[2])}));
        PUT3(m->diffuse_map, ray, ((float[3]){srgb(j, diffuse[0]), srgb(j, diffuse[1]), srgb(j, diffuse[2])}));
        PUT1(m->n_dot_v_map, ray, ndv);
        PUT3(m->target_normal_map, ray, normal);
        PUT1(m->disp_map, ray, disp);
        PUT1(m->acc_map, ray, acc);
        PUT1(m->depth_map, ray, depth);
        PUT1(m->target_depth_map, ray, depth);
    }
    free(keep);
}

/* render_rays, ibl_nerf_renderer.py:629-732 (perturb = 0) for rays r0 .. r0 + nr */
static void render_block(const Job* j, const Net* nc, const Net* nf, Work* k, long r0, int nr, const iblnerf_outputs* out,
                         float* zc, float* zf, float* wc, float* zs) {
    const iblnerf_options* opt = j->opt;
    const int Sc = opt->n_samples, Ni = opt->n_importance, Sf = Sc + Ni;
    float tl[1024];
    torch_linspace(0.f, 1.f, Sc, tl);
    for (int s = 0; s < Sc; ++s)
        zc[s] = opt->lindisp ? 1.f / (1.f / j->near_ * (1.f - tl[s]) + 1.f / j->far_ * tl[s]) : j->near_ * (1.f - tl[s]) + j->far_ * tl[s];    /* :672-674 */
    for (int i = 1; i < nr; ++i) memcpy(zc + (size_t)i * Sc, zc, sizeof(float) * Sc);
    if (Ni <= 0) { pass(j, nc, k, r0, nr, zc, Sc, zc, Sc, &out->fine, NULL); return; }
    pass(j, nc, k, r0, nr, zc, Sc, zc, Sc, opt->coarse_outputs ? &out->coarse : NULL, wc);
    float u[1024], mids[1024], tmp[2048];
    torch_linspace(0.f, 1.f, Ni, u);
    for (int i = 0; i < nr; ++i) {
        const float* z = zc + (size_t)i * Sc;
        for (int s = 0; s + 1 < Sc; ++s) mids[s] = 0.5f * (z[s + 1] + z[s]);                                        /* :701 */
        sample_pdf_row(mids, wc + (size_t)i * Sc + 1, Sc - 1, Ni, u, zs + (size_t)i * Ni, tmp);                     /* :702-703 weights[..., 1:-1] */
        float* f = zf + (size_t)i * Sf;                                                                             /* :707 sort(cat(z_vals, z_samples)) */
        memcpy(f, z, sizeof(float) * Sc);
        memcpy(f + Sc, zs + (size_t)i * Ni, sizeof(float) * Ni);
        for (int a = 1; a < Sf; ++a) {
            const float v = f[a];
            int b = a - 1;
            while (b >= 0 && f[b] > v) { f[b + 1] = f[b]; --b; }
            f[b + 1] = v;
        }
        if (out->z_std) {                                                                                           /* :718 torch.std(unbiased=False) */
            double mean = 0.0, var = 0.0;
            for (int s = 0; s < Ni; ++s) mean += zs[(size_t)i * Ni + s];
            mean /= Ni;
            for (int s = 0; s < Ni; ++s) { const double dd = zs[(size_t)i * Ni + s] - mean; var += dd * dd; }
            out->z_std[r0 + i] = (float)sqrt(var / Ni);
        }
    }
    pass(j, nf, k, r0, nr, zf, Sf, zc, Sc, &out->fine, NULL);
}

int iblnerf_render_cpu(const iblnerf_options* opt, const float* blob_coarse, const float* blob_fine, size_t n_floats, const float* lut_rgb,
                       const float* rays_o, const float* rays_d, int64_t n_rays, float near_, float far_,
                       const iblnerf_overrides* ov, const iblnerf_outputs* out, int n_threads) {
    pick_isa();
    if (!opt || !blob_coarse || !lut_rgb || !rays_o || !rays_d || !out || n_rays < 0) return fail(IBLNERF_ERR_INVALID, "render_cpu: null argument");
    if (n_floats != N_PARAMS) return fail(IBLNERF_ERR_INVALID, "render_cpu: blob has %zu floats, the IBLNeRF state dict has %d", n_floats, N_PARAMS);
    if (opt->n_samples < 2 || opt->n_samples > 511 || opt->n_importance < 0 || opt->n_samples + opt->n_importance > 511)
        return fail(IBLNERF_ERR_INVALID, "render_cpu: sample counts outside 2..511 (torch.sum's cascade levels start at 512: not restated)");
    if (opt->n_importance > 0 && !blob_fine) return fail(IBLNERF_ERR_INVALID, "render_cpu: n_importance > 0 needs the fine network");
    if (opt->normal_mode != IBLNERF_NORMAL_DEPTH_GRADIENT_EPSILON && opt->normal_mode != IBLNERF_NORMAL_GROUND_TRUTH)
        return fail(IBLNERF_ERR_INVALID, "render_cpu: normal_mode %d is not restated here (the numpy oracle has it)", opt->normal_mode);
    if (opt->normal_mode == IBLNERF_NORMAL_GROUND_TRUTH && !(ov && ov->d_gt_normal)) return fail(IBLNERF_ERR_INVALID, "render_cpu: ground-truth normals need overrides.d_gt_normal");
    if (ov && (ov->d_gt_albedo || ov->d_gt_roughness || ov->d_gt_irradiance || ov->d_gt_depth))
        return fail(IBLNERF_ERR_INVALID, "render_cpu: the *_from_gt rows are not restated here (the numpy oracle has them)");
    if (ov && ov->mode) {
        if (ov->mode < 0 || ov->mode > 2 || ov->num_objects <= 0 || ov->num_objects > 8 || !ov->d_mask) return fail(IBLNERF_ERR_INVALID, "render_cpu: overrides need a mask and 1..8 objects");   /* :222, :232 */
        if ((ov->mode == 2 || ov->edit_depth) && !ov->d_depth) return fail(IBLNERF_ERR_INVALID, "render_cpu: depth override rows missing");
        if ((ov->mode == 2 || ov->edit_normal) && !ov->d_normal) return fail(IBLNERF_ERR_INVALID, "render_cpu: normal override rows missing");
        if (ov->mode == 1 && ov->edit_albedo && ov->edit_albedo_by_img && !ov->d_albedo) return fail(IBLNERF_ERR_INVALID, "render_cpu: albedo override rows missing");
        if (ov->n_roughness_list < 0 || ov->n_roughness_list > 8) return fail(IBLNERF_ERR_INVALID, "render_cpu: roughness list length");
    }
    Net nc, nf;
    if (net_init(&nc, blob_coarse, opt->color_independent_to_direction)) return fail(IBLNERF_ERR_STATE, "render_cpu: out of memory");
    if (net_init(&nf, blob_fine ? blob_fine : blob_coarse, opt->color_independent_to_direction)) { net_free(&nc); return fail(IBLNERF_ERR_STATE, "render_cpu: out of memory"); }
    Job job = {opt, lut_rgb, rays_o, rays_d, ov, near_, far_};
    const int Sc = opt->n_samples, Ni = opt->n_importance, Sf = Sc + Ni;
    const int RB = 12;                                   /* rays per task: 12 * 192 fine points = 12 dense-layer blocks */
    const long n_blocks = (n_rays + RB - 1) / RB;
    int bad = 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel num_threads(n_threads)
    {
        Work k;
        float* zc = (float*)xalloc(sizeof(float) * RB * Sc);
        float* zf = (float*)xalloc(sizeof(float) * RB * Sf);
        float* wc = (float*)xalloc(sizeof(float) * RB * Sc);
        float* zs = (float*)xalloc(sizeof(float) * RB * (Ni > 0 ? Ni : 1));
        const int ok = !work_init(&k, RB, Sf, Sc) && zc && zf && wc && zs;
        if (!ok) {
#pragma omp atomic write
            bad = 1;
        }
#pragma omp barrier
        if (!bad) {
#pragma omp for schedule(dynamic, 1)
            for (long b = 0; b < n_blocks; ++b) {
                const long r0 = b * RB;
                const int nr = (int)(n_rays - r0 < RB ? n_rays - r0 : RB);
                render_block(&job, &nc, &nf, &k, r0, nr, out, zc, zf, wc, zs);
            }
        }
        work_free(&k);
        free(zc); free(zf); free(wc); free(zs);
    }
    net_free(&nc);
    net_free(&nf);
    return bad ? fail(IBLNERF_ERR_STATE, "render_cpu: out of memory") : 0;
}

int iblnerf_network_query_cpu(const float* blob, size_t n_floats, int color_independent, const float* pts, int64_t n_rays, int n_samples,
                              const float* dirs, float* out, int n_threads) {
    pick_isa();
    if (!blob || !pts || !out || n_rays < 0 || n_samples <= 0) return fail(IBLNERF_ERR_INVALID, "network_query_cpu: null argument");
    if (n_floats != N_PARAMS) return fail(IBLNERF_ERR_INVALID, "network_query_cpu: blob has %zu floats, the IBLNeRF state dict has %d", n_floats, N_PARAMS);
    Net n;
    if (net_init(&n, blob, color_independent)) return fail(IBLNERF_ERR_STATE, "network_query_cpu: out of memory");
    const int RB = 16;
    const long n_blocks = (n_rays + RB - 1) / RB;
    const int ch = dirs ? 18 : 1;
    int bad = 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel num_threads(n_threads)
    {
        Scratch s;
        if (scratch_init(&s)) {
#pragma omp atomic write
            bad = 1;
        }
#pragma omp barrier
        if (!bad) {
#pragma omp for schedule(dynamic, 1)
            for (long b = 0; b < n_blocks; ++b) {
                const long r0 = b * RB;
                const long nr = n_rays - r0 < RB ? n_rays - r0 : RB;
                mlp_eval(&n, &s, pts + 3 * r0 * n_samples, nr * n_samples, dirs ? dirs + 3 * r0 : NULL, n_samples, out + (size_t)r0 * n_samples * ch);
            }
        }
        scratch_free(&s);
    }
    net_free(&n);
    return bad ? fail(IBLNERF_ERR_STATE, "network_query_cpu: out of memory") : 0;
}

int iblnerf_sample_pdf_cpu(const float* bins, const float* weights, int64_t n, int n_bins, int n_samples, float* out) {
    if (!bins || !weights || !out || n < 0 || n_bins < 2 || n_bins > 512 || n_samples < 1 || n_samples > 4096) return fail(IBLNERF_ERR_INVALID, "sample_pdf_cpu: bad argument");
    float* u = (float*)malloc(sizeof(float) * n_samples);
    float* tmp = (float*)malloc(sizeof(float) * 2 * (n_bins + 1));
    torch_linspace(0.f, 1.f, n_samples, u);
    for (int64_t i = 0; i < n; ++i) sample_pdf_row(bins + i * n_bins, weights + i * (n_bins - 1), n_bins, n_samples, u, out + i * n_samples, tmp);
    free(u); free(tmp);
    return 0;
}

/* nerf_models/nerf_renderer_helper.py:36-45 */
int iblnerf_get_rays_cpu(int H, int W, const float* K, const float* c2w, float* rays_o, float* rays_d) {
    if (H <= 0 || W <= 0 || !K || !c2w || !rays_o || !rays_d) return fail(IBLNERF_ERR_INVALID, "get_rays_cpu: bad argument");
    float* ii = (float*)malloc(sizeof(float) * W);
    float* jj = (float*)malloc(sizeof(float) * H);
    torch_linspace(0.f, (float)(W - 1), W, ii);
    torch_linspace(0.f, (float)(H - 1), H, jj);
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const float dir[3] = {(ii[c] - K[2]) / K[0], -(jj[r] - K[5]) / K[4], -1.f};
            float* d = rays_d + 3 * ((size_t)r * W + c);
            float* o = rays_o + 3 * ((size_t)r * W + c);
            for (int a = 0; a < 3; ++a) {
                d[a] = (dir[0] * c2w[4 * a] + dir[1] * c2w[4 * a + 1]) + dir[2] * c2w[4 * a + 2];
                o[a] = c2w[4 * a + 3];
            }
        }
    free(ii); free(jj);
    return 0;
}
