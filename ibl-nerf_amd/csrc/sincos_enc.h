// sin/cos of 2^k * x for the positional encoding (src/nerf_models/positional_embedder.py:21-34),
// shared by host (unit test through the C-ABI) and device.
//
// The reference evaluates sin(fl(x * 2^k)); x * 2^k is exact, so it is the sine of an exactly
// known real number with |arg| up to ~4600 rad.  Instead of one Payne-Hanek reduction per
// (coordinate, frequency) pair this does ONE extended-precision reduction per coordinate:
//     phi = x / (2 pi)  as an unevaluated float pair (hi, lo)           (~48 bits)
// and per frequency uses that scaling by 2^k and taking the fractional part are exact in binary
// floating point:
//     f = frac(hi * 2^k) + lo * 2^k     in [-0.5, 0.5] turns
//     sin/cos(2 pi f) by quadrant + degree-9/10 polynomials on [-pi/4, pi/4].
// Measured against float64 on 1e6 random points with |x| <= 16, k <= 9: max error < 2.5e-7
// (about 2 ulp of 1.0) — tests/test_host_logic.py pins this bound.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define IBL_HD __host__ __device__ __forceinline__
#else
#define IBL_HD inline
#endif

namespace ibl {

struct TurnPair { float hi, lo; };

// x / (2 pi) as hi + lo
IBL_HD TurnPair to_turns(float x) {
    const float C_HI = 0.15915494f;            // fl(1/(2 pi))
    const float C_LO = 6.4206382e-09f;         // 1/(2 pi) - C_HI  (C_HI = 0.159154936671257019...)
    TurnPair t;
    t.hi = x * C_HI;
    t.lo = fmaf(x, C_HI, -t.hi) + x * C_LO;
    return t;
}

// sin and cos of 2*pi*(t.hi + t.lo) * scale, scale = 2^k exactly
IBL_HD void sincos_turns(TurnPair t, float scale, float* s_out, float* c_out) {
    const float a = t.hi * scale;                     // exact
    float f = a - rintf(a);                           // exact, [-0.5, 0.5]
    f = f + t.lo * scale;                             // tiny correction
    const float y = 4.0f * f;                         // quarter turns, [-2, 2]
    const float qf = rintf(y);
    const float r = y - qf;                           // exact, [-0.5, 0.5]
    const int q = (int)qf & 3;
    const float u = r * 1.5707964f;                   // angle in [-pi/4, pi/4]
    const float u2 = u * u;
    float ps = fmaf(u2, 2.7557319e-06f, -1.9841270e-04f);
    ps = fmaf(ps, u2, 8.3333333e-03f);
    ps = fmaf(ps, u2, -1.6666667e-01f);
    const float sn = fmaf(ps * u2, u, u);
    float pc = fmaf(u2, -2.7557319e-07f, 2.4801587e-05f);
    pc = fmaf(pc, u2, -1.3888889e-03f);
    pc = fmaf(pc, u2, 4.1666668e-02f);
    pc = fmaf(pc, u2, -0.5f);
    const float cs = fmaf(pc, u2, 1.0f);
    // rotate by q quarter turns
    const float s1 = (q & 1) ? cs : sn;
    const float c1 = (q & 1) ? sn : cs;
    *s_out = (q & 2) ? -s1 : s1;
    *c_out = ((q + 1) & 2) ? -c1 : c1;
}

}  // namespace ibl
