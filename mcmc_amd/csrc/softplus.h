// softplus.h -- branch-free fp64 L(a) = log1p(exp(-a)), a >= 0, for the logistic log density.
//
// The reference evaluates, per observation (vignettes/demo.Rnw:61-62),
//     logp = eta<0 ? eta - log1p(exp(eta)) : -log1p(exp(-eta))
//     logq = eta<0 ? -log1p(exp(eta))      : -eta - log1p(exp(-eta))
// i.e. with s = +1 for y==1 and -1 for y==0:  min(s*eta, 0) - L(|eta|).
//
// On gfx950 the fp64 MFMA and the fp64 VALU share one pipe (profiles/r1_mfma64_probe.txt): every
// fp64 VALU op of this epilogue is paid in MFMA time, so the op count is what matters.  Library
// exp()/log1p() cost ~150 fp64 ops.  Here, with a = a_j + r on a grid a_j = j/128 (|r| <= 1/256):
//     L(a) = L(a_j) + log1p( sigma_j * expm1(-r) ),   sigma_j = e^{-a_j} / (1 + e^{-a_j})
// one 16-byte table entry {L_j, -sigma_j} (73.7 KB in LDS, one ds_read_b128), a degree-5 expm1
// and a degree-5 log1p: 15 fp64 ops + 1 integer op.  Both series act on small arguments
// (|v| <= 0.002), so L keeps its relative accuracy when it is tiny (a large).
// For a > 36 the value is held at L(36) = 2.3e-16 (absolute error <= 2.3e-16).
//
// Accuracy (tools/check_softplus.py against long-double log1pl(expl(-a))): <= 2 ulp (mean 0.4) on [0, 36].
// The same source compiles for the host (g++) so the arithmetic can be checked without a GPU.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SP_HD __host__ __device__ __forceinline__
#else
#define SP_HD static inline
#endif

#define SP_AMAX 36
#define SP_INV_H 128
#define SP_N (SP_AMAX * SP_INV_H + 1)

struct SoftplusTables {
    double e[2 * SP_N];   // {L_j, -sigma_j}
};

SP_HD double sp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
SP_HD int32_t sp_lo(double x) { int64_t b; memcpy(&b, &x, 8); return (int32_t)b; }

// min(|eta|, SP_AMAX).  On the device one v_min_f64 with the |.| source modifier: written as
// inline asm because the compiler's fmin() first canonicalises its operand (an extra
// v_max_f64 x,x per call) when it cannot prove the MFMA result is not a signalling NaN.
SP_HD double sp_clamp_abs(double eta)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double a;
    asm("v_min_f64 %0, |%1|, %2" : "=v"(a) : "v"(eta), "s"((double)SP_AMAX));
    return a;
#else
    const double a = fabs(eta);
    return a < (double)SP_AMAX ? a : (double)SP_AMAX;
#endif
}

// a must already be clamped to [0, SP_AMAX] (sp_clamp_abs); tab may live in LDS or host memory
SP_HD double softplus_neg(double a, const double *tab)
{
    const double SHIFT = 6755399441055744.0;   // 1.5 * 2^52: low mantissa bits of (x + SHIFT) = rint(x)
    const double jd = sp_fma(a, (double)SP_INV_H, SHIFT);
    const int32_t j = sp_lo(jd);
    const double jf = jd - SHIFT;
    const double r = sp_fma(jf, -1.0 / SP_INV_H, a);          // exact: a - j/128
    const double Lj = tab[2 * j], nsig = tab[2 * j + 1];
    // expm1(-r) = -r * (1 - r/2 + r^2/6 - r^3/24 + r^4/120)
    double q = sp_fma(r, 1.0 / 120, -1.0 / 24);
    q = sp_fma(r, q, 1.0 / 6);
    q = sp_fma(r, q, -0.5);
    q = sp_fma(r, q, 1.0);
    const double v = (nsig * r) * q;                           // sigma_j * expm1(-r)
    // log1p(v) = v * (1 - v/2 + v^2/3 - v^3/4 + v^4/5)
    double P = sp_fma(v, 0.2, -0.25);
    P = sp_fma(v, P, 1.0 / 3);
    P = sp_fma(v, P, -0.5);
    P = sp_fma(v, P, 1.0);
    return sp_fma(v, P, Lj);
}

// host-side table construction in long double
static inline void softplus_tables_init(SoftplusTables *t)
{
    for (int j = 0; j < SP_N; j++) {
        const long double aj = (long double)j / SP_INV_H;
        const long double ex = expl(-aj);
        t->e[2 * j] = (double)log1pl(ex);
        t->e[2 * j + 1] = (double)(-ex / (1.0L + ex));
    }
}
