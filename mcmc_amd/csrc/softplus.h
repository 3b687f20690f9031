// softplus.h -- branch-free fp64 L(a) = log1p(exp(-a)), a >= 0, for the logistic log density.
//
// The reference evaluates, per observation (vignettes/demo.Rnw:61-62),
//     logp = eta<0 ? eta - log1p(exp(eta)) : -log1p(exp(-eta))
//     logq = eta<0 ? -log1p(exp(eta))      : -eta - log1p(exp(-eta))
// i.e. with s = +1 for y==1 and -1 for y==0:  min(s*eta, 0) - L(|eta|).
// Library exp()/log1p() cost ~150 fp64 VALU instructions with branches; on the MFMA path the
// epilogue must hide under 16 v_mfma_f64_16x16x4 per 4 evaluations, so L is built from two
// table-driven range reductions and two short Horner chains (28 fp64 ops, ~8 int ops):
//
//   t = exp(-a):  -a = k*ln2/32 + r, |r| <= ln2/64;  t = 2^(k>>5) * T[k&31] * (1 + expm1(r))
//                 T = 2^(j/32) (32 doubles = one 256-B LDS bank row: ds_read_b64 is conflict-free)
//   L = log1p(t): w = 1+t, dlt = t-(w-1) (exact);  j = top 7 mantissa bits of w (0..128)
//                 c_j ~ 1/w (c_0 = 1 exactly), s = fma(w,c,-1) + dlt*c, |s| <= 2^-7
//                 L = -log(c_j) + s*P(s),  P = 1 - s/2 + ... - s^7/8
//
// Accuracy (tools/check_softplus.py against long-double log1pl(expl(-a))): <= 4 ulp (mean 0.5) for a in
// [0, 700], relative accuracy kept for tiny results (a large: L -> t).
//
// The same source compiles for the host (g++) so the arithmetic can be checked without a GPU.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define SP_HD __host__ __device__ __forceinline__
#else
#define SP_HD static inline
#include <math.h>
#endif

#define SP_EXP_N 32
#define SP_LOG_N 129

struct SoftplusTables {
    double expT[SP_EXP_N];        // 2^(j/32)
    double logT[2 * SP_LOG_N];    // {c_j, -log(c_j)} pairs
};

SP_HD double sp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

SP_HD int32_t sp_hi(double x) { int64_t b; memcpy(&b, &x, 8); return (int32_t)(b >> 32); }
SP_HD int32_t sp_lo(double x) { int64_t b; memcpy(&b, &x, 8); return (int32_t)b; }
SP_HD double sp_mk(int32_t hi, int32_t lo)
{
    int64_t b = ((int64_t)hi << 32) | (uint32_t)lo;
    double x; memcpy(&x, &b, 8); return x;
}

// expT / logT may live in LDS (device) or plain memory (host)
SP_HD double softplus_neg(double a_in, const double *expT, const double *logT)
{
    const double SHIFT = 6755399441055744.0;            // 1.5 * 2^52
    const double INV = 46.16624130844683;               // 32/ln2
    const double LN2_32_HI = 0x1.62e42fe000000p-6;      // ln2/32 with the low 24 bits zero: k*HI exact
    const double LN2_32_LO = 5.689487495325457e-11;      // ln2/32 - HI
    double a = a_in < 700.0 ? a_in : 700.0;
    double kd = sp_fma(a, -INV, SHIFT);
    int32_t k = sp_lo(kd);
    double kf = kd - SHIFT;
    double r = sp_fma(-kf, LN2_32_HI, -a);
    r = sp_fma(-kf, LN2_32_LO, r);
    double T = expT[k & 31];
    double r2 = r * r;
    double q = sp_fma(r, 1.0 / 720, 1.0 / 120);
    q = sp_fma(r, q, 1.0 / 24);
    q = sp_fma(r, q, 1.0 / 6);
    q = sp_fma(r, q, 0.5);
    double p = sp_fma(r2, q, r);                          // expm1(r)
    double t = sp_fma(T, p, T);
    t = sp_mk(sp_hi(t) + ((k >> 5) << 20), sp_lo(t));     // * 2^(k>>5); k<=0, t stays normal for a<=700

    double w = 1.0 + t;
    double dlt = t - (w - 1.0);
    int32_t j = (sp_hi(w) - 0x3FF00000) >> 13;            // 0..128
    double c = logT[2 * j], nlogc = logT[2 * j + 1];
    double s = sp_fma(w, c, -1.0);
    s = sp_fma(dlt, c, s);
    double P = sp_fma(s, -1.0 / 8, 1.0 / 7);
    P = sp_fma(s, P, -1.0 / 6);
    P = sp_fma(s, P, 1.0 / 5);
    P = sp_fma(s, P, -1.0 / 4);
    P = sp_fma(s, P, 1.0 / 3);
    P = sp_fma(s, P, -0.5);
    P = sp_fma(s, P, 1.0);
    return sp_fma(s, P, nlogc);
}

#include <math.h>
// host-side table construction (long double log for the -log(c) column)
static inline void softplus_tables_init(SoftplusTables *t)
{
    for (int j = 0; j < SP_EXP_N; j++) t->expT[j] = (double)exp2l((long double)j / 32.0L);
    for (int j = 0; j < SP_LOG_N; j++) {
        double c;
        if (j == 0) c = 1.0;
        else if (j == 128) c = 0.5;
        else c = 1.0 / (1.0 + (j + 0.5) / 128.0);
        t->logT[2 * j] = c;
        t->logT[2 * j + 1] = (j == 0) ? 0.0 : (double)(-logl((long double)c));
    }
}
