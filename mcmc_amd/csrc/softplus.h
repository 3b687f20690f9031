// softplus.h -- branch-free fp64 M(eta) = |eta|/2 + log1p(exp(-|eta|)) for the logistic log density.
//
// The reference evaluates, per observation (vignettes/demo.Rnw:61-62),
//     logp = eta<0 ? eta - log1p(exp(eta)) : -log1p(exp(-eta))
//     logq = eta<0 ? -log1p(exp(eta))      : -eta - log1p(exp(-eta))
// i.e. with s = +1 for y==1 and -1 for y==0:  min(s*eta, 0) - L(|eta|) = s*eta/2 - M(eta),
// L(a) = log1p(exp(-a)).  The s*eta/2 part is linear in theta (a column-sum dot product, added by the update
// kernel), so the per-observation work is M alone.
//
// On gfx950 the fp64 MFMA and the fp64 VALU share one pipe (profiles/r2_fp64_sustain_probe.txt): every
// fp64 VALU op of this epilogue is paid in MFMA time, so the op count is what matters.  Library
// exp()/log1p() cost ~150 fp64 ops; the first version of this file (table of {L_j, sigma_j} + an expm1 and a
// log1p series, clamp and a separate sum of |eta|) cost 17.  Here M is a degree-5 polynomial per cell of a grid
// a_j = j/56.5, j = 0..2034, selected by the nearest grid point (|eta| - a_j = r, |r| <= H = 1/113):
//   * the degree-6 Taylor polynomial of M around a_j (t_k = M^(k)(a_j) / k!) with its last term economised: on [-H, H]
//         r^6 = H^6 T6(r/H)/32 + 1.5 H^2 r^4 - 0.5625 H^4 r^2 + H^6/32,   |T6| <= 1,
//     so t4 += 1.5 H^2 t6, t2 -= 0.5625 H^4 t6, t0 += H^6 t6 / 32; what is dropped is <= |t6| H^6 / 32 + |t7| H^7 < 6e-18
//     (|M^(6)| <= 1/4), a twentieth of an ulp of M >= ln 2;
//   * then re-expanded in powers of w = |eta| ITSELF (in long double, rounded once):
//         M(w) ~ d0 + w (d1 + w (d2 + w (d3 + w (d4 + w d5)))),   d_m = sum_k t_k C(k,m) (-a_j)^(k-m).
//     The usual objection to an uncentred polynomial -- cancellation between large terms -- does not bite here: beyond
//     the linear part w/2 every t_k decays like e^(-a_j)/k!, so the expanded terms are bounded by e^(-a) a^k / k! <= 0.18
//     against M >= 0.69.  Measured (tools/check_softplus.py against long double): <= 4 ulp of M, mean 0.42 (the centred
//     form before it: <= 2, mean 0.31), for one subtraction and one table double less per value.
// Eight doubles per entry {d0..d5, 0, 0} = 130 KB of LDS, as two arrays of 32-byte records ({d0,d1,d2,d3} and {d4,d5,0,0}:
// three ds_read_b128 off one address register).
// Measured on the box (exp builds of the MFMA kernel): v_mfma_f64_16x16x4 occupies the SIMD's VALU for its 64
// cycles, so EVERY VALU instruction of the epilogue -- integer ones too -- adds its issue time to the kernel,
// while the table reads themselves cost almost nothing.  The epilogue is therefore written for instruction
// count: 1 v_min_f64 (clamp for the index), 1 fma (index), 1 shift (address), 5 fma (|eta| as an operand modifier),
// and the caller's running sum = **9 VALU instructions** per value (centred degree 5: 10; degree-6 Taylor: 11; the
// {L_j, sigma_j} + expm1/log1p-series version before that: 20).
// Beyond the table (|eta| > 36) the index stays in the last cell (the clamp) and the last record holds the line
// L(36) + w/2 (L(36) = 2.3e-16 is constant to 1 ulp of M there; its higher coefficients are 1e-300 rather than 0 so
// that |eta| = Inf gives +Inf, not NaN; |eta| up to 1e50 is exact to rounding).
// NaN propagates.
//
// Accuracy (tools/check_softplus.py against long double): <= 4 ulp of M (mean 0.42) for |eta| up to 1e30.
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
#define SP_SCALE 56.5                        // grid points per unit (exact in binary)
#define SP_N 2035                            // SP_AMAX * SP_SCALE + 1
#define SP_DEG 6                             // degree of the Taylor polynomial the table is built from (evaluated: 5)
#define SP_ACLAMP 36.0078125                 // < 36 + 0.5/SP_SCALE: the index never leaves the table

struct SoftplusTables {
    double e[4 * 2 * SP_N];   // SP_N records {d0, d1, d2, d3}, then SP_N records {d4, d5, 0, 0}
};
#define SP_TAB_DOUBLES (4 * 2 * SP_N)

SP_HD double sp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// tab may live in LDS or host memory.  Layout: SP_N records {d0, d1, d2, d3}, then SP_N records {d4, d5, 0, 0}.
SP_HD double softplus_half_abs(double eta, const double *tab)
{
    const double SHIFT = 6755399441055744.0;   // 1.5 * 2^52: low mantissa bits of (x + SHIFT) = rint(x)
    // index from |eta| clamped just inside the last cell (one v_min_f64 with the |.| modifier; NaN -> last cell),
    // polynomial in the unclamped |eta|: past the table the last record is the line L(36) + w/2
#if defined(__HIP_DEVICE_COMPILE__)
    double ac;
    asm("v_min_f64 %0, |%1|, %2" : "=v"(ac) : "v"(eta), "s"((double)SP_ACLAMP));
#else
    const double ab = __builtin_fabs(eta);
    const double ac = ab < (double)SP_ACLAMP ? ab : (double)SP_ACLAMP;    // NaN: comparison false -> clamp
#endif
    const double jd = sp_fma(ac, SP_SCALE, SHIFT);
    int64_t bits;
    memcpy(&bits, &jd, 8);
    const uint32_t j = (uint32_t)bits;           // 0 .. SP_N-1 by construction of the clamp
    const double *t0 = tab + 4 * j, *t1 = t0 + 4 * SP_N;
    const double w = __builtin_fabs(eta);
    double q = sp_fma(w, t1[1], t1[0]);
    q = sp_fma(w, q, t0[3]);
    q = sp_fma(w, q, t0[2]);
    q = sp_fma(w, q, t0[1]);
    return sp_fma(w, q, t0[0]);
}

#if defined(__HIPCC__)
// The same function for a table in LDS, written so that the compiler cannot spend instructions on operand
// plumbing (measured with SQ_INSTS_VALU: 14.5 VALU per value before, 11 now): the index fma takes its scale from an
// SGPR pair and its magic constant from a loop-invariant VGPR pair (the compiler's own choice was v_fmac_f64 plus two
// v_mov_b32 that reload the constant for every value), and that constant already carries the table's LDS byte
// offset / 32, so the low mantissa word of jd, shifted by 5, IS the LDS address of record j (one v_lshlrev_b32
// instead of shift + add).  softplus_lds_shift() makes the constant once per kernel.
typedef __attribute__((address_space(3))) const double sp_lds_cd;
__device__ __forceinline__ double softplus_lds_shift(const double *tab_in_lds)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const double *)tab_in_lds;
    double shiftb = 6755399441055744.0 + (double)(off >> 5);      // records are 32 bytes; the table is 32-byte aligned
    asm volatile("" : "+v"(shiftb));     // keep it in a VGPR pair: then the index fma is one v_fma_f64 (SGPR scale, VGPR addend)
    return shiftb;
#else
    return 0.0;        // device-only (the host pass of hipcc only needs the declaration)
#endif
}
// Split in three so that a caller can software-pipeline: address (3 VALU), the three table reads, evaluation
// (6 VALU with the caller's running sum).  softplus_half_abs_lds() is the three in a row.
__device__ __forceinline__ uint32_t softplus_lds_addr(double eta, double shiftb)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double ac, jd;
    asm("v_min_f64 %0, |%2|, %3\n\tv_fma_f64 %1, %0, %4, %5"
        : "=&v"(ac), "=v"(jd) : "v"(eta), "s"((double)SP_ACLAMP), "s"((double)SP_SCALE), "v"(shiftb));
    const uint32_t addr = (uint32_t)__double2loint(jd) << 5;   // low mantissa word (record index + table offset / 32), as a byte address: a 32-bit shift
    __builtin_assume((addr & 31u) == 0u);
    return addr;
#else
    return 0u;         // device-only (the host pass of hipcc only needs the declaration)
#endif
}
struct SoftplusRec { double d0, d1, d2, d3, d4, d5; };
__device__ __forceinline__ void softplus_lds_load(uint32_t addr, SoftplusRec &t)
{
#if defined(__HIP_DEVICE_COMPILE__)
    sp_lds_cd *t0 = (sp_lds_cd *)(uintptr_t)addr, *t1 = t0 + 4 * SP_N;
    t.d0 = t0[0]; t.d1 = t0[1]; t.d2 = t0[2]; t.d3 = t0[3];
    t.d4 = t1[0]; t.d5 = t1[1];
#endif
}
__device__ __forceinline__ double softplus_lds_eval(double eta, const SoftplusRec &t)
{
    const double w = __builtin_fabs(eta);
    double q = sp_fma(w, t.d5, t.d4);
    q = sp_fma(w, q, t.d3);
    q = sp_fma(w, q, t.d2);
    q = sp_fma(w, q, t.d1);
    return sp_fma(w, q, t.d0);
}
__device__ __forceinline__ double softplus_half_abs_lds(double eta, double shiftb)
{
    SoftplusRec t;
    softplus_lds_load(softplus_lds_addr(eta, shiftb), t);
    return softplus_lds_eval(eta, t);
}
// the same without inline asm (asm is convergent in device code, which keeps the compiler from unrolling a loop with a
// run-time trip count around it): for loops over observations outside the MFMA kernel, where an instruction more or
// less does not matter
__device__ __forceinline__ double softplus_half_abs_lds_plain(double eta, double shiftb)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double ac = fmin(__builtin_fabs(eta), (double)SP_ACLAMP);      // NaN -> the clamp -> last cell, as v_min_f64
    const double jd = sp_fma(ac, SP_SCALE, shiftb);
    const uint32_t addr = (uint32_t)__double2loint(jd) << 5;
    SoftplusRec t;
    softplus_lds_load(addr, t);
    return softplus_lds_eval(eta, t);
#else
    return 0.0;
#endif
}
#endif

// host-side table construction in long double.  L^(k)(a) is a polynomial p_k in s = e^-a / (1 + e^-a):
// p_1 = -s and p_{k+1}(s) = p_k'(s) * (-s (1 - s))  (ds/da = -s(1-s)).
static inline void softplus_tables_init(SoftplusTables *t)
{
    long double poly[SP_DEG + 1][SP_DEG + 2];     // poly[k][m]: coefficient of s^m in p_k
    memset(poly, 0, sizeof poly);
    poly[1][1] = -1.0L;
    for (int k = 1; k < SP_DEG; k++)
        for (int m = 1; m <= k; m++) {
            // d/ds (c s^m) * (-s + s^2) = c m (-s^m + s^(m+1))
            poly[k + 1][m] += -poly[k][m] * m;
            poly[k + 1][m + 1] += poly[k][m] * m;
        }
    long double fact = 1.0L;
    long double kfact[SP_DEG + 1];
    kfact[0] = 1.0L;
    for (int k = 1; k <= SP_DEG; k++) { fact *= k; kfact[k] = fact; }
    for (int j = 0; j < SP_N; j++) {
        const double aj = (double)((long double)j / (long double)SP_SCALE);
        const long double ex = expl(-(long double)aj);
        const long double s = ex / (1.0L + ex);
        long double tk[SP_DEG + 1];                 // Taylor coefficients of M at a_j
        tk[0] = (long double)aj / 2 + log1pl(ex);
        for (int k = 1; k <= SP_DEG; k++) {
            long double v = 0.0L;
            for (int m = k; m >= 1; m--) v = v * s + poly[k][m];     // Horner in s, p_k has no constant term
            v *= s;
            tk[k] = v / kfact[k] + (k == 1 ? 0.5L : 0.0L);
        }
        const long double H = 0.5L / (long double)SP_SCALE, H2 = H * H;
        tk[4] += 1.5L * H2 * tk[6];                 // economise t6 r^6 (header)
        tk[2] -= 0.5625L * H2 * H2 * tk[6];
        tk[0] += H2 * H2 * H2 * tk[6] / 32.0L;
        // re-expand sum_{k<=5} t_k (w - a_j)^k in powers of w
        long double dm[SP_DEG];
        for (int m = 0; m < SP_DEG; m++) dm[m] = 0.0L;
        for (int k = 0; k < SP_DEG; k++) {
            long double binom = 1.0L, pw = 1.0L;                 // C(k, k - i) and (-a_j)^i for m = k - i
            for (int i = 0; i <= k; i++) {
                dm[k - i] += tk[k] * binom * pw;
                binom = binom * (k - i) / (i + 1);
                pw *= -(long double)aj;
            }
        }
        double c[SP_DEG];
        for (int k = 0; k < SP_DEG; k++) c[k] = (double)dm[k];
        if (j == SP_N - 1) {                       // the line beyond the table
            c[0] = (double)log1pl(ex);
            c[1] = 0.5;
            for (int k = 2; k < SP_DEG; k++) c[k] = 1e-300;
        }
        double *r0 = t->e + 4 * j, *r1 = t->e + 4 * SP_N + 4 * j;
        r0[0] = c[0]; r0[1] = c[1]; r0[2] = c[2]; r0[3] = c[3];
        r1[0] = c[4]; r1[1] = c[5]; r1[2] = 0.0; r1[3] = 0.0;
    }
}
