// targets.h -- device evaluation of the log unnormalised density descriptors that replace the
// reference's R closure (logh(), metrop.c:303-323 / temper.c:955-975).
//
// A density is evaluated cooperatively by the G lanes (G = 1, 4, 16 or 64, all inside one
// wavefront) that own one chain: each lane takes a strided slice of the observations /
// matrix rows / mixture components and the partial sums are combined with an xor butterfly of
// DPP shuffles, which leaves the bit-identical total on every lane, so all G lanes take the
// same accept/reject decision without a broadcast.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include "../../include/mcmchip.h"
#include "softplus.h"

struct TargetView {
    int kind, d;
    long n;
    const double *x, *y;   // may point into LDS
    const double *par;
    double prior_c;
    int morph_flags;
    double morph_b, morph_r, morph_p;
    const double *morph_center;
    double sp_shift = 0.0;   // != 0: the softplus table is staged in LDS (softplus_lds_shift); 0: library exp/log1p
    int xy_in_lds = 0;       // x and y point into LDS: read them with ds_read, not through the flat aperture
    int x_folded = 0;        // the LDS copy of x holds s_i x_i, s_i = +1 (y_i == 1) / -1: see logistic_terms
};

// lane i <- lane (i + n) mod 16 of its row of 16 (DPP row_ror) / a quad permutation: ~8 cycles instead of the ~120 of
// a ds_bpermute round trip
template <int CTRL> __device__ __forceinline__ double dpp_mov64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// xor butterfly, strides G/2 .. 1.  Within a row of 16 lanes the strides 8, 4, 2, 1 (in that order) are done with
// row rotations instead: after the stride-s step the partial sums have period s along the row, so lane i and its
// rotation partner hold the two halves of the same coset, exactly the pair the xor butterfly adds (commutative):
// bit-identical totals on every lane, as before.
template <int G> __device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int m = G / 2; m >= 16; m >>= 1) v += __shfl_xor(v, m, G);
    if (G >= 16) v += dpp_mov64<0x128>(v);      // row_ror:8
    if (G >= 16) v += dpp_mov64<0x124>(v);      // row_ror:4
    if (G >= 4) v += dpp_mov64<0x4E>(v);        // quad_perm [2,3,0,1]: xor 2
    if (G >= 4) v += dpp_mov64<0xB1>(v);        // quad_perm [1,0,3,2]: xor 1
    return v;
}

template <int G> __device__ __forceinline__ double group_max(double v)
{
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) v = fmax(v, __shfl_xor(v, m, G));
    return v;
}

template <int G> __device__ __forceinline__ int group_and(int v)
{
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) v &= __shfl_xor(v, m, G);
    return v;
}

// isotropic finite normal mixture, par = K, then K x {a_k, b_k, mu_k[d]}
// PT: const double * (global or generic) or an address-space-3 pointer when the parameter block is staged in LDS
// (a generic pointer into LDS turns every read into a FLAT load, far slower than ds_read)
template <int G, typename PT = const double *> __device__ double normmix_group(PT par, int d, const double *v, int g)
{
    int K = (int)par[0];
    PT rec = par + 1;
    // pass 1: max of the exponents, pass 2: sum of exp (components strided over lanes)
    double m = -INFINITY;
    for (int k = g; k < K; k += G) {
        PT r = rec + (size_t)k * (2 + d);
        double s = 0.0;
        for (int j = 0; j < d; j++) { double t = v[j] - r[2 + j]; s += t * t; }
        m = fmax(m, r[0] - r[1] * s);
    }
    m = group_max<G>(m);
    if (m == -INFINITY) return -INFINITY;
    double acc = 0.0;
    for (int k = g; k < K; k += G) {
        PT r = rec + (size_t)k * (2 + d);
        double s = 0.0;
        for (int j = 0; j < d; j++) { double t = v[j] - r[2 + j]; s += t * t; }
        acc += exp(r[0] - r[1] * s - m);
    }
    acc = group_sum<G>(acc);
    return m + log(acc);
}

// log-likelihood terms of the logistic posterior (vignettes/demo.Rnw:59-65) over this lane's observations.
// eta is accumulated j-ascending with separately rounded multiply and add, like the reference's x %*% beta.
// D > 0: dimension known at compile time (state in registers, j loop unrolled); D = 0: run-time d.
// l1 = log1p(exp(-|eta|)) comes from the library, or -- TAB: table staged in LDS, a kernel-uniform choice -- as
// M(eta) - |eta|/2 with the <= 2 ulp table expansion of M = |eta|/2 + l1 (softplus.h): the absolute error is that
// of rounding eta itself; fmax keeps l1 = 0 where M and |eta|/2 are both +Inf.
// FOLD (the staged copy of X carries the sign of the response, s_i x_i): eta' = s_i eta exactly, and both arms of the
// reference's two ifelse are min(eta', 0) - l1 -- the same value bit for bit (y = 1: eta < 0 ? eta - l1 : -l1;
// y = 0, eta' = -eta: eta < 0 ? -l1 : -eta - l1) for 2 vector instructions instead of the ~22 that selecting between
// four candidates and two accumulators took (C2: 46 -> 24 instructions per observation).  All terms go to lp.
template <int G, int D, bool TAB, bool XL, bool FOLD = false>
__device__ __forceinline__ void logistic_terms(const TargetView &t, const double *v, int g, double &lp, double &lq)
{
    // XL: X and y live in LDS.  A generic pointer makes every read a FLAT instruction, whose LDS aperture path is far
    // slower than ds_read; the cast hands the compiler the address space.
    typedef __attribute__((address_space(3))) const double lds_cd;
    lds_cd *xl = XL ? (lds_cd *)t.x : nullptr, *yl = XL ? (lds_cd *)t.y : nullptr;
    const int d = D > 0 ? D : t.d;
    double vr[D > 0 ? D : 1];
    if (D > 0) {
#pragma unroll
        for (int j = 0; j < D; j++) vr[j] = v[j];
    }
    if constexpr (FOLD && D > 0) {
        // The latency-bound case (C2: 6-7 rows per lane, one wavefront per SIMD): blocks of 4 rows per lane, fully
        // unrolled with a uniform trip count, so that the rows' LDS reads, eta chains and table reads overlap -- a loop
        // unrolled by 4 with a one-row remainder loop ran the last 3 of 7 rows one after the other, each paying its two
        // LDS round trips and its dependent chain alone (30.8 -> 29.8 ms at C2; blocks of 8 have 24 table reads in
        // flight, more than lgkmcnt counts: 37.9 ms).  Rows past n are clamped for the reads and contribute + 0.0 (the
        // lane's sum is the same bits as before).  The sum starts from the first product instead of 0 + it, which
        // differs only in the sign of a zero eta.
        constexpr int U = 4;
        const int nrow = (int)t.n;
        for (int b0 = 0; b0 < nrow; b0 += U * G) {
            double eta[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = b0 + g + u * G;
                lds_cd *xli = xl + (size_t)(i < nrow ? i : nrow - 1) * D;
                eta[u] = mul_nofma(xli[0], vr[0]);
#pragma unroll
                for (int j = 1; j < D; j++) eta[u] = add_nofma(eta[u], mul_nofma(xli[j], vr[j]));
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double l1 = fmax(softplus_half_abs_lds_plain(eta[u], t.sp_shift) - 0.5 * fabs(eta[u]), 0.0);
                const double term = fmin(eta[u], 0.0) - l1;
                lp += (b0 + g + u * G < nrow) ? term : 0.0;
            }
        }
        return;
    }
    // (staged in LDS: a few thousand rows at most, 32-bit loop control)
    typedef typename std::conditional<XL, int, long>::type idx_t;
    const idx_t nrow = (idx_t)t.n;
#pragma unroll 4
    for (idx_t i = g; i < nrow; i += G) {
        const double *xi = t.x + (size_t)i * d;
        lds_cd *xli = xl + (size_t)i * d;
        double eta = 0.0;
        if (D > 0) {
#pragma unroll
            for (int j = 0; j < D; j++) eta = add_nofma(eta, mul_nofma(XL ? xli[j] : xi[j], vr[j]));
        } else {
#pragma unroll 4
            for (int j = 0; j < d; j++) eta = add_nofma(eta, mul_nofma(XL ? xli[j] : xi[j], v[j]));
        }
        // branch-free form of the reference's ifelse (vignettes/demo.Rnw:61-62): same operations per branch
        const double l1 = TAB ? fmax(softplus_half_abs_lds_plain(eta, t.sp_shift) - 0.5 * fabs(eta), 0.0)
                              : log1p(exp(-fabs(eta)));
        if constexpr (FOLD) {
            lp += fmin(eta, 0.0) - l1;
        } else {
            if ((XL ? yl[i] : t.y[i]) == 1.0) lp += (eta < 0) ? eta - l1 : -l1;
            else lq += (eta < 0) ? -l1 : -eta - l1;
        }
    }
}

// metrop targets; v = state[d]
// forced inline: as a real call the TargetView travels through scratch memory and every field, X and y included, is
// re-read from there through flat loads (measured: the C2 iteration spent 8 500 of its 12 000 cycles in this call)
template <int G> __device__ __forceinline__ double lud_group(const TargetView &t, const double *v, int g)
{
    const int d = t.d;
    switch (t.kind) {
    case MCMCHIP_T_LOGISTIC: {
        // vignettes/demo.Rnw:59-65; eta accumulated j-ascending with separately rounded
        // multiply and add, like the reference's x %*% beta
        double lp = 0.0, lq = 0.0, ss = 0.0;
        // the state goes to registers and the j loop is unrolled for the usual small dimensions (with a run-time d
        // every term of every eta waits for its own pair of loads)
        const bool tab = t.sp_shift != 0.0;
        switch (d) {
#define LOGIT_CASE(DD) \
        case DD: \
            if (tab && t.xy_in_lds) logistic_terms<G, DD, true, true, true>(t, v, g, lp, lq);   /* staged X is folded */ \
            else if (tab) logistic_terms<G, DD, true, false>(t, v, g, lp, lq); \
            else logistic_terms<G, DD, false, false>(t, v, g, lp, lq); \
            break;
        LOGIT_CASE(1) LOGIT_CASE(2) LOGIT_CASE(3) LOGIT_CASE(4) LOGIT_CASE(5) LOGIT_CASE(6) LOGIT_CASE(7) LOGIT_CASE(8)
#undef LOGIT_CASE
        default:
            if (tab && t.xy_in_lds) logistic_terms<G, 0, true, true, true>(t, v, g, lp, lq);
            else if (tab) logistic_terms<G, 0, true, false>(t, v, g, lp, lq);
            else logistic_terms<G, 0, false, false>(t, v, g, lp, lq);
        }
        for (int j = g; j < d; j += G) ss += v[j] * v[j];
        lp = group_sum<G>(lp);
        if (!(tab && t.xy_in_lds)) lq = group_sum<G>(lq);      // (the folded path puts every term into lp: lq is 0)
        ss = group_sum<G>(ss);
        return (lp + lq) - ss * t.prior_c;
    }
    case MCMCHIP_T_LOGISTIC_NAIVE: {
        // tests/logit.R:40-48
        double lp = 0.0, lq = 0.0, ss = 0.0;
        for (long i = g; i < t.n; i += G) {
            const double *xi = t.x + (size_t)i * d;
            double eta = 0.0;
            for (int j = 0; j < d; j++) eta = add_nofma(eta, mul_nofma(xi[j], v[j]));
            double e = exp(eta);
            double p = e / (1.0 + e);
            if (t.y[i] == 1.0) lp += log(p);
            else lq += log(1.0 - p);
        }
        lp = group_sum<G>(lp);
        lq = group_sum<G>(lq);
        double logl = lp + lq;
        if (t.prior_c == 0.0) return logl;
        for (int j = g; j < d; j += G) ss += v[j] * v[j];
        ss = group_sum<G>(ss);
        return logl - ss * t.prior_c;
    }
    case MCMCHIP_T_MVNORM: {
        const double *mu = t.par, *P = t.par + d;
        double q = 0.0;
        for (int i = g; i < d; i += G) {
            double ti = 0.0;
            for (int j = 0; j < d; j++) ti += P[(size_t)i * d + j] * (v[j] - mu[j]);
            q += (v[i] - mu[i]) * ti;
        }
        q = group_sum<G>(q);
        return -0.5 * q;
    }
    case MCMCHIP_T_NORMMIX:
        return normmix_group<G>(t.par, d, v, g);
    case MCMCHIP_T_SIMPLEX: {
        // tests/accept-batch.R:8: all(x >= 0) && sum(x) <= 1
        double s = 0.0;
        int ok = 1;
        for (int j = g; j < d; j += G) { ok &= (v[j] >= 0); s += v[j]; }
        ok = group_and<G>(ok);
        s = group_sum<G>(s);
        return (ok && s <= 1.0) ? t.par[0] : -INFINITY;
    }
    case MCMCHIP_T_BALL: {
        // tests/circle.R:11-16
        double s = 0.0;
        for (int j = g; j < d; j += G) s += v[j] * v[j];
        s = group_sum<G>(s);
        double fred = 1.0 - s;
        return (fred > 0) ? log(fred) : -INFINITY;
    }
    case MCMCHIP_T_BINOMIAL_LOGIT: {
        // vignettes/morph.Rnw:361-365: sum(x * MB) - sum(n * log(1 + exp(MB)))
        double s1 = 0.0, s2 = 0.0;
        for (long i = g; i < t.n; i += G) {
            const double *xi = t.x + (size_t)i * d;
            double mb = 0.0;
            for (int j = 0; j < d; j++) mb = add_nofma(mb, mul_nofma(xi[j], v[j]));
            s1 += t.y[i] * mb;
            s2 += t.par[i] * log(1 + exp(mb));
        }
        s1 = group_sum<G>(s1);
        s2 = group_sum<G>(s2);
        return s1 - s2;
    }
    case MCMCHIP_T_STUDENT_T: {
        // tests/morph.metrop.R:9: sum_j dt(x_j, df, log=TRUE)
        const double df = t.par[0];
        double s = 0.0;
        for (int j = g; j < d; j += G) s += log1p(v[j] * v[j] / df);
        s = group_sum<G>(s);
        return d * (lgamma((df + 1) / 2) - lgamma(df / 2) - 0.5 * log(df * M_PI)) - (df + 1) / 2 * s;
    }
    default:
        return NAN;
    }
}

// ---- morph (R/morph.R): f.inv and its derivative at a norm, the isotropic inverse and log.jacobian ----
__device__ inline double morph_expo_finv(const TargetView &t, double x)      // morph.R:77
{
    return x <= t.morph_r ? x : x + pow(x - t.morph_r, t.morph_p);
}
__device__ inline double morph_expo_dfinv(const TargetView &t, double x)     // morph.R:78
{
    return x <= t.morph_r ? 1.0 : 1.0 + t.morph_p * pow(x - t.morph_r, t.morph_p - 1.0);
}
__device__ inline double morph_sub_finv(const TargetView &t, double x)       // morph.R:53-55
{
    const double b = t.morph_b, e1 = 2.718281828459045;
    return x > 1 / b ? exp(b * x) - e1 / 3 : pow(x * b, 3.0) * e1 / 6 + x * b * e1 / 2;
}
__device__ inline double morph_sub_dfinv(const TargetView &t, double x)      // morph.R:56-58
{
    const double b = t.morph_b, e1 = 2.718281828459045;
    return x > 1 / b ? b * exp(b * x) : b * ((x * b) * (x * b)) * e1 / 2 + b * e1 / 2;
}
__device__ inline double morph_finv(const TargetView &t, double x)           // morph.R:133-152
{
    const bool ex = t.morph_flags & MCMCHIP_MORPH_EXPO, su = t.morph_flags & MCMCHIP_MORPH_SUBEXPO;
    if (ex && su) return morph_sub_finv(t, morph_expo_finv(t, x));
    if (ex) return morph_expo_finv(t, x);
    if (su) return morph_sub_finv(t, x);
    return x;
}
__device__ inline double morph_dfinv(const TargetView &t, double x)
{
    const bool ex = t.morph_flags & MCMCHIP_MORPH_EXPO, su = t.morph_flags & MCMCHIP_MORPH_SUBEXPO;
    if (ex && su) return morph_expo_dfinv(t, x) * morph_sub_dfinv(t, morph_expo_finv(t, x));
    if (ex) return morph_expo_dfinv(t, x);
    if (su) return morph_sub_dfinv(t, x);
    return 1.0;
}

// w = inverse(y) = f.inv(|y|) y/|y| + center (morph.R:5-14,162), written by the chain's G lanes into w (LDS,
// same wavefront: no barrier needed); returns |y|
template <int G> __device__ __forceinline__ double morph_inverse_group(const TargetView &t, const double *y, double *w, int g)
{
    const int d = t.d;
    const bool shaped = t.morph_flags & (MCMCHIP_MORPH_EXPO | MCMCHIP_MORPH_SUBEXPO);
    double nrm = 0.0, fac = 1.0;
    if (shaped) {
        double ss = 0.0;
        if (G == 1) { for (int j = 0; j < d; j++) ss += y[j] * y[j]; }       // the reference's summation order
        else { for (int j = g; j < d; j += G) ss += y[j] * y[j]; ss = group_sum<G>(ss); }
        nrm = sqrt(ss);
    }
    const double fi = (shaped && nrm != 0) ? morph_finv(t, nrm) : 0.0;
    for (int j = g; j < d; j += G) {
        double v = y[j];
        if (shaped) v = (nrm == 0) ? 0.0 : fi * y[j] / nrm;
        if (t.morph_flags & MCMCHIP_MORPH_CENTER) v += t.morph_center[j];
        w[j] = v;
    }
    return nrm;
}

// what logh() sees under morph.metrop (morph.R:166-179): lud(inverse(y)) + log.jacobian(y); -Inf, NaN and +Inf of
// the descriptor pass through unchanged for the caller's checks
template <int G> __device__ __forceinline__ double lud_morphed_group(const TargetView &t, const double *y, double *w, int g)
{
    if (!t.morph_flags) return lud_group<G>(t, y, g);
    const double nrm = morph_inverse_group<G>(t, y, w, g);
    const double foo = lud_group<G>(t, w, g);
    if (foo == -INFINITY || isnan(foo) || foo == INFINITY) return foo;
    if (!(t.morph_flags & (MCMCHIP_MORPH_EXPO | MCMCHIP_MORPH_SUBEXPO))) return foo;      // morph.R:131
    const int d = t.d;
    double lj;
    if (nrm == 0) lj = d * log(morph_dfinv(t, nrm));                                       // morph.R:22-23
    else lj = log(morph_dfinv(t, nrm)) + (d - 1) * (log(morph_finv(t, nrm)) - log(nrm));   // morph.R:25
    return foo + lj;
}

// model-selection logistic posterior (tests/temp-par.R:39-56, vignettes/bfst.Rnw:472-489), evaluated by the G
// lanes of one wavefront in lockstep (all lanes pass the same arguments; th may be LDS or global):
// par = k, k x p masks, k log pseudo-priors, constant.  Every lane returns the same value.
template <int G>
__device__ double lud_models_group(const double *par, int p, const double *x, const double *y, long n, double prior_c,
                                   int i, const double *th, int g)
{
    const int k = (int)par[0];
    const double *mask = par + 1 + (size_t)(i - 1) * p, *lpp = par + 1 + (size_t)k * p;
    double lp = 0.0, lq = 0.0;
#pragma unroll 2
    for (long r = g; r < n; r += G) {
        const double *xi = x + (size_t)r * p;
        double eta = 0.0;
        for (int j = 0; j < p; j++) eta = add_nofma(eta, mul_nofma(xi[j], mask[j] != 0 ? th[j] : 0.0));
        const double l1 = log1p(exp(-fabs(eta)));      // the reference's ifelse, branch-free (same operations)
        if (y[r] == 1.0) lp += (eta < 0) ? eta - l1 : -l1;
        else lq += (eta < 0) ? -l1 : -eta - l1;
    }
    lp = group_sum<G>(lp);
    lq = group_sum<G>(lq);
    double ss = 0.0;
    for (int j = 0; j < p; j++) ss += th[j] * th[j];
    return (lp + lq) - ss * prior_c + lpp[k] + lpp[i - 1];
}

// temper targets; state = (i, theta[p]) with 1-based component i.  Small p: every lane of the
// caller evaluates the whole thing (no cooperation), so this is a plain scalar function.
template <typename PT = const double *>
__device__ inline double lud_temper(int kind, int p, PT par, int ncomp, int i, const double *th)
{
    switch (kind) {
    case MCMCHIP_T_WITCH: {
        // tests/temp-par-witch.R:24-40.  par = bnd[k], inside[k], corner[k]: the two values the density
        // takes in component i were evaluated on the host (capi.hip) with the reference's expressions
        //   inside = -(d+1)log2 - log1p(-(1-bnd)^d),  corner = -(d+1)log2 - d log(1-bnd)
        bool out = false, all = true;
        const double bnd = par[i - 1];
        for (int j = 0; j < p; j++) {
            const double a = fabs(th[j]);
            out |= a > 1.0;
            all &= a > bnd;
        }
        if (out) return -INFINITY;
        return all ? par[2 * ncomp + i - 1] : par[ncomp + i - 1];
    }
    case MCMCHIP_T_TEMPERED_NORMMIX: {
        int k = (int)par[0];
        double beta = par[i];
        return beta * normmix_group<1, PT>(par + 1 + k, p, th, 0);
    }
    default:
        return NAN;
    }
}
