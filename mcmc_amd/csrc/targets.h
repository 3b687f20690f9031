// targets.h -- device evaluation of the log unnormalised density descriptors that replace the
// reference's R closure (logh(), metrop.c:303-323 / temper.c:955-975).
//
// A density is evaluated cooperatively by the G lanes (G = 1, 4, 16 or 64, all inside one
// wavefront) that own one chain: each lane takes a strided slice of the observations /
// matrix rows / mixture components and the partial sums are combined with an xor butterfly of
// DPP shuffles, which leaves the bit-identical total on every lane, so all G lanes take the
// same accept/reject decision without a broadcast.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mcmchip.h"

struct TargetView {
    int kind, d;
    long n;
    const double *x, *y;   // may point into LDS
    const double *par;
    double prior_c;
};

template <int G> __device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, G);
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
template <int G> __device__ double normmix_group(const double *par, int d, const double *v, int g)
{
    int K = (int)par[0];
    const double *rec = par + 1;
    // pass 1: max of the exponents, pass 2: sum of exp (components strided over lanes)
    double m = -INFINITY;
    for (int k = g; k < K; k += G) {
        const double *r = rec + (size_t)k * (2 + d);
        double s = 0.0;
        for (int j = 0; j < d; j++) { double t = v[j] - r[2 + j]; s += t * t; }
        m = fmax(m, r[0] - r[1] * s);
    }
    m = group_max<G>(m);
    if (m == -INFINITY) return -INFINITY;
    double acc = 0.0;
    for (int k = g; k < K; k += G) {
        const double *r = rec + (size_t)k * (2 + d);
        double s = 0.0;
        for (int j = 0; j < d; j++) { double t = v[j] - r[2 + j]; s += t * t; }
        acc += exp(r[0] - r[1] * s - m);
    }
    acc = group_sum<G>(acc);
    return m + log(acc);
}

// metrop targets; v = state[d]
template <int G> __device__ double lud_group(const TargetView &t, const double *v, int g)
{
    const int d = t.d;
    switch (t.kind) {
    case MCMCHIP_T_LOGISTIC: {
        // vignettes/demo.Rnw:59-65; eta accumulated j-ascending with separately rounded
        // multiply and add, like the reference's x %*% beta
        double lp = 0.0, lq = 0.0, ss = 0.0;
        for (long i = g; i < t.n; i += G) {
            const double *xi = t.x + (size_t)i * d;
            double eta = 0.0;
            for (int j = 0; j < d; j++) eta = add_nofma(eta, mul_nofma(xi[j], v[j]));
            if (t.y[i] == 1.0) lp += (eta < 0) ? eta - log1p(exp(eta)) : -log1p(exp(-eta));
            else lq += (eta < 0) ? -log1p(exp(eta)) : -eta - log1p(exp(-eta));
        }
        for (int j = g; j < d; j += G) ss += v[j] * v[j];
        lp = group_sum<G>(lp);
        lq = group_sum<G>(lq);
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
    default:
        return NAN;
    }
}

// temper targets; state = (i, theta[p]) with 1-based component i.  Small p: every lane of the
// caller evaluates the whole thing (no cooperation), so this is a plain scalar function.
__device__ inline double lud_temper(int kind, int p, const double *par, int ncomp, int i, const double *th)
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
        return beta * normmix_group<1>(par + 1 + k, p, th, 0);
    }
    default:
        return NAN;
    }
}
