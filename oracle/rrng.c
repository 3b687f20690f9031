/*
 * rrng.c -- TEST INFRASTRUCTURE (oracle).  Not part of the product; nothing under
 * mcmc_amd/ may link or call this.
 *
 * Emulation of the R random number streams that the reference's tests and
 * transcripts were produced with, so that the reference's own printed goldens
 * (tests/NAME.Rout.save, devel/bug/foo.Rout) can pin the oracle without R.
 *
 * R itself is a third-party dependency that is absent from /root/reference
 * (DESCRIPTION:8 "Depends: R (>= 3.0.2)"; goldens were produced by R 3.3.3 - 3.6.3).
 * What is restated here is R's published algorithm (src/main/RNG.c, src/nmath/snorm.c,
 * src/nmath/qnorm.c of those R versions):
 *   - set.seed(): initial scrambling seed = 69069*seed+1, 50 times, then one more LCG
 *     step per seed word;
 *   - "Mersenne-Twister" (default kind): MT19937, 32-bit output * 2.3283064365386963e-10,
 *     with R's fixup into the open interval (0,1);
 *   - "Marsaglia-Multicarry": I1 = 36969*(I1&0xffff)+(I1>>16), I2 = 18000*(I2&0xffff)+(I2>>16),
 *     ((I1<<16)^(I2&0xffff)) * 2.328306437080797e-10, same fixup;
 *   - normal.kind "Inversion": u = unif_rand(); u = (int)(2^27*u) + unif_rand();
 *     qnorm(u/2^27) with Wichura's AS 241 (PPND16).
 * Call sites in the reference: metrop.c:232,378; temper.c:537,553,585,649,681-682,753,791,821,996.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define R_MT 0
#define R_MM 1

typedef struct orc_rstream {
    int kind;
    uint32_t mt[624];
    int mti;
    uint32_t i1, i2;
    uint64_t n_unif; /* underlying uniforms consumed */
} orc_rstream;

static const double i2_32m1 = 2.328306437080797e-10; /* 1/(2^32 - 1) */

static double fixup(double x)
{
    if (x <= 0.0) return 0.5 * i2_32m1;
    if ((1.0 - x) <= 0.0) return 1.0 - 0.5 * i2_32m1;
    return x;
}

orc_rstream *orc_r_new(int kind, uint32_t seed)
{
    orc_rstream *r = (orc_rstream *)calloc(1, sizeof(orc_rstream));
    r->kind = kind;
    for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
    if (kind == R_MT) {
        /* 625 seed words: word 0 is the position (forced to 624 by FixupSeeds), 1..624 the state */
        seed = 69069u * seed + 1u; /* word 0, discarded */
        for (int j = 0; j < 624; j++) {
            seed = 69069u * seed + 1u;
            r->mt[j] = seed;
        }
        r->mti = 624;
    } else {
        seed = 69069u * seed + 1u; r->i1 = seed;
        seed = 69069u * seed + 1u; r->i2 = seed;
        if (r->i1 == 0) r->i1 = 1;
        if (r->i2 == 0) r->i2 = 1;
    }
    return r;
}

void orc_r_free(orc_rstream *r) { free(r); }

orc_rstream *orc_r_clone(const orc_rstream *r)
{
    orc_rstream *c = (orc_rstream *)malloc(sizeof(orc_rstream));
    memcpy(c, r, sizeof(orc_rstream));
    return c;
}

uint64_t orc_r_count(const orc_rstream *r) { return r->n_unif; }

static uint32_t mt_next(orc_rstream *r)
{
    static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
    uint32_t y;
    if (r->mti >= 624) {
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            y = (r->mt[kk] & 0x80000000u) | (r->mt[kk + 1] & 0x7fffffffu);
            r->mt[kk] = r->mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < 623; kk++) {
            y = (r->mt[kk] & 0x80000000u) | (r->mt[kk + 1] & 0x7fffffffu);
            r->mt[kk] = r->mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        y = (r->mt[623] & 0x80000000u) | (r->mt[0] & 0x7fffffffu);
        r->mt[623] = r->mt[396] ^ (y >> 1) ^ mag01[y & 1u];
        r->mti = 0;
    }
    y = r->mt[r->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

double orc_r_unif(orc_rstream *r)
{
    r->n_unif++;
    if (r->kind == R_MT) {
        return fixup((double)mt_next(r) * 2.3283064365386963e-10);
    } else {
        r->i1 = 36969u * (r->i1 & 0177777u) + (r->i1 >> 16);
        r->i2 = 18000u * (r->i2 & 0177777u) + (r->i2 >> 16);
        return fixup((double)((r->i1 << 16) ^ (r->i2 & 0177777u)) * i2_32m1);
    }
}

/* Wichura (1988) AS 241, PPND16: lower tail, non-log.
 * Checked against an independent inverse normal cdf (tests/test_oracle_golden.py): the
 * central and middle branches (p >= exp(-25), i.e. all but 3e-11 of the draws) agree to
 * 1e-14; the far-tail branch (r > 5) as restated here agrees only to ~3e-7 absolute, so one
 * of its coefficients may differ from R's in a late digit.  No golden exercises it. */
double orc_r_qnorm(double p)
{
    double q = p - 0.5, r, val;
    if (fabs(q) <= 0.425) {
        r = .180625 - q * q;
        val = q * (((((((r * 2509.0809287301226727 +
                         33430.575583588128105) * r + 67265.770927008700853) * r +
                       45921.953931549871457) * r + 13731.693765509461125) * r +
                     1971.5909503065514427) * r + 133.14166789178437745) * r +
                   3.387132872796366608) /
              (((((((r * 5226.495278852545925 +
                     28729.085735721942674) * r + 39307.89580009271061) * r +
                   21213.794301586595867) * r + 5394.1960214247511077) * r +
                 687.1870074920579083) * r + 42.313330701600911252) * r + 1.);
        return val;
    }
    r = (q < 0) ? p : 1.0 - p;
    r = sqrt(-log(r));
    if (r <= 5.) {
        r += -1.6;
        val = (((((((r * 7.7454501427834140764e-4 +
                     .0227238449892691845833) * r + .24178072517745061177) * r +
                   1.27045825245236838258) * r + 3.64784832476320460504) * r +
                 5.7694972214606914055) * r + 4.6303378461565452959) * r +
               1.42343711074968357734) /
              (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r +
                    .0151986665636164571966) * r + .14810397642748007459) * r +
                  .68976733498510000455) * r + 1.6763848301838038494) * r +
                2.05319162663775882187) * r + 1.);
    } else {
        r += -5.;
        val = (((((((r * 2.01033439929228813265e-7 +
                     2.71155556874348757815e-5) * r + .0012426609473880784386) * r +
                   .026532189526576123093) * r + .29656057182850489123) * r +
                 1.7848265399172913358) * r + 5.4637849111641143699) * r +
               6.6579046435011037772) /
              (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r +
                    1.8463183175100546818e-5) * r + 7.868691311456132591e-4) * r +
                  .0148753612908506148525) * r + .13692988092273580531) * r +
                .59983224010985833) * r + 1.);
    }
    if (q < 0.0) val = -val;
    return val;
}

double orc_r_norm(orc_rstream *r)
{
    const double BIG = 134217728.0; /* 2^27 */
    double u = orc_r_unif(r);
    u = (int)(BIG * u) + orc_r_unif(r);
    return orc_r_qnorm(u / BIG);
}

/* bulk helpers for the python side (rnorm(n) / runif(n)) */
void orc_r_norm_fill(orc_rstream *r, double *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) out[i] = orc_r_norm(r);
}
void orc_r_unif_fill(orc_rstream *r, double *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) out[i] = orc_r_unif(r);
}
