/* capi_demo.c -- the C-ABI of libmcmchip.so used from plain C, the host language of the reference's
 * .Call layer (the C files of package/mcmc/src): no Python, no torch, only include/mcmchip.h.
 *
 *   capi_demo data.bin n d nchain nbatch blen seed
 *
 * data.bin holds n*d doubles of the model matrix (row-major) followed by n doubles of 0/1 responses.
 * Runs metrop() on the logistic posterior (vignettes/demo.Rnw:59-65 with a N(0, 2^2) prior) for `nchain`
 * chains with device Philox numbers, then initseq() and olbm() on the first chain's batch means of the first
 * coordinate, and prints every number with 17 significant digits, one "name: values" line each
 * (tests/test_gpu_parity.py::test_c_demo_matches_python_api compares them with the Python mirror's).
 * Build: see examples/Makefile. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mcmchip.h"

static void die(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, mcmchip_last_error());
    exit(1);
}

static void print_vec(const char *name, const double *v, long n)
{
    printf("%s:", name);
    for (long i = 0; i < n; i++) printf(" %.17g", v[i]);
    printf("\n");
}

int main(int argc, char **argv)
{
    if (argc != 8) {
        fprintf(stderr, "usage: %s data.bin n d nchain nbatch blen seed\n", argv[0]);
        return 2;
    }
    const long n = atol(argv[2]);
    const int d = atoi(argv[3]);
    const long nchain = atol(argv[4]);
    const int nbatch = atoi(argv[5]), blen = atoi(argv[6]);
    const unsigned long long seed = strtoull(argv[7], NULL, 10);

    double *x = malloc(sizeof(double) * n * d), *y = malloc(sizeof(double) * n);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(x, sizeof(double), (size_t)n * d, f) != (size_t)n * d || fread(y, sizeof(double), n, f) != (size_t)n) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    fclose(f);

    if (mcmchip_init(0)) die("mcmchip_init");
    printf("version: %s\n", mcmchip_version());

    mcmchip_target_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.kind = MCMCHIP_T_LOGISTIC;
    desc.d = d;
    desc.n = n;
    desc.x = x;
    desc.y = y;
    desc.prior_c = 0.125;                      /* sum(beta^2) / (2 * 2^2) */
    mcmchip_target *tg = NULL;
    if (mcmchip_target_create(&desc, 0, &tg)) die("mcmchip_target_create");

    double *init = calloc((size_t)nchain * d, sizeof(double));
    double scale = 0.4;
    mcmchip_metrop_args a;
    memset(&a, 0, sizeof a);
    a.nchain = nchain; a.nbatch = nbatch; a.blen = blen; a.nspac = 1;
    a.initial = init;
    a.scale_kind = MCMCHIP_SCALE_CONSTANT; a.scale = &scale;
    a.out_kind = MCMCHIP_OUT_IDENTITY;
    a.rng_mode = MCMCHIP_RNG_PHILOX; a.seed = seed;
    a.path = MCMCHIP_PATH_STREAM;

    mcmchip_metrop_out o;
    memset(&o, 0, sizeof o);
    o.accept = malloc(sizeof(double) * nchain);
    o.batch = malloc(sizeof(double) * nchain * nbatch * d);
    o.final = malloc(sizeof(double) * nchain * d);
    o.accept_batch = malloc(sizeof(double) * nchain * nbatch);
    if (mcmchip_metrop(tg, &a, &o)) die("mcmchip_metrop");
    print_vec("accept", o.accept, nchain);
    print_vec("final", o.final, nchain * d);

    /* a rejected call reports the reference's message and leaves the library usable (metrop.c:60-61) */
    a.nbatch = 0;
    if (mcmchip_metrop(tg, &a, &o) == 0) { fprintf(stderr, "nbatch = 0 was accepted\n"); return 1; }
    printf("error: %s\n", mcmchip_last_error());
    a.nbatch = nbatch;

    /* the first coordinate's batch means of chain 0, centred as R/initseq.R:4 does */
    double *series = malloc(sizeof(double) * nbatch), mean = 0.0;
    for (int i = 0; i < nbatch; i++) { series[i] = o.batch[(size_t)i * d]; mean += series[i]; }
    mean /= nbatch;
    for (int i = 0; i < nbatch; i++) series[i] -= mean;
    const long cap = nbatch / 2 > 0 ? nbatch / 2 : 1;
    int64_t m = 0;
    double g0, vpos, vdec, vcon;
    double *Gp = malloc(sizeof(double) * cap), *Gd = malloc(sizeof(double) * cap), *Gc = malloc(sizeof(double) * cap);
    if (mcmchip_initseq(series, nbatch, 1, 0, cap, &m, &g0, Gp, Gd, Gc, &vpos, &vdec, &vcon)) die("mcmchip_initseq");
    printf("initseq.m: %lld\n", (long long)m);
    print_vec("gamma0", &g0, 1);
    print_vec("Gamma.pos", Gp, (long)m);
    print_vec("Gamma.con", Gc, (long)m);
    print_vec("var.con", &vcon, 1);

    /* olbm on chain 0's batch means (nbatch x d: the C-ABI wants it column-major, x[t + n*j]) */
    double *bm = malloc(sizeof(double) * nbatch * d), *mu = malloc(sizeof(double) * d), *var = malloc(sizeof(double) * d * d);
    for (int i = 0; i < nbatch; i++)
        for (int j = 0; j < d; j++) bm[i + (size_t)nbatch * j] = o.batch[(size_t)i * d + j];
    const long len = nbatch / 5 > 0 ? nbatch / 5 : 1;
    if (mcmchip_olbm(bm, nbatch, d, len, mu, var, 0, 0)) die("mcmchip_olbm");
    print_vec("olbm.mean", mu, d);
    print_vec("olbm.var", var, (long)d * d);

    mcmchip_target_destroy(tg);
    return 0;
}
