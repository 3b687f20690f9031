/* oracle.h -- TEST INFRASTRUCTURE: CPU restatement of the reference hot path (see oracle.c). */
#ifndef ORC_ORACLE_H
#define ORC_ORACLE_H
#include <stdint.h>
#include "../include/mcmchip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* oracle-only rng mode: draws popped sequentially from an emulated R stream (rrng.c) */
#define ORC_RNG_RSTREAM 3

/* oracle-only temper output functionals (the R closures of the witch tests) */
#define ORC_TOUT_NONE 0
#define ORC_TOUT_WITCH_PAR 1
#define ORC_TOUT_WITCH_SER 2
#define ORC_TOUT_INDEX 3      /* a->out_index mask over the identity output */
#define ORC_TOUT_MOMENTS2 4   /* c(v, v^2) of the identity output */

typedef struct orc_rstream orc_rstream;
orc_rstream *orc_r_new(int kind /*0 Mersenne-Twister, 1 Marsaglia-Multicarry*/, uint32_t seed);
void orc_r_free(orc_rstream *r);
orc_rstream *orc_r_clone(const orc_rstream *r);
uint64_t orc_r_count(const orc_rstream *r);
double orc_r_unif(orc_rstream *r);
double orc_r_norm(orc_rstream *r);
double orc_r_qnorm(double p);
void orc_r_norm_fill(orc_rstream *r, double *out, int64_t n);
void orc_r_unif_fill(orc_rstream *r, double *out, int64_t n);

void orc_philox_normals(uint64_t seed, uint64_t chain, uint64_t iter, int d, double *z);
double orc_philox_uniform(uint64_t seed, uint64_t chain, uint64_t iter);

double orc_lud(const mcmchip_target_desc *t, const double *x);
/* morph (R/morph.R): scalar maps and the isotropic vector maps around them */
double orc_morph_f(const mcmchip_target_desc *t, double x);
double orc_morph_finv(const mcmchip_target_desc *t, double x);
double orc_morph_dfinv(const mcmchip_target_desc *t, double x);
void orc_morph_inverse(const mcmchip_target_desc *t, const double *y, double *x);
void orc_morph_transform(const mcmchip_target_desc *t, const double *x, double *y);
double orc_morph_logjac(const mcmchip_target_desc *t, const double *y);
double orc_lud_temper(const mcmchip_target_desc *t, int i, const double *theta);
void orc_logdens(const mcmchip_target_desc *t, const double *states, int64_t nchain, double *out);

int orc_metrop(const mcmchip_target_desc *t, const mcmchip_metrop_args *a, mcmchip_metrop_out *o,
               orc_rstream *rs, int nthreads, char *err, int errlen);
int orc_temper(const mcmchip_target_desc *t, const mcmchip_temper_args *a, mcmchip_temper_out *o,
               orc_rstream *rs, int out_kind, int nthreads, char *err, int errlen);
int64_t orc_initseq(const double *x, int64_t len, int64_t cap, double *gamma0, double *Gpos,
                    double *Gdec, double *Gcon, double *var_pos, double *var_dec, double *var_con);
int orc_olbm(const double *x, int64_t n, int p, int64_t len, double *mean, double *var, int nocalc);

#ifdef __cplusplus
}
#endif
#endif
