/*
 * glue_core.h -- the part of the R glue that needs no R header: everything between the SEXP wrappers
 * (metrop_descriptor.c, temper_descriptor.c, diag_descriptor.c) and the C-ABI of include/mcmchip.h, on plain arrays.
 * Compiled with gcc and tested from tests/test_rglue_cpu.py (ctypes); the SEXP wrappers only unpack R objects, call
 * these functions, call libmcmchip and pack the result list.
 *
 * What it restates from the reference (argument rules and error texts, so that a descriptor call fails the way a
 * closure call does):
 *   scalar arguments       src/getScalarInteger.c, src/getScalarLogical.c, metrop.c:54-65, temper.c:66-76
 *   metrop scale           metrop.c:333-365 (proposal_setup)
 *   temper scale           temper.c:78-87, 915-953 (check_valid_scale)
 *   outfun index vectors   metrop.c:416-503 (out_setup)
 *   result lists           metrop.c:124-128,174-179; temper.c:196-277,282-464; initseq.c:107-124
 */
#ifndef MCMC_GLUE_CORE_H
#define MCMC_GLUE_CORE_H

#include <stdint.h>
#include "../include/mcmchip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GLUE_OK 0
#define GLUE_ERROR 1            /* message in err: the wrapper passes it to Rf_error() after releasing its buffers */

/* ---- 1. a descriptor object (an R list of class "mcmc_descriptor", R/descriptor.R) as plain fields ---------------
 * matrices are column-major as R stores them; absent fields are NULL / 0.  */
typedef struct glue_fields {
    const char *kind;            /* "logistic", "logistic.naive", "mvnorm", "normmix", "simplex", "ball", "witch",
                                    "tempered.normmix", "student.t", "logistic.models", "binomial.logit" */
    int d;                       /* dimension of the state (of theta for tempering kinds) */
    const double *x;             /* design matrix, x_nrow x x_ncol, column-major */
    int x_nrow, x_ncol;
    const double *y;             /* response (0/1, or success counts for binomial.logit) */
    int y_len;
    double prior_sd;             /* logistic kinds: iid N(0, prior_sd^2) prior on the coefficients; <= 0: flat */
    const double *mu;            /* mvnorm: mean (d) */
    const double *precision;     /* mvnorm: precision matrix d x d (symmetric, so the storage order is moot) */
    int nmix;                    /* normmix / tempered.normmix: number of mixture components K */
    const double *weights;       /* K */
    const double *means;         /* K x d, column-major */
    const double *sds;           /* K (isotropic components) */
    const double *bounds;        /* witch: k bounds in (0, 1) */
    const double *betas;         /* tempered.normmix: k inverse temperatures */
    int ncomp;                   /* witch / tempered.normmix / logistic.models: number of tempering components k */
    const int *models;           /* logistic.models: k x d 0/1 matrix, column-major (column 1 = intercept) */
    const double *log_pseudo_prior;   /* logistic.models: k values or NULL (zeros) */
    double constant;             /* logistic.models: additive constant */
    double df;                   /* student.t */
    const double *totals;        /* binomial.logit: x_nrow totals */
    double simplex_value;        /* simplex: the constant log density inside the simplex */
    /* morph (R/morph.R): any of b, r/p, center */
    int has_b, has_r, has_center;
    double morph_b, morph_r, morph_p;
    const double *morph_center;  /* d values, or 1 value recycled (morph.R:6-8) */
    int morph_center_len;
} glue_fields;

/* the C-ABI descriptor plus the buffers it points into (row-major X, the parameter block, the centre) */
typedef struct glue_descriptor {
    mcmchip_target_desc desc;
    double *x_rowmajor;
    double *par;
    double *center;
} glue_descriptor;

int glue_make_descriptor(const glue_fields *f, glue_descriptor *out, char *err, int errlen);
void glue_free_descriptor(glue_descriptor *g);

/* ---- 2. scalar arguments (getScalarInteger.c / getScalarLogical.c and the positivity checks of the callers) ------ */
int glue_scalar_integer(int is_numeric, int len, double value, const char *argname, int must_be_positive, int *out,
                        char *err, int errlen);
int glue_scalar_logical(int is_logical, int len, int value, const char *argname, int *out, char *err, int errlen);

/* ---- 3. proposal scale ------------------------------------------------------------------------------------------ */
/* metrop.c:333-365: d x d matrix -> FULL, length d -> DIAGONAL, length 1 -> CONSTANT (a 1 x 1 matrix with d == 1 is
 * FULL; a length-1 vector with d == 1 is DIAGONAL: the reference tests LENGTH == d first) */
int glue_metrop_scale(int is_numeric, int len, int is_matrix, int nrow, int ncol, int d, int *scale_kind, char *err,
                      int errlen);
/* temper.c:915-953 for one scale object (comp = -1) or element `comp` (0-based) of a list scale */
int glue_temper_scale(int is_double, const double *v, int len, int is_matrix, int nrow, int ncol, int comp, int nx,
                      int *scale_kind, char *err, int errlen);

/* ---- 4. output functional of metrop (metrop.c:416-503) ----------------------------------------------------------- */
enum { GLUE_OUTFUN_NULL = 0, GLUE_OUTFUN_LOGICAL = 1, GLUE_OUTFUN_NUMERIC = 2, GLUE_OUTFUN_FUNCTION = 3,
       GLUE_OUTFUN_OTHER = 4, GLUE_OUTFUN_MOMENTS2 = 5 /* the string "moments2": c(x, x^2) on the device */ };
/* mask: d entries (filled for LOGICAL / NUMERIC); nout: dimension of g(x).  An R function cannot run on the device:
 * GLUE_OUTFUN_FUNCTION is refused with a message that says so. */
int glue_metrop_outfun(int type, const int *lgl, const double *num, int len, int d, int *out_kind, int32_t *mask,
                       int *nout, char *err, int errlen);

/* ---- 5. result lists: slot names in the reference's order; return the number of slots ---------------------------- */
int glue_metrop_result_names(int debug, const char **names /* room for 11 */);
int glue_temper_result_names(int parallel, int debug, const char **names /* room for 16 */);
int glue_initseq_result_names(const char **names /* room for 7 */);

/* ---- 6. layouts ---------------------------------------------------------------------------------------------------- */
/* R matrix (column-major nrow x ncol) -> row-major, as the C-ABI wants X and chain-major `initial` */
void glue_colmajor_to_rowmajor(const double *a, long nrow, long ncol, double *out);
/* tempering `neighbors`: R logical k x k (int, NA_LOGICAL = INT_MIN counts as TRUE like in C) -> validated 0/1 matrix;
 * temper.c:29-41 */
int glue_neighbors(int is_logical, int is_matrix, int nrow, int ncol, const int *lgl, int32_t *out, char *err, int errlen);
/* serial / parallel shape of `initial` (temper.c:43-64): returns nx (dimension of theta) in *nx */
int glue_temper_initial(int is_double, const double *v, long len, int is_matrix, int nrow, int ncol, int parallel,
                        int ncomp, int *nx, char *err, int errlen);

#ifdef __cplusplus
}
#endif
#endif
