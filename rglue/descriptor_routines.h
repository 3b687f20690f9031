/*
 * descriptor_routines.h -- the rows that the device entry points add to the package's registration tables.
 * src/init.c of the package (reference: init.c:7-20) gains three lines:
 *
 *     #include "descriptor_routines.h"
 *     ... in cMethods[],    before its {NULL, NULL, 0, NULL} row:   MCMC_DESCRIPTOR_C_ROWS
 *     ... in callMethods[], before its {NULL, NULL, 0} row:         MCMC_DESCRIPTOR_CALL_ROWS
 *
 * With `.fixes = "C_"` in NAMESPACE (NAMESPACE:2) R code then sees C_metrop_descriptor, C_temper_descriptor,
 * C_initseq_gpu and C_olbm_gpu next to the reference's C_metrop, C_temper, C_initseq and C_olbm.
 * NOT COMPILED IN THIS REPOSITORY'S IMAGE (no R headers); see rglue/README.md.
 */
#ifndef MCMC_DESCRIPTOR_ROUTINES_H
#define MCMC_DESCRIPTOR_ROUTINES_H

#include <Rinternals.h>
#include <R_ext/Rdynload.h>

SEXP metrop_descriptor(SEXP desc, SEXP initial, SEXP nbatch, SEXP blen, SEXP nspac, SEXP scale, SEXP func2, SEXP debug,
                       SEXP seed);
SEXP temper_descriptor(SEXP desc, SEXP initial, SEXP neighbors, SEXP nbatch, SEXP blen, SEXP nspac, SEXP scale,
                       SEXP func2, SEXP debug, SEXP parallel, SEXP seed);
SEXP initseq_gpu(SEXP x);
void olbm_gpu(double *x, int *nin, int *pin, int *lin, double *mean, double *var, int *nocalcin);

/* olbm_gpu takes the argument types of olbm: the table's existing olbm_types array is reused */
#define MCMC_DESCRIPTOR_C_ROWS \
    {"olbm_gpu", (DL_FUNC) &olbm_gpu, 7, olbm_types},

#define MCMC_DESCRIPTOR_CALL_ROWS \
    {"metrop_descriptor", (DL_FUNC) &metrop_descriptor, 9}, \
    {"temper_descriptor", (DL_FUNC) &temper_descriptor, 11}, \
    {"initseq_gpu", (DL_FUNC) &initseq_gpu, 1},

#endif
