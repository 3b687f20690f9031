/*
 * glue_r.h -- helpers shared by the SEXP wrappers.  NOT COMPILED IN THIS REPOSITORY'S IMAGE: there is no R and no R
 * header here (R.h, Rinternals.h); these files are built by R CMD INSTALL as part of the package's src/ (rglue/README.md).
 * Everything that does not need an R header lives in glue_core.c and is compiled and tested here.
 */
#ifndef MCMC_GLUE_R_H
#define MCMC_GLUE_R_H

#include <R.h>
#include <Rinternals.h>
#include "glue_core.h"

/* element `name` of a named list, or R_NilValue */
SEXP glue_list_get(SEXP list, const char *name);
/* unpack an object of class "mcmc_descriptor" (R/descriptor.R) into plain fields; scratch comes from R_alloc */
void glue_fields_from_sexp(SEXP desc, glue_fields *f);
/* the descriptor's device-resident form: created on first use, cached as an external pointer in desc$cache (an
 * environment) and destroyed by its finalizer (mcmchip_target_destroy) */
mcmchip_target *glue_target(SEXP desc);
/* (key, iteration offset, first global chain/ensemble id) from the numeric vector c(key.hi, key.lo, iter, unit) */
void glue_seed_from_sexp(SEXP seed, uint64_t *key, uint64_t *iter_offset, uint64_t *unit_offset);
SEXP glue_seed_to_sexp(uint64_t key, uint64_t iter_offset, uint64_t unit_offset);
/* select the device once (environment variable MCMCHIP_DEVICE, default 0) and install the interrupt poll */
void glue_init_device(void);
int glue_device_list(int *devices, int cap);   /* MCMCHIP_DEVICES, comma separated */
/* release what the caller lists, then raise the library's last error as an R error */
void glue_raise(int rc);

#endif
