/*
 * diag_descriptor.c -- device versions of the two diagnostics; same signatures as the reference's natives
 *   SEXP initseq(SEXP x)                                              (src/initseq.c:7-128, called by R/initseq.R:4)
 *   void olbm(double *x, int *nin, int *pin, int *lin, double *mean, double *var, int *nocalcin)   (src/olbm.c:22-78)
 * so that R/initseq.R and R/olbm.R call them unchanged (C_initseq_gpu / C_olbm_gpu in place of C_initseq / C_olbm).
 * NOT COMPILED IN THIS REPOSITORY'S IMAGE (no R headers); see rglue/README.md.
 */
#include "glue_r.h"

#include <string.h>

SEXP initseq_gpu(SEXP x)
{
    if (!isNumeric(x)) error("argument must be numeric");
    /* initseq.c:15 tests the finiteness of the argument as given, so an integer vector is refused
     * ("argument must be real", src/isAllFinite.c:12-13): kept */
    if (!isReal(x)) error("argument must be real");
    SEXP xreal = PROTECT(coerceVector(x, REALSXP));
    const R_xlen_t len = XLENGTH(xreal);
    for (R_xlen_t i = 0; i < len; i++)
        if (!R_FINITE(REAL(xreal)[i])) error("all elements of argument must be finite");
    glue_init_device();
    int64_t cap = len / 2 > 0 ? (len / 2 < 4096 ? len / 2 : 4096) : 1, n_out = 0;
    double gamma0, vpos, vdec, vcon;
    double *Gpos = NULL, *Gdec = NULL, *Gcon = NULL;
    for (;;) {
        Gpos = (double *)R_alloc((size_t)cap * 3, sizeof(double));
        Gdec = Gpos + cap;
        Gcon = Gdec + cap;
        const int rc = mcmchip_initseq(REAL(xreal), (int64_t)len, 1, 0, cap, &n_out, &gamma0, Gpos, Gdec, Gcon, &vpos, &vdec, &vcon);
        if (rc == MCMCHIP_ECAP) { cap = n_out; continue; }        /* the required length came back in n_out */
        if (rc) glue_raise(rc);
        break;
    }
    const char *names[7];
    const int nslot = glue_initseq_result_names(names);           /* initseq.c:107-124 */
    SEXP result = PROTECT(allocVector(VECSXP, nslot));
    SEXP rnames = PROTECT(allocVector(STRSXP, nslot));
    for (int i = 0; i < nslot; i++) SET_STRING_ELT(rnames, i, mkChar(names[i]));
    namesgets(result, rnames);
    UNPROTECT(1);
    SET_VECTOR_ELT(result, 0, ScalarReal(gamma0));
    const double *seqs[3] = {Gpos, Gdec, Gcon};
    for (int s = 0; s < 3; s++) {
        SEXP v = allocVector(REALSXP, (R_xlen_t)n_out);
        SET_VECTOR_ELT(result, 1 + s, v);
        memcpy(REAL(v), seqs[s], sizeof(double) * (size_t)n_out);
    }
    SET_VECTOR_ELT(result, 4, ScalarReal(vpos));
    SET_VECTOR_ELT(result, 5, ScalarReal(vdec));
    SET_VECTOR_ELT(result, 6, ScalarReal(vcon));
    UNPROTECT(2);
    return result;
}

/* .C interface, argument types REALSXP, INTSXP, INTSXP, INTSXP, REALSXP, REALSXP, LGLSXP as init.c:7-8 */
void olbm_gpu(double *x, int *nin, int *pin, int *lin, double *mean, double *var, int *nocalcin)
{
    glue_init_device();
    if (lin[0] > nin[0]) error("len > n\n");                      /* olbm.c:31-32 */
    const int rc = mcmchip_olbm(x, (int64_t)nin[0], (int32_t)pin[0], (int64_t)lin[0], mean, var, nocalcin[0] != 0, 0);
    if (rc) glue_raise(rc);
}
