/*
 * metrop_descriptor.c -- .Call entry point for metrop() on a log-density DESCRIPTOR: the device-path counterpart of
 * SEXP metrop(func1, initial, nbatch, blen, nspac, scale, func2, debug, rho1, rho2) (src/metrop.c:19-301).
 * Same argument checks and error texts (through glue_core.c), same result list (names and order of
 * metrop.c:124-128,174-179, same C-side shapes: batch is dim_out x nbatch, the trace matrices d x niter; the R method
 * transposes them exactly like R/metrop.R:61-65).
 * Batched extension: `initial` may be an nchain x d matrix; every result then gains a trailing chain dimension.
 * NOT COMPILED IN THIS REPOSITORY'S IMAGE (no R headers); see rglue/README.md.
 */
#include "glue_r.h"

#include <string.h>

static SEXP alloc_real_dims(int n1, int n2, int n3 /* 0: matrix / vector */, int as_vector)
{
    if (as_vector) return allocVector(REALSXP, (R_xlen_t)n1);
    if (n3 <= 0) return allocMatrix(REALSXP, n1, n2);
    return alloc3DArray(REALSXP, n1, n2, n3);
}

SEXP metrop_descriptor(SEXP desc, SEXP initial, SEXP nbatch, SEXP blen, SEXP nspac, SEXP scale, SEXP func2, SEXP debug,
                       SEXP seed)
{
    char msg[256];
    int int_nbatch, int_blen, int_nspac, int_debug;

    /* argument checks in the reference's order (metrop.c:40-74) */
    if (!isNumeric(initial)) error("argument \"initial\" must be numeric");
    if (glue_scalar_integer(isNumeric(nbatch), LENGTH(nbatch), isNumeric(nbatch) && LENGTH(nbatch) == 1 ? asReal(nbatch) : 0.0,
                            "nbatch", 0, &int_nbatch, msg, sizeof msg)) error("%s", msg);
    if (glue_scalar_integer(isNumeric(blen), LENGTH(blen), isNumeric(blen) && LENGTH(blen) == 1 ? asReal(blen) : 0.0,
                            "blen", 0, &int_blen, msg, sizeof msg)) error("%s", msg);
    if (glue_scalar_integer(isNumeric(nspac), LENGTH(nspac), isNumeric(nspac) && LENGTH(nspac) == 1 ? asReal(nspac) : 0.0,
                            "nspac", 0, &int_nspac, msg, sizeof msg)) error("%s", msg);
    if (!isNumeric(scale)) error("argument \"scale\" must be numeric");
    if (glue_scalar_logical(isLogical(debug), LENGTH(debug), isLogical(debug) && LENGTH(debug) == 1 ? LOGICAL(debug)[0] : 0,
                            "debug", &int_debug, msg, sizeof msg)) error("%s", msg);
    if (int_nbatch <= 0) error("argument \"nbatch\" must be positive");
    if (int_blen <= 0) error("argument \"blen\" must be positive");
    if (int_nspac <= 0) error("argument \"nspac\" must be positive");

    mcmchip_target *target = glue_target(desc);                 /* selects the device on first use */
    glue_fields fields;
    glue_fields_from_sexp(desc, &fields);
    const int batched = isMatrix(initial);
    SEXP state = PROTECT(coerceVector(initial, REALSXP));
    const int d = batched ? ncols(state) : LENGTH(state);
    const int nchain = batched ? nrows(state) : 1;
    for (R_xlen_t i = 0; i < XLENGTH(state); i++)
        if (!R_FINITE(REAL(state)[i])) error("all elements of \"state\" must be finite");
    {
        const int dd = fields.d > 0 ? fields.d : fields.x_ncol;
        if (dd != d) error("length of \"initial\" (%d) differs from the dimension of the descriptor (%d)", d, dd);
    }

    /* scale (metrop.c:333-365) */
    SEXP rscale = PROTECT(coerceVector(scale, REALSXP));
    int scale_kind;
    if (glue_metrop_scale(1, LENGTH(rscale), isMatrix(scale), isMatrix(scale) ? nrows(scale) : 0,
                          isMatrix(scale) ? ncols(scale) : 0, d, &scale_kind, msg, sizeof msg)) error("%s", msg);

    /* outfun (metrop.c:416-503) */
    int out_kind, nout, type;
    int32_t *mask = (int32_t *)R_alloc((size_t)d, sizeof(int32_t));
    const int *lgl = NULL;
    const double *num = NULL;
    int flen = 0;
    SEXP rfun = R_NilValue;
    if (func2 == R_NilValue) type = GLUE_OUTFUN_NULL;
    else if (isFunction(func2)) type = GLUE_OUTFUN_FUNCTION;
    else if (isString(func2) && LENGTH(func2) == 1 && strcmp(CHAR(STRING_ELT(func2, 0)), "moments2") == 0) type = GLUE_OUTFUN_MOMENTS2;
    else if (isLogical(func2)) { type = GLUE_OUTFUN_LOGICAL; lgl = LOGICAL(func2); flen = LENGTH(func2); }
    else if (isNumeric(func2)) { type = GLUE_OUTFUN_NUMERIC; rfun = coerceVector(func2, REALSXP); }
    else type = GLUE_OUTFUN_OTHER;
    PROTECT(rfun);
    if (type == GLUE_OUTFUN_NUMERIC) { num = REAL(rfun); flen = LENGTH(rfun); }
    if (glue_metrop_outfun(type, lgl, num, flen, d, &out_kind, mask, &nout, msg, sizeof msg)) error("%s", msg);

    uint64_t key, iter_offset, chain_offset;
    glue_seed_from_sexp(seed, &key, &iter_offset, &chain_offset);

    /* result list: names and order of metrop.c:124-128,174-179 */
    const char *names[11];
    const int nslot = glue_metrop_result_names(int_debug, names);
    SEXP result = PROTECT(allocVector(VECSXP, nslot + 1));       /* + "device.seed": (key, iterations done, first chain) */
    SEXP rnames = PROTECT(allocVector(STRSXP, nslot + 1));
    for (int i = 0; i < nslot; i++) SET_STRING_ELT(rnames, i, mkChar(names[i]));
    SET_STRING_ELT(rnames, nslot, mkChar("device.seed"));
    namesgets(result, rnames);
    UNPROTECT(1);
    const int niter = int_nbatch * int_blen * int_nspac;         /* int arithmetic as metrop.c:131 */
    SEXP accept = PROTECT(allocVector(REALSXP, nchain));
    SET_VECTOR_ELT(result, 0, accept);
    SEXP path = PROTECT(batched ? alloc3DArray(REALSXP, nout, int_nbatch, nchain) : allocMatrix(REALSXP, nout, int_nbatch));
    SET_VECTOR_ELT(result, 1, path);
    SET_VECTOR_ELT(result, 2, duplicate(state));
    SEXP final_ = PROTECT(batched ? allocMatrix(REALSXP, nchain, d) : allocVector(REALSXP, d));
    SET_VECTOR_ELT(result, 3, final_);
    SEXP accb = PROTECT(batched ? allocMatrix(REALSXP, int_nbatch, nchain) : allocVector(REALSXP, int_nbatch));
    SET_VECTOR_ELT(result, 4, accb);
    UNPROTECT(4);
    SEXP dcur = R_NilValue, dprop = R_NilValue, dlg = R_NilValue, du = R_NilValue, dz = R_NilValue, dacc = R_NilValue;
    if (int_debug) {
        SET_VECTOR_ELT(result, 5, dcur = alloc_real_dims(d, niter, batched ? nchain : 0, 0));
        SET_VECTOR_ELT(result, 6, dprop = alloc_real_dims(d, niter, batched ? nchain : 0, 0));
        SET_VECTOR_ELT(result, 7, dlg = batched ? allocMatrix(REALSXP, niter, nchain) : allocVector(REALSXP, niter));
        SET_VECTOR_ELT(result, 8, du = batched ? allocMatrix(REALSXP, niter, nchain) : allocVector(REALSXP, niter));
        SET_VECTOR_ELT(result, 9, dz = alloc_real_dims(d, niter, batched ? nchain : 0, 0));
        SET_VECTOR_ELT(result, 10, dacc = batched ? allocMatrix(LGLSXP, niter, nchain) : allocVector(LGLSXP, niter));
    }

    /* chain-major copies for the C-ABI: an R matrix nchain x d is column-major */
    double *init_cm = (double *)R_alloc((size_t)nchain * d, sizeof(double));
    double *final_cm = (double *)R_alloc((size_t)nchain * d, sizeof(double));
    if (batched) glue_colmajor_to_rowmajor(REAL(state), nchain, d, init_cm);
    else memcpy(init_cm, REAL(state), sizeof(double) * (size_t)d);

    mcmchip_metrop_args a;
    mcmchip_metrop_out o;
    memset(&a, 0, sizeof a);
    memset(&o, 0, sizeof o);
    a.nchain = nchain;
    a.nbatch = int_nbatch; a.blen = int_blen; a.nspac = int_nspac;
    a.initial = init_cm;
    a.scale_kind = scale_kind;
    a.scale = REAL(rscale);                 /* a d x d R matrix is column-major, as MCMCHIP_SCALE_FULL wants it */
    a.out_kind = out_kind;
    a.out_index = out_kind == MCMCHIP_OUT_INDEX ? mask : NULL;
    a.rng_mode = MCMCHIP_RNG_PHILOX;
    a.seed = key; a.chain_offset = chain_offset; a.iter_offset = iter_offset;
    a.debug = int_debug;
    a.device_ptrs = 0;
    a.path = MCMCHIP_PATH_AUTO;
    o.accept = REAL(accept);
    o.batch = REAL(path);                   /* per chain [nbatch][nout] row-major == nout x nbatch column-major */
    o.final = final_cm;
    o.accept_batch = REAL(accb);
    if (int_debug) {
        o.current = REAL(dcur); o.proposal = REAL(dprop); o.z = REAL(dz);
        o.log_green = REAL(dlg); o.u = REAL(du);
        o.debug_accept = LOGICAL(dacc);     /* int32 0/1 */
    }
    int devices[64];
    const int ndev = glue_device_list(devices, 64);          /* MCMCHIP_DEVICES: several GPUs behind this one .Call */
    const int rc = ndev > 1 ? mcmchip_metrop_multi(target, &a, &o, devices, ndev) : mcmchip_metrop(target, &a, &o);   /* device buffers are released inside before it returns */
    if (rc) glue_raise(rc);
    if (int_debug)                           /* the reference stores NA where no uniform was drawn (metrop.c:252-255) */
        for (R_xlen_t i = 0; i < XLENGTH(du); i++)
            if (ISNAN(REAL(du)[i])) REAL(du)[i] = NA_REAL;
    if (batched)
        for (int c = 0; c < nchain; c++)
            for (int j = 0; j < d; j++) REAL(final_)[c + (size_t)nchain * j] = final_cm[(size_t)c * d + j];
    else memcpy(REAL(final_), final_cm, sizeof(double) * (size_t)d);
    SET_VECTOR_ELT(result, nslot, glue_seed_to_sexp(key, o.iter_offset_out, chain_offset));
    UNPROTECT(4);                            /* state, rscale, rfun, result */
    return result;
}
