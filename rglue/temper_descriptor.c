/*
 * temper_descriptor.c -- .Call entry point for temper() on a tempering DESCRIPTOR: the device-path counterpart of
 * SEXP temper(func1, initial, neighbors, nbatch, blen, nspac, scale, func2, debug, parallel, rho1, rho2)
 * (src/temper.c:17-913).  Same checks and error texts (glue_core.c restates temper.c:21-95,915-953), same result list
 * (temper.c:196-277 and, with debug, 282-464: niter-major trace arrays).
 * One ensemble per call from R (nens = 1): `initial` is the k x p matrix (parallel) or c(i, x) (serial) of the
 * reference; the batched form is reachable through the C-ABI.
 * NOT COMPILED IN THIS REPOSITORY'S IMAGE (no R headers); see rglue/README.md.
 */
#include "glue_r.h"

#include <string.h>

SEXP temper_descriptor(SEXP desc, SEXP initial, SEXP neighbors, SEXP nbatch, SEXP blen, SEXP nspac, SEXP scale,
                       SEXP func2, SEXP debug, SEXP parallel, SEXP seed)
{
    char msg[256];
    int is_parallel, is_debug, int_nbatch, int_blen, int_nspac;
    if (glue_scalar_logical(isLogical(parallel), LENGTH(parallel), isLogical(parallel) && LENGTH(parallel) == 1 ? LOGICAL(parallel)[0] : 0,
                            "parallel", &is_parallel, msg, sizeof msg)) error("%s", msg);
    if (glue_scalar_logical(isLogical(debug), LENGTH(debug), isLogical(debug) && LENGTH(debug) == 1 ? LOGICAL(debug)[0] : 0,
                            "debug", &is_debug, msg, sizeof msg)) error("%s", msg);

    /* neighbors (temper.c:29-41) */
    const int nb_ok = isLogical(neighbors) && isMatrix(neighbors);
    const int ncomp = nb_ok ? nrows(neighbors) : 0;
    int32_t *nbrs = (int32_t *)R_alloc(nb_ok ? (size_t)ncomp * ncols(neighbors) + 1 : 1, sizeof(int32_t));
    if (glue_neighbors(isLogical(neighbors), isMatrix(neighbors), nb_ok ? nrows(neighbors) : 0, nb_ok ? ncols(neighbors) : 0,
                       nb_ok ? LOGICAL(neighbors) : NULL, nbrs, msg, sizeof msg)) error("%s", msg);

    /* initial (temper.c:43-64) */
    int nx = 0;
    if (glue_temper_initial(isReal(initial), isReal(initial) ? REAL(initial) : NULL, isReal(initial) ? XLENGTH(initial) : 0,
                            isMatrix(initial), isMatrix(initial) ? nrows(initial) : 0, isMatrix(initial) ? ncols(initial) : 0,
                            is_parallel, ncomp, &nx, msg, sizeof msg)) error("%s", msg);

    if (glue_scalar_integer(isNumeric(nbatch), LENGTH(nbatch), isNumeric(nbatch) && LENGTH(nbatch) == 1 ? asReal(nbatch) : 0.0,
                            "nbatch", 1, &int_nbatch, msg, sizeof msg)) error("%s", msg);
    if (glue_scalar_integer(isNumeric(blen), LENGTH(blen), isNumeric(blen) && LENGTH(blen) == 1 ? asReal(blen) : 0.0,
                            "blen", 1, &int_blen, msg, sizeof msg)) error("%s", msg);
    if (glue_scalar_integer(isNumeric(nspac), LENGTH(nspac), isNumeric(nspac) && LENGTH(nspac) == 1 ? asReal(nspac) : 0.0,
                            "nspac", 1, &int_nspac, msg, sizeof msg)) error("%s", msg);

    /* scale: one object or a list of k (temper.c:78-87); flattened into k consecutive blocks for the C-ABI */
    int scale_kind = 0;
    double *flat;
    if (isNewList(scale)) {
        if (LENGTH(scale) != ncomp) error("argument \"scale\" must have length k if list");
        size_t total = 0;
        for (int i = 0; i < ncomp; i++) {
            SEXP e = VECTOR_ELT(scale, i);
            int kind_i;
            if (glue_temper_scale(isReal(e), isReal(e) ? REAL(e) : NULL, LENGTH(e), isMatrix(e), isMatrix(e) ? nrows(e) : 0,
                                  isMatrix(e) ? ncols(e) : 0, i, nx, &kind_i, msg, sizeof msg)) error("%s", msg);
            if (i == 0) scale_kind = kind_i;
            else if (kind_i != scale_kind) error("list scale: all components must have the same form (scalar, vector or matrix)");
            total += (size_t)XLENGTH(e);
        }
        flat = (double *)R_alloc(total, sizeof(double));
        size_t off = 0;
        for (int i = 0; i < ncomp; i++) {
            SEXP e = VECTOR_ELT(scale, i);
            memcpy(flat + off, REAL(e), sizeof(double) * (size_t)XLENGTH(e));
            off += (size_t)XLENGTH(e);
        }
        scale_kind |= MCMCHIP_SCALE_LIST;
    } else {
        if (glue_temper_scale(isReal(scale), isReal(scale) ? REAL(scale) : NULL, LENGTH(scale), isMatrix(scale),
                              isMatrix(scale) ? nrows(scale) : 0, isMatrix(scale) ? ncols(scale) : 0, -1, nx, &scale_kind,
                              msg, sizeof msg)) error("%s", msg);
        flat = REAL(scale);
    }

    /* outfun (temper.c:89-95): NULL, or the name of a device-side functional; an R function cannot run on the device */
    int out_kind = MCMCHIP_TOUT_IDENTITY;
    if (!isNull(func2)) {
        if (isFunction(func2)) error("an R function outfun cannot run on the device: use NULL, \"witch.corner\", or the closure path of temper()");
        if (isString(func2) && LENGTH(func2) == 1 && strcmp(CHAR(STRING_ELT(func2, 0)), "witch.corner") == 0) out_kind = MCMCHIP_TOUT_WITCH_CORNER;
        else error("argument \"outfun\" must be function");
    }

    mcmchip_target *target = glue_target(desc);
    glue_fields fields;
    glue_fields_from_sexp(desc, &fields);
    {
        const int dd = fields.d > 0 ? fields.d : fields.x_ncol;
        if (dd != nx) error("dimension of \"initial\" (%d) differs from the dimension of the descriptor (%d)", nx, dd);
    }
    uint64_t key, iter_offset, ens_offset;
    glue_seed_from_sexp(seed, &key, &iter_offset, &ens_offset);

    const int nstate = is_parallel ? ncomp * nx : nx + 1;
    const int nout = out_kind == MCMCHIP_TOUT_WITCH_CORNER ? (is_parallel ? ncomp : 2 * ncomp) : (is_parallel ? ncomp * nx : nx);
    const int niter = int_nbatch * int_blen * int_nspac;          /* int arithmetic as temper.c:141 */

    const char *names[16];
    const int nslot = glue_temper_result_names(is_parallel, is_debug, names);
    SEXP result = PROTECT(allocVector(VECSXP, nslot + 1));
    SEXP rnames = PROTECT(allocVector(STRSXP, nslot + 1));
    for (int i = 0; i < nslot; i++) SET_STRING_ELT(rnames, i, mkChar(names[i]));
    SET_STRING_ELT(rnames, nslot, mkChar("device.seed"));
    namesgets(result, rnames);
    UNPROTECT(1);

    /* regular slots, shapes of temper.c:221-277: every array is column-major with the batch (or iteration) index
     * fastest, which is exactly the per-ensemble layout of the C-ABI */
    SEXP batch = (out_kind == MCMCHIP_TOUT_IDENTITY && is_parallel) ? alloc3DArray(REALSXP, int_nbatch, ncomp, nx)
                                                                     : allocMatrix(REALSXP, int_nbatch, nout);
    SET_VECTOR_ELT(result, 0, batch);
    SEXP acceptx = allocVector(REALSXP, ncomp);
    SET_VECTOR_ELT(result, 1, acceptx);
    SEXP accepti = allocMatrix(REALSXP, ncomp, ncomp);
    SET_VECTOR_ELT(result, 2, accepti);
    SET_VECTOR_ELT(result, 3, duplicate(initial));
    SEXP final_ = duplicate(initial);                              /* same shape and attributes as the state */
    SET_VECTOR_ELT(result, 4, final_);
    SEXP ibatch = R_NilValue;
    int slot = 5;
    if (!is_parallel) {
        ibatch = allocMatrix(REALSXP, int_nbatch, ncomp);
        SET_VECTOR_ELT(result, slot++, ibatch);
    }
    mcmchip_temper_args a;
    mcmchip_temper_out o;
    memset(&a, 0, sizeof a);
    memset(&o, 0, sizeof o);
    if (is_debug) {                                                /* slot order of temper.c:282-464 */
        SEXP v;
        SET_VECTOR_ELT(result, slot++, v = allocVector(LGLSXP, niter)); o.which = LOGICAL(v);
        SET_VECTOR_ELT(result, slot++, v = allocVector(REALSXP, niter)); o.unif_which = REAL(v);
        SET_VECTOR_ELT(result, slot++, v = is_parallel ? alloc3DArray(REALSXP, niter, ncomp, nx) : allocMatrix(REALSXP, niter, nx + 1));
        o.state = REAL(v);
        SET_VECTOR_ELT(result, slot++, v = allocVector(REALSXP, niter)); o.log_hastings = REAL(v);
        SET_VECTOR_ELT(result, slot++, v = allocVector(REALSXP, niter)); o.unif_hastings = REAL(v);
        SET_VECTOR_ELT(result, slot++, v = allocMatrix(REALSXP, niter, nx + 1)); o.proposal = REAL(v);
        SET_VECTOR_ELT(result, slot++, v = allocVector(LGLSXP, niter)); o.acceptd = LOGICAL(v);
        SET_VECTOR_ELT(result, slot++, v = allocMatrix(REALSXP, niter, nx)); o.norm = REAL(v);
        SET_VECTOR_ELT(result, slot++, v = is_parallel ? allocMatrix(REALSXP, niter, 2) : allocVector(REALSXP, niter));
        o.unif_choose = REAL(v);
        if (is_parallel) { SET_VECTOR_ELT(result, slot++, v = allocMatrix(REALSXP, niter, nx + 1)); o.coproposal = REAL(v); }
    }

    a.nens = 1;
    a.ncomp = ncomp;
    a.parallel = is_parallel;
    a.nbatch = int_nbatch; a.blen = int_blen; a.nspac = int_nspac;
    a.neighbors = nbrs;
    a.initial = REAL(initial);              /* k x p column-major == state[i + k*j] of temper.c:563; serial: c(i, x) */
    a.scale_kind = scale_kind;
    a.scale = flat;
    a.rng_mode = MCMCHIP_RNG_PHILOX;
    a.seed = key; a.ens_offset = ens_offset; a.iter_offset = iter_offset;
    a.device_ptrs = 0;
    a.debug = is_debug;
    a.out_kind = out_kind;
    a.path = MCMCHIP_PATH_AUTO;
    o.batch = REAL(batch);
    o.ibatch = is_parallel ? NULL : REAL(ibatch);
    o.acceptx = REAL(acceptx);
    o.accepti = REAL(accepti);
    o.final = REAL(final_);
    int devices[64];
    const int ndev = glue_device_list(devices, 64);          /* MCMCHIP_DEVICES: several GPUs behind this one .Call */
    const int rc = ndev > 1 ? mcmchip_temper_multi(target, &a, &o, devices, ndev) : mcmchip_temper(target, &a, &o);
    if (rc) glue_raise(rc);
    /* NaN markers of the C-ABI -> NA_real_ where the reference stores NA (temper.c:593-609,762,833,896-897) */
    for (int i = 0; i < ncomp * ncomp; i++)
        if (!nbrs[i]) REAL(accepti)[i] = NA_REAL;
    if (is_debug) {
        for (int i = 0; i < niter; i++)
            if (ISNAN(o.unif_hastings[i])) o.unif_hastings[i] = NA_REAL;
        for (int i = 0; i < niter * nx; i++)
            if (ISNAN(o.norm[i])) o.norm[i] = NA_REAL;
        for (int i = 0; i < niter * (is_parallel ? 2 : 1); i++)
            if (ISNAN(o.unif_choose[i])) o.unif_choose[i] = NA_REAL;
    }
    SET_VECTOR_ELT(result, nslot, glue_seed_to_sexp(key, o.iter_offset_out, ens_offset));
    UNPROTECT(1);
    return result;
}
