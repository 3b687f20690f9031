/* glue_r.c -- see glue_r.h.  NOT COMPILED HERE (no R headers in this image). */
#include "glue_r.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <R_ext/Utils.h>

SEXP glue_list_get(SEXP list, const char *name)
{
    if (!isNewList(list)) return R_NilValue;
    SEXP names = getAttrib(list, R_NamesSymbol);
    if (names == R_NilValue) return R_NilValue;
    for (R_xlen_t i = 0; i < XLENGTH(list); i++)
        if (strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return VECTOR_ELT(list, i);
    return R_NilValue;
}

static const double *real_field(SEXP list, const char *name, int *len)
{
    SEXP v = glue_list_get(list, name);
    if (v == R_NilValue) { if (len) *len = 0; return NULL; }
    if (!isReal(v)) error("descriptor field \"%s\" must be double (storage.mode)", name);
    if (len) *len = LENGTH(v);
    return REAL(v);
}

static double scalar_field(SEXP list, const char *name, double dflt)
{
    int len = 0;
    const double *v = real_field(list, name, &len);
    if (!v) return dflt;
    if (len != 1) error("descriptor field \"%s\" must be a scalar", name);
    return v[0];
}

void glue_fields_from_sexp(SEXP desc, glue_fields *f)
{
    memset(f, 0, sizeof *f);
    if (!inherits(desc, "mcmc_descriptor")) error("argument \"obj\" must be a log-density descriptor");
    SEXP kind = glue_list_get(desc, "kind");
    if (!isString(kind) || LENGTH(kind) != 1) error("descriptor field \"kind\" must be a character string");
    f->kind = CHAR(STRING_ELT(kind, 0));
    f->d = (int)scalar_field(desc, "d", 0.0);
    SEXP x = glue_list_get(desc, "x");
    if (x != R_NilValue) {
        if (!isReal(x) || !isMatrix(x)) error("descriptor field \"x\" must be a double matrix");
        f->x = REAL(x);
        f->x_nrow = nrows(x);
        f->x_ncol = ncols(x);
    }
    f->y = real_field(desc, "y", &f->y_len);
    f->prior_sd = scalar_field(desc, "prior.sd", 0.0);
    f->mu = real_field(desc, "mu", NULL);
    f->precision = real_field(desc, "precision", NULL);
    f->weights = real_field(desc, "weights", &f->nmix);
    f->means = real_field(desc, "means", NULL);
    f->sds = real_field(desc, "sds", NULL);
    int nb = 0, nt = 0;
    f->bounds = real_field(desc, "bounds", &nb);
    f->betas = real_field(desc, "betas", &nt);
    f->ncomp = nb > 0 ? nb : nt;
    SEXP models = glue_list_get(desc, "models");
    if (models != R_NilValue) {
        if (!isMatrix(models)) error("descriptor field \"models\" must be a matrix");
        SEXP mi = PROTECT(coerceVector(models, INTSXP));
        const int k = nrows(models), dd = ncols(models);
        int *m = (int *)R_alloc((size_t)k * dd, sizeof(int));
        memcpy(m, INTEGER(mi), sizeof(int) * (size_t)k * dd);
        UNPROTECT(1);
        f->models = m;
        f->ncomp = k;
        if (f->d <= 0) f->d = dd;
    }
    f->log_pseudo_prior = real_field(desc, "log.pseudo.prior", NULL);
    f->constant = scalar_field(desc, "constant", 0.0);
    f->df = scalar_field(desc, "df", 0.0);
    f->totals = real_field(desc, "totals", NULL);
    f->simplex_value = scalar_field(desc, "value", 0.0);
    SEXP morph = glue_list_get(desc, "morph");
    if (morph != R_NilValue) {
        if (glue_list_get(morph, "b") != R_NilValue) { f->has_b = 1; f->morph_b = scalar_field(morph, "b", 0.0); }
        if (glue_list_get(morph, "r") != R_NilValue) {
            f->has_r = 1;
            f->morph_r = scalar_field(morph, "r", 0.0);
            f->morph_p = scalar_field(morph, "p", 3.0);
        }
        if (glue_list_get(morph, "center") != R_NilValue) {
            f->has_center = 1;
            f->morph_center = real_field(morph, "center", &f->morph_center_len);
        }
    }
}

static void target_finalizer(SEXP ptr)
{
    mcmchip_target *t = (mcmchip_target *)R_ExternalPtrAddr(ptr);
    if (t) mcmchip_target_destroy(t);
    R_ClearExternalPtr(ptr);
}

void glue_raise(int rc)
{
    if (rc == MCMCHIP_EINTR) error("interrupted");
    error("%s", mcmchip_last_error());
}

mcmchip_target *glue_target(SEXP desc)
{
    SEXP cache = glue_list_get(desc, "cache");
    if (!isEnvironment(cache)) error("descriptor has no cache environment (use the constructors of R/descriptor.R)");
    SEXP sym = install("handle");
    SEXP have = findVarInFrame(cache, sym);
    if (have != R_UnboundValue && TYPEOF(have) == EXTPTRSXP && R_ExternalPtrAddr(have))
        return (mcmchip_target *)R_ExternalPtrAddr(have);
    glue_init_device();
    glue_fields f;
    glue_fields_from_sexp(desc, &f);
    glue_descriptor g;
    char msg[256];
    if (glue_make_descriptor(&f, &g, msg, sizeof msg)) error("%s", msg);
    mcmchip_target *t = NULL;
    int rc = mcmchip_target_create(&g.desc, 0, &t);       /* copies everything into HBM */
    glue_free_descriptor(&g);
    if (rc) glue_raise(rc);
    SEXP ptr = PROTECT(R_MakeExternalPtr(t, R_NilValue, R_NilValue));
    R_RegisterCFinalizerEx(ptr, target_finalizer, TRUE);
    defineVar(sym, ptr, cache);
    UNPROTECT(1);
    return t;
}

void glue_seed_from_sexp(SEXP seed, uint64_t *key, uint64_t *iter_offset, uint64_t *unit_offset)
{
    if (!isReal(seed) || LENGTH(seed) != 4) error("argument \"seed\" must be c(key.hi, key.lo, iterations, first.unit)");
    const double *s = REAL(seed);
    for (int i = 0; i < 4; i++)
        if (!R_FINITE(s[i]) || s[i] < 0 || s[i] != floor(s[i])) error("argument \"seed\" must hold non-negative whole numbers");
    if (s[0] >= 4294967296.0 || s[1] >= 4294967296.0) error("seed halves must be below 2^32");
    *key = ((uint64_t)s[0] << 32) | (uint64_t)s[1];
    *iter_offset = (uint64_t)s[2];
    *unit_offset = (uint64_t)s[3];
}

SEXP glue_seed_to_sexp(uint64_t key, uint64_t iter_offset, uint64_t unit_offset)
{
    SEXP out = PROTECT(allocVector(REALSXP, 4));
    REAL(out)[0] = (double)(key >> 32);
    REAL(out)[1] = (double)(key & 0xffffffffu);
    REAL(out)[2] = (double)iter_offset;
    REAL(out)[3] = (double)unit_offset;
    UNPROTECT(1);
    return out;
}

/* R_CheckUserInterrupt() longjmps; run under R_ToplevelExec it reports instead, so that the library can release its
 * device buffers before the glue raises the error (metrop.c:269 and temper.c:848 poll every iteration; the device
 * path polls between kernel launches, about four times a second) */
static void check_interrupt(void *unused) { (void)unused; R_CheckUserInterrupt(); }
static int poll_r_interrupt(void *unused) { (void)unused; return R_ToplevelExec(check_interrupt, NULL) == FALSE; }

/* MCMCHIP_DEVICES="0,1,2,3,4,5,6,7": the descriptor natives split the chains (ensembles) of every call over these
 * GPUs of the node through mcmchip_metrop_multi / mcmchip_temper_multi -- one .Call, one R process, several devices
 * (R/metrop.R:48-51 makes exactly one native call per run).  Unset or a single entry: the one device of
 * glue_init_device.  Returns the number of entries written to devices[0 .. cap). */
int glue_device_list(int *devices, int cap)
{
    const char *e = getenv("MCMCHIP_DEVICES");
    int n = 0;
    while (e && *e && n < cap) {
        char *end;
        const long v = strtol(e, &end, 10);
        if (end == e) break;
        devices[n++] = (int)v;
        e = *end == ',' ? end + 1 : end;
    }
    return n;
}

void glue_init_device(void)
{
    static int done = 0;
    if (done) return;
    const char *e = getenv("MCMCHIP_DEVICE");
    int first[1];
    /* the descriptor lives on the first listed device; the others receive replicas on first use */
    int rc = mcmchip_init(e ? atoi(e) : (glue_device_list(first, 1) ? first[0] : 0));
    if (rc) glue_raise(rc);
    mcmchip_set_interrupt_poll(poll_r_interrupt, NULL);
    done = 1;
}
