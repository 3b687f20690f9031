/* glue_core.c -- see glue_core.h.  Plain C99, no R header, no HIP: gcc -O2 -std=c99 -fPIC -shared. */
#include "glue_core.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int fail(char *err, int errlen, const char *fmt, ...)
{
    if (err && errlen > 0) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, (size_t)errlen, fmt, ap);
        va_end(ap);
    }
    return GLUE_ERROR;
}

void glue_colmajor_to_rowmajor(const double *a, long nrow, long ncol, double *out)
{
    for (long i = 0; i < nrow; i++)
        for (long j = 0; j < ncol; j++) out[i * ncol + j] = a[i + nrow * j];
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* descriptors                                                                                                          */
/* ------------------------------------------------------------------------------------------------------------------ */
static const struct { const char *name; int kind; int needs_xy; int temper; } kKinds[] = {
    {"logistic", MCMCHIP_T_LOGISTIC, 1, 0},
    {"logistic.naive", MCMCHIP_T_LOGISTIC_NAIVE, 1, 0},
    {"mvnorm", MCMCHIP_T_MVNORM, 0, 0},
    {"normmix", MCMCHIP_T_NORMMIX, 0, 0},
    {"simplex", MCMCHIP_T_SIMPLEX, 0, 0},
    {"ball", MCMCHIP_T_BALL, 0, 0},
    {"witch", MCMCHIP_T_WITCH, 0, 1},
    {"tempered.normmix", MCMCHIP_T_TEMPERED_NORMMIX, 0, 1},
    {"student.t", MCMCHIP_T_STUDENT_T, 0, 0},
    {"logistic.models", MCMCHIP_T_LOGISTIC_MODELS, 1, 1},
    {"binomial.logit", MCMCHIP_T_BINOMIAL_LOGIT, 1, 0},
};

/* parameter block of an isotropic finite normal mixture (include/mcmchip.h, MCMCHIP_T_NORMMIX):
 * K, then K records {log w_k - d/2 log(2 pi s_k^2), 1 / (2 s_k^2), mu_k[d]} */
static void fill_normmix(const glue_fields *f, double *par)
{
    const int K = f->nmix, d = f->d;
    par[0] = (double)K;
    for (int k = 0; k < K; k++) {
        const double s = f->sds[k];
        double *rec = par + 1 + (size_t)k * (2 + d);
        rec[0] = log(f->weights[k]) - 0.5 * d * log(2.0 * M_PI * s * s);
        rec[1] = 1.0 / (2.0 * s * s);
        for (int j = 0; j < d; j++) rec[2 + j] = f->means[k + (size_t)K * j];
    }
}

int glue_make_descriptor(const glue_fields *f, glue_descriptor *out, char *err, int errlen)
{
    memset(out, 0, sizeof *out);
    if (!f || !f->kind) return fail(err, errlen, "descriptor has no \"kind\"");
    int ki = -1;
    for (size_t i = 0; i < sizeof kKinds / sizeof kKinds[0]; i++)
        if (strcmp(f->kind, kKinds[i].name) == 0) ki = (int)i;
    if (ki < 0) return fail(err, errlen, "unknown descriptor kind \"%s\"", f->kind);
    const int kind = kKinds[ki].kind;
    mcmchip_target_desc *D = &out->desc;
    D->kind = kind;
    int d = f->d;
    if (kKinds[ki].needs_xy) {
        if (!f->x || f->x_nrow <= 0 || f->x_ncol <= 0) return fail(err, errlen, "descriptor \"%s\" needs a model matrix x", f->kind);
        if (!f->y || f->y_len != f->x_nrow) return fail(err, errlen, "length(y) == nrow(x) is not TRUE");
        if (d <= 0) d = f->x_ncol;
        if (d != f->x_ncol) return fail(err, errlen, "ncol(x) must equal the dimension of the state");
        for (long i = 0; i < (long)f->x_nrow * f->x_ncol; i++)
            if (!isfinite(f->x[i])) return fail(err, errlen, "all(is.finite(x)) is not TRUE");
        out->x_rowmajor = (double *)malloc(sizeof(double) * (size_t)f->x_nrow * (size_t)f->x_ncol);
        if (!out->x_rowmajor) return fail(err, errlen, "out of memory");
        glue_colmajor_to_rowmajor(f->x, f->x_nrow, f->x_ncol, out->x_rowmajor);
        D->n = f->x_nrow;
        D->x = out->x_rowmajor;
        D->y = f->y;
        D->prior_c = f->prior_sd > 0.0 ? 1.0 / (2.0 * f->prior_sd * f->prior_sd) : 0.0;
        if (kind != MCMCHIP_T_BINOMIAL_LOGIT)
            for (int i = 0; i < f->y_len; i++)
                if (f->y[i] != 0.0 && f->y[i] != 1.0) return fail(err, errlen, "all(y %%in%% c(0, 1)) is not TRUE");
    }
    if (d <= 0) return fail(err, errlen, "descriptor dimension must be positive");
    D->d = d;

    long npar = 0;
    switch (kind) {
    case MCMCHIP_T_MVNORM:
        if (!f->mu || !f->precision) return fail(err, errlen, "mvnorm descriptor needs mu and precision");
        npar = (long)d + (long)d * d;
        break;
    case MCMCHIP_T_NORMMIX:
    case MCMCHIP_T_TEMPERED_NORMMIX:
        if (f->nmix < 1 || !f->weights || !f->means || !f->sds) return fail(err, errlen, "mixture descriptor needs weights, means, sds");
        for (int k = 0; k < f->nmix; k++)
            if (!(f->weights[k] > 0.0) || !(f->sds[k] > 0.0)) return fail(err, errlen, "mixture weights and sds must be positive");
        npar = 1 + (long)f->nmix * (2 + d);
        if (kind == MCMCHIP_T_TEMPERED_NORMMIX) {
            if (f->ncomp < 2 || !f->betas) return fail(err, errlen, "tempered mixture needs k >= 2 inverse temperatures");
            npar += 1 + f->ncomp;
        }
        break;
    case MCMCHIP_T_SIMPLEX: npar = 1; break;
    case MCMCHIP_T_STUDENT_T:
        if (!(f->df > 0.0)) return fail(err, errlen, "df > 0 is not TRUE");
        npar = 1;
        break;
    case MCMCHIP_T_WITCH:
        if (f->ncomp < 2 || !f->bounds) return fail(err, errlen, "witch descriptor needs k >= 2 bounds");
        for (int i = 0; i < f->ncomp; i++)
            if (!(f->bounds[i] > 0.0 && f->bounds[i] < 1.0)) return fail(err, errlen, "witch.which[%d] must be in (0, 1)", i + 1);
        npar = f->ncomp;
        break;
    case MCMCHIP_T_LOGISTIC_MODELS:
        if (f->ncomp < 2 || !f->models) return fail(err, errlen, "logistic.models descriptor needs a k x d matrix of models, k >= 2");
        npar = 1 + (long)f->ncomp * d + f->ncomp + 1;
        break;
    case MCMCHIP_T_BINOMIAL_LOGIT:
        if (!f->totals) return fail(err, errlen, "binomial descriptor needs one total per row of x");
        npar = f->x_nrow;
        break;
    default: break;
    }
    if (npar > 0) {
        double *par = (double *)malloc(sizeof(double) * (size_t)npar);
        if (!par) { glue_free_descriptor(out); return fail(err, errlen, "out of memory"); }
        out->par = par;
        switch (kind) {
        case MCMCHIP_T_MVNORM:
            memcpy(par, f->mu, sizeof(double) * (size_t)d);
            memcpy(par + d, f->precision, sizeof(double) * (size_t)d * (size_t)d);
            break;
        case MCMCHIP_T_NORMMIX: fill_normmix(f, par); break;
        case MCMCHIP_T_TEMPERED_NORMMIX:
            par[0] = (double)f->ncomp;
            memcpy(par + 1, f->betas, sizeof(double) * (size_t)f->ncomp);
            fill_normmix(f, par + 1 + f->ncomp);
            break;
        case MCMCHIP_T_SIMPLEX: par[0] = f->simplex_value; break;
        case MCMCHIP_T_STUDENT_T: par[0] = f->df; break;
        case MCMCHIP_T_WITCH: memcpy(par, f->bounds, sizeof(double) * (size_t)f->ncomp); break;
        case MCMCHIP_T_LOGISTIC_MODELS: {
            const int k = f->ncomp;
            par[0] = (double)k;
            for (int i = 0; i < k; i++)            /* masks row-major k x d from R's column-major matrix */
                for (int j = 0; j < d; j++) par[1 + (size_t)i * d + j] = f->models[i + (size_t)k * j] ? 1.0 : 0.0;
            for (int i = 0; i < k; i++) par[1 + (size_t)k * d + i] = f->log_pseudo_prior ? f->log_pseudo_prior[i] : 0.0;
            par[1 + (size_t)k * d + k] = f->constant;
            break;
        }
        case MCMCHIP_T_BINOMIAL_LOGIT: memcpy(par, f->totals, sizeof(double) * (size_t)f->x_nrow); break;
        default: break;
        }
        D->par = par;
        D->npar = npar;
    }
    /* morph(b, r, p, center), R/morph.R:121-152 */
    if (f->has_b || f->has_r || f->has_center) {
        if (kKinds[ki].temper) { glue_free_descriptor(out); return fail(err, errlen, "morph applies to metrop targets only"); }
        if (f->has_b) {
            if (!(f->morph_b > 0.0)) { glue_free_descriptor(out); return fail(err, errlen, "b > 0 is not TRUE"); }
            D->morph_flags |= MCMCHIP_MORPH_SUBEXPO;
            D->morph_b = f->morph_b;
        }
        if (f->has_r) {
            if (!(f->morph_p > 2.0)) { glue_free_descriptor(out); return fail(err, errlen, "p > 2 is not TRUE"); }
            if (!(f->morph_r >= 0.0)) { glue_free_descriptor(out); return fail(err, errlen, "r >= 0 is not TRUE"); }
            D->morph_flags |= MCMCHIP_MORPH_EXPO;
            D->morph_r = f->morph_r;
            D->morph_p = f->morph_p;
        }
        if (f->has_center) {
            if (!f->morph_center || !(f->morph_center_len == 1 || f->morph_center_len == d)) {
                glue_free_descriptor(out);
                return fail(err, errlen, "length(center) must be 1 or length(state)");
            }
            out->center = (double *)malloc(sizeof(double) * (size_t)d);
            if (!out->center) { glue_free_descriptor(out); return fail(err, errlen, "out of memory"); }
            for (int j = 0; j < d; j++) out->center[j] = f->morph_center[f->morph_center_len == 1 ? 0 : j];
            D->morph_flags |= MCMCHIP_MORPH_CENTER;
            D->morph_center = out->center;
        }
    }
    return GLUE_OK;
}

void glue_free_descriptor(glue_descriptor *g)
{
    if (!g) return;
    free(g->x_rowmajor);
    free(g->par);
    free(g->center);
    memset(g, 0, sizeof *g);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* scalar arguments                                                                                                     */
/* ------------------------------------------------------------------------------------------------------------------ */
int glue_scalar_integer(int is_numeric, int len, double value, const char *argname, int must_be_positive, int *out,
                        char *err, int errlen)
{
    if (!is_numeric) return fail(err, errlen, "argument \"%s\" must be numeric", argname);
    if (len != 1) return fail(err, errlen, "argument \"%s\" must be scalar", argname);
    /* coerceVector(REALSXP -> INTSXP) truncates toward zero; NA / out of range becomes NA_INTEGER (INT_MIN), which the
     * positivity check of the callers then refuses */
    int v;
    if (isnan(value) || value >= 2147483648.0 || value <= -2147483649.0) v = INT32_MIN;
    else v = (int)value;
    if (must_be_positive && v <= 0) return fail(err, errlen, "argument \"%s\" must be positive", argname);
    *out = v;
    return GLUE_OK;
}

int glue_scalar_logical(int is_logical, int len, int value, const char *argname, int *out, char *err, int errlen)
{
    if (!is_logical) return fail(err, errlen, "argument \"%s\" must be logical", argname);
    if (len != 1) return fail(err, errlen, "argument \"%s\" must be scalar", argname);
    *out = value;
    return GLUE_OK;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* scale                                                                                                                */
/* ------------------------------------------------------------------------------------------------------------------ */
int glue_metrop_scale(int is_numeric, int len, int is_matrix, int nrow, int ncol, int d, int *scale_kind, char *err,
                      int errlen)
{
    if (!is_numeric) return fail(err, errlen, "argument \"scale\" must be numeric");
    if (is_matrix) {
        if (nrow == d && ncol == d) { *scale_kind = MCMCHIP_SCALE_FULL; return GLUE_OK; }
        return fail(err, errlen, "dimensions of \"scale\" matrix not d by d");
    }
    if (len == d) { *scale_kind = MCMCHIP_SCALE_DIAGONAL; return GLUE_OK; }
    if (len == 1) { *scale_kind = MCMCHIP_SCALE_CONSTANT; return GLUE_OK; }
    return fail(err, errlen, "length of \"scale\" vector not d or 1");
}

int glue_temper_scale(int is_double, const double *v, int len, int is_matrix, int nrow, int ncol, int comp, int nx,
                      int *scale_kind, char *err, int errlen)
{
    char who[48];
    if (comp >= 0) snprintf(who, sizeof who, "component %d of scale", comp + 1);
    else snprintf(who, sizeof who, "scale");
    if (!is_double) return fail(err, errlen, "%s not type double", who);
    const long n = is_matrix ? (long)nrow * ncol : len;
    for (long i = 0; i < n; i++)
        if (!isfinite(v[i])) return fail(err, errlen, "%s has non-finite element", who);
    if (is_matrix) {
        if (nrow != nx) return fail(err, errlen, "%s matrix with wrong row dim", who);
        if (ncol != nx) return fail(err, errlen, "%s matrix with wrong col dim", who);
        *scale_kind = MCMCHIP_SCALE_FULL;
        return GLUE_OK;
    }
    if (!(len == 1 || len == nx)) return fail(err, errlen, "%s not matrix, scalar, or vector of length k", who);
    /* propose() of temper.c:1003-1016 treats length 1 as a constant and length nx as a diagonal; with nx == 1 the
     * two coincide (same multiply, same add) */
    *scale_kind = len == 1 ? MCMCHIP_SCALE_CONSTANT : MCMCHIP_SCALE_DIAGONAL;
    return GLUE_OK;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* outfun                                                                                                               */
/* ------------------------------------------------------------------------------------------------------------------ */
int glue_metrop_outfun(int type, const int *lgl, const double *num, int len, int d, int *out_kind, int32_t *mask,
                       int *nout, char *err, int errlen)
{
    if (type == GLUE_OUTFUN_NULL) {
        *out_kind = MCMCHIP_OUT_IDENTITY;
        *nout = d;
        return GLUE_OK;
    }
    if (type == GLUE_OUTFUN_MOMENTS2) {
        *out_kind = MCMCHIP_OUT_MOMENTS2;
        *nout = 2 * d;
        return GLUE_OK;
    }
    if (type == GLUE_OUTFUN_FUNCTION)
        return fail(err, errlen, "an R function outfun cannot run on the device: use an index vector, a logical vector, "
                                 "\"moments2\", or the closure path of metrop()");
    if (type == GLUE_OUTFUN_LOGICAL) {
        if (len != d) return fail(err, errlen, "is.logical(outfun) & (length(outfun) != length(initial))");
        int n = 0;
        for (int i = 0; i < d; i++) {
            mask[i] = lgl[i] != 0;      /* NA_LOGICAL is non-zero: counted like TRUE, as out_index[i] = LOGICAL(func)[i] does */
            n += mask[i];
        }
        *out_kind = MCMCHIP_OUT_INDEX;
        *nout = n;
        return GLUE_OK;
    }
    if (type == GLUE_OUTFUN_NUMERIC) {
        int pos = 0, neg = 0;
        for (int i = 0; i < len; i++) {
            const double v = num[i];
            if (isnan(v)) return fail(err, errlen, "NA or NaN index for outfun");
            if (!isfinite(v)) return fail(err, errlen, "-Inf or Inf index for outfun");
            /* beyond int the reference's `int fooint = foodble` (metrop.c:457) is undefined; on x86-64 it yields INT_MIN
               and the run ends in the not-an-integer error below: say so without the undefined conversion */
            if (fabs(v) >= 2147483648.0) return fail(err, errlen, "is.numeric(outfun) & any(outfun != as.integer(outfun))");
            const int iv = (int)v;
            const int av = iv >= 0 ? iv : -iv;
            if (iv == 0) return fail(err, errlen, "is.numeric(outfun) & any(outfun == 0)");
            if (v != (double)iv) return fail(err, errlen, "is.numeric(outfun) & any(outfun != as.integer(outfun))");
            if (av > d) return fail(err, errlen, "is.numeric(outfun) & any(abs(outfun) > length(initial)");
            if (v > 0) pos++;
            else neg++;
        }
        if (pos > 0 && neg > 0) return fail(err, errlen, "is.numeric(outfun) & any(outfun > 0) & any(outfun < 0)");
        for (int i = 0; i < d; i++) mask[i] = pos > 0 ? 0 : 1;       /* an empty vector keeps everything, as in the reference */
        for (int i = 0; i < len; i++) {
            const int iv = (int)num[i];
            if (pos > 0) mask[iv - 1] = 1;
            else mask[-iv - 1] = 0;
        }
        int n = 0;
        for (int i = 0; i < d; i++) n += mask[i];
        *out_kind = MCMCHIP_OUT_INDEX;
        *nout = n;
        return GLUE_OK;
    }
    return fail(err, errlen, "outfun must be NULL, a function, a numeric vector, or a logical vector");
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* result lists                                                                                                         */
/* ------------------------------------------------------------------------------------------------------------------ */
int glue_metrop_result_names(int debug, const char **names)
{
    static const char *const regular[] = {"accept", "batch", "initial", "final", "accept.batch"};
    static const char *const dbg[] = {"current", "proposal", "log.green", "u", "z", "debug.accept"};
    int n = 0;
    for (int i = 0; i < 5; i++) names[n++] = regular[i];
    if (debug)
        for (int i = 0; i < 6; i++) names[n++] = dbg[i];
    return n;
}

int glue_temper_result_names(int parallel, int debug, const char **names)
{
    static const char *const regular[] = {"batch", "acceptx", "accepti", "initial", "final"};
    static const char *const dbg[] = {"which", "unif.which", "state", "log.hastings", "unif.hastings", "proposal",
                                      "acceptd", "norm", "unif.choose"};
    int n = 0;
    for (int i = 0; i < 5; i++) names[n++] = regular[i];
    if (!parallel) names[n++] = "ibatch";
    if (debug) {
        for (int i = 0; i < 9; i++) names[n++] = dbg[i];
        if (parallel) names[n++] = "coproposal";
    }
    return n;
}

int glue_initseq_result_names(const char **names)
{
    static const char *const all[] = {"gamma0", "Gamma.pos", "Gamma.dec", "Gamma.con", "var.pos", "var.dec", "var.con"};
    for (int i = 0; i < 7; i++) names[i] = all[i];
    return 7;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* tempering arguments                                                                                                  */
/* ------------------------------------------------------------------------------------------------------------------ */
int glue_neighbors(int is_logical, int is_matrix, int nrow, int ncol, const int *lgl, int32_t *out, char *err, int errlen)
{
    if (!is_logical) return fail(err, errlen, "argument \"neighbors\" must be logical");
    if (!is_matrix) return fail(err, errlen, "argument \"neighbors\" must be matrix");
    if (nrow != ncol) return fail(err, errlen, "argument \"neighbors\" must have same row and column dimension");
    const int k = nrow;
    if (k <= 1) return fail(err, errlen, "must have at least two components");
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++)
            if (lgl[i + k * j] != lgl[j + k * i]) return fail(err, errlen, "argument \"neighbors\" must be symmetric matrix");
    for (int i = 0; i < k * k; i++) out[i] = lgl[i] != 0;
    return GLUE_OK;
}

int glue_temper_initial(int is_double, const double *v, long len, int is_matrix, int nrow, int ncol, int parallel,
                        int ncomp, int *nx, char *err, int errlen)
{
    if (!is_double) return fail(err, errlen, "argument \"initial\" must be real");
    for (long i = 0; i < len; i++)
        if (!isfinite(v[i])) return fail(err, errlen, "argument \"initial\" must have all elements finite");
    if (parallel) {
        if (!is_matrix) return fail(err, errlen, "argument \"initial\" must be matrix in parallel case");
        if (nrow != ncomp) return fail(err, errlen, "row dim of args \"initial\" and \"neighbors\" must be same in parallel case");
        *nx = ncol;
        return GLUE_OK;
    }
    if (!(len > 1)) return fail(err, errlen, "argument \"initial\" must have length > 1 in serial case");
    const double reali = v[0];
    const int i = (int)reali;
    if ((double)i != reali) return fail(err, errlen, "1st elem of argument \"initial\" must be integer in serial case");
    if (i <= 0 || i > ncomp) return fail(err, errlen, "1st elem of argument \"initial\" must be in 1, ..., k in serial case");
    *nx = (int)len - 1;
    return GLUE_OK;
}
