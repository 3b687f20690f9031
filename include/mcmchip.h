/*
 * mcmchip.h -- C-ABI of libmcmchip.so, the MI355X (gfx950) implementation of the
 * sampler hot path of the R package `mcmc` (reference: /root/reference/package/mcmc).
 *
 * This is the drop-in boundary.  Every entry point takes plain pointers and
 * sizes (no R types, no torch types).  The reference's `.Call`/`.C` natives
 * (src/init.c:7-20, src/mcmc.h:8-18) would forward to these when the `obj`
 * argument is a log-density *descriptor* instead of an R closure; see
 * INTEGRATION.md for the binding a maintainer adds to src/metrop.c etc.
 *
 * Conventions
 *   - all floating point is IEEE binary64 ("double").
 *   - every function returns MCMCHIP_OK (0) or an error code; the message
 *     (same wording as the reference's error() call where one exists) is
 *     retrievable with mcmchip_last_error() from the same thread.
 *   - `device_ptrs == 0`: data pointers are host memory, the library stages
 *     them through HBM.  `device_ptrs == 1`: the bulk data pointers (initial, z, u, unif,
 *     norm, x of initseq/olbm and the sampler outputs) are device (HIP) memory on the
 *     current device, nothing crosses PCIe.  Small control arrays (scale, out_index,
 *     neighbors) and the small results of initseq/olbm (mean, var, Gamma vectors, n_out)
 *     are always host memory.
 *   - inputs are never modified (reference: metrop.c:67-70 duplicates
 *     `initial`); outputs are caller-allocated.
 *   - batched extension: the reference runs ONE chain per call; here a call
 *     advances `nchain` independent chains in lockstep.  With nchain == 1 the
 *     output arrays have exactly the reference's C-side shapes.
 */
#ifndef MCMCHIP_H
#define MCMCHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCMCHIP_OK       0
#define MCMCHIP_EINVAL   1  /* bad argument (reference: the error() calls in metrop.c:35-65, temper.c:21-95) */
#define MCMCHIP_EDOMAIN  2  /* log density -Inf at initial state, or +Inf/NaN anywhere (metrop.c:196-197,317-320) */
#define MCMCHIP_EHIP     3  /* HIP runtime error */
#define MCMCHIP_ECAP     4  /* caller-provided output capacity too small (initseq) */
#define MCMCHIP_EINTR    5  /* the interrupt callback asked to stop between two launches (R_CheckUserInterrupt, metrop.c:269) */

/* ------------------------------------------------------------------ */
/* log unnormalised density descriptors (replace the R closure `func1`,
 * metrop.c:303-323 / temper.c:955-975)                                 */
/* ------------------------------------------------------------------ */
enum {
    /* vignettes/demo.Rnw:59-65: eta = X beta;
     *   logp = eta<0 ? eta-log1p(exp(eta)) : -log1p(exp(-eta))
     *   logq = eta<0 ? -log1p(exp(eta))    : -eta-log1p(exp(-eta))
     *   sum(logp[y==1]) + sum(logq[y==0]) - prior_c*sum(beta^2)   (demo: prior_c = 1/8) */
    MCMCHIP_T_LOGISTIC = 1,
    /* tests/logit.R:40-48: p = exp(eta)/(1+exp(eta)); sum(log(p[y==1])) + sum(log(1-p[y==0])) - prior_c*sum(beta^2) */
    MCMCHIP_T_LOGISTIC_NAIVE = 2,
    /* -0.5 (x-mu)' P (x-mu);  par = mu[d], P[d*d] (row-major, symmetric precision matrix) */
    MCMCHIP_T_MVNORM = 3,
    /* log sum_k exp(a_k - b_k |x-mu_k|^2);  par = K, then K records {a_k, b_k, mu_k[d]} */
    MCMCHIP_T_NORMMIX = 4,
    /* tests/accept-batch.R:8, devel/bug/foo.Rout:28: all(x>=0) && sum(x)<=1 ? par[0] : -Inf */
    MCMCHIP_T_SIMPLEX = 5,
    /* tests/circle.R:11-16: fred = 1-sum(x^2); fred>0 ? log(fred) : -Inf */
    MCMCHIP_T_BALL = 6,
    /* tests/temp-par-witch.R:24-40 (temper only, state = (i, theta)): par = bnd[k]
     *   any(|theta|>1) -> -Inf; all(|theta|>bnd_i) -> -(d+1)log2 - d log(1-bnd_i);
     *   else -(d+1)log2 - log1p(-(1-bnd_i)^d) */
    MCMCHIP_T_WITCH = 7,
    /* temper only: h(i,x) = beta_i * normmix(x);  par = k, beta[k], then a NORMMIX par block */
    MCMCHIP_T_TEMPERED_NORMMIX = 8,
    /* the heavy-tailed target of tests/morph.metrop.R:9,13 and devel/reallylong.R:4:
     * sum_j dt(x_j, df, log=TRUE);  par = df
     *   = d*(lgamma((df+1)/2) - lgamma(df/2) - log(df*pi)/2) - (df+1)/2 * sum_j log1p(x_j^2/df) */
    MCMCHIP_T_STUDENT_T = 9,
    /* temper only, state = (i, beta[d]): the model-selection posterior of tests/temp-par.R:39-56,
     * tests/temp-ser.R and vignettes/bfst.Rnw:472-489.  x (n x d), y as for LOGISTIC;
     * par = k, then k x d 0/1 model masks (row-major; column 0 is the intercept), then k log pseudo-priors,
     * then one additive constant.
     *   eta = X (beta * mask_i);  logp/logq as LOGISTIC;
     *   sum(logp[y==1]) + sum(logq[y==0]) - prior_c*sum(beta^2) + constant + log_pseudo_prior[i]
     * temp-par.R: prior_c = 1/2, constant = 0, pseudo-priors 0;  bfst.Rnw: sum(dnorm(beta, 0, 2, log=TRUE))
     * = -sum(beta^2)/8 - d*log(2*sqrt(2*pi)), i.e. prior_c = 1/8 and that constant */
    MCMCHIP_T_LOGISTIC_MODELS = 10,
    /* vignettes/morph.Rnw:361-365 (binomial regression with a conjugate prior, the worked example of
     * morph.metrop): x = M (n x d), y = posterior success counts x_l, par = posterior totals n_l (n doubles)
     *   MB = M beta;  sum(y * MB) - sum(par * log(1 + exp(MB))) */
    MCMCHIP_T_BINOMIAL_LOGIT = 11
};

/* variable transformation of morph.metrop() (R/morph.R:121-181, R/morph.metrop.R): the chain runs on the
 * morphed state y; the log density it sees is lud(inverse(y)) + log.jacobian(y) (morph.R:166-179), the output
 * functional sees inverse(y) (morph.R:95-106).  inverse(y) = f.inv(|y|) y/|y| + center (isotropic, morph.R:5-14,162)
 *   EXPO:    f.inv(s) = s <= r ? s : s + (s-r)^p                                   (morph.R:77)
 *   SUBEXPO: f.inv(s) = s > 1/b ? exp(b s) - e/3 : (s b)^3 e/6 + s b e/2           (morph.R:53-55)
 *   both:    subexpo.f.inv(expo.f.inv(s))                                          (morph.R:150)
 * log.jacobian(y) = log f.inv'(|y|) + (d-1)(log f.inv(|y|) - log|y|), d log f.inv'(0) at 0 (morph.R:16-28).
 * `initial`, `final` and the debug trace of mcmchip_metrop are morphed states; transform() (morph.R:161,
 * closed form for p = 3, Newton otherwise) runs once on the host. */
enum { MCMCHIP_MORPH_EXPO = 1, MCMCHIP_MORPH_SUBEXPO = 2, MCMCHIP_MORPH_CENTER = 4 };

typedef struct mcmchip_target_desc {
    int32_t kind;
    int32_t d;          /* dimension of the state (of theta for temper targets) */
    int64_t n;          /* observations (logistic kinds), else 0 */
    const double *x;    /* n x d, row-major (logistic kinds) */
    const double *y;    /* n, values 0/1 (logistic kinds) */
    double prior_c;     /* logistic kinds: subtract prior_c * sum(beta^2) */
    const double *par;  /* kind-specific parameter block, see enum */
    int64_t npar;
    int32_t morph_flags;          /* 0: no transformation; else OR of MCMCHIP_MORPH_* (metrop kinds only) */
    int32_t morph_pad_;
    double morph_b, morph_r, morph_p;   /* b > 0 (morph.R:51); r >= 0, p > 2 (morph.R:75-76) */
    const double *morph_center;   /* d doubles (MCMCHIP_MORPH_CENTER), else ignored */
} mcmchip_target_desc;

typedef struct mcmchip_target mcmchip_target; /* opaque: descriptor resident in HBM */

/* ------------------------------------------------------------------ */
/* proposal scale (metrop.c:333-404, temper.c:915-953,990-1026)         */
/* ------------------------------------------------------------------ */
enum {
    MCMCHIP_SCALE_CONSTANT = 1,  /* 1 double */
    MCMCHIP_SCALE_DIAGONAL = 2,  /* d doubles */
    MCMCHIP_SCALE_FULL     = 3,  /* d*d doubles, column-major as R stores it; p = x + S z accumulated column by column (metrop.c:391-399) */
    MCMCHIP_SCALE_LIST     = 8   /* temper only, OR-ed with one of the above: k consecutive blocks, one per component (temper.c:998-999) */
};

/* output functional g(x) (metrop.c:416-548) */
enum {
    MCMCHIP_OUT_IDENTITY = 0,  /* func2 == NULL */
    MCMCHIP_OUT_INDEX    = 1,  /* logical / +-index vector, already resolved to a 0/1 mask of length d (metrop.c:435-497) */
    MCMCHIP_OUT_MOMENTS2 = 2   /* c(x, x^2), the vignette's outfun (vignettes/demo.Rnw:264) */
};

/* random stream */
enum {
    /* parity mode: normals z[chain][iter][d], uniforms u[chain][iter]; the uniform of an
     * iteration is read from its slot only when the reference would draw one (metrop.c:226-233) */
    MCMCHIP_RNG_INJECTED = 1,
    /* device mode: counter-based Philox4x32-10 keyed by (seed), counter = (iteration, global chain id, draw) */
    MCMCHIP_RNG_PHILOX = 2
};

/* which kernels evaluate the log density.  STREAM: the persistent per-chain kernel (every descriptor, debug trace).
 * MFMA: the contraction X . Theta on the fp64 matrix cores, for the descriptors that carry a design matrix of d <= 256
 * columns -- MCMCHIP_T_LOGISTIC and MCMCHIP_T_BINOMIAL_LOGIT (metrop and logdens, morphed or not; no debug trace) and
 * MCMCHIP_T_LOGISTIC_MODELS (logdens only).  AUTO takes MFMA when n * d >= 65 536, or n >= 4096 with >= 4096 chains (at
 * least 64 chains and n >= 1024 in any case).  The two sum the observations in different orders: a sharded run that
 * must reproduce an unsharded one bit for bit pins the path. */
enum { MCMCHIP_PATH_AUTO = 0, MCMCHIP_PATH_STREAM = 1, MCMCHIP_PATH_MFMA = 2 };

/* ------------------------------------------------------------------ */
/* metrop  -- replaces SEXP metrop(func1, initial, nbatch, blen, nspac,
 * scale, func2, debug, rho1, rho2)  (src/mcmc.h:8-9, metrop.c:19-301)  */
/* ------------------------------------------------------------------ */
typedef struct mcmchip_metrop_args {
    int64_t nchain;
    int32_t nbatch, blen, nspac;     /* all > 0 (metrop.c:60-65) */
    const double *initial;           /* nchain x d, chain-major; all finite (metrop.c:72-73) */
    int32_t scale_kind;
    const double *scale;
    int32_t out_kind;
    const int32_t *out_index;        /* d entries 0/1 when out_kind == INDEX */
    int32_t rng_mode;
    const double *z;                 /* INJECTED: nchain x niter x d */
    const double *u;                 /* INJECTED: nchain x niter */
    uint64_t seed;                   /* PHILOX key */
    uint64_t chain_offset;           /* PHILOX: global id of local chain 0 (results independent of sharding) */
    uint64_t iter_offset;            /* PHILOX: iterations already done (restart: split runs == one long run, tests/saveseed.R) */
    int32_t debug;                   /* fill the debug trace (metrop.c:129-180,237-257) */
    int32_t device_ptrs;
    int32_t path;                    /* MCMCHIP_PATH_* */
} mcmchip_metrop_args;

typedef struct mcmchip_metrop_out {
    double *accept;         /* nchain                      (metrop.c:289) */
    double *batch;          /* nchain x nbatch x nout : batch[(c*nbatch+ib)*nout+j]; per chain this is the
                               reference's dim_out x nbatch column-major `path` (metrop.c:103,279-280) */
    double *final;          /* nchain x d                  (metrop.c:291-297) */
    double *accept_batch;   /* nchain x nbatch             (metrop.c:282-283) */
    /* debug trace, NULL unless args.debug; iteration index l = ispac + nspac*(jbatch + blen*ibatch) (metrop.c:238) */
    double *current;        /* nchain x niter x d : state BEFORE the update */
    double *proposal;       /* nchain x niter x d */
    double *z;              /* nchain x niter x d */
    double *log_green;      /* nchain x niter : lud(proposal) - lud(current), may be -Inf */
    double *u;              /* nchain x niter : uniform drawn, NaN where the reference records NA (metrop.c:252-255) */
    int32_t *debug_accept;  /* nchain x niter */
    uint64_t iter_offset_out; /* args.iter_offset + nbatch*blen*nspac */
    int32_t nout;           /* dim of g(x) */
    int32_t path_used;
} mcmchip_metrop_out;

/* ------------------------------------------------------------------ */
/* temper -- replaces SEXP temper(func1, initial, neighbors, nbatch, blen,
 * nspac, scale, func2, debug, parallel, rho1, rho2) (mcmc.h:10-12,
 * temper.c:17-913).  `nens` independent ensembles advance in lockstep.   */
/* ------------------------------------------------------------------ */
/* output functionals of temper (func2 of temper.c:17; NULL = identity, temper.c:851-867) */
enum {
    MCMCHIP_TOUT_IDENTITY = 0,   /* parallel: the k x p state; serial: x (p values) */
    /* the corner indicator of the witch's-hat tests (WITCH descriptor only, bounds = its par):
     * parallel (tests/temp-par-witch.R:52-59): k values, as.numeric(all(|theta_i| > bnd_i)) per component;
     * serial (tests/temp-ser-witch.R:51-62): 2k values c(bar, baz), bar[i] = all(|theta| > bnd_i), baz[i] = 1 for
     * the current component i, zeros elsewhere */
    MCMCHIP_TOUT_WITCH_CORNER = 1,
    /* the index-vector and moment functionals that metrop has (metrop.c:435-497, vignettes/demo.Rnw:264), applied to
     * the identity output v (parallel: the k*p state entries in the state's own order state[i + k*j]; serial: x):
     * INDEX keeps v[j] where args.out_index[j] != 0, MOMENTS2 is c(v, v^2).  The reference's temper() takes an R
     * function here (temper.c:977-988); these two cover the common ones without a closure. */
    MCMCHIP_TOUT_INDEX = 2,
    MCMCHIP_TOUT_MOMENTS2 = 3
};
/* temper kernels (args.path; AUTO = 0 picks): LDS = one wavefront per ensemble with the state in LDS (any shape,
 * serial tempering, debug trace); RUN = lane-per-component registers, runs of iterations on disjoint components
 * executed together (parallel, k <= 64, p <= 4, closed-form descriptors, no trace).  Same results bit for bit. */
enum { MCMCHIP_PATH_LDS = 1, MCMCHIP_PATH_RUN = 2, MCMCHIP_PATH_TMFMA = 3 };
/* TMFMA (temper_mfma.hip): MCMCHIP_T_LOGISTIC_MODELS over a large data set.  The ensembles advance in lockstep and the
 * one or two log densities an iteration needs per ensemble (temper.c:547-845) are one launch of the fp64 matrix-core
 * density kernel over 2 nens states; same draws in the same order as the other two, log densities equal to rounding
 * (another summation order over the observations).  Identity / INDEX / MOMENTS2 output, no trace.  AUTO takes it from
 * MCMCHIP_TMFMA_MIN_N observations on (measured, 1024 ensembles of 8 sub-models: 106 against 134 us per iteration at
 * n = 4096, 119 / 424 at 16 384, 0.69 / 10.1 ms at 262 144; tools/temper_paths.py). */
#define MCMCHIP_TMFMA_MIN_N 8192

typedef struct mcmchip_temper_args {
    int64_t nens;
    int32_t ncomp;                   /* k >= 2 (temper.c:35-37) */
    int32_t parallel;                /* 1: parallel tempering, state k x p; 0: serial, state (i, x) */
    int32_t nbatch, blen, nspac;
    const int32_t *neighbors;        /* k x k, 0/1, symmetric (temper.c:29-41) */
    const double *initial;           /* parallel: nens x (k x p column-major, state[i + k*j], temper.c:563);
                                        serial:   nens x (1+p) = (i, x) with 1 <= i <= k (temper.c:55-63) */
    int32_t scale_kind;              /* CONSTANT | DIAGONAL | FULL, optionally | LIST */
    const double *scale;
    int32_t rng_mode;
    /* INJECTED: unif[ens][iter][4] = {u_which, u_choose(1), u_choose2, u_hastings} slots, each read
     * only when the reference draws it (temper.c:537,553,585,649,681-682,753,791,821);
     * norm[ens][iter][p] */
    const double *unif;
    const double *norm;
    uint64_t seed, ens_offset, iter_offset;
    int32_t device_ptrs;
    int32_t debug;                   /* fill the debug trace (temper.c:282-464, vignettes/debug.Rnw:64-243) */
    int32_t out_kind;                /* output functional (temper.c:977-988 evaluates an R closure): MCMCHIP_TOUT_* */
    int32_t path;                    /* MCMCHIP_PATH_AUTO, or force one kernel: MCMCHIP_PATH_LDS | MCMCHIP_PATH_RUN */
    const int32_t *out_index;        /* MCMCHIP_TOUT_INDEX: 0/1 mask over the identity output (k*p | p entries), host memory */
} mcmchip_temper_args;

typedef struct mcmchip_temper_out {
    double *batch;     /* out_kind != IDENTITY: nens x (nbatch x nout), per ensemble batch[kb + nbatch*j]
                          (nout: CORNER k | 2k, INDEX the number of kept entries, MOMENTS2 twice the identity's);
                          parallel: nens x (nbatch x k x p), per ensemble batch[kb + nbatch*(i + k*j)] (temper.c:871-875);
                          serial:   nens x (nbatch x p),     per ensemble batch[kb + nbatch*j]       (temper.c:877-879) */
    double *ibatch;    /* serial only: nens x (nbatch x k), ibatch[kb + nbatch*i] (temper.c:881-886); may be NULL if parallel */
    double *acceptx;   /* nens x k ; 0/0 = NaN for a component never tried (temper.c:889-890) */
    double *accepti;   /* nens x (k x k column-major); NaN (reference: NA_REAL) for non-neighbours (temper.c:892-897) */
    double *final;     /* same shape as initial */
    uint64_t iter_offset_out;
    /* debug trace, NULL unless args.debug.  Per ensemble, niter-major like the reference:
     * element (iter, j) at [iter + niter*j] (temper.c:533,591,604); NaN where the reference stores NA */
    int32_t *which;          /* nens x niter            : 1 = within-component update (temper.c:541) */
    double *unif_which;      /* nens x niter */
    double *state;           /* nens x niter x nstate   : state BEFORE the update (temper.c:528-535) */
    double *log_hastings;    /* nens x niter */
    double *unif_hastings;   /* nens x niter            : NaN unless log hastings < 0 (temper.c:583-587) */
    double *proposal;        /* nens x niter x (p+1)    : (i, y) | swap: (j, x_j) | jump: (j, x) */
    int32_t *acceptd;        /* nens x niter */
    double *norm;            /* nens x niter x p        : NaN for swap/jump iterations (temper.c:762,833) */
    double *unif_choose;     /* parallel: nens x niter x 2 (second NaN for within, temper.c:607-608); serial: nens x niter */
    double *coproposal;      /* parallel only: nens x niter x (p+1) : (i, x_i) (temper.c:597,728) */
} mcmchip_temper_out;

/* ------------------------------------------------------------------ */
/* library / target lifetime                                            */
/* ------------------------------------------------------------------ */
int mcmchip_init(int device);                 /* select device, create streams; idempotent */
const char *mcmchip_last_error(void);
const char *mcmchip_version(void);

/* copy a descriptor into HBM (desc pointers are host memory unless desc_device_ptrs) and
 * pre-tile X for the fp64 MFMA path; replaces capturing `func1`/`rho1` (metrop.c:35-38) */
int mcmchip_target_create(const mcmchip_target_desc *desc, int desc_device_ptrs, mcmchip_target **out);
void mcmchip_target_destroy(mcmchip_target *t);

/* evaluate the log density of `nchain` states (chain-major nchain x d; for temper kinds (1+d): (i, theta)).
 * replaces logh() (metrop.c:303-323); -Inf allowed, +Inf/NaN are reported in the values, not as errors */
int mcmchip_logdens(const mcmchip_target *t, const double *states, int64_t nchain, int path,
                    int device_ptrs, double *out);

/* mcmchip_logdens_noalloc -- NO COUNTERPART IN THE REFERENCE.  For callers that cannot call hipMalloc:
 * tight serving loops, harnesses that forbid allocation (it launches on the library's own stream and waits for it, so
 * it is NOT something to call under HIP graph capture).  The fp64 MFMA log density of
 * mcmchip_logdens(path = MFMA, device_ptrs = 1) with the same launch plan and therefore the same bits, but its
 * scratch (dimension-major states + per-split partial sums) lives in a fixed device array of
 * MCMCHIP_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends; the call makes no
 * hipMalloc / hipFree / event creation.  `states` (nchain x d) and `out` (nchain) are DEVICE pointers on the current
 * device; t must be a logistic descriptor with d <= 256.  Needs (dpad + nunits) * ceil(nchain / W) * W doubles (W = 256
 * chains a workgroup up to d = 128, 128 beyond; dpad = d rounded up to 8 / 16 / 64 for d <= 64 / <= 128 / <= 256),
 * nunits <= 256 the units the observations are summed in (e.g. 8192 chains at d = 64 with 256 units: 20 MiB); beyond the array it returns MCMCHIP_ECAP
 * with the required size in the message.  Calls on one GPU share the array: calls that overlap on one GPU race. */
#define MCMCHIP_NOALLOC_SCRATCH_BYTES (32u << 20)
int mcmchip_logdens_noalloc(const mcmchip_target *t, const double *states, int64_t nchain, double *out);

int mcmchip_metrop(const mcmchip_target *t, const mcmchip_metrop_args *args, mcmchip_metrop_out *out);

int mcmchip_temper(const mcmchip_target *t, const mcmchip_temper_args *args, mcmchip_temper_out *out);

/* The same calls over several GPUs of the node, from ONE process -- the reference's front ends make one native call
 * per run (R/metrop.R:48-51, R/temper.R:66-68) and an R session is one process.  devices[0 .. ndev): HIP device
 * indices (a device may be listed more than once: its blocks then run side by side on it).  The nchain chains (nens
 * ensembles) are cut into ndev contiguous blocks that differ by at most one unit; one host thread per block runs the
 * call above on its device with the descriptor replicated there (once, kept with `t` and released with it), Philox
 * keyed by the global chain / ensemble id (args.chain_offset + block start), and writes its results at the block's
 * place in the caller's arrays.  Results do not depend on the device list (metrop: with path AUTO the kernel is chosen
 * from the whole call's chain count; bit for bit on the MFMA path and in temper, to rounding on the stream path,
 * whose summation order follows the block's chain count).  mcmchip_init need not have been called for the listed
 * devices.  device_ptrs = 1: every pointer must be readable and writable from every listed device.  The interrupt
 * callback keeps running on the calling thread.  mcmchip_last_kernel_time then reports the first block's kernels. */
int mcmchip_metrop_multi(const mcmchip_target *t, const mcmchip_metrop_args *args, mcmchip_metrop_out *out,
                         const int32_t *devices, int32_t ndev);
int mcmchip_temper_multi(const mcmchip_target *t, const mcmchip_temper_args *args, mcmchip_temper_out *out,
                         const int32_t *devices, int32_t ndev);

/* initseq -- replaces SEXP initseq(SEXP x) (mcmc.h:14, initseq.c:7-128), batched over `ncol`
 * series stored column-major x[t + len*c].  Input is used as given: the caller centres it
 * (R/initseq.R:4).  Output vectors have data-dependent length n_out[c] <= len/2
 * (initseq.c:24-47); caller passes capacity `cap` per column and arrays ncol x cap.
 * Returns MCMCHIP_ECAP (n_out filled with the needed lengths) if cap is too small. */
int mcmchip_initseq(const double *x, int64_t len, int64_t ncol, int device_ptrs,
                    int64_t cap, int64_t *n_out,
                    double *gamma0, double *Gamma_pos, double *Gamma_dec, double *Gamma_con,
                    double *var_pos, double *var_dec, double *var_con);

/* initseq of a series that is NOT centred yet: R/initseq.R:3-4 (x - mean(x), then the native call) in one call that reads
 * the series once per round and leaves it as it is -- for a batch resident in HBM centring in place costs a read pass and
 * a read-write pass over the series before the first lag product (C5: 123 ms next to initseq's 73; this call: 80).  The lag sums are taken
 * about a provisional centre per column (the mean of 64 runs of 1024 rows spread over the column) and carried over to the
 * mean exactly: with z = x - centre, delta = mean(z),
 *     sum_t (z[t] - delta)(z[t+k] - delta) = sum_t z[t] z[t+k] - (len + k) delta^2 + delta (H_k + T_k),
 * H_k / T_k the sums of the first / last k values of z (a few thousand rows per column, gathered on the device, summed on the host).  Same
 * arguments, results and errors as mcmchip_initseq; `mean` (host, ncol doubles, or NULL) receives the column means.
 * Results agree with mcmchip_initseq on the centred series to rounding (~1e-13 relative; the tests hold it to the 1e-10
 * of the oracle comparison). */
int mcmchip_initseq_uncentred(const double *x, int64_t len, int64_t ncol, int device_ptrs,
                              int64_t cap, int64_t *n_out,
                              double *gamma0, double *Gamma_pos, double *Gamma_dec, double *Gamma_con,
                              double *var_pos, double *var_dec, double *var_con, double *mean /* host, or NULL */);

/* The two halves of initseq for a series sharded BY TIME over several GPUs (no counterpart in the
 * single-process reference): the raw lag sums  sums[c * nlags + l] = sum_t x[t + len*c] * x[t + l + len*c]  over the
 * whole array given (terms past its end are zero), l = 0 .. nlags-1, which add up over shards (a shard owning rows
 * [lo, hi) holds a halo of nlags rows and subtracts the sums of the halo alone, mcmc_amd/dist.py), and the stop rule
 * plus the three sequences (initseq.c:24-105) applied to the TOTAL sums of a series of length `len`.  The second is
 * host-only.  n_out[c] = -1: column c has not reached its stopping lag within nlags, come back with more. */
int mcmchip_lagsums(const double *x, int64_t len, int64_t ncol, int64_t nlags, int device_ptrs, double *sums /* host */);
int mcmchip_initseq_from_lagsums(const double *sums, int64_t len, int64_t ncol, int64_t nlags, int64_t cap, int64_t *n_out,
                                 double *gamma0, double *Gamma_pos, double *Gamma_dec, double *Gamma_con,
                                 double *var_pos, double *var_dec, double *var_con);

/* A series kept in HBM between calls -- no counterpart in the reference, where the batch matrix lives in R's heap.
 * Uploads the n x p host matrix x once and returns a device pointer laid out as initseq / lagsums / olbm want it
 * (column-major, x[t + n*c], the reference's X(I,J) of olbm.c:19) for use with device_ptrs = 1: initseq and olbm on the
 * same batch then cross PCIe once instead of twice.  row_major != 0: x is in C order (x[t*p + c], what numpy and most
 * non-R callers hold); it is uploaded as it is and transposed on the device, which is ~10x faster than transposing a
 * large matrix on the host first.  The pointer is released with mcmchip_series_free (NULL is allowed). */
int mcmchip_series_upload(const double *x, int64_t n, int64_t p, int row_major, double **dev);
int mcmchip_series_free(double *dev);
/* x - mean(x) per column, in place on a resident series (R/initseq.R:3 does it in R before the native call); the
 * means go to `mean` (host, p doubles) when it is not NULL.  No counterpart in the reference's C. */
int mcmchip_series_centre(double *dev, int64_t n, int64_t p, double *mean);

/* olbm -- replaces void olbm(double *x, int *nin, int *pin, int *lin, double *mean,
 * double *var, int *nocalcin) (mcmc.h:16-17, olbm.c:22-78).  x is n x p column-major.
 * nocalc != 0: `mean` is input (used as given); nocalc == 0: means computed and returned (for n >= 2^20 and p <= 256
 * in the same pass over x as the matrix, not in a pass of their own as in olbm.c:38-43; they come out correctly
 * rounded to an ulp, where the reference's sequential sum is good to ~sqrt(n) ulp).
 * p <= 8 runs as a stream on the vector units (olbm_small.hip: 0.4 - 5 ms per 1e8 rows; MCMCHIP_OLBM_SMALL=0 keeps the
 * matrix-core kernel).
 * 3 <= p <= 192 with windows of >= 256 rows (at least four window lengths of windows; p <= 8 only from n p = 4e7 on) runs on
 * the matrix cores with one lane per (tooth, column) and no scan (olbm_lane.hip: 0.8 ms (p = 4) .. 14 ms (p = 64) .. 79 ms
 * (p = 192) per 1e8 rows, 1.25 x the input read; MCMCHIP_OLBM_LANE=0 keeps the kernels that served before).
 * Windows of >= 1024 rows of 193 .. 256 columns (at least two window lengths of windows) are computed with x read once
 * instead of twice (olbm_comb.hip; MCMCHIP_OLBM_COMB=0 keeps the two-stream kernel).
 * Accuracy, stated here because it is a deviation a caller should know: the device agrees with an 80-bit evaluation
 * of the definition to ~1e-14 of sqrt(var_ii var_jj).  olbm.c itself is not that good -- it adds the n terms of a mean
 * one after the other and carries ONE running window sum through all n - len updates (olbm.c:38-43, 52-62) -- so the
 * device is held to the reference's arithmetic (the oracle) at 1e-10 for len <= 10 000 and len <= n / 4, and at 1e-9 for
 * longer windows, where olbm.c is ~1.4e-10 away from the 80-bit value itself (tools/olbm_conditioning.py,
 * tests/test_gpu_round2.py::test_olbm_one_pass_route_vs_oracle, ::test_olbm_comb_route_vs_oracle). */
int mcmchip_olbm(const double *x, int64_t n, int32_t p, int64_t len, double *mean, double *var,
                 int32_t nocalc, int device_ptrs);

/* Host-only diagnostic (no counterpart in the reference, no GPU call): builds the work plan olbm uses for windows of
 * >= 1024 rows (olbm_comb.hip: teeth one window length apart swept together, split over `groups` workgroups) and checks
 * it -- every window in exactly one piece, pieces in time order without gaps, step counts consistent.  Returns the
 * number of violations (0 = sound; -1: bad argument). */
int mcmchip_olbm_comb_plan_check(int64_t nwin, int64_t len, int32_t teeth, int64_t groups, int32_t *npieces,
                                 int64_t *min_steps, int64_t *max_steps);
/* The same check for the plan of olbm_lane.hip (four teeth per wave, units of the sweep dealt round-robin to `groups`
 * workgroups).  Host-only, no counterpart in the reference. */
int mcmchip_olbm_lane_plan_check(int64_t nwin, int64_t len, int64_t groups, int32_t *npieces, int64_t *min_steps,
                                 int64_t *max_steps);
/* ... and for p <= 8, where `sub` = 2 or 4 consecutive units of that plan share an instruction (olbm_few_kernel). */
int mcmchip_olbm_few_plan_check(int64_t nwin, int64_t len, int64_t groups, int32_t sub, int32_t *npieces, int64_t *min_steps,
                                int64_t *max_steps);

/* Cross-chain diagnostics -- NO COUNTERPART IN THE REFERENCE, which runs one chain (its users read colMeans(out$batch)
 * and apply(out$batch, 2, sd) / sqrt(out$nbatch), vignettes/demo.Rnw:210-222).  A batched call returns nchain x nbatch x
 * nout batch means (mcmchip_metrop_out.batch); this reduces them over chains, per output functional j:
 *   m_c = mean of chain c's batch means, s2_c = their variance (divisor nbatch - 1)
 *   grand_mean = mean_c m_c;  between = sum_c (m_c - grand_mean)^2 / (nchain - 1);  within = mean_c s2_c
 *   mcse_between = sqrt(between / nchain)           the chains as replicates
 *   mcse_within  = sqrt(within / (nbatch nchain))   the batch-means estimate of demo.Rnw:218-222, pooled over chains
 *   rhat = sqrt(((nbatch-1)/nbatch within + between) / within)   (potential scale reduction on the batch-mean scale)
 *   mcse_initseq (optional, non-NULL asks for it) = sqrt(mean_c var.con_c / (nbatch nchain)), var.con_c from initseq
 *     (R/initseq.R, initseq.c) on chain c's centred batch means: for a few long chains (nchain <= 65535,
 *     nchain * nout * nbatch <= 2^28)
 * Every output pointer may be NULL; the vectors are host memory, chain_mean / chain_var (nchain x nout) follow
 * device_ptrs like `batch`.  stats (5 x nout doubles, host: count, mean, M2, sum of s2_c, sum of var.con_c per
 * functional) are the sufficient statistics a sharded run exchanges: mcmchip_chain_summary_combine (host only) merges
 * the stats of `nranks` shards IN RANK ORDER and fills the same vectors.  One chain: between / mcse_between / rhat are
 * NaN (0/0); one batch: within is NaN. */
typedef struct mcmchip_chain_summary_out {
    double *grand_mean, *between, *within, *mcse_between, *mcse_within, *rhat;   /* nout each */
    double *mcse_initseq;          /* nout, or NULL: skip the per-chain initseq */
    double *stats;                 /* 5 x nout */
    double *chain_mean, *chain_var; /* nchain x nout */
} mcmchip_chain_summary_out;
int mcmchip_chain_summary(const double *batch, int64_t nchain, int32_t nbatch, int32_t nout, int device_ptrs,
                          mcmchip_chain_summary_out *out);
int mcmchip_chain_summary_combine(const double *stats /* nranks x 5 x nout */, int32_t nranks, int32_t nbatch, int32_t nout,
                                  mcmchip_chain_summary_out *out /* chain_mean, chain_var ignored; stats = the total */);

/* Interrupts and launch length.  The reference calls R_CheckUserInterrupt() every iteration (metrop.c:269,
 * temper.c:848).  Here a sampler call is a sequence of kernel launches of about a quarter of a second each (one
 * short calibration launch first; the state is carried between launches, results are bit for bit those of a single
 * launch), and between two launches the host calls `poll(ctx)`: a non-zero return abandons the call, which releases
 * its device buffers and returns MCMCHIP_EINTR.  An R glue passes a function that runs R_CheckUserInterrupt under
 * R_ToplevelExec.  poll == NULL (default): no callback.
 * mcmchip_set_launch_limit: max_iters > 0 fixes the iterations per launch (tests; tempering rounds it up to whole
 * 64-iteration windows), 0 restores the time-based rule; budget_seconds > 0 changes the quarter second. */
int mcmchip_set_interrupt_poll(int (*poll)(void *ctx), void *ctx);
int mcmchip_set_launch_limit(int64_t max_iters, double budget_seconds);

/* timing of the most recent call's dominant kernel, measured with HIP events on the
 * library's own stream (bench.py roofline): average ms per launch and launch count */
int mcmchip_last_kernel_time(double *avg_ms, int64_t *launches, char *name, int name_cap);

/* synchronise the library's stream */
int mcmchip_sync(void);

#ifdef __cplusplus
}
#endif
#endif /* MCMCHIP_H */
