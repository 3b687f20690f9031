// kernels.h -- internal launch interfaces between the C-ABI glue (capi.hip) and the kernel files.
#pragma once
#include "common.h"

// everything the samplers need, with all pointers already in HBM
struct MetropDev {
    // target
    int kind, d;
    long n;
    const double *x, *y, *par;
    long npar;
    double prior_c;
    int morph_flags;          // morph.metrop: the chain runs on the morphed state (R/morph.R:121-181)
    double morph_b, morph_r, morph_p;
    const double *morph_center;
    const struct SoftplusTables *sp_tables;   // softplus tables in HBM (stream path: staged in LDS when they fit)
    int no_table;             // 1: library exp/log1p even where the table would fit (MCMCHIP_NO_SOFTPLUS_TABLE=1, tests)
    // run shape
    long nchain;
    int nbatch, blen, nspac;
    int scale_kind;
    const double *scale;
    int nout, sq_from;        // g(x)_k = x[out_map[k]] (squared when k >= sq_from)
    const int *out_map;
    int rng_mode;
    const double *z, *u;      // injected streams
    unsigned long long seed, chain_offset, iter_offset;
    int debug;
    // one launch of the stream kernel: iterations [l_begin, l_end); resume = 1 continues from `final_` and `carry`
    // (per chain: cached log density, accepted, accepted in this batch, batch accumulator[nout])
    long l_begin, l_end;
    int resume;
    double *carry;
    // chain-major inputs/outputs
    const double *initial;
    double *final_, *accept, *batch, *accept_batch;
    double *dbg_current, *dbg_proposal, *dbg_z, *dbg_lg, *dbg_u;
    int *dbg_accept;
    int *flag;
};

// metrop_stream.hip: one persistent launch, G lanes per chain, whole nbatch*blen*nspac loop in-kernel
// batch_home != nullptr: the caller's host array for p.batch ([chain][batch][nout], pageable); a run of several launches
// whose batch means are large copies them there launch by launch, each copy under the next launch (on `side`), and sets
// *went_home; otherwise the caller copies at the end
int launch_metrop_stream(const MetropDev &p, hipStream_t s, void *batch_home = nullptr, hipStream_t side = nullptr,
                         bool *went_home = nullptr);
// log density of nchain states (chain-major; temper kinds: (i, theta))
int launch_logdens_stream(const mcmchip_target *t, const double *states, long nchain, double *out,
                          hipStream_t s);

// metrop_mfma.hip: lockstep iterations, fp64 MFMA X.Theta contraction + update kernel per iteration
int run_metrop_mfma(const mcmchip_target *t, const MetropDev &p, DeviceCtx *dc);
int launch_logdens_mfma(const mcmchip_target *t, const double *states, long nchain, double *out,
                        DeviceCtx *dc);
int build_mfma_tiles(mcmchip_target *t, hipStream_t s);
// allocation-free variant of launch_logdens_mfma on the per-device static scratch array (set up by mcmchip_init)
int mfma_scratch_setup(DeviceCtx *dc);
int launch_logdens_mfma_noalloc(const mcmchip_target *t, const double *states, long nchain, double *out, DeviceCtx *dc);

// temper.hip
struct TemperDev {
    int kind, p;              // target kind, dimension of theta
    const double *par;
    long npar;
    const double *x, *y;      // LOGISTIC_MODELS: design matrix n x p and responses
    long n;
    double prior_c;
    long nens;
    int k, parallel, nbatch, blen, nspac;
    const int *neighbors;     // k x k
    int maxnn;                // largest number of neighbours of a component (host-counted)
    const double *initial;
    int scale_kind;
    const double *scale;
    int rng_mode;
    const double *unif, *norm;
    unsigned long long seed, ens_offset, iter_offset;
    double *batch, *ibatch, *acceptx, *accepti, *final_;
    int debug;
    int out_kind;             // MCMCHIP_TOUT_*
    // MCMCHIP_TOUT_INDEX / _MOMENTS2: g_i = v[out_map[i]] (squared when i >= sq_from), v the identity output
    const int *out_map;
    int nout_map, sq_from;
    int path;                 // MCMCHIP_PATH_AUTO | _LDS | _RUN
    // one launch: iterations [it_begin, it_end) (multiples of 64 except the end of the run); resume = 1 restores the
    // ensemble's record from `carry`
    long it_begin, it_end;
    int resume;
    double *carry;
    int *dbg_which, *dbg_acceptd;
    double *dbg_unif_which, *dbg_state, *dbg_lh, *dbg_uh, *dbg_proposal, *dbg_norm, *dbg_unif_choose, *dbg_coproposal;
    int *flag;
};
int launch_temper(const TemperDev &p, hipStream_t s);
// temper_mfma.hip: the ensembles in lockstep, log densities of the model-selection posterior on the matrix cores
bool temper_mfma_applies(const mcmchip_target *t, const TemperDev &q);
int run_temper_mfma(const mcmchip_target *t, const TemperDev &q, DeviceCtx *dc);
// metrop_mfma.hip, for that driver
int mfma_density_plan(const mcmchip_target *t, long nchain, DeviceCtx *dc, long *ldc, int *nunits);
int mfma_density_launch(const mcmchip_target *t, long nchain, const double *theta, double *partial, DeviceCtx *dc);

// capi.hip: VAL_* bits of a device-resident input (non-finite values; serial-tempering component index)
int device_validate(DeviceCtx *dc, const double *v, long n, long stride, int k, int *bits);

// diag.hip
int run_initseq(const double *x, long len, long ncol, long cap, long *n_out, double *gamma0,
                double *Gpos, double *Gdec, double *Gcon, double *vpos, double *vdec, double *vcon,
                DeviceCtx *dc, bool about_mean = false, double *mean_out = nullptr);
int run_lagsums(const double *x, long len, long ncol, long nlags, double *sums, DeviceCtx *dc);
int run_initseq_from_lagsums(const double *sums, long len, long ncol, long nlags, long cap, long *n_out, double *gamma0,
                             double *Gpos, double *Gdec, double *Gcon, double *vpos, double *vdec, double *vcon);
int run_series_centre(double *x, long n, long p, double *mean, DeviceCtx *dc);
int run_olbm(const double *x, long n, int p, long len, double *mean, double *var, int nocalc,
             DeviceCtx *dc);

// summary.hip: cross-chain reduction of batch means; stats_host = 5 x nout sufficient statistics
int run_chain_summary(const double *batch, long nchain, int nbatch, int nout, int want_initseq, double *chain_mean_out,
                      double *chain_var_out, double *stats_host, DeviceCtx *dc);
void combine_chain_stats(const double *stats, int nranks, int nout, double *total);
void chain_summary_vectors(const double *stats, int nbatch, int nout, mcmchip_chain_summary_out *o, bool have_initseq);
