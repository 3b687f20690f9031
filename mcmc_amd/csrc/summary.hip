// summary.hip -- cross-chain diagnostics of a batched run (mcmchip_chain_summary).
//
// No counterpart in the reference, which runs one chain: a call here returns nchain x nbatch x nout batch means
// (metrop.c:279-280 per chain), and 65 536 chains produce nothing a user can read without a reduction over chains.
// Per output functional j, with b[c][i] the batch means of chain c (i < nbatch):
//     m_c = mean_i b[c][i]                         the chain's estimate (colMeans(out$batch), vignettes/demo.Rnw:210)
//     s2_c = sum_i (b[c][i] - m_c)^2 / (nbatch - 1)   variance of its batch means (demo.Rnw:218-222: MCSE^2 = s2_c / nbatch)
//     grand mean = mean_c m_c,  between = sum_c (m_c - grand mean)^2 / (nchain - 1),  within = mean_c s2_c
//     MCSE of the grand mean two ways: sqrt(between / nchain) (chains as replicates) and
//       sqrt(within / (nbatch nchain)) (batch means, pooled over chains); their ratio is the usual check that the
//       chains agree: rhat = sqrt(((nbatch-1)/nbatch within + between) / within) on the batch-mean scale
//     optionally the initial-sequence estimator of R/initseq.R per chain and functional (series = the chain's batch
//       means, centred), pooled: sqrt(mean_c var.con_c / (nbatch nchain))
// The reduction carries {count, mean, M2 = sum of squared deviations, sum of s2_c, sum of var.con_c} per functional
// and combines them pairwise (Chan et al.: no sums of raw squares, so a mean of 1 with a spread of 1e-4 loses nothing)
// in a fixed order -- a lane's chains, the lanes of a workgroup, the workgroups, then the ranks of a sharded run
// (mcmc_amd/dist.py::chain_summary_sharded all-gathers 5 nout doubles per rank and combines them in rank order):
// reruns give the same bits.
#include "kernels.h"
#include <math.h>
#include <vector>

namespace {
struct Moments {
    double n, mean, m2, w;
};

__host__ __device__ inline Moments combine(const Moments &a, const Moments &b)
{
    if (b.n == 0.0) return a;
    if (a.n == 0.0) return b;
    Moments r;
    r.n = a.n + b.n;
    const double delta = b.mean - a.mean;
    r.mean = a.mean + delta * (b.n / r.n);
    r.m2 = a.m2 + b.m2 + delta * delta * (a.n * b.n / r.n);
    r.w = a.w + b.w;
    return r;
}
}  // namespace

// block = JT functionals x (256 / JT) chain lanes (JT a power of two <= 64); a lane walks its chains of the group,
// two passes over a chain's nbatch values (mean, then squared deviations)
__global__ __launch_bounds__(256) void chain_stats_kernel(const double *__restrict__ batch, long nchain, int nbatch, int nout,
                                                          int jt_log2, long chains_per_group, double *__restrict__ chain_mean,
                                                          double *__restrict__ chain_var, double *__restrict__ partial)
{
    __shared__ Moments sh[256];
    const int JT = 1 << jt_log2, CT = 256 >> jt_log2;
    const int jx = threadIdx.x & (JT - 1), cy = threadIdx.x >> jt_log2;
    const int j = blockIdx.x * JT + jx;
    const long c0 = (long)blockIdx.y * chains_per_group;
    long c1 = c0 + chains_per_group;
    if (c1 > nchain) c1 = nchain;
    Moments acc = {0.0, 0.0, 0.0, 0.0};
    if (j < nout)
        for (long c = c0 + cy; c < c1; c += CT) {
            const double *b = batch + (size_t)c * nbatch * nout + j;
            double s = 0.0;
            for (int i = 0; i < nbatch; i++) s += b[(size_t)i * nout];
            const double m = s / nbatch;
            double ss = 0.0;
            for (int i = 0; i < nbatch; i++) {
                const double d = b[(size_t)i * nout] - m;
                ss += d * d;
            }
            const double s2 = ss / (double)(nbatch - 1);            // nbatch = 1: 0 / 0 = NaN, as var() of one value is NA
            if (chain_mean) chain_mean[(size_t)c * nout + j] = m;
            if (chain_var) chain_var[(size_t)c * nout + j] = s2;
            acc = combine(acc, Moments{1.0, m, 0.0, s2});
        }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (cy == 0 && j < nout) {
        Moments t = sh[jx];
        for (int k = 1; k < CT; k++) t = combine(t, sh[(k << jt_log2) + jx]);
        double *p = partial + ((size_t)blockIdx.y * nout + j) * 4;
        p[0] = t.n; p[1] = t.mean; p[2] = t.m2; p[3] = t.w;
    }
}

// the groups in order: stats[k * nout + j], k = count, mean, M2, sum of within-chain variances
__global__ void chain_stats_finish_kernel(const double *__restrict__ partial, int ngroups, int nout, double *__restrict__ stats)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nout) return;
    Moments t = {0.0, 0.0, 0.0, 0.0};
    for (int g = 0; g < ngroups; g++) {
        const double *p = partial + ((size_t)g * nout + j) * 4;
        t = combine(t, Moments{p[0], p[1], p[2], p[3]});
    }
    stats[j] = t.n;
    stats[nout + j] = t.mean;
    stats[2 * nout + j] = t.m2;
    stats[3 * nout + j] = t.w;
}

// series[(c * nout + j) * nbatch + i] = batch[(c * nbatch + i) * nout + j] - chain_mean[c * nout + j]: every (chain,
// functional) a centred column for initseq (R/initseq.R:4), through a 32 x 32 LDS tile so both sides coalesce
__global__ __launch_bounds__(256) void chain_series_kernel(const double *__restrict__ batch, const double *__restrict__ chain_mean,
                                                           int nbatch, int nout, double *__restrict__ series)
{
    __shared__ double tile[32][33];
    const long c = blockIdx.z;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        if (i < nbatch && j < nout) tile[r][tx] = batch[((size_t)c * nbatch + i) * nout + j];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + tx;
        if (i < nbatch && j < nout) series[((size_t)c * nout + j) * nbatch + i] = tile[tx][r] - chain_mean[(size_t)c * nout + j];
    }
}

// stats (5 x nout, one rank or several combined) -> the summary vectors (any may be null)
static void summary_from_stats(const double *st, int nbatch, int nout, double *grand_mean, double *between, double *within,
                               double *mcse_between, double *mcse_within, double *rhat, double *mcse_initseq)
{
    for (int j = 0; j < nout; j++) {
        const double n = st[j], mean = st[nout + j], m2 = st[2 * nout + j], w = st[3 * nout + j], vc = st[4 * nout + j];
        const double B = m2 / (n - 1.0), W = w / n;               // one chain: 0 / 0 = NaN
        if (grand_mean) grand_mean[j] = mean;
        if (between) between[j] = B;
        if (within) within[j] = W;
        if (mcse_between) mcse_between[j] = sqrt(B / n);
        if (mcse_within) mcse_within[j] = sqrt(W / ((double)nbatch * n));
        if (rhat) rhat[j] = sqrt((((double)nbatch - 1.0) / (double)nbatch * W + B) / W);
        if (mcse_initseq) mcse_initseq[j] = sqrt((vc / n) / ((double)nbatch * n));
    }
}

int run_chain_summary(const double *batch, long nchain, int nbatch, int nout, int want_initseq, double *chain_mean_out,
                      double *chain_var_out, double *stats_host, DeviceCtx *dc)
{
    hipStream_t s = dc->stream;
    int jt_log2 = 0;
    while ((1 << jt_log2) < nout && jt_log2 < 6) jt_log2++;
    const int JT = 1 << jt_log2, CT = 256 >> jt_log2;
    long ngroups = (nchain + CT - 1) / CT;
    if (ngroups > 1024) ngroups = 1024;
    long cpg = (nchain + ngroups - 1) / ngroups;
    cpg = (cpg + CT - 1) / CT * CT;
    ngroups = (nchain + cpg - 1) / cpg;
    DevBuf partial, stats, cmean, series;
    HIP_TRY(partial.alloc((size_t)ngroups * nout * 4 * 8));
    HIP_TRY(stats.alloc((size_t)5 * nout * 8));
    double *cm = chain_mean_out;
    if (want_initseq && !cm) {
        HIP_TRY(cmean.alloc((size_t)nchain * nout * 8));
        cm = cmean.as<double>();
    }
    timing_reset("chain_stats_kernel");
    timing_begin(s);
    hipLaunchKernelGGL(chain_stats_kernel, dim3((unsigned)((nout + JT - 1) / JT), (unsigned)ngroups), dim3(256), 0, s, batch, nchain,
                       nbatch, nout, jt_log2, cpg, cm, chain_var_out, partial.as<double>());
    timing_end(s);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(chain_stats_finish_kernel, dim3((unsigned)((nout + 63) / 64)), dim3(64), 0, s, partial.as<double>(),
                       (int)ngroups, nout, stats.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(stats_host, stats.p, (size_t)4 * nout * 8, hipMemcpyDeviceToHost, s));
    for (int j = 0; j < nout; j++) stats_host[4 * nout + j] = 0.0;
    if (want_initseq) {
        // the chains' batch means as nchain * nout centred columns of length nbatch, initseq on all of them, var.con pooled
        const long ncol = nchain * nout;
        const long cap = nbatch / 2 > 0 ? nbatch / 2 : 1;
        HIP_TRY(series.alloc((size_t)ncol * nbatch * 8));
        hipLaunchKernelGGL(chain_series_kernel, dim3((unsigned)((nbatch + 31) / 32), (unsigned)((nout + 31) / 32), (unsigned)nchain),
                           dim3(256), 0, s, batch, cm, nbatch, nout, series.as<double>());
        HIP_TRY(hipGetLastError());
        std::vector<long> n_out((size_t)ncol);
        std::vector<double> g0((size_t)ncol), vp((size_t)ncol), vd((size_t)ncol), vc((size_t)ncol);
        std::vector<double> Gp((size_t)ncol * cap), Gd((size_t)ncol * cap), Gc((size_t)ncol * cap);
        if (int rc = run_initseq(series.as<double>(), nbatch, ncol, cap, n_out.data(), g0.data(), Gp.data(), Gd.data(), Gc.data(),
                                 vp.data(), vd.data(), vc.data(), dc))
            return rc;
        for (int j = 0; j < nout; j++) {
            double t = 0.0;
            for (long c = 0; c < nchain; c++) t += vc[(size_t)c * nout + j];       // chains in order
            stats_host[4 * nout + j] = t;
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    return MCMCHIP_OK;
}

// ranks in order (host only)
void combine_chain_stats(const double *stats, int nranks, int nout, double *total)
{
    for (int j = 0; j < nout; j++) {
        Moments t = {0.0, 0.0, 0.0, 0.0};
        double vc = 0.0;
        for (int r = 0; r < nranks; r++) {
            const double *p = stats + (size_t)r * 5 * nout;
            t = combine(t, Moments{p[j], p[nout + j], p[2 * nout + j], p[3 * nout + j]});
            vc += p[4 * nout + j];
        }
        total[j] = t.n;
        total[nout + j] = t.mean;
        total[2 * nout + j] = t.m2;
        total[3 * nout + j] = t.w;
        total[4 * nout + j] = vc;
    }
}

void chain_summary_vectors(const double *stats, int nbatch, int nout, mcmchip_chain_summary_out *o, bool have_initseq)
{
    summary_from_stats(stats, nbatch, nout, o->grand_mean, o->between, o->within, o->mcse_between, o->mcse_within, o->rhat,
                       have_initseq ? o->mcse_initseq : nullptr);
}
