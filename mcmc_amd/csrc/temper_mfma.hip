// temper_mfma.hip -- serial and parallel tempering over the model-selection logistic posterior
// (MCMCHIP_T_LOGISTIC_MODELS: tests/temp-par.R:39-56, vignettes/bfst.Rnw:472-489) with the log densities on the fp64
// matrix cores, for data sets too large for one wavefront per ensemble.
//
// temper_kernel (temper.hip) gives an ensemble one wavefront, whose 64 lanes share the n observations of every
// density evaluation: at n = 1e5 an iteration is 1600 softplus terms per lane, one ensemble after the other's
// dependent chain.  Here the ensembles advance in LOCKSTEP, as the chains of metrop_mfma.hip do: an iteration of
// temper() (temper.c:517-887) needs one new log density per ensemble (within-component update, serial jump) or two
// (swap: h(j, x_i) and h(i, x_j), temper.c:738-739), so every iteration is ONE launch of logdens_mfma_kernel over
// 2 nens "chains" (slot 0 and 1 of each ensemble; masks folded into the states, as for mcmchip_logdens) and one launch
// of a per-ensemble kernel (one lane per ensemble) that finishes the iteration -- Hastings ratio, accept rule, swap /
// commit, counters, batch sums -- and draws and stages the next proposal.  State, cached log densities
// (temper.c:123-124), counters and batch accumulators live in HBM, ensemble-minor, so that lanes coalesce.
//
// Same draws in the same order as temper_kernel (injected streams, or Philox addressed by (ensemble,
// iteration, block)), same acceptance-table and batch-means layout; the log densities differ from temper_kernel's in
// the summation order of the observations (rounding), which decides nothing unless a uniform falls within an ulp of
// the Hastings ratio.  No debug trace on this path.
#include "kernels.h"
#include "philox.h"
#include <math.h>
#include <vector>

namespace {

struct TmParams {
    long nens, lde;                 // ensembles; stride of the ensemble-minor arrays
    int k, p, parallel, maxnn, nstate, nout;
    int nbatch, blen, nspac;
    int scale_kind;
    const double *scale;
    int rng_mode;
    const double *unif, *norm;
    unsigned long long seed, ens_offset, iter_offset;
    const int *nn, *nb;             // neighbour counts [k] and ascending neighbour lists [k][maxnn] (temper.c:497-510)
    const int *neighbors;           // k x k
    const double *par;              // k, k x p masks, k log pseudo-priors, constant
    double prior_c, pad_term;
    // the density launch: chain c of the batch reads theta[:, c] and leaves partial[:, c]
    long ldc;
    int nunits;
    double *theta;
    const double *partial, *colsum;
    // per-ensemble records
    double *S, *L, *Y, *buf, *ibuf;
    unsigned *axn, *axd, *ain, *aid;
    int *mv;                        // the staged move: within?, ci, cj, neighbour slot
    const double *initial;
    double *batch, *ibatch, *acceptx, *accepti, *final_;
    int out_kind, nout_map, sq_from;
    const int *out_map;
    int *flag;
};

__device__ __forceinline__ int tm_pick1(double u, int m)
{
    int i = (int)trunc(m * u) + 1;   // temper.c:555-556
    if (i > m) i--;
    return i;
}

// theta[:, c] = beta * mask_comp (eta = X (beta * mask), tests/temp-par.R:48-50); beta_j = b[j * bs]
__device__ __forceinline__ void tm_stage(const TmParams &q, long c, int comp, const double *b, long bs)
{
    const double *mask = q.par + 1 + (size_t)(comp - 1) * q.p;
    for (int j = 0; j < q.p; j++) q.theta[(size_t)j * q.ldc + c] = mask[j] != 0.0 ? b[(size_t)j * bs] : 0.0;
}
__device__ __forceinline__ void tm_stage_zero(const TmParams &q, long c)
{
    for (int j = 0; j < q.p; j++) q.theta[(size_t)j * q.ldc + c] = 0.0;
}

// h(comp, beta) from the density launch's partial sums of chain c (the association of metrop_update_kernel's
// models branch): sum_i [eta_i/2 - M(eta_i)] - prior_c |beta|^2 + constant + log pseudo-prior
__device__ __forceinline__ double tm_value(const TmParams &q, long c, int comp, const double *b, long bs)
{
    double part = 0.0;
    for (int u = 0; u < q.nunits; u++) part += q.partial[(size_t)u * q.ldc + c];
    double lin = 0.0, ss = 0.0;
    for (int j = 0; j < q.p; j++) {
        lin = fma(q.theta[(size_t)j * q.ldc + c], q.colsum[j], lin);
        ss += b[(size_t)j * bs] * b[(size_t)j * bs];
    }
    const double *lpp = q.par + 1 + (size_t)q.k * q.p;
    const double ly = (0.5 * lin - part) + q.pad_term;
    return (ly - ss * q.prior_c) + lpp[q.k] + lpp[comp - 1];
}

// the cached log densities of the initial state (temper.c:98-119): parallel, chain k e + i holds (i + 1, x_i)
__global__ __launch_bounds__(256) void tm_init_stage_kernel(TmParams q)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= q.nens) return;
    for (int i = 0; i < q.nstate; i++) q.S[(size_t)i * q.lde + e] = q.initial[(size_t)e * q.nstate + i];
    for (int i = 0; i < q.k; i++) {
        q.axn[(size_t)i * q.lde + e] = q.axd[(size_t)i * q.lde + e] = 0u;
        q.ibuf[(size_t)i * q.lde + e] = 0.0;
        q.L[(size_t)i * q.lde + e] = NAN;
        for (int s = 0; s < q.maxnn; s++) q.ain[((size_t)i * q.maxnn + s) * q.lde + e] = q.aid[((size_t)i * q.maxnn + s) * q.lde + e] = 0u;
    }
    for (int i = 0; i < q.nout; i++) q.buf[(size_t)i * q.lde + e] = 0.0;
    if (q.parallel) {
        for (int i = 0; i < q.k; i++) tm_stage(q, (long)q.k * e + i, i + 1, q.S + (size_t)i * q.lde + e, (long)q.k * q.lde);
    } else {
        tm_stage(q, e, (int)q.S[e], q.S + q.lde + e, q.lde);
    }
}

__global__ __launch_bounds__(256) void tm_init_finish_kernel(TmParams q)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= q.nens) return;
    const int ncomp = q.parallel ? q.k : 1;
    for (int i = 0; i < ncomp; i++) {
        const int comp = q.parallel ? i + 1 : (int)q.S[e];
        const double v = q.parallel ? tm_value(q, (long)q.k * e + i, comp, q.S + (size_t)i * q.lde + e, (long)q.k * q.lde)
                                    : tm_value(q, e, comp, q.S + q.lde + e, q.lde);
        q.L[(size_t)(comp - 1) * q.lde + e] = v;
        if (v == -INFINITY) atomicOr(q.flag, FLAG_NEGINF_INIT);
        else if (v == INFINITY) atomicOr(q.flag, FLAG_POSINF);
        else if (isnan(v)) atomicOr(q.flag, FLAG_NAN);
    }
}

// iteration `it`: finish it (it >= 0: the density launch behind its proposal has run), then draw and stage it + 1
__global__ __launch_bounds__(256) void tm_step_kernel(TmParams q, long it)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= q.nens) return;
    const int k = q.k, p = q.p, par = q.parallel, maxnn = q.maxnn;
    const long lde = q.lde, niter = (long)q.nbatch * q.blen * q.nspac;
    const unsigned long long gens = q.ens_offset + (unsigned long long)e;
    const bool inj = q.rng_mode == MCMCHIP_RNG_INJECTED;
    const long c0 = 2 * e, c1 = 2 * e + 1;
    auto Sx = [&](int i) -> double & { return q.S[(size_t)i * lde + e]; };
    auto Lx = [&](int i) -> double & { return q.L[(size_t)i * lde + e]; };
    if (it >= 0) {
        const int within = q.mv[e], ci = q.mv[lde + e], cj = q.mv[2 * lde + e], slot_ = q.mv[3 * lde + e];
        double lh, ly = 0.0, sco = 0.0, spr = 0.0;
        if (within) {
            ly = tm_value(q, c0, ci, q.Y + e, lde);
            lh = ly - Lx(ci - 1);
            if (ly == INFINITY || isnan(ly)) atomicOr(q.flag, isnan(ly) ? FLAG_NAN : FLAG_POSINF);
        } else if (par) {
            sco = tm_value(q, c0, cj, q.S + (size_t)(ci - 1) * lde + e, (long)k * lde);   // (j, x_i) first (temper.c:738-739)
            spr = tm_value(q, c1, ci, q.S + (size_t)(cj - 1) * lde + e, (long)k * lde);   // (i, x_j)
            lh = spr + sco - Lx(cj - 1) - Lx(ci - 1);                                       // temper.c:742-744
            if (isnan(sco) || isnan(spr) || sco == INFINITY || spr == INFINITY) atomicOr(q.flag, FLAG_NAN);
        } else {
            ly = tm_value(q, c0, cj, q.S + lde + e, lde);
            lh = ly - Lx(ci - 1) + log((double)q.nn[ci - 1]) - log((double)q.nn[cj - 1]);  // temper.c:810-812
            if (ly == INFINITY || isnan(ly)) atomicOr(q.flag, isnan(ly) ? FLAG_NAN : FLAG_POSINF);
        }
        if (isnan(lh) || (isinf(lh) && lh > 0)) { atomicOr(q.flag, FLAG_HASTINGS); lh = -INFINITY; }
        bool accept = true;
        if (lh < 0.0) {                                   // temper.c:582-587
            double u2, u3;
            if (inj) u3 = q.unif[((size_t)e * niter + it) * 4 + 3];
            else philox_u2(q.seed, PHILOX_TAG_TEMPER, gens, q.iter_offset + (unsigned long long)it, 1u, u2, u3);
            accept = u3 < exp(lh);
        }
        if (within) {
            if (accept) {
                for (int j = 0; j < p; j++) {
                    if (par) Sx((ci - 1) + k * j) = q.Y[(size_t)j * lde + e];
                    else Sx(1 + j) = q.Y[(size_t)j * lde + e];
                }
                Lx(ci - 1) = ly;
                q.axn[(size_t)(ci - 1) * lde + e]++;
            }
            q.axd[(size_t)(ci - 1) * lde + e]++;
        } else if (par) {
            if (accept) {
                for (int j = 0; j < p; j++) {
                    const double a = Sx((ci - 1) + k * j), b = Sx((cj - 1) + k * j);
                    Sx((ci - 1) + k * j) = b;
                    Sx((cj - 1) + k * j) = a;
                }
                Lx(ci - 1) = spr;
                Lx(cj - 1) = sco;
                q.ain[((size_t)(ci - 1) * maxnn + slot_) * lde + e]++;
            }
            q.aid[((size_t)(ci - 1) * maxnn + slot_) * lde + e]++;
        } else {
            if (accept) {
                Sx(0) = cj;
                Lx(cj - 1) = ly;
                q.ain[((size_t)(ci - 1) * maxnn + slot_) * lde + e]++;
            }
            q.aid[((size_t)(ci - 1) * maxnn + slot_) * lde + e]++;
        }
        // batch means (temper.c:851-886)
        const int is = (int)(it % q.nspac);
        const long lb = it / q.nspac;
        const int jb = (int)(lb % q.blen);
        const long kb = lb / q.blen;
        if (is == q.nspac - 1) {
            const bool mapped = q.out_kind == MCMCHIP_TOUT_INDEX || q.out_kind == MCMCHIP_TOUT_MOMENTS2;
            for (int i = 0; i < q.nout; i++) {
                double v;
                if (mapped) {
                    v = Sx(q.out_map[i] + (par ? 0 : 1));
                    if (i >= q.sq_from) v = mul_nofma(v, v);
                    q.buf[(size_t)i * lde + e] = add_nofma(q.buf[(size_t)i * lde + e], v);
                } else {
                    q.buf[(size_t)i * lde + e] += par ? Sx(i) : Sx(i + 1);
                }
            }
            if (!par) q.ibuf[(size_t)((int)Sx(0) - 1) * lde + e] += 1.0;
            if (jb == q.blen - 1) {
                for (int i = 0; i < q.nout; i++) {
                    q.batch[(size_t)e * q.nbatch * q.nout + kb + (size_t)q.nbatch * i] = q.buf[(size_t)i * lde + e] / q.blen;
                    q.buf[(size_t)i * lde + e] = 0.0;
                }
                if (!par)
                    for (int i = 0; i < k; i++) {
                        q.ibatch[(size_t)e * q.nbatch * k + kb + (size_t)q.nbatch * i] = q.ibuf[(size_t)i * lde + e] / q.blen;
                        q.ibuf[(size_t)i * lde + e] = 0.0;
                    }
            }
        }
    }
    const long nx = it + 1;
    if (nx < niter) {
        // ---- draw and stage iteration nx: u_which, u_choose [, u_choose2 | normals]
        const unsigned long long git = q.iter_offset + (unsigned long long)nx;
        double u0, u1, u2 = 0.0, u3;
        if (inj) {
            const double *up = q.unif + ((size_t)e * niter + nx) * 4;
            u0 = up[0]; u1 = up[1]; u2 = up[2];
        } else {
            philox_u2(q.seed, PHILOX_TAG_TEMPER, gens, git, 0u, u0, u1);
        }
        const int within = u0 < 0.5;                   // temper.c:537-538
        int ci, cj = 0, slot_ = 0;
        if (within) {
            ci = par ? tm_pick1(u1, k) : (int)Sx(0);
            const int sk = q.scale_kind & 7;
            const int slen1 = sk == MCMCHIP_SCALE_CONSTANT ? 1 : (sk == MCMCHIP_SCALE_DIAGONAL ? p : p * p);
            const double *sc = q.scale + ((q.scale_kind & MCMCHIP_SCALE_LIST) ? (size_t)(ci - 1) * slen1 : 0);
            double z[16];
            if (inj) {
                for (int j = 0; j < p; j++) z[j] = q.norm[((size_t)e * niter + nx) * p + j];
            } else {
                for (int j2 = 0; 2 * j2 < p; j2++) {
                    double a, b;
                    philox_n2(q.seed, PHILOX_TAG_TEMPER, gens, git, (uint32_t)(2 + j2), a, b);
                    z[2 * j2] = a;
                    if (2 * j2 + 1 < p) z[2 * j2 + 1] = b;
                }
            }
            for (int j = 0; j < p; j++) {                // temper.c:1003-1024
                const double xj = par ? Sx((ci - 1) + k * j) : Sx(1 + j);
                double yj;
                if (sk == MCMCHIP_SCALE_CONSTANT) yj = add_nofma(xj, mul_nofma(sc[0], z[j]));
                else if (sk == MCMCHIP_SCALE_DIAGONAL) yj = add_nofma(xj, mul_nofma(sc[j], z[j]));
                else {
                    yj = xj;
                    for (int i = 0; i < p; i++) yj = add_nofma(yj, mul_nofma(sc[j + p * i], z[i]));
                }
                q.Y[(size_t)j * lde + e] = yj;
            }
            tm_stage(q, c0, ci, q.Y + e, lde);
            tm_stage_zero(q, c1);
        } else if (par) {
            if (!inj) philox_u2(q.seed, PHILOX_TAG_TEMPER, gens, git, 1u, u2, u3);
            ci = tm_pick1(u1, k);
            slot_ = tm_pick1(u2, q.nn[ci - 1]) - 1;
            cj = q.nb[(ci - 1) * maxnn + slot_] + 1;
            tm_stage(q, c0, cj, q.S + (size_t)(ci - 1) * lde + e, (long)k * lde);
            tm_stage(q, c1, ci, q.S + (size_t)(cj - 1) * lde + e, (long)k * lde);
        } else {
            ci = (int)Sx(0);
            slot_ = tm_pick1(u1, q.nn[ci - 1]) - 1;
            cj = q.nb[(ci - 1) * maxnn + slot_] + 1;
            tm_stage(q, c0, cj, q.S + lde + e, lde);
            tm_stage_zero(q, c1);
        }
        q.mv[e] = within; q.mv[lde + e] = ci; q.mv[2 * lde + e] = cj; q.mv[3 * lde + e] = slot_;
    } else {
        // ---- the run is over: acceptance tables (temper.c:889-897), final state
        for (int i = 0; i < k; i++)
            q.acceptx[(size_t)e * k + i] = (double)q.axn[(size_t)i * lde + e] / (double)q.axd[(size_t)i * lde + e];   // 0/0 = NaN
        for (int j = 0; j < k; j++)
            for (int i = 0; i < k; i++) {
                double v = NAN;                                      // NA for non-neighbours
                if (q.neighbors[i + k * j]) {
                    int slot = 0;
                    while (q.nb[i * maxnn + slot] != j) slot++;
                    v = (double)q.ain[((size_t)i * maxnn + slot) * lde + e] / (double)q.aid[((size_t)i * maxnn + slot) * lde + e];
                }
                q.accepti[(size_t)e * k * k + i + k * j] = v;
            }
        for (int i = 0; i < q.nstate; i++) q.final_[(size_t)e * q.nstate + i] = Sx(i);
    }
}

}  // namespace

bool temper_mfma_applies(const mcmchip_target *t, const TemperDev &q)
{
    return t->kind == MCMCHIP_T_LOGISTIC_MODELS && t->xt && !q.debug && q.p <= 16 &&
           (q.out_kind == MCMCHIP_TOUT_IDENTITY || q.out_kind == MCMCHIP_TOUT_INDEX || q.out_kind == MCMCHIP_TOUT_MOMENTS2);
}

int run_temper_mfma(const mcmchip_target *t, const TemperDev &d, DeviceCtx *dc)
{
    if (!temper_mfma_applies(t, d))
        return set_error(MCMCHIP_EINVAL, "the MFMA tempering path takes the logistic-models descriptor (d <= 16), identity or index output, debug = FALSE");
    hipStream_t s = dc->stream;
    const int k = d.k, p = d.p, maxnn = d.maxnn;
    const long nens = d.nens, lde = ((nens + 255) / 256) * 256;
    const bool mapped = d.out_kind == MCMCHIP_TOUT_INDEX || d.out_kind == MCMCHIP_TOUT_MOMENTS2;
    const int nstate = d.parallel ? k * p : p + 1, nout = mapped ? d.nout_map : (d.parallel ? k * p : p);
    // neighbour lists on the host (the caller validated the matrix)
    std::vector<int> hneigh((size_t)k * k), hn((size_t)k + (size_t)k * maxnn, 0);
    HIP_TRY(hipMemcpyAsync(hneigh.data(), d.neighbors, hneigh.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < k; i++) {
        int c = 0;
        for (int j = 0; j < k; j++)
            if (hneigh[(size_t)i + (size_t)k * j]) hn[(size_t)k + (size_t)i * maxnn + c++] = j;    // ascending (temper.c:504-510)
        hn[(size_t)i] = c;
    }
    // the two shapes of density launch: the iteration's 2 nens chains, and the initial k nens (parallel) / nens (serial)
    const long nch_it = 2 * nens, nch_in = d.parallel ? (long)k * nens : nens;
    long ldc_it = 0, ldc_in = 0;
    int nunits = 0;
    if (int rc = mfma_density_plan(t, nch_it, dc, &ldc_it, &nunits)) return rc;
    if (int rc = mfma_density_plan(t, nch_in, dc, &ldc_in, &nunits)) return rc;
    const long ldc_max = ldc_it > ldc_in ? ldc_it : ldc_in;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t b_theta = up((size_t)t->dpad * ldc_max * 8), b_part = up((size_t)nunits * ldc_max * 8);
    const size_t nd = (size_t)nstate + k + p + nout + k;                          // doubles per ensemble: S, L, Y, buf, ibuf
    const size_t nu = 2 * (size_t)k + 2 * (size_t)k * maxnn;                      // unsigned: axn, axd, ain, aid
    const size_t b_d = up(nd * lde * 8), b_u = up(nu * lde * 4), b_mv = up((size_t)4 * lde * 4), b_n = up(hn.size() * 4);
    DevBuf all;
    HIP_TRY(all.alloc(b_theta + b_part + b_d + b_u + b_mv + b_n));
    char *cur = all.as<char>();
    auto carve = [&](size_t b) { char *r = cur; cur += b; return r; };
    double *theta = (double *)carve(b_theta), *partial = (double *)carve(b_part), *dd = (double *)carve(b_d);
    unsigned *uu = (unsigned *)carve(b_u);
    int *mv = (int *)carve(b_mv), *dn = (int *)carve(b_n);
    HIP_TRY(hipMemcpyAsync(dn, hn.data(), hn.size() * 4, hipMemcpyHostToDevice, s));

    TmParams q;
    memset(&q, 0, sizeof q);
    q.nens = nens; q.lde = lde; q.k = k; q.p = p; q.parallel = d.parallel; q.maxnn = maxnn; q.nstate = nstate; q.nout = nout;
    q.nbatch = d.nbatch; q.blen = d.blen; q.nspac = d.nspac;
    q.scale_kind = d.scale_kind; q.scale = d.scale;
    q.rng_mode = d.rng_mode; q.unif = d.unif; q.norm = d.norm;
    q.seed = d.seed; q.ens_offset = d.ens_offset; q.iter_offset = d.iter_offset;
    q.nn = dn; q.nb = dn + k; q.neighbors = d.neighbors;
    q.par = t->par; q.prior_c = t->prior_c; q.pad_term = (double)t->npad_rows * 0.6931471805599453;
    q.nunits = nunits; q.theta = theta; q.partial = partial; q.colsum = t->colsum;
    q.S = dd; q.L = q.S + (size_t)nstate * lde; q.Y = q.L + (size_t)k * lde; q.buf = q.Y + (size_t)p * lde;
    q.ibuf = q.buf + (size_t)nout * lde;
    q.axn = uu; q.axd = q.axn + (size_t)k * lde; q.ain = q.axd + (size_t)k * lde; q.aid = q.ain + (size_t)k * maxnn * lde;
    q.mv = mv;
    q.initial = d.initial;
    q.batch = d.batch; q.ibatch = d.ibatch; q.acceptx = d.acceptx; q.accepti = d.accepti; q.final_ = d.final_;
    q.out_kind = d.out_kind; q.nout_map = d.nout_map; q.sq_from = d.sq_from; q.out_map = d.out_map;
    q.flag = d.flag;

    const dim3 grid((unsigned)((nens + 255) / 256)), block(256);
    // ---- cached log densities of the initial state
    q.ldc = ldc_in;
    HIP_TRY(hipMemsetAsync(theta, 0, b_theta, s));             // padded rows and chains: zeros
    hipLaunchKernelGGL(tm_init_stage_kernel, grid, block, 0, s, q);
    HIP_TRY(hipGetLastError());
    timing_reset("logdens_mfma_kernel");
    timing_begin(s);
    int rc = mfma_density_launch(t, nch_in, theta, partial, dc);
    timing_end(s);
    if (rc) return rc;
    hipLaunchKernelGGL(tm_init_finish_kernel, grid, block, 0, s, q);
    HIP_TRY(hipGetLastError());
    // ---- the iterations
    q.ldc = ldc_it;
    HIP_TRY(hipMemsetAsync(theta, 0, b_theta, s));
    const long niter = (long)d.nbatch * d.blen * d.nspac;
    const bool polled = lib_ctx().poll != nullptr;
    long group = 1, in_group = 0;
    hipLaunchKernelGGL(tm_step_kernel, grid, block, 0, s, q, -1L);      // stages iteration 0
    HIP_TRY(hipGetLastError());
    for (long it = 0; it < niter; it++) {
        timing_begin(s);
        rc = mfma_density_launch(t, nch_it, theta, partial, dc);
        timing_end(s);
        if (rc) return rc;
        hipLaunchKernelGGL(tm_step_kernel, grid, block, 0, s, q, it);
        HIP_TRY(hipGetLastError());
        if (polled && ++in_group >= group && it + 1 < niter) {          // temper.c:848
            const double ms = timing_last_ms();
            HIP_TRY(hipStreamSynchronize(s));
            if (int irc = poll_interrupt()) return irc;
            group = ms > 0.0 ? (long)(lib_ctx().launch_budget_s * 1e3 / ms) : 1;
            if (group < 1) group = 1;
            in_group = 0;
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    return MCMCHIP_OK;
}
