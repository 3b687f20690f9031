// metrop_stream.hip -- batched random-walk Metropolis, "stream" path.
//
// What it replaces: the triple loop of the reference's metrop() (metrop.c:199-285) together with
// propose() (367-404), the accept rule (225-235), outfun()/batch means (273-283) and the debug
// trace (237-257) -- for `nchain` independent chains at once.
//
// Mapping onto gfx950: ONE persistent launch runs the whole nbatch*blen*nspac loop.  A chain is
// owned by G lanes of one wavefront (G = 1/4/16/64 chosen on the host from the work per
// density evaluation); the lanes split the observations (or matrix rows / mixture components)
// of the log density and combine with an xor butterfly, so every lane of the group holds the
// same log density and takes the same accept decision.  The chain's state, proposal, normals
// and batch accumulator live in LDS; for small data sets (the 100 x 5 `logit` data: 4.8 KB) X
// and y are staged into LDS once per workgroup, so an iteration touches HBM only for the
// injected random numbers and the optional trace.  Bound: fp64 transcendental throughput
// (2n exp/log1p per chain-iteration), not HBM.
//
// Arithmetic fidelity: the proposal is x + s*z with separately rounded multiply and add
// (mul_nofma/add_nofma: no FMA contraction, as in the reference's x86-64 build); full-matrix
// scale accumulates column by column in i order (metrop.c:391-399).
#include "metrop_stream_impl.h"
#include <stdlib.h>
#include <chrono>

static int pick_group(long work, long nchain)
{
    long want = 65536 / (nchain > 0 ? nchain : 1);   // aim at >= 64k lanes in flight
    if (want > work) want = work;
    if (want >= 64) return 64;
    if (want >= 16) return 16;
    if (want >= 4) return 4;
    return 1;
}

int launch_metrop_stream(const MetropDev &p, hipStream_t s, void *batch_home, hipStream_t side, bool *went_home)
{
    const int d = p.d;
    long work = d;
    if (p.kind == MCMCHIP_T_LOGISTIC || p.kind == MCMCHIP_T_LOGISTIC_NAIVE || p.kind == MCMCHIP_T_BINOMIAL_LOGIT) work = p.n;
    else if (p.kind == MCMCHIP_T_NORMMIX) work = 8;
    else if (p.morph_flags && d <= 16) work = 1;   // small morphed targets: one lane per chain keeps the norm's summation order
    int G = pick_group(work, p.nchain);
    if (const char *e = getenv("MCMCHIP_STREAM_G")) {       // experiments: force the lanes per chain (1, 4, 16, 64)
        const int f = atoi(e);
        if (f == 1 || f == 4 || f == 16 || f == 64) G = f;
    }
    StreamLayout lay;
    const int blocks_it = (d + 1) / 2 + 1;                    // Philox blocks per iteration: normals + the uniform
    auto ahead = [&](int GG) { return p.rng_mode == MCMCHIP_RNG_INJECTED ? 0 : (GG / blocks_it > 1 ? (GG / blocks_it > 8 ? 8 : GG / blocks_it) : 1); };
    auto per_chain_of = [&](int GG) { return 3 * d + p.nout + (p.morph_flags ? d : 0) + ahead(GG) * (2 * ((d + 1) / 2) + 1); };
    lay.per_chain = per_chain_of(G);
    // The logistic posterior spends its time in log1p(exp(-|eta|)), ~130 fp64 instructions from the library; with the
    // 130 KB softplus table of the MFMA path staged in LDS it is ~15 (C2: 72 -> ms).  The table
    // takes most of the LDS (one workgroup per CU), which the 5-9x shorter density more than pays for.
    const size_t tab_bytes = (size_t)SP_TAB_DOUBLES * 8;
    const bool want_tab = p.kind == MCMCHIP_T_LOGISTIC && p.sp_tables != nullptr && !p.no_table;
    const size_t lds_max = 160 * 1024;
    size_t budget = 64 * 1024;
    size_t xbytes = (p.x ? (size_t)(p.n * d + p.n) * 8 : 0);
    lay.tab_in_lds = 0;
    for (;;) {
        lay.per_chain = per_chain_of(G);
        lay.rng_ahead = ahead(G);
        size_t chains = (size_t)(256 / G) * lay.per_chain * 8;
        if (want_tab && tab_bytes + chains <= lds_max) {
            lay.tab_in_lds = 1;
            lay.x_in_lds = (xbytes > 0 && tab_bytes + xbytes + chains <= lds_max);
            lay.lds_bytes = tab_bytes + chains + (lay.x_in_lds ? xbytes : 0);
            budget = lds_max;
            break;
        }
        if (chains <= budget / 2 || G == 64) {
            lay.x_in_lds = (xbytes > 0 && xbytes + chains <= budget);
            lay.lds_bytes = chains + (lay.x_in_lds ? xbytes : 0);
            break;
        }
        G = (G == 1) ? 4 : (G == 4 ? 16 : 64);
    }
    if (lay.lds_bytes > budget)
        return set_error(MCMCHIP_EINVAL, "state dimension %d too large for the stream path", d);
    const int cpb = 256 / G;
    dim3 grid((unsigned)((p.nchain + cpb - 1) / cpb)), block(256);
    // The run as a sequence of launches (see the kernel): a short calibration launch, then launches of about a quarter
    // of a second, with a host poll point between them.  A run that fits the first launch needs no carry buffer.
    const long niter = (long)p.nbatch * p.blen * p.nspac;
    MetropDev q = p;
    DevBuf carry;
    long done = 0, last_iters = 0;
    double last_ms = 0.0;
    // Batch means on their way to pageable host memory, 64 MB and more of them (C2: 1.6 GB behind a 30 ms kernel, 45 ms on
    // the PCIe link): the run is cut into 4 launches after the calibration launch (more for a run longer than the launch budget) and the batches a launch completed go
    // home (copy_rows_to_pageable_host, on `side`) while the next launch runs.  In the result's layout
    // [chain][batch][nout] they are one run of (b1 - b0) * nout doubles per chain.
    const long per_batch = (long)p.blen * p.nspac;
    const size_t pitch = (size_t)p.nbatch * p.nout * 8;
    bool piecewise = batch_home && side && went_home && p.batch && p.nout > 0 && (size_t)p.nchain * pitch >= ((size_t)64 << 20) &&
                     !getenv("MCMCHIP_NO_PIECEWISE_COPY");
    long seg_iters = 0;                       // launch length while copying piecewise
    long home_b = 0;                          // batches [0, home_b) are in the caller's array or on their way
    // pieces whose launches are queued but whose copies are not: up to `queue_ahead` <= AHEAD launches (one launch budget in all, see
    // below) are queued behind the piece being copied, so the device does not wait for the host between launches
    constexpr int AHEAD = 8;
    int queue_ahead = 1;
    struct Piece { long b0, b1; hipEvent_t ev; };
    Piece pend[AHEAD + 1];
    int npend = 0;
    hipEvent_t ev[AHEAD + 1] = {};
    int cur = 0;
    const bool trace = getenv("MCMCHIP_TRACE") != nullptr;
    auto send_oldest = [&]() -> int {
        const Piece pc = pend[0];
        for (int i = 1; i < npend; i++) pend[i - 1] = pend[i];
        npend--;
        if (pc.b1 <= pc.b0) return MCMCHIP_OK;
        HIP_TRY(hipStreamWaitEvent(side, pc.ev, 0));
        const auto t0 = std::chrono::steady_clock::now();
        if (trace) (void)hipEventSynchronize(pc.ev);
        const auto t1 = std::chrono::steady_clock::now();
        HIP_TRY(copy_rows_to_pageable_host((char *)batch_home + (size_t)pc.b0 * p.nout * 8, (const char *)p.batch + (size_t)pc.b0 * p.nout * 8,
                                           (size_t)p.nchain, (size_t)(pc.b1 - pc.b0) * p.nout * 8, pitch, side));
        if (trace)
            fprintf(stderr, "[metrop_stream] piece of %.1f MB: waited %.2f ms for its launch, copy %.2f ms\n",
                    (double)p.nchain * (double)(pc.b1 - pc.b0) * p.nout * 8 / 1e6, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        return MCMCHIP_OK;
    };
    struct EvGuard {
        hipEvent_t *e;
        ~EvGuard() { for (int i = 0; i <= AHEAD; i++) if (e[i]) (void)hipEventDestroy(e[i]); }
    } guard{ev};
    while (done < niter) {
        long n;
        if (piecewise && seg_iters > 0) n = seg_iters < niter - done ? seg_iters : niter - done;
        else n = next_launch_iters(niter - done, last_iters, last_ms, 1);
        q.l_begin = done;
        q.l_end = done + n;
        q.resume = done > 0;
        if (q.l_end < niter && !carry.p) {
            HIP_TRY(carry.alloc((size_t)p.nchain * (3 + p.nout) * 8));
            q.carry = carry.as<double>();
        }
        int rc;
        timing_begin(s);
        switch (G) {
        case 1: rc = launch_stream_g1(q, lay, grid, s); break;
        case 4: rc = launch_stream_g4(q, lay, grid, s); break;
        case 16: rc = launch_stream_g16(q, lay, grid, s); break;
        default: rc = launch_stream_g64(q, lay, grid, s); break;
        }
        timing_end(s);
        if (rc) return rc;
        done += n;
        if (piecewise && seg_iters > 0) {
            // this launch is queued: its batches join the pieces waiting, the oldest goes home once AHEAD launches are behind it
            if (!ev[cur]) HIP_TRY(hipEventCreateWithFlags(&ev[cur], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ev[cur], s));
            pend[npend++] = Piece{home_b, done / per_batch, ev[cur]};
            home_b = done / per_batch;
            cur = (cur + 1) % (AHEAD + 1);
            if (npend > queue_ahead)
                if (int crc = send_oldest()) return crc;
            if (done < niter)
                if (int irc = poll_interrupt()) return irc;
            continue;
        }
        if (done < niter) {
            last_ms = timing_last_ms();          // waits for the launch: the host poll point
            last_iters = n;
            if (int irc = poll_interrupt()) return irc;
            if (piecewise) {
                // after the calibration launch: a quarter of the rest per launch, whole batches, at most one launch budget (a
                // long run keeps its launches as long as they were: each costs 0.8 ms at C2), as many launches queued ahead
                // of the copy as fit one budget; rows of at least 1 KB per chain (else the copy stays behind the run)
                const long budget = next_launch_iters(1L << 60, last_iters, last_ms, 1);        // iterations of one launch budget
                // (4 pieces: a launch more costs 0.8 ms at C2 -- table and data back into LDS -- and what there is to hide, the
                // copy less the page faults of a fresh destination, which all fall into the first piece, is 4 ms per 200 MB)
                long pieces = 4;
                if (const char *e = getenv("MCMCHIP_PIECES")) pieces = atol(e) > 0 ? atol(e) : 4;      // experiments
                long want = ((niter - done + pieces - 1) / pieces + per_batch - 1) / per_batch * per_batch;
                if (want > budget) want = budget / per_batch * per_batch;
                if (want >= per_batch && (size_t)(want / per_batch) * p.nout * 8 >= 1024) {
                    seg_iters = want;
                    queue_ahead = (int)(budget / want < 1 ? 1 : (budget / want > AHEAD ? AHEAD : budget / want));
                } else piecewise = false;
            }
        }
    }
    if (piecewise && seg_iters > 0) {
        while (npend > 0)
            if (int crc = send_oldest()) return crc;
        *went_home = true;
    }
    if (carry.p) HIP_TRY(hipStreamSynchronize(s));      // the carry buffer is released on return
    return MCMCHIP_OK;
}
// ---------------------------------------------------------------------------------------------
// log density of a batch of states (replaces logh(), metrop.c:303-323): 16 lanes per state
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logdens_stream_kernel(TargetView tv, int temper_kind, long ncomp_l,
                                                             const double *states, long nchain, double *out)
{
    extern __shared__ double ld_w[];      // morph only: 16 x d doubles
    constexpr int G = 16;
    const int g = threadIdx.x % G;
    const int ncomp = (int)ncomp_l;
    long c = (long)blockIdx.x * (256 / G) + threadIdx.x / G;
    const bool live = c < nchain;
    if (!live) c = nchain - 1;
    double v;
    if (temper_kind) {
        const double *s = states + (size_t)c * (tv.d + 1);
        if (tv.kind == MCMCHIP_T_LOGISTIC_MODELS)
            v = lud_models_group<G>(tv.par, tv.d, tv.x, tv.y, tv.n, tv.prior_c, (int)s[0], s + 1, g);
        else
            v = lud_temper(tv.kind, tv.d, tv.par, ncomp, (int)s[0], s + 1);
    } else {
        v = lud_morphed_group<G>(tv, states + (size_t)c * tv.d, ld_w + (size_t)(threadIdx.x / G) * tv.d, g);
    }
    if (live && g == 0) out[c] = v;
}

int launch_logdens_stream(const mcmchip_target *t, const double *states, long nchain, double *out,
                          hipStream_t s)
{
    TargetView tv;
    tv.kind = t->kind; tv.d = t->d; tv.n = t->n; tv.x = t->x; tv.y = t->y; tv.par = t->par;
    tv.prior_c = t->prior_c;
    tv.morph_flags = t->morph_flags; tv.morph_b = t->morph_b; tv.morph_r = t->morph_r; tv.morph_p = t->morph_p;
    tv.morph_center = t->morph_center;
    const size_t wbytes = t->morph_flags ? (size_t)16 * t->d * 8 : 0;
    if (wbytes > 64 * 1024) return set_error(MCMCHIP_EINVAL, "state dimension %d too large for a morphed log density", t->d);
    int temper_kind = (t->kind == MCMCHIP_T_WITCH || t->kind == MCMCHIP_T_TEMPERED_NORMMIX ||
                       t->kind == MCMCHIP_T_LOGISTIC_MODELS);
    dim3 grid((unsigned)((nchain + 15) / 16)), block(256);
    hipLaunchKernelGGL(logdens_stream_kernel, grid, block, wbytes, s, tv, temper_kind,
                       (long)(t->kind == MCMCHIP_T_WITCH ? t->npar : 0), states, nchain, out);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}
