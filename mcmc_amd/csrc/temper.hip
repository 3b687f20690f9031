// temper.hip -- serial and parallel tempering for `nens` independent ensembles.
//
// What it replaces: the loop nest of the reference's temper() (temper.c:517-887): per
// iteration ONE elementary update -- with probability 1/2 a within-component random-walk
// Metropolis step on one component (temper.c:547-675), else a swap of two neighbouring
// components (parallel, 679-781) or a jump to a neighbour (serial, 783-845) -- plus batch
// means (851-886) and the acceptance-rate tables (889-897).
//
// Mapping onto gfx950: one wavefront (a 64-thread workgroup) per ensemble.  The k x p state
// matrix, the cached log densities (temper.c:123-124), the batch accumulator and the per-pair
// acceptance counters all live in LDS, so a swap is two LDS row exchanges and nothing touches
// HBM inside the iteration loop except injected random numbers in parity mode.  Control flow
// is wave-uniform (every lane of the wave follows the same within/swap branch: no divergence);
// the vector parts (normals, proposal, row swap, batch accumulation over k*p elements) are
// spread over the lanes.  The path is latency-bound (a dependent chain Philox -> index ->
// density -> exp -> compare per iteration), so throughput comes from running all ensembles
// concurrently (1024 ensembles = 4 waves per CU).
//
// RNG draw order per iteration is the reference's: u_which, then u_choose
// [, u_choose2 | normals], then u_hastings iff log Hastings < 0 (temper.c:584-587): a -Inf
// proposal DOES draw and reject; equality accepts without a draw -- the opposite of metrop.
#include "kernels.h"
#include "philox.h"
#include "targets.h"

#define TEMPER_MAXP 16
#define TEMPER_MAXK 64

struct TemperLds {
    int state, L, Y, Z, buf, ibuf, axn, axd, ain, aid, nn, nb, ubuf, zbuf, par, scale, end;  // offsets in 8-byte words
    int npar_lds, nscale_lds;     // > 0: the descriptor parameters / the scale are staged in LDS (the serial chain of
                                  // an iteration then never waits on global memory)
};

__device__ __forceinline__ int pick1(double u, int m)
{
    int i = (int)trunc(m * u) + 1;   // temper.c:555-556
    if (i > m) i--;
    return i;
}

__global__ __launch_bounds__(64) void temper_kernel(TemperDev q, TemperLds lay, int maxnn)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long e = blockIdx.x;
    const int k = q.k, p = q.p, par = q.parallel;
    const int nstate = par ? k * p : p + 1;
    const int corner = q.out_kind == MCMCHIP_TOUT_WITCH_CORNER;
    const bool mapped = q.out_kind == MCMCHIP_TOUT_INDEX || q.out_kind == MCMCHIP_TOUT_MOMENTS2;
    const int nout = mapped ? q.nout_map : (corner ? (par ? k : 2 * k) : (par ? k * p : p));
    double *S = lds + lay.state, *L = lds + lay.L, *Y = lds + lay.Y, *Z = lds + lay.Z;
    double *buf = lds + lay.buf, *ibuf = lds + lay.ibuf;
    unsigned *axn = (unsigned *)(lds + lay.axn), *axd = (unsigned *)(lds + lay.axd);
    unsigned *ain = (unsigned *)(lds + lay.ain), *aid = (unsigned *)(lds + lay.aid);
    int *nn = (int *)(lds + lay.nn), *nb = (int *)(lds + lay.nb);
    double *Ubuf = lds + lay.ubuf, *Zbuf = lds + lay.zbuf;   // Philox draws of the next 64 iterations
    const int pz = (p + 1) & ~1;

    const double *gpar = q.par, *gscale = q.scale;
    if (lay.npar_lds > 0) {
        double *lp = lds + lay.par;
        for (int i = lane; i < lay.npar_lds; i += 64) lp[i] = q.par[i];
        gpar = lp;
    }
    if (lay.nscale_lds > 0) {
        double *ls = lds + lay.scale;
        for (int i = lane; i < lay.nscale_lds; i += 64) ls[i] = q.scale[i];
        gscale = ls;
    }
    // log density of (component comp, theta): called with wave-uniform arguments only.  The data-carrying
    // kind spreads its observations over the 64 lanes; the small closed-form kinds are scalar per lane.
    const bool coop = q.kind == MCMCHIP_T_LOGISTIC_MODELS;
    typedef __attribute__((address_space(3))) const double lds_cd;
    const bool par_lds = lay.npar_lds > 0;
    auto LUD = [&](int comp, const double *th) -> double {
        if (coop) return lud_models_group<64>(gpar, p, q.x, q.y, q.n, q.prior_c, comp, th, lane);
        // parameters staged in LDS are read with ds_read (through a generic pointer they would be FLAT loads)
        if (par_lds) return lud_temper<lds_cd *>(q.kind, p, (lds_cd *)gpar, k, comp, th);
        return lud_temper(q.kind, p, gpar, k, comp, th);
    };

    // A launch runs iterations [it_begin, it_end) (launch_temper cuts a run into launches with a host poll point between
    // them, temper.c:848; boundaries are multiples of 64 iterations, the Philox window).  A resumed launch restores
    // the ensemble's whole LDS record -- state, cached log densities, batch sums, acceptance counters -- from `carry`.
    double *carry = q.carry ? q.carry + (size_t)e * lay.ubuf : nullptr;
    if (q.resume) {
        for (int i = lane; i < lay.ubuf; i += 64) lds[i] = carry[i];
    } else {
        for (int i = lane; i < nstate; i += 64) S[i] = q.initial[(size_t)e * nstate + i];
        for (int i = lane; i < k; i += 64) {
            axn[i] = axd[i] = 0;
            ibuf[i] = 0.0;
            for (int c2 = 0; c2 < maxnn; c2++) ain[i * maxnn + c2] = aid[i * maxnn + c2] = 0;
        }
    }
    __syncthreads();
    for (int i = lane; i < k; i += 64) {
        int c = 0;
        for (int j = 0; j < k; j++)
            if (q.neighbors[i + k * j]) nb[i * maxnn + c++] = j;   // ascending order (temper.c:504-510)
        nn[i] = c;
    }
    __syncthreads();
    // cached log densities (temper.c:98-119)
    if (q.resume) {
    } else if (par && coop) {
        for (int i = 0; i < k; i++) {                    // one wave-cooperative evaluation per component
            for (int j = lane; j < p; j += 64) Y[j] = S[i + k * j];
            __syncthreads();
            const double v = LUD(i + 1, Y);
            if (lane == 0) {
                L[i] = v;
                if (v == -INFINITY) atomicOr(q.flag, FLAG_NEGINF_INIT);
                else if (v == INFINITY) atomicOr(q.flag, FLAG_POSINF);
                else if (isnan(v)) atomicOr(q.flag, FLAG_NAN);
            }
            __syncthreads();
        }
    } else if (par) {
        for (int i = lane; i < k; i += 64) {
            double th[TEMPER_MAXP];
            for (int j = 0; j < p; j++) th[j] = S[i + k * j];
            double v = lud_temper(q.kind, p, gpar, k, i + 1, th);
            L[i] = v;
            if (v == -INFINITY) atomicOr(q.flag, FLAG_NEGINF_INIT);
            else if (v == INFINITY) atomicOr(q.flag, FLAG_POSINF);
            else if (isnan(v)) atomicOr(q.flag, FLAG_NAN);
        }
    } else {
        for (int i = lane; i < k; i += 64) L[i] = NAN;
        __syncthreads();
        {
            const int i = (int)S[0];
            const double v = coop ? LUD(i, S + 1) : (lane == 0 ? lud_temper(q.kind, p, gpar, k, i, S + 1) : 0.0);
            if (lane == 0) {
                L[i - 1] = v;
                if (v == -INFINITY) atomicOr(q.flag, FLAG_NEGINF_INIT);
                else if (v == INFINITY) atomicOr(q.flag, FLAG_POSINF);
                else if (isnan(v)) atomicOr(q.flag, FLAG_NAN);
            }
        }
    }
    __syncthreads();

    const long niter = (long)q.nbatch * q.blen * q.nspac;
    const unsigned long long gens = q.ens_offset + (unsigned long long)e;
    const int sk = q.scale_kind & 7;
    const int slen1 = sk == MCMCHIP_SCALE_CONSTANT ? 1 : (sk == MCMCHIP_SCALE_DIAGONAL ? p : p * p);
    const bool slist = (q.scale_kind & MCMCHIP_SCALE_LIST) != 0;
    const bool inj = q.rng_mode == MCMCHIP_RNG_INJECTED;

    long it = q.it_begin;
    int is = (int)(it % q.nspac), jb = (int)((it / q.nspac) % q.blen);
    long kb = it / q.nspac / q.blen;
    for (; it < q.it_end;) {
        if (is == 0 && jb == 0) {                       // a batch starts
            for (int i = lane; i < nout; i += 64) buf[i] = 0.0;
            for (int i = lane; i < k; i += 64) ibuf[i] = 0.0;
        }
        {
            {
                const unsigned long long git = q.iter_offset + (unsigned long long)it;
                double u0, u1, u2 = 0.0, u3 = 0.0;   // which, choose1, choose2, hastings
                const double *up;
                if (inj) {
                    up = q.unif + ((size_t)e * niter + it) * 4;
                } else {
                    // device RNG: every 64 iterations the 64 lanes draw, in one pass, all Philox blocks of the
                    // next 64 iterations (lane t <-> iteration it + t) into LDS, so the dependent chain of an
                    // iteration only contains LDS reads.  Same counter addressing as a draw-on-demand scheme.
                    if ((it & 63) == 0) {
                        const unsigned long long g2 = git + (unsigned long long)lane;
                        double a, b;
                        philox_u2(q.seed, PHILOX_TAG_TEMPER, gens, g2, 0u, a, b);
                        Ubuf[lane * 4 + 0] = a; Ubuf[lane * 4 + 1] = b;
                        philox_u2(q.seed, PHILOX_TAG_TEMPER, gens, g2, 1u, a, b);
                        Ubuf[lane * 4 + 2] = a; Ubuf[lane * 4 + 3] = b;
                        for (int j2 = 0; 2 * j2 < p; j2++) {
                            philox_n2(q.seed, PHILOX_TAG_TEMPER, gens, g2, (uint32_t)(2 + j2), a, b);
                            Zbuf[lane * pz + 2 * j2] = a; Zbuf[lane * pz + 2 * j2 + 1] = b;
                        }
                        __syncthreads();
                    }
                    up = Ubuf + (it & 63) * 4;
                }
                u0 = up[0]; u1 = up[1];
#define TDBG(arr, w, j) (arr)[(size_t)e * (w) * niter + it + (size_t)niter * (j)]
                if (q.debug) {       // temper.c:528-543: state before the update, the within/swap draw
                    for (int i = lane; i < nstate; i += 64) TDBG(q.dbg_state, nstate, i) = S[i];
                    if (lane == 0) { q.dbg_which[(size_t)e * niter + it] = u0 < 0.5; q.dbg_unif_which[(size_t)e * niter + it] = u0; }
                }
                const bool within = u0 < 0.5;          // temper.c:537-538
                double lh;
                int ci, cj = 0, slot_ = 0;             // 1-based components; slot_ = position of cj among ci's neighbours
                double ly = 0.0, sco = 0.0, spr = 0.0;
                if (within) {
                    ci = par ? pick1(u1, k) : (int)S[0];
                    // normals (temper.c:995-996) and proposal (1003-1024)
                    if (inj) {
                        for (int j = lane; j < p; j += 64) Z[j] = q.norm[((size_t)e * niter + it) * p + j];
                    } else {
                        for (int j = lane; j < p; j += 64) Z[j] = Zbuf[(it & 63) * pz + j];
                    }
                    __syncthreads();
                    const double *sc = gscale + (slist ? (size_t)(ci - 1) * slen1 : 0);
                    for (int j = lane; j < p; j += 64) {
                        const double xj = par ? S[(ci - 1) + k * j] : S[1 + j];
                        double yj;
                        if (sk == MCMCHIP_SCALE_CONSTANT) yj = add_nofma(xj, mul_nofma(sc[0], Z[j]));
                        else if (sk == MCMCHIP_SCALE_DIAGONAL) yj = add_nofma(xj, mul_nofma(sc[j], Z[j]));
                        else {
                            yj = xj;
                            for (int i = 0; i < p; i++) yj = add_nofma(yj, mul_nofma(sc[j + p * i], Z[i]));
                        }
                        Y[j] = yj;
                    }
                    __syncthreads();
                    ly = LUD(ci, Y);
                    lh = ly - L[ci - 1];
                    if (q.debug) {   // temper.c:589-609 / 653-665
                        for (int j = lane; j < p; j += 64) {
                            TDBG(q.dbg_proposal, p + 1, j + 1) = Y[j];
                            TDBG(q.dbg_norm, p, j) = Z[j];
                            if (par) TDBG(q.dbg_coproposal, p + 1, j + 1) = S[(ci - 1) + k * j];
                        }
                        if (lane == 0) {
                            TDBG(q.dbg_proposal, p + 1, 0) = ci;
                            if (par) { TDBG(q.dbg_coproposal, p + 1, 0) = ci; TDBG(q.dbg_unif_choose, 2, 0) = u1; TDBG(q.dbg_unif_choose, 2, 1) = NAN; }
                            else q.dbg_unif_choose[(size_t)e * niter + it] = NAN;
                        }
                    }
                    if (ly == INFINITY || isnan(ly)) { if (lane == 0) atomicOr(q.flag, isnan(ly) ? FLAG_NAN : FLAG_POSINF); }
                } else if (par) {
                    u2 = up[2];
                    ci = pick1(u1, k);
                    slot_ = pick1(u2, nn[ci - 1]) - 1;
                    cj = nb[(ci - 1) * maxnn + slot_] + 1;
                    // the two rows go through the LDS buffers Y (x_i) and Z (x_j): no private arrays
                    for (int j = lane; j < p; j += 64) { Y[j] = S[(ci - 1) + k * j]; Z[j] = S[(cj - 1) + k * j]; }
                    __syncthreads();
                    const double *xi = Y, *xj = Z;
                    sco = LUD(cj, xi);   // (j, x_i) first (temper.c:738-739)
                    spr = LUD(ci, xj);   // (i, x_j)
                    lh = spr + sco - L[cj - 1] - L[ci - 1];       // temper.c:742-744
                    if (q.debug) {   // coproposal = (i, x_i), proposal = (j, x_j) (temper.c:720-731,757-766)
                        for (int j = lane; j < p; j += 64) {
                            TDBG(q.dbg_proposal, p + 1, j + 1) = xj[j];
                            TDBG(q.dbg_coproposal, p + 1, j + 1) = xi[j];
                            TDBG(q.dbg_norm, p, j) = NAN;
                        }
                        if (lane == 0) {
                            TDBG(q.dbg_proposal, p + 1, 0) = cj;
                            TDBG(q.dbg_coproposal, p + 1, 0) = ci;
                            TDBG(q.dbg_unif_choose, 2, 0) = u1;
                            TDBG(q.dbg_unif_choose, 2, 1) = u2;
                        }
                    }
                    if (isnan(sco) || isnan(spr) || sco == INFINITY || spr == INFINITY) { if (lane == 0) atomicOr(q.flag, FLAG_NAN); }
                } else {
                    ci = (int)S[0];
                    slot_ = pick1(u1, nn[ci - 1]) - 1;
                    cj = nb[(ci - 1) * maxnn + slot_] + 1;
                    ly = LUD(cj, S + 1);
                    lh = ly - L[ci - 1] + log((double)nn[ci - 1]) - log((double)nn[cj - 1]);   // temper.c:810-812
                    if (q.debug) {   // temper.c:825-835
                        for (int j = lane; j < p; j += 64) { TDBG(q.dbg_proposal, p + 1, j + 1) = S[1 + j]; TDBG(q.dbg_norm, p, j) = NAN; }
                        if (lane == 0) { TDBG(q.dbg_proposal, p + 1, 0) = cj; q.dbg_unif_choose[(size_t)e * niter + it] = u1; }
                    }
                    if (ly == INFINITY || isnan(ly)) { if (lane == 0) atomicOr(q.flag, isnan(ly) ? FLAG_NAN : FLAG_POSINF); }
                }
                if (isnan(lh) || (isinf(lh) && lh > 0)) { if (lane == 0) atomicOr(q.flag, FLAG_HASTINGS); lh = -INFINITY; }
                const double lh_trace = lh;
                bool accept = true;
                double uh_trace = NAN;
                if (lh < 0.0) {                                   // temper.c:582-587
                    u3 = up[3];
                    uh_trace = u3;
                    accept = u3 < exp(lh);
                }
                if (q.debug && lane == 0) {
                    q.dbg_lh[(size_t)e * niter + it] = lh_trace;
                    q.dbg_uh[(size_t)e * niter + it] = uh_trace;
                    q.dbg_acceptd[(size_t)e * niter + it] = accept;
                }
#undef TDBG
                __syncthreads();   // every lane has read S / L / Y for this decision
                if (within) {
                    if (accept) {
                        for (int j = lane; j < p; j += 64) {
                            if (par) S[(ci - 1) + k * j] = Y[j];
                            else S[1 + j] = Y[j];
                        }
                        if (lane == 0) { L[ci - 1] = ly; axn[ci - 1]++; }
                    }
                    if (lane == 0) axd[ci - 1]++;
                } else if (par) {
                    if (accept) {
                        for (int j = lane; j < p; j += 64) {      // LDS row swap
                            const double a = S[(ci - 1) + k * j], b = S[(cj - 1) + k * j];
                            S[(ci - 1) + k * j] = b;
                            S[(cj - 1) + k * j] = a;
                        }
                        if (lane == 0) { L[ci - 1] = spr; L[cj - 1] = sco; }
                    }
                    if (lane == 0) {
                        if (accept) ain[(ci - 1) * maxnn + slot_]++;
                        aid[(ci - 1) * maxnn + slot_]++;
                    }
                } else {
                    if (lane == 0) {
                        if (accept) { S[0] = cj; L[cj - 1] = ly; ain[(ci - 1) * maxnn + slot_]++; }
                        aid[(ci - 1) * maxnn + slot_]++;
                    }
                }
                __syncthreads();
                it++;
            }
            if (++is < q.nspac) continue;
            is = 0;
            // batch accumulation once per blen step (temper.c:851-867)
            if (mapped) {                       // index vector / c(v, v^2) over the identity output
                for (int i = lane; i < nout; i += 64) {
                    double v = S[q.out_map[i] + (par ? 0 : 1)];
                    if (i >= q.sq_from) v = mul_nofma(v, v);
                    buf[i] = add_nofma(buf[i], v);
                }
            } else if (!corner) {
                for (int i = lane; i < nout; i += 64) buf[i] += par ? S[i] : S[i + 1];
            } else if (par) {                   // tests/temp-par-witch.R:52-59
                for (int i = lane; i < k; i += 64) {
                    bool all = true;
                    for (int j = 0; j < p; j++) all &= fabs(S[i + k * j]) - gpar[i] > 0;
                    buf[i] += all ? 1.0 : 0.0;
                }
            } else {                            // tests/temp-ser-witch.R:51-62: c(bar, baz)
                const int ic = (int)S[0];
                bool all = true;
                for (int j = 0; j < p; j++) all &= fabs(S[1 + j]) > gpar[ic - 1];
                for (int i = lane; i < 2 * k; i += 64) {
                    const int comp = i < k ? i : i - k;
                    buf[i] += (comp == ic - 1) ? (i < k ? (all ? 1.0 : 0.0) : 1.0) : 0.0;
                }
            }
            if (!par && lane == 0) ibuf[(int)S[0] - 1] += 1.0;
        }
        if (++jb < q.blen) continue;
        jb = 0;
        __syncthreads();
        for (int i = lane; i < nout; i += 64)
            q.batch[(size_t)e * q.nbatch * nout + kb + (size_t)q.nbatch * i] = buf[i] / q.blen;   // temper.c:871-879
        if (!par)
            for (int i = lane; i < k; i += 64)
                q.ibatch[(size_t)e * q.nbatch * k + kb + (size_t)q.nbatch * i] = ibuf[i] / q.blen; // temper.c:881-886
        __syncthreads();
        kb++;
    }
    __syncthreads();
    if (carry)
        for (int i = lane; i < lay.ubuf; i += 64) carry[i] = lds[i];
    if (q.it_end < niter) return;
    for (int i = lane; i < k; i += 64) q.acceptx[(size_t)e * k + i] = (double)axn[i] / (double)axd[i];  // 0/0 = NaN (temper.c:889-890)
    for (int idx = lane; idx < k * k; idx += 64) {
        const int i = idx % k, j = idx / k;
        double v = NAN;                                              // NA for non-neighbours (temper.c:896-897)
        if (q.neighbors[i + k * j]) {
            int slot = 0;
            while (nb[i * maxnn + slot] != j) slot++;
            v = (double)ain[i * maxnn + slot] / (double)aid[i * maxnn + slot];
        }
        q.accepti[(size_t)e * k * k + i + k * j] = v;
    }
    for (int i = lane; i < nstate; i += 64) q.final_[(size_t)e * nstate + i] = S[i];
}


// ---------------------------------------------------------------------------------------------
// Run-parallel variant for the common shape of parallel tempering: k <= 64 components of dimension P <= 4 with
// closed-form densities (C4: 64 temperatures, witch's hat p = 3 / tempered mixture p = 2).  Lane c of the ensemble's
// wavefront OWNS component c: its state x[P], its cached log density (temper.c:123-124), its batch accumulator
// and its within-acceptance counters are registers of that lane.  Elementary updates that touch DISJOINT components commute: a within-component
// step reads and writes one row of the state, a swap two rows, and each iteration owns its own random
// numbers (fixed Philox counters / injected slots).  So a maximal RUN of consecutive iterations whose
// component sets are pairwise disjoint can be executed in one pass, every touched component updated by
// its own lane at the same time; the result is the sequential one, bit for bit.  Per window of 64
// iterations: lane t draws iteration t's numbers and decodes (within | swap, i, j, slot); a scalar scan
// over the 64 component masks cuts the window into runs (also at batch ends); per run the iterations
// scatter their index to their components' lanes through LDS, each such lane gathers its z / u, fetches the
// partner row by ds_bpermute for a swap, and ONE log-density evaluation, one exp and one predicated
// commit serve the whole run.  With k = 64 a run holds ~6 iterations.  Batch sums are added per
// accumulation point in time order with the value each lane held at that point (before or after its
// update), so they too equal the sequential sums exactly.
template <int P> __device__ __forceinline__ double witch_value(double bnd, double inside, double cornerv, const double *th)
{
    bool out = false, all = true;                  // same tests as lud_temper's WITCH case
#pragma unroll
    for (int j = 0; j < P; j++) {
        const double a = fabs(th[j]);
        out |= a > 1.0;
        all &= a > bnd;
    }
    return out ? -INFINITY : (all ? cornerv : inside);
}

struct TemperRunLds {
    // offsets in 8-byte words; the window buffers (zbuf .. hdr) exist twice, `wstride` words apart
    int zbuf, u3buf, desc, tm, prevc, hdr, wstride, nnT, nbT, ain, aid, par, scale, end;
    int npar_lds, nscale_lds;
};

// wavefront-scope hand-over inside one wavefront's private LDS traffic (LDS operations of a wavefront complete in order)
#define WAVE_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)

// Three wavefronts per ensemble: wave 0 CONSUMES window w (owns the state, executes the runs) while wave 1 draws the
// uniforms of window w+1, decodes the moves and builds the run masks, and wave 2 draws its normals; the window
// buffers are double-buffered in LDS and the three meet at one workgroup barrier per window.
template <int P, int KIND>
__global__ __launch_bounds__(192) void temper_run_kernel(TemperDev q, TemperRunLds lay, int maxnn)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long e = blockIdx.x;
    const int k = q.k;
    constexpr int PZ = (P + 1) & ~1;
    constexpr bool WITCH = KIND == MCMCHIP_T_WITCH;
    const bool corner = q.out_kind == MCMCHIP_TOUT_WITCH_CORNER;
    const int nout = corner ? k : k * P;
    int *nnT = (int *)(lds + lay.nnT), *nbT = (int *)(lds + lay.nbT);
    unsigned *ain = (unsigned *)(lds + lay.ain), *aid = (unsigned *)(lds + lay.aid);
    const double *gpar = q.par, *gscale = q.scale;
    if (lay.npar_lds > 0) {
        double *lp = lds + lay.par;
        for (int i = threadIdx.x; i < lay.npar_lds; i += 192) lp[i] = q.par[i];
        gpar = lp;
    }
    if (lay.nscale_lds > 0) {
        double *ls = lds + lay.scale;
        for (int i = threadIdx.x; i < lay.nscale_lds; i += 192) ls[i] = q.scale[i];
        gscale = ls;
    }
    // A launch runs windows [it_begin / 64, ceil(it_end / 64)).  Carry record of an ensemble (doubles): per lane
    // x[P], L, acc[P], axn, axd (2 P + 3 values, lane-major), then the k * maxnn pair counters ain, aid.
    constexpr int CW = 2 * P + 3;
    double *carry = q.carry ? q.carry + (size_t)e * (64 * CW + 2 * k * maxnn) : nullptr;
    for (int i = threadIdx.x; i < k * maxnn; i += 192) {
        ain[i] = q.resume ? (unsigned)carry[64 * CW + i] : 0u;
        aid[i] = q.resume ? (unsigned)carry[64 * CW + k * maxnn + i] : 0u;
    }
    const bool live = lane < k;
    const int c = live ? lane : 0;                 // idle lanes shadow component 0 and never commit
    const int sk = q.scale_kind & 7;
    const bool slist = (q.scale_kind & MCMCHIP_SCALE_LIST) != 0;
    const bool inj = q.rng_mode == MCMCHIP_RNG_INJECTED;
    if (wave == 0 && live) {
        int n = 0;
        for (int j = 0; j < k; j++)
            if (q.neighbors[c + k * j]) nbT[c * maxnn + n++] = j;     // ascending (temper.c:504-510)
        nnT[c] = n;
    }
    __syncthreads();

    const long niter = (long)q.nbatch * q.blen * q.nspac;
    const long w_begin = q.it_begin / 64, w_end = (q.it_end + 63) / 64;
    const unsigned long long gens = q.ens_offset + (unsigned long long)e;
    const unsigned long long per = (unsigned long long)q.blen * (unsigned long long)q.nspac;   // iterations per batch
    const unsigned nspac = (unsigned)q.nspac;
    const bool small_per = per < 0x7fffff00ull;

    if (wave != 0) {
        // ================= producers =================
        unsigned sp = (unsigned)((unsigned long long)(w_begin * 64) % nspac);      // it0 mod nspac   (wave 1)
        unsigned long long bp = (unsigned long long)(w_begin * 64) % per;          // it0 mod per
        for (long w = w_begin; w < w_end; w++) {
            double *wb_ = lds + (w & 1) * lay.wstride;
            const long it0 = w * 64;
            const int nvalid = (int)(niter - it0 < 64 ? niter - it0 : 64);
            const bool valid = lane < nvalid;
            const unsigned long long g2 = q.iter_offset + (unsigned long long)it0 + (unsigned long long)lane;
            if (wave == 2) {
                // ---- lane t: the normals of iteration it0 + t
                double zt[PZ];
#pragma unroll
                for (int j = 0; j < PZ; j++) zt[j] = 0.0;
                if (valid) {
                    if (inj) {
                        const double *zp = q.norm + ((size_t)e * niter + it0 + lane) * P;
#pragma unroll
                        for (int j = 0; j < P; j++) zt[j] = zp[j];
                    } else {
#pragma unroll
                        for (int j2 = 0; 2 * j2 < P; j2++)
                            philox_n2(q.seed, PHILOX_TAG_TEMPER, gens, g2, (uint32_t)(2 + j2), zt[2 * j2], zt[2 * j2 + 1]);
                    }
                }
                double *Zb = wb_ + lay.zbuf;
#pragma unroll
                for (int j = 0; j < PZ; j++) Zb[lane * PZ + j] = zt[j];
            } else {
                // ---- lane t: the uniforms and the decoded move of iteration it0 + t
                double u0 = 1.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
                if (valid) {
                    if (inj) {
                        const double *up = q.unif + ((size_t)e * niter + it0 + lane) * 4;
                        u0 = up[0]; u1 = up[1]; u2 = up[2]; u3 = up[3];
                    } else {
                        philox_u2(q.seed, PHILOX_TAG_TEMPER, gens, g2, 0u, u0, u1);
                        philox_u2(q.seed, PHILOX_TAG_TEMPER, gens, g2, 1u, u2, u3);
                    }
                }
                const bool t_within = u0 < 0.5;                        // temper.c:537-538
                const int t_l = valid ? pick1(u1, k) - 1 : 0;
                int t_m = t_l, t_slot = 0;
                if (valid && !t_within) {
                    t_slot = pick1(u2, nnT[t_l]) - 1;
                    t_m = nbT[t_l * maxnn + t_slot];
                }
                // accumulation points (every nspac iterations, temper.c:851-867) and batch ends, as lane masks
                const bool t_acc = (sp + (unsigned)lane + 1u) % nspac == 0u;
                const bool t_bend = small_per ? ((unsigned)bp + (unsigned)lane + 1u) % (unsigned)per == 0u
                                              : bp + (unsigned long long)lane + 1ull == per;   // per >= 2^31: one per window at most
                const unsigned long long Am = __ballot(valid && t_acc);
                const unsigned long long Bm = __ballot(valid && t_bend);
                sp = (sp + 64u) % nspac;
                if (small_per) bp = (unsigned long long)(((unsigned)bp + 64u) % (unsigned)per);
                else { bp += 64ull; if (bp >= per) bp -= per; }
                unsigned long long *Tm = (unsigned long long *)(wb_ + lay.tm);
                (wb_ + lay.u3buf)[lane] = u3;
                ((unsigned *)(wb_ + lay.desc))[lane] =
                    (unsigned)t_l | ((unsigned)t_m << 8) | ((unsigned)t_slot << 16) | ((unsigned)t_within << 24);
                Tm[lane] = 0ull;
                WAVE_SYNC();
                // transpose "iteration t touches components {i, j}" into "component c is touched by iterations T_c"
                if (valid) {
                    atomicOr(&Tm[t_l], 1ull << lane);
                    if (!t_within) atomicOr(&Tm[t_m], 1ull << lane);
                }
                WAVE_SYNC();
                // the last earlier iteration of the window that shares a component with iteration t
                const unsigned long long conf = (Tm[t_l] | Tm[t_m]) & ((1ull << lane) - 1ull);
                ((int *)(wb_ + lay.prevc))[lane] = conf ? 63 - __builtin_clzll(conf) : -1;
                if (lane == 0) {
                    unsigned long long *h = (unsigned long long *)(wb_ + lay.hdr);
                    h[0] = Am; h[1] = Bm;
                }
            }
            __syncthreads();               // window w handed over (the consumer is one window behind)
        }
        __syncthreads();                   // the consumer's last window
        return;
    }

    // ================= consumer =================
    double x[P], acc[P], sc[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
        x[j] = q.initial[(size_t)e * k * P + c + k * j];
        acc[j] = 0.0;
        // constant scale = diagonal scale with equal entries (same multiply, same add)
        sc[j] = sk == MCMCHIP_SCALE_CONSTANT ? q.scale[slist ? c : 0]
              : sk == MCMCHIP_SCALE_DIAGONAL ? q.scale[(slist ? c * P : 0) + j] : 0.0;
    }
    const double *scf = gscale + (slist ? (size_t)c * P * P : 0);      // full-matrix scale of this lane's component
    unsigned axn = 0, axd = 0;
    double wb = 0.0, win = 0.0, wco = 0.0;
    if (WITCH) { wb = q.par[c]; win = q.par[k + c]; wco = q.par[2 * k + c]; }
    typedef __attribute__((address_space(3))) const double lds_cd;
    const bool par_lds = lay.npar_lds > 0;
    auto LUD_OWN = [&](const double *th) -> double {                  // log density of (component c, th)
        if (WITCH) return witch_value<P>(wb, win, wco, th);
        // parameters staged in LDS are read with ds_read (through a generic pointer they would be FLAT loads)
        if (par_lds) return lud_temper<lds_cd *>(KIND, P, (lds_cd *)gpar, k, c + 1, th);
        return lud_temper(KIND, P, gpar, k, c + 1, th);
    };
    double L;
    if (!q.resume) {
        L = LUD_OWN(x);                                               // temper.c:98-119
        if (live) {
            if (L == -INFINITY) atomicOr(q.flag, FLAG_NEGINF_INIT);
            else if (L == INFINITY) atomicOr(q.flag, FLAG_POSINF);
            else if (isnan(L)) atomicOr(q.flag, FLAG_NAN);
        }
    } else {
        const double *cr = carry + lane * CW;
#pragma unroll
        for (int j = 0; j < P; j++) { x[j] = cr[j]; acc[j] = cr[P + 1 + j]; }
        L = cr[P];
        axn = (unsigned)cr[2 * P + 1];
        axd = (unsigned)cr[2 * P + 2];
    }
    long kb = (long)((unsigned long long)(w_begin * 64) / per);      // batches completed before this launch
    __syncthreads();                       // the first window is ready
    for (long w = w_begin; w < w_end; w++) {
        const double *wb_ = lds + (w & 1) * lay.wstride;
        const double *Zbuf = wb_ + lay.zbuf, *U3 = wb_ + lay.u3buf;
        const unsigned *desc = (const unsigned *)(wb_ + lay.desc);
        const long it0 = w * 64;
        const int nvalid = (int)(niter - it0 < 64 ? niter - it0 : 64);
        const unsigned long long T = ((const unsigned long long *)(wb_ + lay.tm))[lane];
        const int prevc = ((const int *)(wb_ + lay.prevc))[lane];
        const unsigned long long *h = (const unsigned long long *)(wb_ + lay.hdr);
        const unsigned long long Am = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)h[0]) |
                                      ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(h[0] >> 32)) << 32);
        const unsigned long long Bm = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)h[1]) |
                                      ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(h[1] >> 32)) << 32);

        // this lane's (component's) first touching iteration of the window and its draws
        int nti = T ? __builtin_ctzll(T) : 64;
        unsigned cd = desc[nti & 63];
        double cuh = U3[nti & 63], cz[P];
#pragma unroll
        for (int j = 0; j < P; j++) cz[j] = Zbuf[(nti & 63) * PZ + j];

        // ---- cut the window into maximal runs of pairwise disjoint iterations and execute each run
        int rs = 0;
        while (rs < nvalid) {
            // the run ends before the first later iteration that conflicts with one inside it, after a batch end,
            // or with the window
            const unsigned long long cm = __ballot(prevc >= rs) & ~((2ull << rs) - 1ull);
            int t = cm ? __builtin_ctzll(cm) : 64;
            const unsigned long long bmr = Bm & ~((1ull << rs) - 1ull);
            const int be = bmr ? __builtin_ctzll(bmr) + 1 : 65;
            const bool bend_prev = be <= t;
            if (bend_prev) t = be;
            if (t > nvalid) t = nvalid;
            {
                // ===== run [rs, t): lane as component -- nti is its first touching iteration at or after rs
                // (64: none), already fetched; it takes part if that iteration lies inside the run
                const bool active = nti < t;
                const int ti = nti;
                const int d_l = cd & 0xff, d_m = (cd >> 8) & 0xff, d_slot = (cd >> 16) & 0xff;
                const int role = (cd >> 24) ? 0 : (lane == d_l ? 1 : 2);  // 0 within, 1 swap (component i), 2 swap (component j)
                const bool sw = active && role != 0;
                const int partner = sw ? (role == 1 ? d_m : d_l) : lane;
                double xq[P], th[P], xold[P];
#pragma unroll
                for (int j = 0; j < P; j++) { xq[j] = __shfl(x[j], partner); xold[j] = x[j]; }
                const double Lq = __shfl(L, partner);
                // while the exchange is in flight: fetch the draws of this lane's NEXT touching iteration
                const unsigned long long rem2 = T & ~((2ull << (ti & 63)) - 1ull);
                const int nti2 = (active && rem2) ? __builtin_ctzll(rem2) : 64;
                const int tl = nti2 & 63;
                const unsigned nd = desc[tl];
                const double nuh = U3[tl];
                double nz[P];
#pragma unroll
                for (int j = 0; j < P; j++) nz[j] = Zbuf[tl * PZ + j];
                // within: proposal (temper.c:1003-1024); swap: the partner's row
                if (sk == MCMCHIP_SCALE_FULL) {
#pragma unroll
                    for (int j = 0; j < P; j++) {
                        double yj = x[j];
#pragma unroll
                        for (int i = 0; i < P; i++) yj = add_nofma(yj, mul_nofma(scf[j + P * i], cz[i]));
                        th[j] = sw ? xq[j] : yj;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < P; j++) {
                        const double yj = add_nofma(x[j], mul_nofma(sc[j], cz[j]));
                        th[j] = sw ? xq[j] : yj;
                    }
                }
                const double v = LUD_OWN(th);     // within: lud(i, y); swap: lud(own component, partner's row)
                const double vq = __shfl(v, partner);
                const double spr = role == 1 ? v : vq;                // (i, x_j)
                const double sco = role == 1 ? vq : v;                // (j, x_i)
                const double Li = role == 1 ? L : Lq, Lj = role == 1 ? Lq : L;
                double lh = sw ? spr + sco - Lj - Li : v - L;          // temper.c:742-744 | 574-580
                // every error case of the reference (a +Inf or NaN density, a NaN or +Inf Hastings ratio) ends in
                // a ratio that is not < +Inf: one test on the fast path, the exact flags on the rare one
                const bool bad = !(lh < INFINITY);
                if (__any(active && bad)) {
                    if (sw) {
                        if (isnan(sco) || isnan(spr) || sco == INFINITY || spr == INFINITY) { if (role == 1) atomicOr(q.flag, FLAG_NAN); }
                    } else if (active && (v == INFINITY || isnan(v))) atomicOr(q.flag, isnan(v) ? FLAG_NAN : FLAG_POSINF);
                    if (bad && active && role != 2) atomicOr(q.flag, FLAG_HASTINGS);
                }
                if (bad) lh = -INFINITY;
                // accept iff lh >= 0 or u < exp(lh) (temper.c:582-587).  A single-precision exp (relative error
                // < 2e-5 on [-80, 0]) settles every decision whose u lies outside a 1e-4 band around it; only
                // a lane inside the band sends the run through the double-precision exp.  Same decisions.
                bool accept = true;
                if (lh < 0.0) {
                    const double ef = (double)__expf((float)lh);
                    const bool far = lh > -80.0;
                    const bool sure = far ? (cuh < ef * (1.0 - 1e-4) || cuh > ef * (1.0 + 1e-4)) : (cuh > 1e-30);
                    accept = far && cuh < ef;
                    if (__any(active && !sure)) accept = cuh < exp(lh);
                }
                if (active) {
                    if (accept) {
                        L = v;
#pragma unroll
                        for (int j = 0; j < P; j++) x[j] = th[j];
                    }
                    if (role == 0) { axd++; if (accept) axn++; }
                    else if (role == 1) {
                        if (accept) atomicAdd(&ain[lane * maxnn + d_slot], 1u);
                        atomicAdd(&aid[lane * maxnn + d_slot], 1u);
                    }
                    nti = nti2; cd = nd; cuh = nuh;
#pragma unroll
                    for (int j = 0; j < P; j++) cz[j] = nz[j];
                }
                // batch sums: the accumulation points of the run in time order, each with the value this lane
                // held at that point (its update happens at iteration ti)
                unsigned long long am = Am & ((t >= 64 ? ~0ull : ((1ull << t) - 1ull)) & ~((1ull << rs) - 1ull));
                const int tau = active ? ti : -1;
                while (am) {
                    const int tp = __builtin_ctzll(am);
                    am &= am - 1ull;
                    const bool pre = tp < tau;
                    if (!corner) {
#pragma unroll
                        for (int j = 0; j < P; j++) acc[j] += pre ? xold[j] : x[j];
                    } else {                                           // tests/temp-par-witch.R:52-59
                        bool all = true;
#pragma unroll
                        for (int j = 0; j < P; j++) all &= fabs(pre ? xold[j] : x[j]) - wb > 0;
                        acc[0] += all ? 1.0 : 0.0;
                    }
                }
                if (bend_prev) {                                       // the run ended a batch (temper.c:871-875)
                    if (live) {
                        if (!corner) {
#pragma unroll
                            for (int j = 0; j < P; j++)
                                q.batch[(size_t)e * q.nbatch * nout + kb + (size_t)q.nbatch * (c + k * j)] = acc[j] / q.blen;
                        } else {
                            q.batch[(size_t)e * q.nbatch * nout + kb + (size_t)q.nbatch * c] = acc[0] / q.blen;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < P; j++) acc[j] = 0.0;
                    kb++;
                }
                rs = t;
            }
        }
        __syncthreads();                   // window w consumed; window w+1 is ready
    }
    if (carry) {
        double *cr = carry + lane * CW;
#pragma unroll
        for (int j = 0; j < P; j++) { cr[j] = x[j]; cr[P + 1 + j] = acc[j]; }
        cr[P] = L;
        cr[2 * P + 1] = (double)axn;
        cr[2 * P + 2] = (double)axd;
        for (int i = lane; i < k * maxnn; i += 64) {
            carry[64 * CW + i] = (double)ain[i];
            carry[64 * CW + k * maxnn + i] = (double)aid[i];
        }
    }
    if (q.it_end < niter) return;
    if (live) {
        q.acceptx[(size_t)e * k + c] = (double)axn / (double)axd;        // 0/0 = NaN (temper.c:889-890)
        int slot = 0;
        for (int j = 0; j < k; j++) {
            double v = NAN;                                              // NA for non-neighbours (temper.c:896-897)
            if (q.neighbors[c + k * j]) {
                v = (double)ain[c * maxnn + slot] / (double)aid[c * maxnn + slot];
                slot++;
            }
            q.accepti[(size_t)e * k * k + c + k * j] = v;
        }
#pragma unroll
        for (int j = 0; j < P; j++) q.final_[(size_t)e * k * P + c + k * j] = x[j];
    }
}

template <int P, int KIND>
static int launch_temper_run(const TemperDev &q, int maxnn, hipStream_t s)
{
    const int k = q.k;
    TemperRunLds lay;
    int o = 0;
    lay.zbuf = o; o += 64 * ((P + 1) & ~1);
    lay.u3buf = o; o += 64;
    lay.desc = o; o += 32;
    lay.tm = o; o += 64;
    lay.prevc = o; o += 32;
    lay.hdr = o; o += 2;
    lay.wstride = o;
    o *= 2;                                                             // two window buffers
    lay.nnT = o; o += (k + 1) / 2;
    lay.nbT = o; o += (k * maxnn + 1) / 2;
    lay.ain = o; o += (k * maxnn + 1) / 2;
    lay.aid = o; o += (k * maxnn + 1) / 2;
    const long slen1h = (q.scale_kind & 7) == MCMCHIP_SCALE_FULL ? P * P : 0;     // constant / diagonal scales live in registers
    const long nscale = (q.scale_kind & MCMCHIP_SCALE_LIST) ? slen1h * k : slen1h;
    lay.npar_lds = (KIND != MCMCHIP_T_WITCH && q.npar > 0 && q.npar <= 2048) ? (int)q.npar : 0;
    lay.nscale_lds = (int)nscale;
    lay.par = o; o += lay.npar_lds;
    lay.scale = o; o += lay.nscale_lds;
    lay.end = o;
    hipLaunchKernelGGL((temper_run_kernel<P, KIND>), dim3((unsigned)q.nens), dim3(192), (size_t)o * 8, s, q, lay, maxnn);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

static int temper_nout(const TemperDev &q)
{
    if (q.out_kind == MCMCHIP_TOUT_INDEX || q.out_kind == MCMCHIP_TOUT_MOMENTS2) return q.nout_map;
    if (q.out_kind == MCMCHIP_TOUT_WITCH_CORNER) return q.parallel ? q.k : 2 * q.k;
    return q.parallel ? q.k * q.p : q.p;
}

// LDS layout of temper_kernel for this call (one place: the launch and the size of the carry record both use it)
static TemperLds temper_lds_layout(const TemperDev &q)
{
    const int k = q.k, p = q.p, maxnn = q.maxnn;
    const int nstate = q.parallel ? k * p : p + 1;
    const int nout = temper_nout(q);
    TemperLds lay;
    int o = 0;
    lay.state = o; o += nstate;
    lay.L = o; o += k;
    lay.Y = o; o += p;
    lay.Z = o; o += p;
    lay.buf = o; o += nout;
    lay.ibuf = o; o += k;
    lay.axn = o; o += (k + 1) / 2;
    lay.axd = o; o += (k + 1) / 2;
    lay.ain = o; o += (k * maxnn + 1) / 2;
    lay.aid = o; o += (k * maxnn + 1) / 2;
    lay.nn = o; o += (k + 1) / 2;
    lay.nb = o; o += (k * maxnn + 1) / 2;
    lay.ubuf = o; o += 64 * 4;
    lay.zbuf = o; o += 64 * ((p + 1) & ~1);
    // descriptor parameters and scale: into LDS when small (they are read inside the serial chain)
    const long npar_dev = q.kind == MCMCHIP_T_WITCH ? 3L * k : q.npar;
    const long slen1h = (q.scale_kind & 7) == MCMCHIP_SCALE_CONSTANT ? 1 : ((q.scale_kind & 7) == MCMCHIP_SCALE_DIAGONAL ? p : p * p);
    const long nscale = (q.scale_kind & MCMCHIP_SCALE_LIST) ? slen1h * k : slen1h;
    lay.npar_lds = (npar_dev > 0 && npar_dev <= 2048) ? (int)npar_dev : 0;
    lay.nscale_lds = nscale <= 1024 ? (int)nscale : 0;
    lay.par = o; o += lay.npar_lds;
    lay.scale = o; o += lay.nscale_lds;
    lay.end = o;
    return lay;
}

// doubles of an ensemble's carry record between two launches: temper_run_kernel keeps the draws of a window and its
// counters, temper_kernel its whole LDS record up to the random-number buffers
static size_t temper_carry_doubles(const TemperDev &q, bool run_kernel)
{
    if (run_kernel) return (size_t)64 * (2 * q.p + 3) + 2 * (size_t)q.k * q.maxnn;
    return (size_t)temper_lds_layout(q).ubuf;
}

// one launch of the chosen kernel over q.it_begin .. q.it_end
static int launch_temper_once(const TemperDev &q, bool run_kernel, hipStream_t s)
{
    const int p = q.p, maxnn = q.maxnn;
    if (run_kernel) {
#define RUN_CASE(PP) \
        case PP: return q.kind == MCMCHIP_T_WITCH ? launch_temper_run<PP, MCMCHIP_T_WITCH>(q, maxnn, s) \
                                                  : launch_temper_run<PP, MCMCHIP_T_TEMPERED_NORMMIX>(q, maxnn, s);
        switch (p) { RUN_CASE(1) RUN_CASE(2) RUN_CASE(3) RUN_CASE(4) }
#undef RUN_CASE
        return set_error(MCMCHIP_EINVAL, "run-parallel tempering path: bad dimension %d", p);
    }
    const TemperLds lay = temper_lds_layout(q);
    size_t bytes = (size_t)lay.end * 8;
    if (bytes > 64 * 1024) return set_error(MCMCHIP_EINVAL, "tempering state (%zu bytes) does not fit the LDS budget", bytes);
    hipLaunchKernelGGL(temper_kernel, dim3((unsigned)q.nens), dim3(64), bytes, s, q, lay, maxnn);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

int launch_temper(const TemperDev &q0, hipStream_t s)
{
    const TemperDev &q = q0;
    if (q.p > TEMPER_MAXP) return set_error(MCMCHIP_EINVAL, "tempering path supports state dimension <= %d", TEMPER_MAXP);
    if (q.k > 4096) return set_error(MCMCHIP_EINVAL, "too many components");
    // maximum degree of the neighbour graph, counted by mcmchip_temper on the caller's host array (a TRUE diagonal
    // is legal, temper.c:38-41 only asks for symmetry, so a row may hold up to k neighbours)
    const int k = q.k, p = q.p;
    const int maxnn = q.maxnn;
    if (maxnn < 1 || maxnn > k) return set_error(MCMCHIP_EINVAL, "bad neighbour degree %d", maxnn);
    // run-parallel kernel for the common parallel-tempering shape (faster than temper_kernel from k = 4 up, measured);
    // everything else takes temper_kernel
    const bool run_ok = q.parallel && !q.debug && k <= 64 && p >= 1 && p <= 4 && maxnn <= 8 &&
                        (q.out_kind == MCMCHIP_TOUT_IDENTITY || q.out_kind == MCMCHIP_TOUT_WITCH_CORNER) &&
                        (q.kind == MCMCHIP_T_WITCH || q.kind == MCMCHIP_T_TEMPERED_NORMMIX);
    if (q.path == MCMCHIP_PATH_RUN && !run_ok)
        return set_error(MCMCHIP_EINVAL, "run-parallel tempering path needs parallel = 1, debug = 0, k <= 64, p <= 4, "
                                         "at most 8 neighbours per component, a closed-form descriptor and the identity or corner output");
    const bool run_kernel = run_ok && q.path != MCMCHIP_PATH_LDS;
    // The run as a sequence of launches (temper.c:848 polls for an interrupt every iteration; here the host does between
    // launches): a short calibration launch, then about a quarter of a second each, cut at multiples of 64 iterations.
    const long niter = (long)q.nbatch * q.blen * q.nspac;
    TemperDev qq = q;
    DevBuf carry;
    long done = 0, last_iters = 0;
    double last_ms = 0.0;
    while (done < niter) {
        const long n = next_launch_iters(niter - done, last_iters, last_ms, 64);
        qq.it_begin = done;
        qq.it_end = done + n;
        qq.resume = done > 0;
        if (qq.it_end < niter && !carry.p) {
            HIP_TRY(carry.alloc((size_t)q.nens * temper_carry_doubles(q, run_kernel) * 8));
            qq.carry = carry.as<double>();
        }
        timing_begin(s);
        int rc = launch_temper_once(qq, run_kernel, s);
        timing_end(s);
        if (rc) return rc;
        done += n;
        if (done < niter) {
            last_ms = timing_last_ms();          // waits for the launch: the host poll point
            last_iters = n;
            if (int irc = poll_interrupt()) return irc;
        }
    }
    if (carry.p) HIP_TRY(hipStreamSynchronize(s));      // the carry buffer is released on return
    return MCMCHIP_OK;
}
