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
#include "kernels.h"
#include "philox.h"
#include "targets.h"

struct StreamLayout {
    int x_in_lds;        // X and y staged in LDS
    int per_chain;       // doubles of LDS per chain: x, y, z, buf
    int tab_in_lds;      // logistic posterior: the softplus table (softplus.h, 130 KB) sits at the start of LDS
    size_t lds_bytes;
};

// The G lanes that own a chain sit in one wavefront and LDS operations of a wavefront complete in order, so the
// chain's private LDS buffers need no workgroup barrier inside the iteration loop: a wavefront-scope fence keeps
// the compiler from moving LDS accesses across the hand-over points.
#define CHAIN_SYNC()                                             \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)

// KSEL >= 0 fixes the descriptor kind at compile time (the logistic posterior gets its own instantiation: with every
// kind's density inlined behind one switch the register allocation of the generic kernel spills scalars and vectors
// inside the iteration loop); KSEL = -1 is the generic kernel.
template <int G, int KSEL>
__global__ __launch_bounds__(256) void metrop_stream_kernel(MetropDev p, StreamLayout lay)
{
    extern __shared__ __attribute__((aligned(32))) double lds[];
    constexpr int CPB = 256 / G;
    const int d = p.d, nout = p.nout;
    const int slot = threadIdx.x / G, g = threadIdx.x % G;
    long chain = (long)blockIdx.x * CPB + slot;
    const bool live = chain < p.nchain;
    if (!live) chain = p.nchain - 1;   // keep the lanes in lockstep (barriers), suppress the stores

    TargetView tv;
    tv.kind = KSEL >= 0 ? KSEL : p.kind; tv.d = d; tv.n = p.n; tv.par = p.par; tv.prior_c = p.prior_c;
    tv.morph_flags = KSEL >= 0 ? 0 : p.morph_flags; tv.morph_b = p.morph_b; tv.morph_r = p.morph_r; tv.morph_p = p.morph_p;
    tv.morph_center = p.morph_center;
    double *base = lds;
    if (lay.tab_in_lds) {              // 32-byte aligned at offset 0; published by the barrier below
        for (int i = threadIdx.x; i < SP_TAB_DOUBLES; i += 256) lds[i] = p.sp_tables->e[i];
        tv.sp_shift = softplus_lds_shift(lds);
        base = lds + SP_TAB_DOUBLES;
    }
    if (lay.x_in_lds) {
        double *xs = base, *ys = base + p.n * d;
        for (long i = threadIdx.x; i < p.n * d; i += 256) xs[i] = p.x[i];
        for (long i = threadIdx.x; i < p.n; i += 256) ys[i] = p.y[i];
        tv.x = xs; tv.y = ys;
        base = ys + p.n;
    } else {
        tv.x = p.x; tv.y = p.y;
    }
    double *X = base + (size_t)slot * lay.per_chain;  // current state
    double *Y = X + d;                                 // proposal
    double *Z = Y + d;                                 // normals
    double *B = Z + d;                                 // batch accumulator
    double *W = B + nout;                              // morph only: inverse(state), what lud and outfun see
    for (int j = g; j < d; j += G) X[j] = p.initial[(size_t)chain * d + j];
    __syncthreads();

    double cur = lud_morphed_group<G>(tv, X, W, g);
    if (live && g == 0) {
        if (cur == -INFINITY) atomicOr(p.flag, FLAG_NEGINF_INIT);
        else if (cur == INFINITY) atomicOr(p.flag, FLAG_POSINF);
        else if (isnan(cur)) atomicOr(p.flag, FLAG_NAN);
    }

    const long niter = (long)p.nbatch * p.blen * p.nspac;
    const unsigned long long gchain = p.chain_offset + (unsigned long long)chain;
    double acc = 0.0, tries = 0.0;
    long l = 0;
    for (int ib = 0; ib < p.nbatch; ib++) {
        double acc_b = 0.0, tries_b = 0.0;
        for (int k = g; k < nout; k += G) B[k] = 0.0;
        for (int jb = 0; jb < p.blen; jb++) {
            for (int is = 0; is < p.nspac; is++, l++) {
                // ---- d normals, always (metrop.c:377-378)
                if (p.rng_mode == MCMCHIP_RNG_INJECTED) {
                    const double *zp = p.z + ((size_t)chain * niter + l) * d;
                    for (int j = g; j < d; j += G) Z[j] = zp[j];
                } else {
                    for (int j2 = g; 2 * j2 < d; j2 += G) {
                        double za, zb;
                        philox_n2(p.seed, PHILOX_TAG_METROP, gchain, p.iter_offset + l, (uint32_t)j2, za, zb);
                        Z[2 * j2] = za;
                        if (2 * j2 + 1 < d) Z[2 * j2 + 1] = zb;
                    }
                }
                if (p.scale_kind == MCMCHIP_SCALE_FULL) CHAIN_SYNC();
                // ---- proposal (metrop.c:380-403)
                if (p.scale_kind == MCMCHIP_SCALE_CONSTANT) {
                    const double s0 = p.scale[0];
                    for (int j = g; j < d; j += G) Y[j] = add_nofma(X[j], mul_nofma(s0, Z[j]));
                } else if (p.scale_kind == MCMCHIP_SCALE_DIAGONAL) {
                    for (int j = g; j < d; j += G) Y[j] = add_nofma(X[j], mul_nofma(p.scale[j], Z[j]));
                } else {
                    for (int j = g; j < d; j += G) {
                        double a = X[j];
                        for (int i = 0; i < d; i++) a = add_nofma(a, mul_nofma(p.scale[j + (size_t)d * i], Z[i]));
                        Y[j] = a;
                    }
                }
                CHAIN_SYNC();
                // ---- log density of the proposal (logh(), metrop.c:223)
                const double ly = lud_morphed_group<G>(tv, Y, W, g);
                if (ly == INFINITY || isnan(ly)) {
                    if (live && g == 0) atomicOr(p.flag, isnan(ly) ? FLAG_NAN : FLAG_POSINF);
                }
                // ---- accept rule (metrop.c:225-235)
                bool accept = false;
                double u = NAN;
                if (ly != -INFINITY && !isnan(ly)) {
                    if (ly > cur) {
                        accept = true;
                    } else {
                        const double green = exp(ly - cur);
                        if (p.rng_mode == MCMCHIP_RNG_INJECTED) {
                            u = p.u[(size_t)chain * niter + l];
                        } else {
                            double ub;
                            philox_u2(p.seed, PHILOX_TAG_METROP, gchain, p.iter_offset + l, PHILOX_BLOCK_UNIF, u, ub);
                        }
                        accept = u < green;
                    }
                }
                if (p.debug && live) {
                    const size_t o = ((size_t)chain * niter + l) * d;
                    for (int j = g; j < d; j += G) {
                        p.dbg_current[o + j] = X[j];
                        p.dbg_proposal[o + j] = Y[j];
                        p.dbg_z[o + j] = Z[j];
                    }
                    if (g == 0) {
                        p.dbg_lg[(size_t)chain * niter + l] = ly - cur;
                        p.dbg_u[(size_t)chain * niter + l] = u;
                        p.dbg_accept[(size_t)chain * niter + l] = accept;
                    }
                }
                if (accept) {
                    for (int j = g; j < d; j += G) X[j] = Y[j];
                    cur = ly;
                    acc += 1.0; acc_b += 1.0;
                }
                tries += 1.0; tries_b += 1.0;
                CHAIN_SYNC();
            }
            // ---- outfun + batch accumulation, once per blen step (metrop.c:273-275)
            const double *S = X;
            if (KSEL < 0 && p.morph_flags) {            // morph$outfun: g(inverse(state)), morph.R:95-106
                morph_inverse_group<G>(tv, X, W, g);
                S = W;
            }
            for (int k = g; k < nout; k += G) {
                double v = S[p.out_map[k]];
                if (k >= p.sq_from) v = mul_nofma(v, v);
                B[k] = add_nofma(B[k], v);
            }
        }
        if (live) {
            for (int k = g; k < nout; k += G)
                p.batch[((size_t)chain * p.nbatch + ib) * nout + k] = B[k] / p.blen;   // metrop.c:279-280
            if (g == 0) p.accept_batch[(size_t)chain * p.nbatch + ib] = acc_b / tries_b; // metrop.c:282-283
        }
    }
    if (live) {
        if (g == 0) p.accept[chain] = acc / tries;
        for (int j = g; j < d; j += G) p.final_[(size_t)chain * d + j] = X[j];
    }
}

static int pick_group(long work, long nchain)
{
    long want = 65536 / (nchain > 0 ? nchain : 1);   // aim at >= 64k lanes in flight
    if (want > work) want = work;
    if (want >= 64) return 64;
    if (want >= 16) return 16;
    if (want >= 4) return 4;
    return 1;
}

int launch_metrop_stream(const MetropDev &p, hipStream_t s)
{
    const int d = p.d;
    long work = d;
    if (p.kind == MCMCHIP_T_LOGISTIC || p.kind == MCMCHIP_T_LOGISTIC_NAIVE || p.kind == MCMCHIP_T_BINOMIAL_LOGIT) work = p.n;
    else if (p.kind == MCMCHIP_T_NORMMIX) work = 8;
    else if (p.morph_flags && d <= 16) work = 1;   // small morphed targets: one lane per chain keeps the norm's summation order
    int G = pick_group(work, p.nchain);
    StreamLayout lay;
    lay.per_chain = 3 * d + p.nout + (p.morph_flags ? d : 0);
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
    const bool special = p.kind == MCMCHIP_T_LOGISTIC && !p.morph_flags;      // the specialised instantiation
#define STREAM_LAUNCH(GG) \
    do { \
        auto kern = special ? metrop_stream_kernel<GG, MCMCHIP_T_LOGISTIC> : metrop_stream_kernel<GG, -1>; \
        if (lay.lds_bytes > 48 * 1024) \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lay.lds_bytes)); \
        hipLaunchKernelGGL(kern, grid, block, lay.lds_bytes, s, p, lay); \
    } while (0)
    switch (G) {
    case 1: STREAM_LAUNCH(1); break;
    case 4: STREAM_LAUNCH(4); break;
    case 16: STREAM_LAUNCH(16); break;
    default: STREAM_LAUNCH(64); break;
    }
#undef STREAM_LAUNCH
    HIP_TRY(hipGetLastError());
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
