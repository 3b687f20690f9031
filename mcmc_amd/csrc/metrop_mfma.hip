// metrop_mfma.hip -- batched random-walk Metropolis, fp64 MFMA path (large n, many chains).
//
// What it replaces: the same loop nest as metrop_stream.hip (metrop.c:199-285), but here the
// log density of the logistic posterior (vignettes/demo.Rnw:59-65) dominates: for C lockstep
// chains eta = X (n x d) . Theta (d x C) is a genuine dense fp64 contraction, 2*n*d flop per
// chain-iteration, followed by n softplus evaluations per chain.
//
// Per Metropolis iteration two kernels run on the library stream:
//   logdens_mfma_kernel  grid (ceil(C / 256), observation splits): a workgroup of MFMA_NW = 8 wavefronts (two per SIMD) owns 256
//       chains; wave w owns MFMA_NC = 2 tiles of 16 chains whose Theta fragments stay in VGPRs for the whole launch
//       (loop invariant).  Every wave reads the X fragments of its share of the observation tiles straight from
//       global memory into registers, one tile ahead (16-byte loads; all waves of a CU and all CUs of an XCD read the
//       same tiles at about the same time, so X comes from HBM once per XCD and from the vector cache / L2
//       otherwise): no LDS staging, no barrier in the loop.  Each 16-observation tile feeds 16 * NC
//       v_mfma_f64_16x16x4_f64 per wave.  The accumulator tile (4 observations x 1 chain per lane) goes straight
//       through the softplus epilogue (softplus.h: 9 VALU instructions per value, a degree-5 polynomial in |eta| for
//       |eta|/2 + log1p(exp(-|eta|)) on the cell of the nearest of 2035 grid points, table in LDS, 130 KB) and into a per-lane
//       running sum: eta never leaves registers.  The epilogue of tile t runs as four clusters of vector
//       instructions between the four MFMA groups of tile t+1 (TilePlan below: why clusters and not an even
//       interleave).
//   metrop_update_kernel one lane per chain: fixed-order sum of the per-unit partials, prior term, accept rule
//       (metrop.c:225-235), state/batch-means bookkeeping, Philox normals and the next proposal.
//
// Reproducibility: the observations are cut into units (at most 256, a function of the data alone: make_plan); the
// density kernel writes one partial sum per (unit, chain), accumulated in a fixed order, and the update kernel adds the
// units in order.  A log density therefore has the same bits whatever the number of chains of the call, the observation
// split of the launch, or the number of calls / GPUs the chains are sharded over
// (tests/test_gpu_round2.py::test_mfma_logdens_bits_do_not_depend_on_the_call_shape).  What can still differ between a
// sharded and an unsharded run is the PATH: the AUTO rule between the stream and MFMA kernels looks at the chain
// count, and the two kernels sum in different orders; a sharded run that must reproduce an unsharded one pins `path`.
//
// HBM layout: X is stored once per target as sign-folded fragment-major tiles
//   xt[tile][kstep/2][lane][kstep%2] = s_row * X[16*tile + (lane&15)][4*kstep + (lane>>4)],  s = +1 (y==1) / -1 (y==0)
// so that min(s*eta,0) - L(|eta|) needs no label in the loop and the A fragments of two k-steps are one coalesced
// 16-byte-per-lane load (1 KiB a wave).  State, proposal and batch accumulators are dimension-major
// ([d][C], "SoA") so that both the per-chain update kernel and the B-fragment loads coalesce.
//
// Roofline: fp64 MFMA.  Algorithmic flops per launch = 2 * n * d * C.
#include "kernels.h"
#include "philox.h"
#include "targets.h"
#include <type_traits>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define MFMA_NW 8          // wavefronts per workgroup (2 per SIMD: one covers the other's waits on loads and table reads)
#define MFMA_NC 2          // 16-chain tiles per wavefront -> 256 chains per workgroup
// granularity of the tile count (build_mfma_tiles pads n to it with zero rows) and of an observation split
static inline int mfma_tc(int ks) { return ks <= 2 ? 16 : (ks <= 4 ? 8 : (ks <= 8 ? 4 : (ks <= 16 ? 2 : 1))); }

// ---------------------------------------------------------------------------------------------
__global__ void tile_x_kernel(const double *__restrict__ x, const double *__restrict__ y, long n, int d,
                              int ksteps, long ntiles, double *__restrict__ xt)
{
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = ntiles * ksteps * 64;
    if (idx >= total) return;
    const int j = (int)(idx & 1);
    int lane = (int)((idx >> 1) & 63);
    long tk = idx >> 7;                    // (tile, k-step pair)
    int ks = 2 * (int)(tk % (ksteps / 2)) + j;
    long t = tk / (ksteps / 2);
    long row = t * 16 + (lane & 15);
    int col = 4 * ks + (lane >> 4);
    double v = 0.0;
    if (row < n && col < d) {
        v = x[(size_t)row * d + col];
        if (y && y[row] != 1.0) v = -v;
    }
    xt[idx] = v;
}

// column sums of the sign-folded X from the tiles: one workgroup per (kstep, k) column, fixed-order reduction
__global__ __launch_bounds__(256) void tile_colsum_kernel(const double *__restrict__ xt, int ksteps, long ntiles,
                                                          double *__restrict__ colsum)
{
    __shared__ double red[256];
    const int col = blockIdx.x;                 // 0 .. dpad-1
    const int ks = col >> 2, kk = col & 3;
    double s = 0.0;
    // element (tile t, row r) of column col sits at xt[((t*ksteps/2 + ks/2)*64 + kk*16 + r)*2 + ks%2]
    for (long i = threadIdx.x; i < ntiles * 16; i += 256) {
        const long t = i >> 4;
        const int r = (int)(i & 15);
        s += xt[(((size_t)t * (ksteps / 2) + (ks >> 1)) * 64 + kk * 16 + r) * 2 + (ks & 1)];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) colsum[col] = red[0];
}

// MCMCHIP_T_BINOMIAL_LOGIT (vignettes/morph.Rnw:361-365): sum(y * eta) - sum(w * log(1 + exp(eta))), w = par.  With
// log(1 + e^eta) = eta/2 + M(eta), M(eta) = |eta|/2 + L(|eta|) (even in eta: no sign fold of X), the log density is
//     sum_i (y_i - w_i/2) eta_i - sum_i w_i M(eta_i):
// the first sum is linear in theta (colsum holds sum_i (2 y_i - w_i) X_ij, so that the update kernel's lin/2 is it), the
// second is what the density kernel accumulates with the row weights as one more operand of the running sum's fma.
// wt[tile][g][r] = w[16 tile + g + 4 r] (0 past n): the four rows lane group g = lane >> 4 holds in its accumulator
// registers r = 0..3 are 32 contiguous bytes.
__global__ void tile_w_kernel(const double *__restrict__ w, long n, long ntiles, double *__restrict__ wt)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ntiles * 16) return;
    const long t = idx >> 4;
    const int g = (int)((idx >> 2) & 3), r = (int)(idx & 3);
    const long row = t * 16 + g + 4 * r;
    wt[idx] = row < n ? w[row] : 0.0;
}

// one workgroup per column j: colsum[j] = sum_i (2 y_i - w_i) x[i][j], fixed-order reduction; padded columns 0
__global__ __launch_bounds__(256) void weighted_colsum_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                                               const double *__restrict__ w, long n, int d,
                                                               double *__restrict__ colsum)
{
    __shared__ double red[256];
    const int col = blockIdx.x;
    double s = 0.0;
    if (col < d)
        for (long i = threadIdx.x; i < n; i += 256) s = fma(2.0 * y[i] - w[i], x[(size_t)i * d + col], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) colsum[col] = red[0];
}

int build_mfma_tiles(mcmchip_target *t, hipStream_t s)
{
    const bool weighted = t->kind == MCMCHIP_T_BINOMIAL_LOGIT;
    // d > 64: k-steps in {20, 24, 28, 32}; d > 128: {48, 64} (whole turns of the kernel's 16-k-step fragment ring)
    t->dpad = t->d <= 64 ? ((t->d + 7) / 8) * 8 : (t->d <= 128 ? ((t->d + 15) / 16) * 16 : ((t->d + 63) / 64) * 64);
    if (t->dpad < 8) t->dpad = 8;
    if (weighted && t->dpad == 128) t->dpad = 192;   // the weighted kernel has no registers to spare at 32 k-steps x 2 chain tiles
    t->ksteps = t->dpad / 4;
    long tiles = (t->n + 15) / 16;
    const int tc = mfma_tc(t->ksteps);
    tiles = ((tiles + tc - 1) / tc) * tc;
    t->ntiles = tiles;
    t->npad_rows = tiles * 16 - t->n;
    size_t bytes = (size_t)tiles * t->ksteps * 64 * sizeof(double);
    HIP_TRY(hipMalloc((void **)&t->xt, bytes));
    long total = tiles * t->ksteps * 64;
    hipLaunchKernelGGL(tile_x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, t->x,
                       weighted ? (const double *)nullptr : t->y, (long)t->n, t->d, t->ksteps, tiles, t->xt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMalloc((void **)&t->colsum, (size_t)t->dpad * sizeof(double)));
    if (weighted) {
        HIP_TRY(hipMalloc((void **)&t->wt, (size_t)tiles * 16 * sizeof(double)));
        hipLaunchKernelGGL(tile_w_kernel, dim3((unsigned)((tiles * 16 + 255) / 256)), dim3(256), 0, s, t->par, (long)t->n, tiles, t->wt);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(weighted_colsum_kernel, dim3((unsigned)t->dpad), dim3(256), 0, s, t->x, t->y, t->par, (long)t->n, t->d, t->colsum);
        HIP_TRY(hipGetLastError());
        t->npad_rows = 0;            // rows of weight 0 contribute nothing
        return MCMCHIP_OK;
    }
    hipLaunchKernelGGL(tile_colsum_kernel, dim3((unsigned)t->dpad), dim3(256), 0, s, t->xt, t->ksteps, tiles, t->colsum);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Schedule of one tile (measured with tools/fp64_sustain_probe, profiles/r2_fp64_sustain_probe.txt): the fp64 MFMA
// and the fp64 VALU share the SIMD's pipe, a lone wave keeps it full (64 cycles per MFMA back to back), vector
// instructions ADD their 4.2 cycles each, and every switch between the two kinds costs about 6 cycles more.  The
// first version of this kernel spread the softplus of the previous tile evenly between the MFMAs (one MFMA, 2-3 VALU):
// 32 switches a tile.  Here a tile is MFMA groups of GK k-steps (GK * NC instructions) separated by VALU clusters:
//     cluster q (after group q-1): evaluate the values whose table records were requested one cluster ago,
//                                  form the table addresses of the next values (softplus.h: 6 + 3 instructions a value)
//     then (behind the first k-step of the next group) issue the LDS reads those addresses need: they have the rest of
//     the group (>= 256 cycles) to land; behind each group its X registers are refilled with the next tile's fragments.
// The values of the last cluster have nothing behind them to hide their table reads: they stay pending (their eta in
// the accumulator registers, which the tile after next only overwrites with its first MFMA; their records in `pend`)
// and are evaluated in cluster 0 of the next tile.
template <int KS, int NC> struct TilePlan {
    static constexpr int GK = KS >= 16 ? 4 : (KS >= 4 ? KS / 4 : 1);  // k-steps per MFMA group (>= 4 groups: <= 2 values a cluster)
    static constexpr int NG = (KS + GK - 1) / GK;                     // groups = clusters per tile
    static constexpr int NV = 4 * NC;                                 // softplus values per lane and tile
    static constexpr int ad(int v) { return v * NG / NV; }            // cluster that forms the address of value v
    static constexpr int npend()
    {
        int n = 0;
        for (int v = 0; v < NV; v++) n += ad(v) == NG - 1;
        return n;
    }
};

template <int KS, int NC, int NW, int TC, bool W = false>
__global__ __launch_bounds__(NW * 64) void logdens_mfma_kernel(const double *__restrict__ xt,
                                                               const double *__restrict__ prop, long ldc,
                                                               double *__restrict__ partial,
                                                               int tiles_per_unit, int units_per_split, int nunits,
                                                               long ntiles, const SoftplusTables *__restrict__ tab,
                                                               const double *__restrict__ wt)
{
    using P = TilePlan<KS, NC>;
    constexpr int GK = P::GK, NG = P::NG, NV = P::NV, NP = P::npend();
    // X fragment registers: up to d = 128 one per k-step (the whole next tile is in flight behind this one); beyond, a
    // ring of 16 k-steps -- the Theta fragments alone take 128 registers at d = 256 -- refilled 4 groups ahead of use
    constexpr int AR = KS > 32 ? 16 : KS, LEAD = KS > 32 ? AR / GK : NG;
    static_assert(KS <= 32 || (AR % GK == 0 && KS % GK == 0 && KS % AR == 0), "ring of whole groups");
    // LDS holds the softplus table only.  The X fragments go from global memory straight into registers: the waves
    // of a workgroup (and all workgroups of an XCD) read the same tiles at about the same time, so one of them pays
    // the HBM / L2 miss and the rest hit the CU's vector cache or the XCD's L2; no staging, no barrier in the loop.
    __shared__ __attribute__((aligned(32))) double sTab[SP_TAB_DOUBLES];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < SP_TAB_DOUBLES; i += NW * 64) sTab[i] = tab->e[i];

    // loop-invariant B fragments: Theta[k = 4ks + (lane>>4)][chain = chain0 + 16c + (lane&15)]
    const long chain0 = ((long)blockIdx.x * NW + wave) * 16 * NC;
    double b[KS][NC];
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int c = 0; c < NC; c++)
            b[ks][c] = prop[(size_t)(4 * ks + (lane >> 4)) * ldc + chain0 + 16 * c + (lane & 15)];

    // The observations are cut into UNITS of tiles_per_unit tiles, a size that depends on the data alone (make_plan).
    // One partial sum is written per (unit, chain), always accumulated in the same order, and the update kernel adds
    // the units in order: a log density has the same bits whatever the number of chains of the call, the observation
    // split of the launch or the number of GPUs the chains are sharded over.  A workgroup owns whole units.
    const int u_begin = (int)blockIdx.y * units_per_split;
    const int u_end = u_begin + units_per_split < nunits ? u_begin + units_per_split : nunits;

    const double sp_shift = softplus_lds_shift(sTab);
    double sum[NC];               // sum of M(eta) = |eta|/2 + L(|eta|)

    __syncthreads();   // publishes the table

    // min(eta,0) - L(|eta|) = eta/2 - M(eta): sum(eta) is linear in Theta (colsum . theta, added by the update
    // kernel), so only M(eta) = |eta|/2 + L(|eta|) is accumulated here
    double4_t acc2[2][NC];        // ping-pong accumulators: tile tt uses acc2[tt & 1]
    SoftplusRec pend[NP > 0 ? NP : 1];   // table records of the pending values (an all-zero record evaluates to 0)

    // a[]: this lane's X fragments of the tile about to be multiplied; tile_block refills each group's registers with
    // the NEXT tile's fragments (nx: this lane's column of it in global memory) as soon as the group has been issued,
    // so a load has NG-1 groups (about 1.6 us at d = 64) to arrive.  accN: accumulators of this tile (on entry they
    // still hold the tile before last, whose pending values cluster 0 evaluates).  accO: the finished previous tile.
    double a[AR];
    // k-steps [k0, k1) of a tile (colp: the tile's first double, uniform; each lane adds its 16-byte slot last, so the
    // address is scalar base + lane offset): 16-byte loads of the pairs the range covers whole, 8-byte loads at odd ends
    const int lane2 = lane * 2;
    auto refill = [&](const double *__restrict__ colp, int k0, int k1) {
#pragma unroll
        for (int kp = k0 >> 1; 2 * kp < k1; kp++) {
            const bool lo = 2 * kp >= k0, hi = 2 * kp + 1 < k1;
            const double *q = colp + kp * 128;
            if (lo && hi) {
                const double2 p = *(const double2 *)(q + lane2);
                a[(2 * kp) % AR] = p.x;
                a[(2 * kp + 1) % AR] = p.y;
            } else if (lo) {
                a[(2 * kp) % AR] = q[lane2];
            } else if (hi) {
                a[(2 * kp + 1) % AR] = q[lane2 + 1];
            }
        }
    };
    // W: the row weights of tile tt sit in w2[tt & 1], like its accumulators (register r <-> row (lane >> 4) + 4 r).  A
    // tile's weights are loaded during its own block -- behind cluster 0, which still needs the array for the pending
    // values of the tile before last -- and used in the next block and in cluster 0 of the one after.
    double w2[2][4];
    const int wlane = (lane >> 4) * 4;
    auto load_w = [&](const double *__restrict__ wp, double *w) {
        const double2 lo = *(const double2 *)(wp + wlane), hi = *(const double2 *)(wp + wlane + 2);
        w[0] = lo.x; w[1] = lo.y; w[2] = hi.x; w[3] = hi.y;
    };
    auto add_w = [&](double &acc_sum, double wv, double m) {
        if constexpr (W) acc_sum = fma(wv, m, acc_sum);
        else acc_sum += m;
    };
    // cx / nx: this tile and the next one in global memory (the ring's refills reach into both); wx: this tile's weights
    auto tile_block = [&](const double *__restrict__ cx, const double *__restrict__ nx, double4_t *accN, const double4_t *accO, auto have_old,
                          const double *__restrict__ wx, double *wN, const double *wO) {
        constexpr bool OLD = decltype(have_old)::value;
        uint32_t addr[NV];
        SoftplusRec rec[NV];
        // the compiler's s_waitcnt before an MFMA group waits for EVERY outstanding LDS read, so nothing may be issued
        // between a cluster and the group behind it: the table reads go behind the first k-step of the group
        auto mfma_ks = [&](int ks) {
#pragma unroll
            for (int c = 0; c < NC; c++)
                accN[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks % AR], b[ks][c], ks == 0 ? (double4_t){0.0, 0.0, 0.0, 0.0} : accN[c], 0, 0, 0);
        };
#pragma unroll
        for (int g = 0; g < NG; g++) {
            // cluster g: evaluate the values addressed one cluster ago (cluster 0: the pending ones of the tile before
            // last, still in accN), address the next ones
            if (OLD) {
                int ip = 0;
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    if (g == 0 && P::ad(v) == NG - 1) add_w(sum[v / 4], wN[v % 4], softplus_lds_eval(accN[v / 4][v % 4], pend[ip++]));
                    if (g > 0 && P::ad(v) == g - 1) add_w(sum[v / 4], wO[v % 4], softplus_lds_eval(accO[v / 4][v % 4], rec[v]));
                }
#pragma unroll
                for (int v = 0; v < NV; v++)
                    if (P::ad(v) == g) addr[v] = softplus_lds_addr(accO[v / 4][v % 4], sp_shift);
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_ks(g * GK);
            __builtin_amdgcn_sched_barrier(0);
            if (OLD) {
                int ip = 0;
#pragma unroll
                for (int v = 0; v < NV; v++)
                    if (P::ad(v) == g) {
                        if (g == NG - 1) softplus_lds_load(addr[v], pend[ip++]);
                        else softplus_lds_load(addr[v], rec[v]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = g * GK + 1; ks < (g + 1) * GK && ks < KS; ks++) mfma_ks(ks);
            __builtin_amdgcn_sched_barrier(0);
            {   // the registers of group g are free: they take group g + LEAD (of this tile, or of the next one)
                constexpr int NGc = NG;
                const int gl = (g + LEAD) % NGc;
                refill(g + LEAD < NGc ? cx : nx, gl * GK, (gl + 1) * GK < KS ? (gl + 1) * GK : KS);
                if constexpr (W)
                    if (g == 0) load_w(wx, wN);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // after the last tile (accumulators accL; accP still holds the tile before it with its pending values)
    auto drain = [&](const double4_t *accL, const double4_t *accP, const double *wL, const double *wP) {
        int ip = 0;
#pragma unroll
        for (int v = 0; v < NV; v++)
            if (P::ad(v) == NG - 1) add_w(sum[v / 4], wP[v % 4], softplus_lds_eval(accP[v / 4][v % 4], pend[ip++]));
#pragma unroll
        for (int v = 0; v < NV; v++) add_w(sum[v / 4], wL[v % 4], softplus_half_abs_lds(accL[v / 4][v % 4], sp_shift));
    };
    const std::true_type with_old;
    const std::false_type first_tile;

    for (int u = u_begin; u < u_end; u++) {
        // a unit starts the pipeline afresh (first tile without a previous one, drain at the end): ~1000 cycles per
        // unit of >= 2 * 2600 * tiles_per_unit
#pragma unroll
        for (int c = 0; c < NC; c++) sum[c] = 0.0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int c = 0; c < NC; c++) acc2[i][c] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < NP; i++) pend[i] = SoftplusRec{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        // tiles t0 .. t0+nt-1: the first has no previous tile, then pairs (odd tile -> acc2[1], even -> acc2[0]).  The
        // tile after the very last is never needed: its loads are pointed at the last tile again (inside the allocation).
        const long t0 = (long)u * tiles_per_unit;
        const long left = ntiles - t0;
        const int nt = left < tiles_per_unit ? (int)left : tiles_per_unit;   // 32-bit tile counters: scalar compares
        const double *x0 = xt + (size_t)t0 * (KS * 64);
        // the loads behind the unit's last tile fetch the first tile of the next unit of this workgroup (contiguous in
        // memory), so only the first unit waits for its first fragments
        const int lim = u + 1 < u_end ? nt : nt - 1;
        auto col = [&](int i) { return x0 + (size_t)(i < lim ? i : lim) * (KS * 64); };
        auto wcol = [&](int i) { return W ? wt + (size_t)(t0 + (i < lim ? i : lim)) * 16 : wt; };
#pragma unroll
        for (int r = 0; r < 4; r++) w2[0][r] = w2[1][r] = 0.0;      // the zero pending records of a fresh unit meet finite weights
        if (u == u_begin) refill(col(0), 0, AR);
        tile_block(col(0), col(1), acc2[0], acc2[1], first_tile, wcol(0), w2[0], w2[1]);
        int i = 1;
        for (; i + 1 < nt; i += 2) {
            tile_block(col(i), col(i + 1), acc2[1], acc2[0], with_old, wcol(i), w2[1], w2[0]);
            tile_block(col(i + 1), col(i + 2), acc2[0], acc2[1], with_old, wcol(i + 1), w2[0], w2[1]);
        }
        if (i < nt) {
            tile_block(col(i), col(i + 1), acc2[1], acc2[0], with_old, wcol(i), w2[1], w2[0]);
            drain(acc2[1], acc2[0], w2[1], w2[0]);
        } else {
            drain(acc2[0], acc2[1], w2[0], w2[1]);
        }
        // rows of the accumulator live on lane>>4 (4 groups) x 4 registers: fold the 4 groups
#pragma unroll
        for (int c = 0; c < NC; c++) {
            double v = sum[c];                             // partial = sum(|eta|)/2 + sum(L); the update kernel negates
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) partial[(size_t)u * ldc + chain0 + 16 * c + lane] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
struct UpdParams {
    long nchain, ldc;
    int d, dpad;
    int nunits;             // partial sums per chain (one per unit of observations, added in order)
    double pad_term;        // npad_rows * ln 2 (each zero row contributes exactly -L(0))
    double prior_c;
    int nbatch, blen, nspac;
    int scale_kind;
    const double *scale;
    int nout, sq_from;
    const int *out_map;
    int rng_mode;
    const double *z, *u;
    unsigned long long seed, chain_offset, iter_offset;
    const double *partial;
    const double *colsum;   // [dpad] column sums of the sign-folded X
    double *state, *prop;   // [dpad][ldc]
    double *cur, *acc, *acc_b, *buf;  // [ldc], [ldc], [ldc], [nout][ldc]
    double *batch, *accept_batch, *accept, *final_;  // chain-major outputs
    double *logdens_out;    // when non-null: just emit the log density (mcmchip_logdens)
    // MCMCHIP_T_LOGISTIC_MODELS (log density only): the states as the caller gave them, (i, beta[d]) per chain, and the
    // descriptor's parameter block (k, k x d masks, k log pseudo-priors, constant); prop holds beta * mask_i
    const double *models_states, *models_par;
    // morph.metrop (R/morph.R:121-181): state / prop hold the morphed chain; theta = inverse(prop) is what the density
    // kernel contracts with X, theta_state = inverse(state) is what the output functional sees (morph.R:95-106), lj the
    // proposal's log.jacobian (morph.R:16-28).  No morph: theta == prop, theta_state == state, lj == nullptr.
    int morph_flags;
    double morph_b, morph_r, morph_p;
    const double *morph_center;
    double *theta, *theta_state, *lj;
    int *flag;
};

// theta[:, c] = inverse(prop[:, c]) and lj[c] = log.jacobian(prop[:, c]); one lane per chain, the sum of squares in the
// reference's order (morph.R:8-13)
__device__ __forceinline__ void morph_proposal(const UpdParams &q, long c)
{
    TargetView t;
    t.morph_flags = q.morph_flags; t.morph_b = q.morph_b; t.morph_r = q.morph_r; t.morph_p = q.morph_p;
    const int d = q.d;
    const long ldc = q.ldc;
    const bool shaped = q.morph_flags & (MCMCHIP_MORPH_EXPO | MCMCHIP_MORPH_SUBEXPO);
    double nrm = 0.0;
    if (shaped) {
        double ss = 0.0;
        for (int j = 0; j < d; j++) ss += q.prop[(size_t)j * ldc + c] * q.prop[(size_t)j * ldc + c];
        nrm = sqrt(ss);
    }
    const double fi = (shaped && nrm != 0) ? morph_finv(t, nrm) : 0.0;
    for (int j = 0; j < d; j++) {
        double v = q.prop[(size_t)j * ldc + c];
        if (shaped) v = (nrm == 0) ? 0.0 : fi * v / nrm;
        if (q.morph_flags & MCMCHIP_MORPH_CENTER) v += q.morph_center[j];
        q.theta[(size_t)j * ldc + c] = v;
    }
    double lj = 0.0;
    if (shaped) {
        if (nrm == 0) lj = d * log(morph_dfinv(t, nrm));                                       // morph.R:22-23
        else lj = log(morph_dfinv(t, nrm)) + (d - 1) * (log(morph_finv(t, nrm)) - log(nrm));   // morph.R:25
    }
    q.lj[c] = lj;
}

__global__ __launch_bounds__(256) void morph_init_kernel(UpdParams q)      // the initial state: theta, theta_state, lj
{
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= q.nchain) return;
    morph_proposal(q, c);
    if (q.theta_state != q.theta)
        for (int j = 0; j < q.d; j++) q.theta_state[(size_t)j * q.ldc + c] = q.theta[(size_t)j * q.ldc + c];
}

__device__ __forceinline__ void make_proposal(const UpdParams &q, long c, long l_next)
{
    const int d = q.d;
    const long ldc = q.ldc;
    const long niter = (long)q.nbatch * q.blen * q.nspac;
    if (q.scale_kind == MCMCHIP_SCALE_CONSTANT || q.scale_kind == MCMCHIP_SCALE_DIAGONAL) {
        // a pair of normals goes straight into its two coordinates (the same operations as below, without the detour of
        // d doubles through the lane's scratch memory)
        const bool inj = q.rng_mode == MCMCHIP_RNG_INJECTED, cst = q.scale_kind == MCMCHIP_SCALE_CONSTANT;
        const double *zp = inj ? q.z + ((size_t)c * niter + l_next) * d : nullptr;
        const double s0 = q.scale[0];
        for (int j = 0; j < d; j += 2) {
            double za, zb = 0.0;
            if (inj) {
                za = zp[j];
                if (j + 1 < d) zb = zp[j + 1];
            } else
                philox_n2(q.seed, PHILOX_TAG_METROP, q.chain_offset + (unsigned long long)c,
                          q.iter_offset + (unsigned long long)l_next, (uint32_t)(j / 2), za, zb);
            q.prop[(size_t)j * ldc + c] = add_nofma(q.state[(size_t)j * ldc + c], mul_nofma(cst ? s0 : q.scale[j], za));
            if (j + 1 < d)
                q.prop[(size_t)(j + 1) * ldc + c] = add_nofma(q.state[(size_t)(j + 1) * ldc + c], mul_nofma(cst ? s0 : q.scale[j + 1], zb));
        }
        if (q.morph_flags) morph_proposal(q, c);
        return;
    }
    double zl[MCMCHIP_MFMA_MAXD];
    if (q.rng_mode == MCMCHIP_RNG_INJECTED) {
        const double *zp = q.z + ((size_t)c * niter + l_next) * d;
        for (int j = 0; j < d; j++) zl[j] = zp[j];
    } else {
        for (int j = 0; j < d; j += 2) {
            double za, zb;
            philox_n2(q.seed, PHILOX_TAG_METROP, q.chain_offset + (unsigned long long)c,
                      q.iter_offset + (unsigned long long)l_next, (uint32_t)(j / 2), za, zb);
            zl[j] = za;
            if (j + 1 < d) zl[j + 1] = zb;
        }
    }
    if (q.scale_kind == MCMCHIP_SCALE_CONSTANT) {
        const double s0 = q.scale[0];
        for (int j = 0; j < d; j++)
            q.prop[(size_t)j * ldc + c] = add_nofma(q.state[(size_t)j * ldc + c], mul_nofma(s0, zl[j]));
    } else if (q.scale_kind == MCMCHIP_SCALE_DIAGONAL) {
        for (int j = 0; j < d; j++)
            q.prop[(size_t)j * ldc + c] = add_nofma(q.state[(size_t)j * ldc + c], mul_nofma(q.scale[j], zl[j]));
    } else {
        for (int j = 0; j < d; j++) {
            double a = q.state[(size_t)j * ldc + c];
            for (int i = 0; i < d; i++) a = add_nofma(a, mul_nofma(q.scale[j + (size_t)d * i], zl[i]));
            q.prop[(size_t)j * ldc + c] = a;
        }
    }
    if (q.morph_flags) morph_proposal(q, c);
}

// l == -1: initial state (prop holds a copy of the state): record cur, draw the first proposal
__global__ __launch_bounds__(256) void metrop_update_kernel(UpdParams q, long l)
{
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= q.nchain) return;
    const int d = q.d;
    const long ldc = q.ldc;
    // log density = sum_i [ (eta_i - |eta_i|)/2 - L(|eta_i|) ] - prior_c |theta|^2, with
    // sum_i eta_i = colsum . theta (eta is linear in theta) and the rest from the MFMA kernel's per-unit partials
    // (loads in batches of 32, the sums in the same order: with few chains the kernel is one long chain of dependent loads
    // per lane -- 256 chains, 254 units: 60 us one load at a time, more than the density kernel it follows)
    double part = 0.0;
#pragma unroll 32
    for (int s = 0; s < q.nunits; s++) part += q.partial[(size_t)s * ldc + c];
    double ss = 0.0, lin = 0.0;
#pragma unroll 8
    for (int j = 0; j < d; j++) {
        const double v = q.theta[(size_t)j * ldc + c];
        ss += v * v;
        lin = fma(v, q.colsum[j], lin);
    }
    double ly = (0.5 * lin - part) + q.pad_term;
    if (q.models_par) {         // tests/temp-par.R:39-56: the prior is over all of beta, masked or not
        const double *st = q.models_states + (size_t)c * (d + 1);
        const int k = (int)q.models_par[0];
        const double *lpp = q.models_par + 1 + (size_t)k * d;
        ss = 0.0;
        for (int j = 0; j < d; j++) ss += st[1 + j] * st[1 + j];
        ly = (ly - ss * q.prior_c) + lpp[k] + lpp[(int)st[0] - 1];
        q.logdens_out[c] = ly;
        return;
    }
    ly -= ss * q.prior_c;
    // morph.R:166-179: lud(inverse(y)) + log.jacobian(y); -Inf, NaN and +Inf of the density pass through
    if (q.lj && !(ly == -INFINITY || isnan(ly) || ly == INFINITY)) ly += q.lj[c];
    if (q.logdens_out) { q.logdens_out[c] = ly; return; }
    if (ly == INFINITY) atomicOr(q.flag, FLAG_POSINF);
    else if (isnan(ly)) atomicOr(q.flag, FLAG_NAN);
    const long niter = (long)q.nbatch * q.blen * q.nspac;
    if (l < 0) {
        if (ly == -INFINITY) atomicOr(q.flag, FLAG_NEGINF_INIT);
        q.cur[c] = ly;
        q.acc[c] = 0.0;
        q.acc_b[c] = 0.0;
        for (int k = 0; k < q.nout; k++) q.buf[(size_t)k * ldc + c] = 0.0;
        make_proposal(q, c, 0);
        return;
    }
    const double cur = q.cur[c];
    bool accept = false;
    if (ly != -INFINITY && !isnan(ly)) {       // metrop.c:225-235
        if (ly > cur) {
            accept = true;
        } else {
            const double green = exp(ly - cur);
            double u, ub;
            if (q.rng_mode == MCMCHIP_RNG_INJECTED) u = q.u[(size_t)c * niter + l];
            else philox_u2(q.seed, PHILOX_TAG_METROP, q.chain_offset + (unsigned long long)c,
                           q.iter_offset + (unsigned long long)l, PHILOX_BLOCK_UNIF, u, ub);
            accept = u < green;
        }
    }
    if (accept) {
        for (int j = 0; j < d; j++) q.state[(size_t)j * ldc + c] = q.prop[(size_t)j * ldc + c];
        if (q.theta_state != q.state)
            for (int j = 0; j < d; j++) q.theta_state[(size_t)j * ldc + c] = q.theta[(size_t)j * ldc + c];
        q.cur[c] = ly;
        q.acc[c] += 1.0;
        q.acc_b[c] += 1.0;
    }
    const int is = (int)(l % q.nspac);
    const long lb = l / q.nspac;
    const int jb = (int)(lb % q.blen);
    const long ib = lb / q.blen;
    if (is == q.nspac - 1) {                   // metrop.c:273-275
        // one pass: the batch sum is loaded once and either stored back or, at the end of a batch, divided and written
        // out (metrop.c:279-283) -- the same operations; as two loops over global memory (store, reload, divide, store)
        // the bookkeeping took 10 of the kernel's 26 us at 256 chains
        const bool batch_end = jb == q.blen - 1;
        double *bo = q.batch + ((size_t)c * q.nbatch + ib) * q.nout;
#pragma unroll 4
        for (int k = 0; k < q.nout; k++) {
            double v = q.theta_state[(size_t)q.out_map[k] * ldc + c];
            if (k >= q.sq_from) v = mul_nofma(v, v);
            const double sum = add_nofma(q.buf[(size_t)k * ldc + c], v);
            if (batch_end) bo[k] = sum / q.blen;
            q.buf[(size_t)k * ldc + c] = batch_end ? 0.0 : sum;
        }
        if (batch_end) {
            q.accept_batch[(size_t)c * q.nbatch + ib] = q.acc_b[c] / ((double)q.blen * q.nspac);
            q.acc_b[c] = 0.0;
        }
    }
    if (l + 1 < niter) {
        make_proposal(q, c, l + 1);
    } else {
        q.accept[c] = q.acc[c] / (double)niter;
        for (int j = 0; j < d; j++) q.final_[(size_t)c * d + j] = q.state[(size_t)j * ldc + c];
    }
}

// chain-major [C][d] -> dimension-major [dpad][ldc], zero padding (both state and prop)
__global__ void soa_init_kernel(const double *__restrict__ init, long nchain, int d, int dpad, long ldc,
                                double *__restrict__ state, double *__restrict__ prop)
{
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)dpad * ldc) return;
    long c = idx % ldc;
    int j = (int)(idx / ldc);
    double v = (c < nchain && j < d) ? init[(size_t)c * d + j] : 0.0;
    if (state) state[idx] = v;
    prop[idx] = v;
}

// MCMCHIP_T_LOGISTIC_MODELS: states (i, beta[d]) chain-major -> beta * mask_i dimension-major (eta = X (beta * mask_i))
__global__ void soa_init_models_kernel(const double *__restrict__ states, long nchain, int d, int dpad, long ldc,
                                       const double *__restrict__ par, double *__restrict__ prop)
{
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)dpad * ldc) return;
    long c = idx % ldc;
    int j = (int)(idx / ldc);
    double v = 0.0;
    if (c < nchain && j < d) {
        const double *st = states + (size_t)c * (d + 1);
        const int i = (int)st[0];
        if (par[1 + (size_t)(i - 1) * d + j] != 0.0) v = st[1 + j];
    }
    prop[idx] = v;
}

// ---------------------------------------------------------------------------------------------
struct MfmaPlan {
    int ks, nc;
    long ldc, chain_blocks;
    int tiles_per_unit, nunits;      // functions of the target alone
    int units_per_split, nsplit;     // how this call spreads the units over workgroups
};

static MfmaPlan make_plan(const mcmchip_target *t, long nchain, int num_cu)
{
    MfmaPlan pl;
    pl.ks = t->ksteps;
    pl.nc = pl.ks > 32 ? 1 : MFMA_NC;        // d > 128: one 16-chain tile per wave (its Theta fragments are 128 registers)
    const int tc = mfma_tc(pl.ks);
    long per_wg = 16L * pl.nc * MFMA_NW;
    pl.chain_blocks = (nchain + per_wg - 1) / per_wg;
    pl.ldc = pl.chain_blocks * per_wg;
    // units: at most 256 per target, at least two tile granules each (table load and B preload of a workgroup cost a
    // few us).  Nothing here may depend on the chain count: the unit is the grain of the summation order.
    long upt = (t->ntiles + 255) / 256;
    upt = ((upt + tc - 1) / tc) * tc;
    if (upt < 2L * tc) upt = 2L * tc;
    pl.tiles_per_unit = (int)upt;
    pl.nunits = (int)((t->ntiles + upt - 1) / upt);
    // one workgroup per CU (LDS-bound occupancy).  A grid that is a whole number of chip-fulls needs no
    // observation split; otherwise spread the units over workgroups until the tail round is small
    // (with 8 splits a 4096 x 5 posterior over 4096 chains ran on 32 workgroups).
    // (Tried: an 8-way split decoded so that each XCD streams its own eighth of X.  The HBM-side read count did not
    // move -- an XCD then re-streams its eighth once per round of 32 workgroups, 8 rounds -- and the launch got 0.7 %
    // slower; the L2s request X from the fabric 8 to 16 times per launch either way, under 1 % of the HBM rate.
    // Also tried: one round of workgroups, each walking its XCD's units for 8 chain blocks so that a unit of X (2 MB)
    // is re-read from the L2 -- Theta (33 MB at C3) then has to be reloaded per (unit, chain block) and crosses the
    // fabric once per unit: 7.1 GB instead of 4.1 GB, and 0.7 % slower.  Theta in registers for the whole launch and
    // X streamed once per XCD is the cheaper side of that trade.)
    long want = 1;
    if (pl.chain_blocks % num_cu != 0) want = (4L * num_cu + pl.chain_blocks - 1) / pl.chain_blocks;
    if (want > pl.nunits) want = pl.nunits;
    pl.units_per_split = (int)((pl.nunits + want - 1) / want);
    pl.nsplit = (pl.nunits + pl.units_per_split - 1) / pl.units_per_split;
    return pl;
}

template <int KS, int TC>
static void launch_ld(const MfmaPlan &pl, const mcmchip_target *t, const double *prop, double *partial,
                      DeviceCtx *dc)
{
    constexpr int NC = KS > 32 ? 1 : MFMA_NC;
    dim3 grid((unsigned)pl.chain_blocks, (unsigned)pl.nsplit), block(MFMA_NW * 64);
    if (t->wt)
        hipLaunchKernelGGL((logdens_mfma_kernel<KS, NC, MFMA_NW, TC, true>), grid, block, 0, dc->stream, t->xt, prop, pl.ldc,
                           partial, pl.tiles_per_unit, pl.units_per_split, pl.nunits, t->ntiles, dc->d_tables, t->wt);
    else
        hipLaunchKernelGGL((logdens_mfma_kernel<KS, NC, MFMA_NW, TC, false>), grid, block, 0, dc->stream, t->xt, prop, pl.ldc,
                           partial, pl.tiles_per_unit, pl.units_per_split, pl.nunits, t->ntiles, dc->d_tables,
                           (const double *)nullptr);
}

static int launch_logdens(const MfmaPlan &pl, const mcmchip_target *t, const double *prop, double *partial,
                          DeviceCtx *dc)
{
    switch (pl.ks) {
    case 2: launch_ld<2, 16>(pl, t, prop, partial, dc); break;
    case 4: launch_ld<4, 8>(pl, t, prop, partial, dc); break;
    case 6: launch_ld<6, 4>(pl, t, prop, partial, dc); break;
    case 8: launch_ld<8, 4>(pl, t, prop, partial, dc); break;
    case 10: launch_ld<10, 2>(pl, t, prop, partial, dc); break;
    case 12: launch_ld<12, 2>(pl, t, prop, partial, dc); break;
    case 14: launch_ld<14, 2>(pl, t, prop, partial, dc); break;
    case 16: launch_ld<16, 2>(pl, t, prop, partial, dc); break;
    case 20: launch_ld<20, 1>(pl, t, prop, partial, dc); break;
    case 24: launch_ld<24, 1>(pl, t, prop, partial, dc); break;
    case 28: launch_ld<28, 1>(pl, t, prop, partial, dc); break;
    case 32: launch_ld<32, 1>(pl, t, prop, partial, dc); break;
    case 48: launch_ld<48, 1>(pl, t, prop, partial, dc); break;
    case 64: launch_ld<64, 1>(pl, t, prop, partial, dc); break;
    default: return set_error(MCMCHIP_EINVAL, "MFMA path supports d <= %d", MCMCHIP_MFMA_MAXD);
    }
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

// the density launch for other drivers (temper_mfma.hip): the plan's chain stride and unit count, and one launch over
// `nchain` columns of theta ([dpad][ldc], zero padded) into partial ([nunits][ldc])
int mfma_density_plan(const mcmchip_target *t, long nchain, DeviceCtx *dc, long *ldc, int *nunits)
{
    if (!t->xt) return set_error(MCMCHIP_EINVAL, "target has no MFMA tiles");
    const MfmaPlan pl = make_plan(t, nchain, dc->num_cu);
    *ldc = pl.ldc;
    *nunits = pl.nunits;
    return MCMCHIP_OK;
}
int mfma_density_launch(const mcmchip_target *t, long nchain, const double *theta, double *partial, DeviceCtx *dc)
{
    const MfmaPlan pl = make_plan(t, nchain, dc->num_cu);
    return launch_logdens(pl, t, theta, partial, dc);
}

// carve the MFMA path's scratch buffers out of the device's cached workspace (256-byte aligned pieces)
struct Workspace {
    DeviceCtx *dc;
    size_t need = 0, off = 0;
    std::vector<size_t> sizes;
    explicit Workspace(DeviceCtx *d) : dc(d) {}
    void want(size_t bytes) { sizes.push_back((bytes + 255) & ~(size_t)255); need += sizes.back(); }
    int reserve()
    {
        if (need > dc->ws_bytes) {
            if (dc->ws) (void)hipFree(dc->ws);
            dc->ws = nullptr;
            dc->ws_bytes = 0;
            HIP_TRY(hipMalloc(&dc->ws, need));
            dc->ws_bytes = need;
        }
        return MCMCHIP_OK;
    }
    size_t next = 0;
    double *take() { double *p = (double *)((char *)dc->ws + off); off += sizes[next++]; return p; }
    void done()          // after the stream has been synchronised
    {
        if (dc->ws_bytes > MCMCHIP_WS_KEEP) {
            (void)hipFree(dc->ws);
            dc->ws = nullptr;
            dc->ws_bytes = 0;
        }
    }
};

int run_metrop_mfma(const mcmchip_target *t, const MetropDev &p, DeviceCtx *dc)
{
    if (!t->xt) return set_error(MCMCHIP_EINVAL, "target has no MFMA tiles (not a logistic target)");
    if (p.debug) return set_error(MCMCHIP_EINVAL, "debug trace is only available on the stream path");
    hipStream_t s = dc->stream;
    MfmaPlan pl = make_plan(t, p.nchain, dc->num_cu);
    const long ldc = pl.ldc;
    Workspace ws(dc);
    ws.want((size_t)t->dpad * ldc * 8);
    ws.want((size_t)t->dpad * ldc * 8);
    ws.want((size_t)pl.nunits * ldc * 8);
    ws.want((size_t)ldc * 8);
    ws.want((size_t)ldc * 8);
    ws.want((size_t)ldc * 8);
    ws.want((size_t)(p.nout > 0 ? p.nout : 1) * ldc * 8);
    if (t->morph_flags) {
        ws.want((size_t)t->dpad * ldc * 8);
        ws.want((size_t)t->dpad * ldc * 8);
        ws.want((size_t)ldc * 8);
    }
    if (int rc = ws.reserve()) return rc;
    double *state = ws.take(), *prop = ws.take(), *partial = ws.take(), *cur = ws.take(), *acc = ws.take(),
           *acc_b = ws.take(), *buf = ws.take();
    double *theta = prop, *theta_state = state, *lj = nullptr;
    if (t->morph_flags) { theta = ws.take(); theta_state = ws.take(); lj = ws.take(); }

    long total = (long)t->dpad * ldc;
    hipLaunchKernelGGL(soa_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p.initial,
                       p.nchain, t->d, t->dpad, ldc, state, prop);
    HIP_TRY(hipGetLastError());

    UpdParams q;
    memset(&q, 0, sizeof q);
    q.nchain = p.nchain; q.ldc = ldc; q.d = t->d; q.dpad = t->dpad; q.nunits = pl.nunits;
    q.pad_term = (double)t->npad_rows * 0.6931471805599453;
    q.prior_c = t->kind == MCMCHIP_T_BINOMIAL_LOGIT ? 0.0 : t->prior_c;       // morph.Rnw:361-365 has no separate prior term
    q.nbatch = p.nbatch; q.blen = p.blen; q.nspac = p.nspac;
    q.scale_kind = p.scale_kind; q.scale = p.scale;
    q.nout = p.nout; q.sq_from = p.sq_from; q.out_map = p.out_map;
    q.rng_mode = p.rng_mode; q.z = p.z; q.u = p.u;
    q.seed = p.seed; q.chain_offset = p.chain_offset; q.iter_offset = p.iter_offset;
    q.partial = partial;
    q.colsum = t->colsum;
    q.state = state; q.prop = prop;
    q.cur = cur; q.acc = acc; q.acc_b = acc_b; q.buf = buf;
    q.batch = p.batch; q.accept_batch = p.accept_batch; q.accept = p.accept; q.final_ = p.final_;
    q.logdens_out = nullptr;
    q.flag = p.flag;
    q.morph_flags = t->morph_flags; q.morph_b = t->morph_b; q.morph_r = t->morph_r; q.morph_p = t->morph_p;
    q.morph_center = t->morph_center;
    q.theta = theta; q.theta_state = theta_state; q.lj = lj;
    if (t->morph_flags) {
        HIP_TRY(hipMemsetAsync(theta, 0, (size_t)t->dpad * ldc * 8, s));       // padded rows and chains: zeros
        hipLaunchKernelGGL(morph_init_kernel, dim3((unsigned)((p.nchain + 255) / 256)), dim3(256), 0, s, q);
        HIP_TRY(hipGetLastError());
    }

    const long niter = (long)p.nbatch * p.blen * p.nspac;
    dim3 ugrid((unsigned)((p.nchain + 255) / 256)), ublock(256);
    timing_reset("logdens_mfma_kernel");
    // The iterations are queued asynchronously.  With an interrupt callback registered (metrop.c:269) the host waits
    // for the queue about every quarter of a second and polls: the first group is one iteration, which calibrates.
    const bool polled = lib_ctx().poll != nullptr;
    long group = 1, in_group = 0;
    for (long l = -1; l < niter; l++) {
        timing_begin(s);
        int rc = launch_logdens(pl, t, theta, partial, dc);
        timing_end(s);
        if (rc) return rc;
        hipLaunchKernelGGL(metrop_update_kernel, ugrid, ublock, 0, s, q, l);
        HIP_TRY(hipGetLastError());
        if (polled && ++in_group >= group && l + 1 < niter) {
            const double ms = timing_last_ms();      // waits for the last density launch
            HIP_TRY(hipStreamSynchronize(s));
            if (int irc = poll_interrupt()) { ws.done(); return irc; }
            group = ms > 0.0 ? (long)(lib_ctx().launch_budget_s * 1e3 / ms) : 1;
            if (group < 1) group = 1;
            in_group = 0;
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    ws.done();
    return MCMCHIP_OK;
}

int launch_logdens_mfma(const mcmchip_target *t, const double *states, long nchain, double *out,
                        DeviceCtx *dc)
{
    if (!t->xt) return set_error(MCMCHIP_EINVAL, "target has no MFMA tiles (not a logistic target)");
    hipStream_t s = dc->stream;
    MfmaPlan pl = make_plan(t, nchain, dc->num_cu);
    const long ldc = pl.ldc;
    DevBuf state, prop, partial, ljbuf;
    HIP_TRY(state.alloc((size_t)t->dpad * ldc * 8));
    HIP_TRY(prop.alloc((size_t)t->dpad * ldc * 8));
    HIP_TRY(partial.alloc((size_t)pl.nunits * ldc * 8));
    if (t->morph_flags) HIP_TRY(ljbuf.alloc((size_t)ldc * 8));
    long total = (long)t->dpad * ldc;
    const bool models = t->kind == MCMCHIP_T_LOGISTIC_MODELS;
    if (models)
        hipLaunchKernelGGL(soa_init_models_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, states, nchain, t->d,
                           t->dpad, ldc, t->par, prop.as<double>());
    else
        hipLaunchKernelGGL(soa_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, states,
                           nchain, t->d, t->dpad, ldc, state.as<double>(), prop.as<double>());
    HIP_TRY(hipGetLastError());
    UpdParams q;
    memset(&q, 0, sizeof q);
    q.nchain = nchain; q.ldc = ldc; q.d = t->d; q.dpad = t->dpad; q.nunits = pl.nunits;
    // morphed descriptor: prop holds the morphed states, the contraction runs on state = inverse(prop) (its padding is
    // zero from soa_init_kernel, the rest is overwritten)
    q.prop = prop.as<double>();
    q.theta = t->morph_flags ? state.as<double>() : prop.as<double>();
    q.theta_state = q.theta;
    q.morph_flags = t->morph_flags; q.morph_b = t->morph_b; q.morph_r = t->morph_r; q.morph_p = t->morph_p;
    q.morph_center = t->morph_center;
    q.lj = t->morph_flags ? ljbuf.as<double>() : nullptr;
    if (t->morph_flags) {
        hipLaunchKernelGGL(morph_init_kernel, dim3((unsigned)((nchain + 255) / 256)), dim3(256), 0, s, q);
        HIP_TRY(hipGetLastError());
    }
    timing_reset("logdens_mfma_kernel");
    timing_begin(s);
    int rc = launch_logdens(pl, t, q.theta, partial.as<double>(), dc);
    timing_end(s);
    if (rc) return rc;
    q.pad_term = (double)t->npad_rows * 0.6931471805599453;
    q.prior_c = t->kind == MCMCHIP_T_BINOMIAL_LOGIT ? 0.0 : t->prior_c;       // morph.Rnw:361-365 has no separate prior term
    q.nbatch = q.blen = q.nspac = 1;
    q.partial = partial.as<double>();
    q.colsum = t->colsum;
    q.state = state.as<double>();
    q.logdens_out = out;
    if (models) { q.models_states = states; q.models_par = t->par; }
    hipLaunchKernelGGL(metrop_update_kernel, dim3((unsigned)((nchain + 255) / 256)), dim3(256), 0, s, q, -1L);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return MCMCHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// mcmchip_logdens_noalloc: the MFMA log density for callers that cannot allocate (HIP graph capture, serving loops).
// Its scratch -- the dimension-major copy of the states and the per-split partial sums, which the other entry
// points take from hipMalloc -- is this file-scope device array.  Every GPU the library runs on holds its own copy
// (32 MiB) at its own address until the process ends; host code reaches it only through the pointer that
// hipGetSymbolAddress returned for that device (the array's name is a host-side stand-in, not a device address).
// Calls on one GPU share the array: calls that overlap on one GPU race.  Nothing in it has to start at zero (the
// staging kernel writes every padded state, the density kernel every partial sum it later reads).
static __device__ __align__(256) double mfma_scratch[MCMCHIP_NOALLOC_SCRATCH_DOUBLES];

int mfma_scratch_setup(DeviceCtx *dc)          // mcmchip_init: once per device, with that device current
{
    if (dc->noalloc_scratch) return MCMCHIP_OK;
    void *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(mfma_scratch)));
    if (!ptr) return set_error(MCMCHIP_EHIP, "hipGetSymbolAddress returned a null scratch address");
    dc->noalloc_scratch = (double *)ptr;
    return MCMCHIP_OK;
}

int launch_logdens_mfma_noalloc(const mcmchip_target *t, const double *states, long nchain, double *out, DeviceCtx *dc)
{
    if (!t->xt) return set_error(MCMCHIP_EINVAL, "target has no MFMA tiles (not a logistic target with d <= %d)", MCMCHIP_MFMA_MAXD);
    if (!dc->noalloc_scratch) return set_error(MCMCHIP_EHIP, "the scratch array of this device was not set up by mcmchip_init");
    hipStream_t s = dc->stream;
    const MfmaPlan pl = make_plan(t, nchain, dc->num_cu);          // the plan of mcmchip_logdens: same bits
    const long ldc = pl.ldc;
    const size_t nprop = (size_t)t->dpad * ldc, npart = (size_t)pl.nunits * ldc;
    if (nprop + npart > (size_t)MCMCHIP_NOALLOC_SCRATCH_DOUBLES)
        return set_error(MCMCHIP_ECAP, "mcmchip_logdens_noalloc: %ld chains of dimension %d need %zu doubles of scratch, the "
                         "fixed array holds %ld (use mcmchip_logdens)", nchain, t->d, nprop + npart, (long)MCMCHIP_NOALLOC_SCRATCH_DOUBLES);
    double *prop = dc->noalloc_scratch, *partial = dc->noalloc_scratch + nprop;
    const long total = (long)nprop;
    hipLaunchKernelGGL(soa_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, states, nchain, t->d,
                       t->dpad, ldc, (double *)nullptr, prop);
    HIP_TRY(hipGetLastError());
    if (int rc = launch_logdens(pl, t, prop, partial, dc)) return rc;
    UpdParams q;
    memset(&q, 0, sizeof q);
    q.nchain = nchain; q.ldc = ldc; q.d = t->d; q.dpad = t->dpad; q.nunits = pl.nunits;
    q.pad_term = (double)t->npad_rows * 0.6931471805599453;
    q.prior_c = t->kind == MCMCHIP_T_BINOMIAL_LOGIT ? 0.0 : t->prior_c;       // morph.Rnw:361-365 has no separate prior term
    q.nbatch = q.blen = q.nspac = 1;
    q.partial = partial;
    q.colsum = t->colsum;
    q.state = prop; q.prop = prop; q.theta = prop; q.theta_state = prop;
    q.logdens_out = out;
    hipLaunchKernelGGL(metrop_update_kernel, dim3((unsigned)((nchain + 255) / 256)), dim3(256), 0, s, q, -1L);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return MCMCHIP_OK;
}
