// olbm_small.hip -- olbm() (reference olbm.c:22-78) for FEW columns (p <= 8): an HBM-streaming kernel on the vector units.
//
// The matrix-core kernels (diag.hip, olbm_comb.hip) pad the columns to 64 and are built around 256 of them: with p = 3
// they still take what 64 columns take (14 ms at n = 1e8, 170 GB/s of input).  The usual call -- olbm(out$batch, len)
// with a handful of functionals -- is a stream: 2 p loads and p (p + 1) / 2 multiply-adds per window.  Here a thread
// owns KW consecutive windows of ALL columns; a workgroup walks its segment of windows in tiles of 256 KW; the window
// sums come from a scan of the differences x[t + len] - x[t] (olbm.c:61-62) over the tile -- sequentially inside a
// thread, DPP row scan + lane reads inside a wave, LDS across the four waves -- relative to the segment's start, the
// outer products are accumulated in registers, and the unknown window sum at the segment's start enters afterwards
// through the rank-two terms of olbm_comb.hip (sum (s + D)(s + D)^T = sum D D^T + cnt s s^T + s d^T + d s^T): no seed
// pass, no centre in the kernel, one direct sum of the first window per column for the whole call.  The exact means
// (olbm.c:38-43) come out of the same pass: the windows at multiples of len tile the series.
#include "kernels.h"
#include "olbm_shape.h"
#include <math.h>
#include <vector>

namespace {

template <int P> struct SmallShape {
    static constexpr int KW = 4;                           // consecutive windows per thread
    static constexpr int TILE = 256 * KW;
    static constexpr int NPAIR = P * (P + 1) / 2;
    static constexpr int NREC = NPAIR + 3 * P + 2;         // per piece: pairs, d, e, tiling sums, tiling count, windows
};

// inclusive scan over the 16 lanes of a DPP row
__device__ __forceinline__ double row_scan(double v)
{
#define OS_STEP(N)                                                                              \
    {                                                                                           \
        int lo = __double2loint(v), hi = __double2hiint(v);                                     \
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xf, 0xf, true);                     \
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xf, 0xf, true);                     \
        v += __hiloint2double(hi, lo);                                                          \
    }
    OS_STEP(1) OS_STEP(2) OS_STEP(4) OS_STEP(8)
#undef OS_STEP
    return v;
}
__device__ __forceinline__ double lane_read(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <int P>
__global__ __launch_bounds__(256) void olbm_small_kernel(const double *__restrict__ x, long n, int p, long len, long nwin,
                                                         long seg, double *__restrict__ rec)
{
    using S = SmallShape<P>;
    constexpr int KW = S::KW, TILE = S::TILE, NPAIR = S::NPAIR;
    __shared__ double wtot[2][P][4];                       // wave totals of a tile, double buffered by tile parity
    __shared__ double red[4][S::NREC];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, row = lane >> 4;
    const long W0 = (long)blockIdx.x * seg;
    long W1 = W0 + seg;
    if (W1 > nwin) W1 = nwin;
    double acc[NPAIR], dsum[P], tsum[P], run[P];
#pragma unroll
    for (int i = 0; i < NPAIR; i++) acc[i] = 0.0;
#pragma unroll
    for (int c = 0; c < P; c++) dsum[c] = tsum[c] = run[c] = 0.0;
    double tcnt = 0.0;
    // position of this thread's first window modulo len, kept incrementally (windows at multiples of len tile the series)
    long w = W0 + (long)t * KW;
    long rmod = w % len;
    const long step_mod = TILE % len;
    int par = 0;
    for (long wt = W0; wt < W1; wt += TILE, w += TILE, par ^= 1) {
        // ---- differences of this thread's KW windows, all columns (zero past the segment / the series)
        double del[P][KW];
#pragma unroll
        for (int c = 0; c < P; c++) {
            const double *xc = x + (size_t)(c < p ? c : 0) * n;
#pragma unroll
            for (int k = 0; k < KW; k++) {
                const long wk = w + k;
                const bool live = c < p && wk < W1 && wk + len < n;
                del[c][k] = live ? xc[wk + len] - xc[wk] : 0.0;
            }
        }
        // ---- per column: prefix inside the thread, scan over the wave, totals of the waves through LDS
        double base[P];                                    // D of this thread's first window
#pragma unroll
        for (int c = 0; c < P; c++) {
            double tot = del[c][0];
#pragma unroll
            for (int k = 1; k < KW; k++) tot += del[c][k];
            const double inc = row_scan(tot);              // inclusive over the row
            const double t15 = lane_read(inc, 15), t31 = lane_read(inc, 31), t47 = lane_read(inc, 47), t63 = lane_read(inc, 63);
            const double r2 = t15 + t31, r3 = r2 + t47;
            const double rb = row == 0 ? 0.0 : (row == 1 ? t15 : (row == 2 ? r2 : r3));
            base[c] = (inc - tot) + rb;                    // exclusive prefix inside the wave
            if (lane == 0) wtot[par][c][wave] = r3 + t63;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < P; c++) {
            const double a0 = wtot[par][c][0], a1 = wtot[par][c][1], a2 = wtot[par][c][2], a3 = wtot[par][c][3];
            const double wb = wave == 0 ? 0.0 : (wave == 1 ? a0 : (wave == 2 ? a0 + a1 : (a0 + a1) + a2));
            base[c] = run[c] + (wb + base[c]);
            run[c] += ((a0 + a1) + a2) + a3;               // every thread keeps the same running sum
        }
        // ---- outer products of the KW windows
#pragma unroll
        for (int k = 0; k < KW; k++) {
            const bool live = w + k < W1;
            long rk = rmod + k;
            while (rk >= len) rk -= len;                   // at most KW - 1 turns (len = 1)
            const bool tiling = live && rk == 0;
            double D[P];
#pragma unroll
            for (int c = 0; c < P; c++) {
                D[c] = live ? base[c] : 0.0;
                base[c] += del[c][k];
                dsum[c] += D[c];
                tsum[c] += tiling ? D[c] : 0.0;
            }
            tcnt += tiling ? 1.0 : 0.0;
            int i = 0;
#pragma unroll
            for (int a = 0; a < P; a++)
#pragma unroll
                for (int b = a; b < P; b++) acc[i] = fma(D[a], D[b], acc[i]), i++;
        }
        rmod += step_mod;
        if (rmod >= len) rmod -= len;
    }
    // ---- the piece's record: fixed-order reduction over the lanes, then over the waves
    auto wave_sum = [&](double v) {
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
        return v;
    };
#pragma unroll
    for (int i = 0; i < NPAIR; i++) {
        const double v = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
#pragma unroll
    for (int c = 0; c < P; c++) {
        const double v = wave_sum(dsum[c]), u = wave_sum(tsum[c]);
        if (lane == 0) {
            red[wave][NPAIR + c] = v;
            red[wave][NPAIR + P + c] = run[c];             // e: the running sum after the piece (the same in every thread)
            red[wave][NPAIR + 2 * P + c] = u;
        }
    }
    {
        const double v = wave_sum(tcnt);
        if (lane == 0) {
            red[wave][NPAIR + 3 * P] = v;
            red[wave][NPAIR + 3 * P + 1] = (double)(W1 > W0 ? W1 - W0 : 0);
        }
    }
    __syncthreads();
    double *out = rec + (size_t)blockIdx.x * S::NREC;
    for (int i = t; i < S::NREC; i += 256) {
        const bool same = (i >= NPAIR + P && i < NPAIR + 2 * P) || i == NPAIR + 3 * P + 1;     // e, windows: not sums
        out[i] = same ? red[0][i] : ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    }
}

// one workgroup: chain the pieces (s_{j+1} = s_j + e_j), the means from the tiling windows, then the matrix
template <int P>
__global__ __launch_bounds__(256) void olbm_small_finish_kernel(const double *__restrict__ rec, int npiece, int p, long n,
                                                                long len, long nwin, const double *__restrict__ centre,
                                                                const double *__restrict__ s0, const double *__restrict__ tail,
                                                                int want_mean, double *__restrict__ mean,
                                                                double *__restrict__ delta, double *__restrict__ sig,
                                                                double *__restrict__ var)
{
    using S = SmallShape<P>;
    constexpr int NPAIR = S::NPAIR, NREC = S::NREC;
    __shared__ double shift[P];
    __shared__ double csum[P][256], tspart[P][256];
    const int t = threadIdx.x;
    // sig[j][c] <- E_j, the window sum at the start of piece j less the one at window 0: every thread takes a run of
    // pieces (sums of its e, an exclusive scan of the 256 run sums by one thread per column, then the prefixes inside
    // the run) -- a single thread walking 2048 pieces through global memory took a millisecond
    const int ch = (npiece + 255) / 256, j0 = t * ch, j1 = j0 + ch < npiece ? j0 + ch : npiece;
#pragma unroll
    for (int c = 0; c < P; c++) {
        double sm = 0.0;
#pragma unroll 8
        for (int j = j0; j < j1; j++) sm += rec[(size_t)j * NREC + NPAIR + P + c];
        csum[c][t] = sm;
    }
    __syncthreads();
    if (t < P) {
        double E = 0.0;
        for (int i = 0; i < 256; i++) {
            const double v = csum[t][i];
            csum[t][i] = E;
            E += v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < P; c++) {
        double E = csum[c][t], ts = 0.0;
        for (int j = j0; j < j1; j++) {
            const double *r = rec + (size_t)j * NREC;
            sig[(size_t)j * P + c] = E;
            ts += r[NPAIR + 3 * P] * E + r[NPAIR + 2 * P + c];          // the piece's tiling windows: count * E_j + their D's
            E += r[NPAIR + P + c];
        }
        tspart[c][t] = ts;
    }
    __syncthreads();
    if (t < P) {
        double d = 0.0;
        if (want_mean && t < p) {
            double ts = 0.0;
            for (int i = 0; i < 256; i++) ts += tspart[t][i];
            const long ntile = n / len;
            const double nd = ((double)ntile * s0[t] + ts) + tail[t];   // n (mean - centre)
            d = nd / (double)n;
            mean[t] = centre[t] + d;                                    // olbm.c:42
            delta[t] = d;
        }
        shift[t] = t < p ? s0[t] - (double)len * d : 0.0;
    }
    __syncthreads();
    if (t < P * P) {
        const int row = t / P, col = t % P;
        const int a = row < col ? row : col, b = row < col ? col : row;     // the ordered pair: (r,c) and (c,r) the same bits
        int idx = 0;
        for (int i = 0; i < a; i++) idx += P - i;
        idx += b - a;
        double s = 0.0;
#pragma unroll 8
        for (int j = 0; j < npiece; j++) {                              // (loads in batches, the sum in piece order)
            const double *r = rec + (size_t)j * NREC;
            const double sa = shift[a] + sig[(size_t)j * P + a], sb = shift[b] + sig[(size_t)j * P + b];
            const double tt = fma(r[NPAIR + 3 * P + 1], sa, r[NPAIR + a]);
            s += r[idx];
            s = fma(tt, sb, s);
            s = fma(sa, r[NPAIR + b], s);
        }
        if (row < p && col < p) var[row + (size_t)p * col] = s / ((double)nwin * (double)n * (double)len);   // olbm.c:74-75
    }
}

template <int P>
int launch_small(const double *x, long n, int p, long len, const double *dcentre, double *dmean_out, double *ddelta, double *dvar,
                 double *s0, double *tail, DevBuf &recbuf, int *npiece_out, bool walk, DeviceCtx *dc)
{
    using S = SmallShape<P>;
    hipStream_t s = dc->stream;
    const long nwin = n - len + 1;
    // pieces: a few workgroups per CU, whole tiles each
    long groups = 8L * dc->num_cu;
    long seg = (nwin + groups - 1) / groups;
    seg = ((seg + S::TILE - 1) / S::TILE) * S::TILE;
    const int npiece = (int)((nwin + seg - 1) / seg);
    if (walk) {
        HIP_TRY(recbuf.alloc(((size_t)npiece * S::NREC + (size_t)npiece * P) * 8));
        timing_reset("olbm_small_kernel");
        timing_begin(s);
        hipLaunchKernelGGL(olbm_small_kernel<P>, dim3((unsigned)npiece), dim3(256), 0, s, x, n, p, len, nwin, seg, recbuf.as<double>());
        timing_end(s);
        HIP_TRY(hipGetLastError());
    }
    *npiece_out = npiece;
    double *rec = recbuf.as<double>(), *sig = rec + (size_t)npiece * S::NREC;
    hipLaunchKernelGGL(olbm_small_finish_kernel<P>, dim3(1), dim3(256), 0, s, rec, npiece, p, n, len, nwin, dcentre, s0, tail,
                       dmean_out ? 1 : 0, dmean_out, ddelta, sig, dvar);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

int dispatch_small(int p, const double *x, long n, long len, const double *dcentre, double *dmean_out, double *ddelta, double *dvar,
                   double *s0, double *tail, DevBuf &recbuf, int *npiece, bool walk, DeviceCtx *dc)
{
#define OS_CASE(PP) if (p <= PP) return launch_small<PP>(x, n, p, len, dcentre, dmean_out, ddelta, dvar, s0, tail, recbuf, npiece, walk, dc);
    OS_CASE(1) OS_CASE(2) OS_CASE(3) OS_CASE(4) OS_CASE(5) OS_CASE(6) OS_CASE(7) OS_CASE(8)
#undef OS_CASE
    return set_error(MCMCHIP_EINVAL, "olbm_small: p > 8");
}

}  // namespace

bool olbm_small_applies(long n, int p, long len)
{
    static const char *env = getenv("MCMCHIP_OLBM_SMALL");
    if (env && env[0] == '0') return false;
    return p <= 8;
}

int run_olbm_small(const double *x, long n, int p, long len, double *mean, double *var, int nocalc, DeviceCtx *dc)
{
    hipStream_t s = dc->stream;
    DevBuf vecs, recbuf;                                    // var | mean | centre | delta | s0 | tail
    HIP_TRY(vecs.alloc(((size_t)p * p + 5 * (size_t)p) * 8));
    double *dvar = vecs.as<double>(), *dmean = dvar + (size_t)p * p, *dcentre = dmean + p, *ddelta = dcentre + p, *s0 = ddelta + p,
           *tail = s0 + p;
    const bool one_pass = !nocalc && n >= OL_ONEPASS_MIN_N;
    if (nocalc) HIP_TRY(hipMemcpyAsync(dmean, mean, (size_t)p * 8, hipMemcpyHostToDevice, s));
    else if (one_pass) { if (int rc = olbm_pilot_centre(x, n, p, dcentre, s)) return rc; }
    else if (int rc = olbm_exact_means(x, n, p, dmean, s)) return rc;
    const double *centre = one_pass ? dcentre : dmean;
    if (int rc = olbm_first_window_sums(x, n, p, len, centre, s0, tail, s)) return rc;
    int npiece = 0;
    if (int rc = dispatch_small(p, x, n, len, centre, one_pass ? dmean : nullptr, ddelta, dvar, s0, tail, recbuf, &npiece, true, dc)) return rc;
    HIP_TRY(hipMemcpyAsync(var, dvar, (size_t)p * p * 8, hipMemcpyDeviceToHost, s));
    if (!nocalc) HIP_TRY(hipMemcpyAsync(mean, dmean, (size_t)p * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (one_pass) {
        // as in olbm_comb.hip: the sampled centre enters only through s0 - len * delta; beyond three digits lost there the
        // finish (not the walk) is repeated about the exact mean
        std::vector<double> hd((size_t)p);
        HIP_TRY(hipMemcpy(hd.data(), ddelta, (size_t)p * 8, hipMemcpyDeviceToHost));
        bool redo = false;
        for (int c = 0; c < p; c++) {
            const double main_term = var[c + (size_t)p * c] * (double)n;
            const double corr = (double)len * hd[c] * hd[c];
            if (!(corr <= OL_ONEPASS_MAX_LOSS * main_term) && !(fabs(hd[c]) <= 1e-14 * fabs(mean[c]))) redo = true;
        }
        if (redo) {
            if (int rc = olbm_first_window_sums(x, n, p, len, dmean, s0, tail, s)) return rc;
            if (int rc = dispatch_small(p, x, n, len, dmean, nullptr, nullptr, dvar, s0, tail, recbuf, &npiece, false, dc)) return rc;
            HIP_TRY(hipMemcpyAsync(var, dvar, (size_t)p * p * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
    }
    return MCMCHIP_OK;
}
