// diag.hip -- Monte Carlo standard error estimators: initseq() and olbm().
//
// initseq (reference initseq.c:7-128): the O(len x lags) double sum of lag products
// (initseq.c:24-47) is the hot loop; the monotone / convex post-processing (55-105) works on a
// few hundred numbers and stays on the host, in the reference's own sequential order.
//   autocov_kernel streams a column once per block of 32 lags: a workgroup stages a 2048-element
//   tile (plus a lag-shifted tile with a 32-element halo) in LDS, each lane keeps an 8 x 40
//   register window and 32 accumulators, so one pass costs 8 B/element of HBM for 32 lags
//   (16 flop/byte: the pass is fp64-VALU-bound; lags are produced speculatively in blocks
//   because the stopping lag (first negative Gamma, initseq.c:42-46) is data dependent).
//   LDS rows are padded by one double per 8 so the per-lane windows (stride 9 doubles) are
//   bank-conflict-free for ds_read_b64.  Partial sums are reduced in a fixed order.
//
// olbm (reference olbm.c:22-78): sliding window sums and the p x p sum of outer products.
//   Window sums are rebuilt per 64-window slab from a wave prefix scan of x[t+len]-x[t]
//   (coalesced along time because x is column-major), seeded per time segment by a direct
//   centred sum, so rounding does not accumulate along n as in the reference's running
//   recurrence (olbm.c:61-62).  Each workgroup owns one 16 x 16 tile pair of the lower triangle.
#include "kernels.h"
#include <math.h>
#include <vector>

// =============================================================================================
// initseq
// =============================================================================================
#define IS_R 8
#define IS_B 32
#define IS_TILE (256 * IS_R)
#define IS_TILES_PER_BLOCK 32

__global__ __launch_bounds__(256) void autocov_kernel(const double *__restrict__ x, long len, const int *__restrict__ cols,
                                                      long lag_first, long nblk, int nlb, double *__restrict__ partial)
{
    __shared__ double A[IS_TILE + IS_TILE / 8];
    __shared__ double Bt[IS_TILE + IS_B + (IS_TILE + IS_B) / 8 + 8];
    __shared__ double red[4][IS_B];
    const int t = threadIdx.x;
    const int col = cols[blockIdx.z];
    const int lb = blockIdx.y;
    const long lag0 = lag_first + (long)lb * IS_B;
    const double *xs = x + (size_t)col * len;
    double acc[IS_B];
#pragma unroll
    for (int k = 0; k < IS_B; k++) acc[k] = 0.0;

    for (int tile = 0; tile < IS_TILES_PER_BLOCK; tile++) {
        const long a = ((long)blockIdx.x * IS_TILES_PER_BLOCK + tile) * IS_TILE;
        if (a >= len) break;
        for (int e = t; e < IS_TILE; e += 256) {
            const long idx = a + e;
            A[e + (e >> 3)] = idx < len ? xs[idx] : 0.0;
        }
        for (int e = t; e < IS_TILE + IS_B; e += 256) {
            const long idx = a + lag0 + e;
            Bt[e + (e >> 3)] = idx < len ? xs[idx] : 0.0;
        }
        __syncthreads();
        double wa[IS_R], wb[IS_R + IS_B];
#pragma unroll
        for (int m = 0; m < IS_R; m++) wa[m] = A[t * 9 + m];
#pragma unroll
        for (int m = 0; m < IS_R + IS_B; m++) wb[m] = Bt[t * 9 + m + (m >> 3)];
#pragma unroll
        for (int i = 0; i < IS_R; i++)
#pragma unroll
            for (int k = 0; k < IS_B; k++) acc[k] = fma(wa[i], wb[i + k], acc[k]);
        __syncthreads();
    }
    // fixed-shape reduction: xor butterfly inside the wave, then the 4 waves in order
    const int wave = t >> 6, lane = t & 63;
#pragma unroll
    for (int k = 0; k < IS_B; k++) {
        double v = acc[k];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (t < IS_B) {
        const double v = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
        partial[(((size_t)blockIdx.z * nlb + lb) * nblk + blockIdx.x) * IS_B + t] = v;
    }
}

__global__ void autocov_reduce_kernel(const double *__restrict__ partial, long nblk, long total, double *__restrict__ out)
{
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (colslot*nlb + lb)*IS_B + k
    if (idx >= total) return;
    long grp = idx / IS_B;
    int k = (int)(idx % IS_B);
    double s = 0.0;
    for (long b = 0; b < nblk; b++) s += partial[(grp * nblk + b) * IS_B + k];
    out[idx] = s;
}

// host post-processing of one column, in the reference's order (initseq.c:49-105)
static void initseq_finish(const std::vector<double> &Gam, long m, double g0, double *Gpos, double *Gdec,
                           double *Gcon, double *vpos, double *vdec, double *vcon)
{
    std::vector<double> buff(Gam.begin(), Gam.begin() + m);
    for (long j = 0; j < m; ++j) Gpos[j] = buff[j];
    for (long j = 1; j < m; ++j)
        if (buff[j] > buff[j - 1]) buff[j] = buff[j - 1];
    for (long j = 0; j < m; ++j) Gdec[j] = buff[j];
    for (long j = m - 1; j > 0; --j) buff[j] -= buff[j - 1];
    std::vector<double> puff((size_t)(m > 0 ? m : 1));
    std::vector<long> nuff((size_t)(m > 0 ? m : 1));
    long nstep = 0;
    for (long j = 1; j < m; ++j) {
        puff[nstep] = buff[j];
        nuff[nstep] = 1;
        ++nstep;
        while (nstep > 1 && puff[nstep - 1] / nuff[nstep - 1] < puff[nstep - 2] / nuff[nstep - 2]) {
            puff[nstep - 2] += puff[nstep - 1];
            nuff[nstep - 2] += nuff[nstep - 1];
            --nstep;
        }
    }
    for (long js = 0, j = 1; js < nstep; ++js) {
        double muff = puff[js] / nuff[js];
        for (long q = 0; q < nuff[js]; ++j, ++q) buff[j] = buff[j - 1] + muff;
    }
    for (long j = 0; j < m; ++j) Gcon[j] = buff[j];
    double vp = 0.0, vd = 0.0, vc = 0.0;
    for (long j = 0; j < m; ++j) { vp += Gpos[j]; vd += Gdec[j]; vc += Gcon[j]; }
    *vpos = vp * 2.0 - g0;
    *vdec = vd * 2.0 - g0;
    *vcon = vc * 2.0 - g0;
}

int run_initseq(const double *x, long len, long ncol, long cap, long *n_out, double *gamma0, double *Gpos,
                double *Gdec, double *Gcon, double *vpos, double *vdec, double *vcon, DeviceCtx *dc)
{
    hipStream_t s = dc->stream;
    const long half = len / 2;
    const long nblk = (len + (long)IS_TILE * IS_TILES_PER_BLOCK - 1) / ((long)IS_TILE * IS_TILES_PER_BLOCK);
    const int NLB_MAX = 8;
    DevBuf partial, sums, dcols;
    HIP_TRY(partial.alloc((size_t)ncol * NLB_MAX * nblk * IS_B * 8));
    HIP_TRY(sums.alloc((size_t)ncol * NLB_MAX * IS_B * 8));
    HIP_TRY(dcols.alloc((size_t)ncol * sizeof(int)));
    std::vector<std::vector<double>> Gam((size_t)ncol);   // Gamma_i per column, as far as computed
    std::vector<long> done((size_t)ncol, -1);            // output length once the stop rule fired
    std::vector<double> g0((size_t)ncol, 0.0);
    std::vector<int> active;
    for (long c = 0; c < ncol; c++) active.push_back((int)c);
    std::vector<double> host;
    long lag_first = 0;
    int nlb = 2;
    timing_reset("autocov_kernel");
    if (half == 0) active.clear();
    while (!active.empty()) {
        const long nact = (long)active.size();
        HIP_TRY(hipMemcpyAsync(dcols.p, active.data(), (size_t)nact * sizeof(int), hipMemcpyHostToDevice, s));
        dim3 grid((unsigned)nblk, (unsigned)nlb, (unsigned)nact);
        timing_begin(s);
        hipLaunchKernelGGL(autocov_kernel, grid, dim3(256), 0, s, x, len, dcols.as<int>(), lag_first, nblk, nlb,
                           partial.as<double>());
        timing_end(s);
        HIP_TRY(hipGetLastError());
        const long total = nact * nlb * IS_B;
        hipLaunchKernelGGL(autocov_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                           partial.as<double>(), nblk, total, sums.as<double>());
        HIP_TRY(hipGetLastError());
        host.resize((size_t)total);
        HIP_TRY(hipMemcpyAsync(host.data(), sums.p, (size_t)total * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<int> next;
        for (long a = 0; a < nact; a++) {
            const int c = active[(size_t)a];
            const double *g = host.data() + (size_t)a * nlb * IS_B;   // lags lag_first .. lag_first + nlb*32
            bool fin = false;
            for (long q = 0; q < (long)nlb * IS_B / 2 && !fin; q++) {
                const long i = lag_first / 2 + q;
                if (i >= half) { done[(size_t)c] = half; fin = true; break; }
                const double gam1 = g[2 * q] / len, gam2 = g[2 * q + 1] / len;    // initseq.c:30,39
                if (i == 0) g0[(size_t)c] = gam1;
                double G = gam1 + gam2;
                if (G < 0.0) {                                                    // initseq.c:42-46
                    Gam[(size_t)c].push_back(0.0);
                    done[(size_t)c] = i + 1;
                    fin = true;
                } else {
                    Gam[(size_t)c].push_back(G);
                    if (i + 1 >= half) { done[(size_t)c] = half; fin = true; }
                }
            }
            if (!fin) next.push_back(c);
        }
        active.swap(next);
        lag_first += (long)nlb * IS_B;
        nlb = NLB_MAX;
    }
    bool overflow = false;
    for (long c = 0; c < ncol; c++) {
        long m = half == 0 ? 0 : done[(size_t)c];
        n_out[c] = m;
        gamma0[c] = g0[(size_t)c];
        if (m > cap) { overflow = true; continue; }
        initseq_finish(Gam[(size_t)c], m, g0[(size_t)c], Gpos + (size_t)c * cap, Gdec + (size_t)c * cap,
                       Gcon + (size_t)c * cap, vpos + c, vdec + c, vcon + c);
    }
    if (overflow) return set_error(MCMCHIP_ECAP, "initseq: output capacity too small; required lengths are in n_out");
    return MCMCHIP_OK;
}

// =============================================================================================
// olbm
// =============================================================================================
#define OL_SLAB 64          // windows per slab
#define OL_SEG_SLABS 256    // slabs per time segment (16384 windows)

__global__ __launch_bounds__(256) void colsum_kernel(const double *__restrict__ x, long n, long chunk, long nchunk,
                                                     double *__restrict__ partial)
{
    __shared__ double red[4];
    const int col = blockIdx.y;
    const long a = (long)blockIdx.x * chunk;
    long b = a + chunk;
    if (b > n) b = n;
    double s = 0.0;
    for (long t = a + threadIdx.x; t < b; t += 256) s += x[(size_t)col * n + t];
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)col * nchunk + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ void colmean_kernel(const double *__restrict__ partial, long nchunk, long n, int p, double *__restrict__ mean)
{
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= p) return;
    double s = 0.0;
    for (long c = 0; c < nchunk; c++) s += partial[(size_t)col * nchunk + c];
    mean[col] = s / (double)n;     // olbm.c:42
}

// one workgroup = one time segment x one 16x16 tile pair (ti >= tj) of var
__global__ __launch_bounds__(256) void olbm_kernel(const double *__restrict__ x, long n, int p, long len, long nwin,
                                                   const double *__restrict__ mean, int npairs,
                                                   double *__restrict__ partial)
{
    __shared__ double W[OL_SLAB][33];   // centred window sums: [window][0..15 = I cols, 16..31 = J cols]
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    // decode the tile pair
    int pair = blockIdx.y, ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= pair) ti++;
    const int tj = pair - ti * (ti + 1) / 2;
    const long w_begin = (long)blockIdx.x * OL_SLAB * OL_SEG_SLABS;
    long w_end = w_begin + (long)OL_SLAB * OL_SEG_SLABS;
    if (w_end > nwin) w_end = nwin;

    // each wave owns 8 of the 32 columns; base[c] = centred sum of the window starting at w (wave-uniform)
    double base[8];
    int colv[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int lc = wave * 8 + c;
        const int col = (lc < 16) ? ti * 16 + lc : tj * 16 + (lc - 16);
        colv[c] = col < p ? col : -1;
        double s = 0.0;
        if (colv[c] >= 0) {
            const double *xc = x + (size_t)col * n;
            const double m = mean[col];
            for (long q = lane; q < len; q += 64) s += xc[w_begin + q] - m;
        }
        for (int mm = 32; mm > 0; mm >>= 1) s += __shfl_xor(s, mm, 64);
        base[c] = s;
    }
    const int i = t & 15, j = t >> 4;    // output element (row i of tile ti, col j of tile tj)
    double acc = 0.0;
    for (long w0 = w_begin; w0 < w_end; w0 += OL_SLAB) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            // W[w0 + lane] = base + sum_{s < lane} (x[w0+s+len] - x[w0+s])   (olbm.c:61-62 as a prefix scan)
            double dlt = 0.0;
            const long w = w0 + lane;
            if (colv[c] >= 0 && w + len < n) {
                const double *xc = x + (size_t)colv[c] * n;
                dlt = xc[w + len] - xc[w];
            }
            double inc = dlt;    // inclusive scan
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double up = __shfl_up(inc, off, 64);
                if (lane >= off) inc += up;
            }
            const double mine = base[c] + (inc - dlt);
            W[lane][wave * 8 + c] = (w < w_end) ? mine : 0.0;
            base[c] += __shfl(inc, 63, 64);
        }
        __syncthreads();
#pragma unroll 16
        for (int s2 = 0; s2 < OL_SLAB; s2++) acc = fma(W[s2][i], W[s2][16 + j], acc);
        __syncthreads();
    }
    partial[((size_t)blockIdx.x * npairs + pair) * 256 + t] = acc;
}

__global__ void olbm_finish_kernel(const double *__restrict__ partial, long nseg, int npairs, int p, double denom,
                                   double *__restrict__ var)
{
    const int pair = blockIdx.x, t = threadIdx.x;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= pair) ti++;
    const int tj = pair - ti * (ti + 1) / 2;
    double s = 0.0;
    for (long g = 0; g < nseg; g++) s += partial[((size_t)g * npairs + pair) * 256 + t];
    const int r = ti * 16 + (t & 15), c = tj * 16 + (t >> 4);
    if (r < p && c < p) {
        const double v = s / denom;          // olbm.c:74
        var[r + (size_t)p * c] = v;
        var[c + (size_t)p * r] = v;          // olbm.c:75 (symmetric)
    }
}

int run_olbm(const double *x, long n, int p, long len, double *mean, double *var, int nocalc, DeviceCtx *dc)
{
    hipStream_t s = dc->stream;
    DevBuf dmean, dvar, part, cpart;
    HIP_TRY(dmean.alloc((size_t)p * 8));
    HIP_TRY(dvar.alloc((size_t)p * p * 8));
    if (nocalc) {
        HIP_TRY(hipMemcpyAsync(dmean.p, mean, (size_t)p * 8, hipMemcpyHostToDevice, s));
    } else {
        const long chunk = 1 << 20;
        const long nchunk = (n + chunk - 1) / chunk;
        HIP_TRY(cpart.alloc((size_t)p * nchunk * 8));
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nchunk, (unsigned)p), dim3(256), 0, s, x, n, chunk, nchunk,
                           cpart.as<double>());
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(colmean_kernel, dim3((unsigned)((p + 63) / 64)), dim3(64), 0, s, cpart.as<double>(), nchunk,
                           n, p, dmean.as<double>());
        HIP_TRY(hipGetLastError());
    }
    const long nwin = n - len + 1;
    const long seg = (long)OL_SLAB * OL_SEG_SLABS;
    const long nseg = (nwin + seg - 1) / seg;
    const int nt = (p + 15) / 16;
    const int npairs = nt * (nt + 1) / 2;
    HIP_TRY(part.alloc((size_t)nseg * npairs * 256 * 8));
    timing_reset("olbm_kernel");
    timing_begin(s);
    hipLaunchKernelGGL(olbm_kernel, dim3((unsigned)nseg, (unsigned)npairs), dim3(256), 0, s, x, n, p, len, nwin,
                       dmean.as<double>(), npairs, part.as<double>());
    timing_end(s);
    HIP_TRY(hipGetLastError());
    const double denom = (double)nwin * (double)n * (double)len;
    hipLaunchKernelGGL(olbm_finish_kernel, dim3((unsigned)npairs), dim3(256), 0, s, part.as<double>(), nseg, npairs, p,
                       denom, dvar.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(var, dvar.p, (size_t)p * p * 8, hipMemcpyDeviceToHost, s));
    if (!nocalc) HIP_TRY(hipMemcpyAsync(mean, dmean.p, (size_t)p * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return MCMCHIP_OK;
}
