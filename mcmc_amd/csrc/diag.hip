// diag.hip -- Monte Carlo standard error estimator olbm() (initseq lives in initseq.hip).
//
// olbm (reference olbm.c:22-78): sliding window sums and the p x p sum of outer products.
//   p <= 256: olbm_mfma_kernel, all columns in one workgroup, var = V^T V on v_mfma_f64_16x16x4_f64 with
//   the window sums V rebuilt on the fly (see the comment at the kernel).  Window sums are seeded per
//   time segment by a direct centred sum, so rounding does not accumulate along n as in the reference's
//   running recurrence (olbm.c:61-62).  Means not supplied and n >= 2^20: no separate pass for the means
//   (olbm.c:38-43); the window sums are taken about a provisional centre, the exact means come out of them and
//   the matrix is corrected by a rank-two term (see pilot_centre_kernel).
//   p > 256: olbm_kernel, one workgroup per 64 x 64 tile pair of the lower triangle, 4 x 4 outputs per
//   lane in registers.
#include "kernels.h"
#include <math.h>
#include <stdlib.h>
#include <vector>

// =============================================================================================
// olbm
// =============================================================================================
#define OL_SLAB 64          // windows per slab (one wave-wide prefix scan)
#define OL_CT 64            // columns per tile: a workgroup owns a 64 x 64 tile pair of var
#define OL_LD 129           // LDS row stride in doubles (odd-ish: conflict-free column writes)

__global__ __launch_bounds__(256) void colsum_kernel(const double *__restrict__ x, long n, long chunk, long nchunk,
                                                     double *__restrict__ partial)
{
    __shared__ double red[4];
    const int col = blockIdx.y;
    const long a = (long)blockIdx.x * chunk;
    long b = a + chunk;
    if (b > n) b = n;
    double s = 0.0;
#pragma unroll 8
    for (long t = a + threadIdx.x; t < b; t += 256) s += x[(size_t)col * n + t];
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)col * nchunk + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// rows per workgroup of colsum_kernel: at most 256 chunks per column, at least 16 384 rows each (one chunk of 2^20 rows left a
// series of 1e6 rows to ONE workgroup per column, a load at a time: 1 ms of a 1.2 ms olbm call)
static long colsum_chunk(long n)
{
    long c = (n + 255) / 256;
    c = (c + 255) / 256 * 256;
    return c < 16384 ? 16384 : c;
}

__global__ void colmean_kernel(const double *__restrict__ partial, long nchunk, long n, int p, double *__restrict__ mean)
{
    int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= p) return;
    double s = 0.0;
#pragma unroll 8
    for (long c = 0; c < nchunk; c++) s += partial[(size_t)col * nchunk + c];
    mean[col] = s / (double)n;     // olbm.c:42
}

// one workgroup = one time segment x one 64 x 64 tile pair (ti >= tj) of var.
// x - mean(x) per column in place (mcmchip_series_centre): what R/initseq.R:3 does in R before the native call
__global__ __launch_bounds__(256) void centre_kernel(double *__restrict__ x, long n, const double *__restrict__ mean)
{
    double *xc = x + (size_t)blockIdx.y * n;
    const double m = mean[blockIdx.y];
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) xc[t] -= m;
}

int run_series_centre(double *x, long n, long p, double *mean, DeviceCtx *dc)
{
    hipStream_t s = dc->stream;
    if (p > 65535) return set_error(MCMCHIP_EINVAL, "too many columns");
    const long chunk = colsum_chunk(n), nchunk = (n + chunk - 1) / chunk;
    DevBuf cpart, dmean;
    HIP_TRY(cpart.alloc((size_t)p * nchunk * 8));
    HIP_TRY(dmean.alloc((size_t)p * 8));
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nchunk, (unsigned)p), dim3(256), 0, s, x, n, chunk, nchunk,
                       cpart.as<double>());
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(colmean_kernel, dim3((unsigned)((p + 63) / 64)), dim3(64), 0, s, cpart.as<double>(), nchunk, n, (int)p,
                       dmean.as<double>());
    HIP_TRY(hipGetLastError());
    long gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(centre_kernel, dim3((unsigned)gx, (unsigned)p), dim3(256), 0, s, x, n, dmean.as<double>());
    HIP_TRY(hipGetLastError());
    if (mean) HIP_TRY(hipMemcpyAsync(mean, dmean.p, (size_t)p * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return MCMCHIP_OK;
}

// Phase 1 (per 64-window slab): each wave rebuilds the centred window sums of 32 of the 128 columns with a
//   wave prefix scan over x[t+len] - x[t] (lanes = consecutive windows: coalesced 512-B loads) -> LDS.
// Phase 2: every thread holds a 4 x 4 block of the 64 x 64 output in registers and walks the 64 windows
//   (8 broadcast-friendly ds_read_b64 per 16 FMAs).
__global__ __launch_bounds__(256) void olbm_kernel(const double *__restrict__ x, long n, int p, long len, long nwin,
                                                   long seg_windows, const double *__restrict__ mean, int npairs,
                                                   double *__restrict__ partial)
{
    __shared__ double W[OL_SLAB * OL_LD];   // [window][0..63 = I cols, 64..127 = J cols]
    __shared__ double base[2 * OL_CT];      // centred sum of the window that starts the current slab
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    int pair = blockIdx.y, ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= pair) ti++;
    const int tj = pair - ti * (ti + 1) / 2;
    const long w_begin = (long)blockIdx.x * seg_windows;
    long w_end = w_begin + seg_windows;
    if (w_end > nwin) w_end = nwin;

    // seed: direct centred sum of the first window of the segment, 32 columns per wave
    for (int c = 0; c < 32; c++) {
        const int lc = wave * 32 + c;
        const int col = (lc < OL_CT) ? ti * OL_CT + lc : tj * OL_CT + (lc - OL_CT);
        double s = 0.0;
        if (col < p) {
            const double *xc = x + (size_t)col * n;
            const double m = mean[col];
            for (long q = lane; q < len; q += 64) s += xc[w_begin + q] - m;
        }
        for (int mm = 32; mm > 0; mm >>= 1) s += __shfl_xor(s, mm, 64);
        if (lane == 0) base[lc] = s;
    }
    __syncthreads();

    const int a = t & 15, b = t >> 4;     // rows {a + 16m} of tile ti, cols {b + 16m'} of tile tj
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.0;

    // x[w+len] - x[w] of the slab's 64 windows for this wave's 32 columns: all 64 loads are issued
    // together (and the next slab's are issued before phase 2), so HBM latency is paid once per slab
    const double *xcol[32];
#pragma unroll
    for (int c = 0; c < 32; c++) {
        const int lc = wave * 32 + c;
        const int col = (lc < OL_CT) ? ti * OL_CT + lc : tj * OL_CT + (lc - OL_CT);
        xcol[c] = col < p ? x + (size_t)col * n : nullptr;
    }
    double dl[32];
    auto load_slab = [&](long w0) {
        const long w = w0 + lane;
#pragma unroll
        for (int c = 0; c < 32; c++) {
            double v = 0.0;
            if (xcol[c] && w + len < n) v = xcol[c][w + len] - xcol[c][w];
            dl[c] = v;
        }
    };
    load_slab(w_begin);

    for (long w0 = w_begin; w0 < w_end; w0 += OL_SLAB) {
        const long w = w0 + lane;
#pragma unroll
        for (int c = 0; c < 32; c++) {
            const int lc = wave * 32 + c;
            // W[w0 + lane] = base + sum_{s < lane} (x[w0+s+len] - x[w0+s])   (olbm.c:61-62 as a prefix scan)
            const double dlt = dl[c];
            double inc = dlt;    // inclusive scan
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double up = __shfl_up(inc, off, 64);
                if (lane >= off) inc += up;
            }
            const double b0 = base[lc];
            W[lane * OL_LD + lc] = (w < w_end) ? b0 + (inc - dlt) : 0.0;
            if (lane == 63) base[lc] = b0 + inc;
        }
        if (w0 + OL_SLAB < w_end) load_slab(w0 + OL_SLAB);
        __syncthreads();
#pragma unroll 4
        for (int s2 = 0; s2 < OL_SLAB; s2++) {
            const double *row = W + s2 * OL_LD;
            double wi[4], wj[4];
#pragma unroll
            for (int m = 0; m < 4; m++) { wi[m] = row[a + 16 * m]; wj[m] = row[OL_CT + b + 16 * m]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fma(wi[i], wj[j], acc[i][j]);
        }
        __syncthreads();
    }
    double *out = partial + ((size_t)blockIdx.x * npairs + pair) * (OL_CT * OL_CT);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) out[(a + 16 * i) * OL_CT + (b + 16 * j)] = acc[i][j];
}

__global__ void olbm_finish_kernel(const double *__restrict__ partial, long nseg, int npairs, int p, double denom,
                                   double *__restrict__ var)
{
    const int pair = blockIdx.x;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= pair) ti++;
    const int tj = pair - ti * (ti + 1) / 2;
    for (int e = threadIdx.x; e < OL_CT * OL_CT; e += blockDim.x) {
        double s = 0.0;
        for (long g = 0; g < nseg; g++) s += partial[((size_t)g * npairs + pair) * (OL_CT * OL_CT) + e];
        const int r = ti * OL_CT + e / OL_CT, c = tj * OL_CT + e % OL_CT;
        if (r < p && c < p) {
            const double v = s / denom;          // olbm.c:74
            var[r + (size_t)p * c] = v;
            var[c + (size_t)p * r] = v;          // olbm.c:75 (symmetric)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// p <= 256: all columns in one workgroup, outer products on the fp64 matrix cores.
// var = V^T V with V[w][c] = (window sum of column c starting at w) - len*mean[c]: a GEMM whose K axis
// is the window index.  One workgroup (8 waves) walks a segment of windows in 16-window slabs through a
// double-buffered LDS image V[2][16 windows][columns]; there is one barrier per slab and no separate
// producer phase: inside the mfma loop of slab s every wave also
//   - turns the raw loads of slab s+1 (in registers since the previous iteration) into window sums:
//     x[t+len] - x[t] (olbm.c:61-62), an inclusive prefix over the 16 lanes of a DPP row (lanes = consecutive
//     windows of one column, 4 columns per wave-load), plus the running sum the thread keeps per column,
//   - writes them to the other V buffer, and
//   - issues the global loads of slab s+2,
// so HBM latency, the scan and the LDS writes all hide behind the matrix pipe.
//
// Tile schedule.  With F fragments (F*16 >= p columns) var has F(F+1)/2 distinct 16 x 16 tiles; a tile is
// an unordered fragment pair, so they are taken as the circulant family {f, (f+delta) mod F},
// delta = 0..F/2: wave w owns NB = F/8 consecutive base fragments f and all their delta < F/2 tiles, and
// the delta = F/2 pairs (each shared by two bases) are split so every wave gets at most one.  At F = 16
// that is exactly 17 tiles on each of the 8 waves.  The columns of the first F/2+1 fragments are stored
// a second time after column 16F, so (f+delta) never wraps: every fragment a wave needs in a k-step sits
// at (one base register) + (compile-time offset) and a k-step costs NB + F/2 ds_read_b64 for its MFMAs.
#include "olbm_shape.h"

template <int F>
__global__ __launch_bounds__(64 * OM_NW) void olbm_mfma_kernel(const double *__restrict__ x, long n, int p, long len,
                                                               long nwin, long seg_windows,
                                                               const double *__restrict__ centre,
                                                               double *__restrict__ partial,
                                                               double *__restrict__ colpart)
{
    using S = OmShape<F>;
    constexpr int LD = S::LD, PP = S::PP, H = S::H, NB = S::NB, NLD = S::NLD;
    extern __shared__ double om_sm[];
    double *V = om_sm;                           // [2][OM_SLAB][LD]
    double *run = om_sm + 2 * OM_SLAB * LD;      // [PP] seeds
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const long w_begin = (long)blockIdx.x * seg_windows;
    long w_end = w_begin + seg_windows;
    if (w_end > nwin) w_end = nwin;

    // seed: direct centred sum of the segment's first window (olbm.c:52-57 restarted per segment);
    // four columns per wave in flight, lanes strided over the window
    for (int c0 = wave; c0 < PP; c0 += 4 * OM_NW) {
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        if (c0 < p) {
            const double *xc[4];
            double m4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int c = c0 + u * OM_NW < p ? c0 + u * OM_NW : c0;     // clamp: the duplicate is dropped below
                xc[u] = x + (size_t)c * n + w_begin;
                m4[u] = centre[c];
            }
#pragma unroll 4
            for (long q = lane; q < len; q += 64)
#pragma unroll
                for (int u = 0; u < 4; u++) s4[u] += xc[u][q] - m4[u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            double sv = s4[u];
            for (int mm = 32; mm > 0; mm >>= 1) sv += __shfl_xor(sv, mm, 64);
            const int c = c0 + u * OM_NW;
            if (lane == 0 && c < PP) run[c] = c < p ? sv : 0.0;
        }
    }
    __syncthreads();

    om_double4 acc[S::NS];
#pragma unroll
    for (int q = 0; q < S::NS; q++) acc[q] = om_double4{0.0, 0.0, 0.0, 0.0};
    // the K index of the GEMM is a dummy: lane group g = lane/16 of k-step ks takes window row
    // 2*ks + g/2 + 8*(g%2)
    const int g = lane >> 4;
    const double *Vw = V + ((g >> 1) + 8 * (g & 1)) * LD + (lane & 15) + 16 * NB * wave;
    const bool mfma_wave = wave < S::NACT;

    // Producer of the next slab's window sums.  Every vector instruction of a wave is paid in matrix-pipe time
    // (profiles/r2_fp64_sustain_probe.txt), so the scan over the 16 windows of a slab is done where it is cheapest:
    //   loads   lane = (column cs of 4, window wl of 16), 8 bytes a lane, coalesced; slot q covers the wave's columns
    //           4q .. 4q+3 (a wave owns CW contiguous columns)
    //   phase A d = x[t+len] - x[t] (olbm.c:61-62), one subtraction per slot, written to the wave's private LDS image
    //           D[column][window]
    //   phase B lane = (column c of CW, part h of NPART): reads its WPP consecutive differences back (16-byte reads),
    //           sums them sequentially in registers -- one add per window instead of a 4-level DPP scan per element --
    //           adds the running sum of the column and the totals of the parts before it, writes V (and the duplicate
    //           columns) and keeps the new running sum.
    // Round 1 ran the scan with DPP row shifts on the (column, window) lanes: ~25 vector instructions per slot and slab
    // where this takes ~3.5.
    constexpr int CW = S::CW, NPART = S::NPART, WPP = S::WPP, DLD = S::DLD;
    double *D = om_sm + 2 * OM_SLAB * LD + PP + wave * (CW * DLD);
    const int cs = lane >> 4, wl = lane & 15;
    const unsigned lane_lo = (unsigned)((size_t)cs * n + wl), lane_hi = lane_lo + (unsigned)len;
    double hi[NLD], lo[NLD];
    const int pc = lane % CW, ph = lane / CW;            // phase B: column of the wave, part
    const int pcol = wave * CW + pc;                     // global column
    double runc = run[pcol];
    // Raw loads through one buffer descriptor per slot, built once (base = the slot's first column at w_begin, extent =
    // the rest of x): the hardware range check returns 0 past the end of x, so a load needs no branch, no per-lane
    // address arithmetic and no scalar code -- the two lane offsets advance by one slab per slab (two VALU adds a slab;
    // round 1 rebuilt the descriptors per slot and slab, ~25 dependent scalar instructions each).  What a load reads
    // for columns >= p or for windows past the end is real memory of other columns: it only reaches rows/columns of V
    // that the tail mask below and the finish kernel (row, col < p) discard.
    unsigned off_lo = lane_lo * 8u, off_hi = lane_hi * 8u;
    const long total = (long)n * p;
    __amdgpu_buffer_rsrc_t rs[NLD];
#pragma unroll
    for (int q = 0; q < NLD; q++) {
        const long first = (long)(wave * CW + q * 4) * n + w_begin;          // uniform
        long rem = (total - first) * 8;
        rem = rem < 0 ? 0 : (rem > 0xffffffffL ? 0xffffffffL : rem);
        rs[q] = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (first < total ? first : 0)), 0, (int)(unsigned)rem, 0x00020000);
    }
    auto load_slab = [&]() {                 // the slab at the current offsets; then the offsets move on
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            hi[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs[q], off_hi, 0, 0));
            lo[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs[q], off_lo, 0, 0));
        }
        off_hi += OM_SLAB * 8u;
        off_lo += OM_SLAB * 8u;
    };
    const int nwl = (int)(w_end - w_begin);                            // windows of this segment (32-bit: scalar compares)
    auto phase_a = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; q++) D[(q * 4 + cs) * DLD + wl] = hi[q] - lo[q];
    };
    double dv[WPP], base = 0.0;
    auto phase_b_read = [&]() {
        const double *dp = D + pc * DLD + ph * WPP;
#pragma unroll
        for (int k = 0; k < WPP; k += 2) {
            const double2 t = *(const double2 *)(dp + k);
            dv[k] = t.x;
            dv[k + 1] = t.y;
        }
    };
    double excl = 0.0;                       // total of the parts before this lane's (exclusive scan over NPART lanes)
    auto phase_b_prefix = [&]() {
#pragma unroll
        for (int k = 1; k < WPP; k++) dv[k] += dv[k - 1];              // inclusive prefix of the part
        if constexpr (NPART == 2) {
            const double t = __shfl(dv[WPP - 1], lane - CW, 64);
            excl = ph >= 1 ? t : 0.0;
        } else if constexpr (NPART > 2) {                             // log-step inclusive scan, then shifted by one part
            double incl = dv[WPP - 1];
#pragma unroll
            for (int d = 1; d < NPART; d <<= 1) {
                const double t = __shfl(incl, lane - d * CW, 64);
                incl += ph >= d ? t : 0.0;
            }
            const double t = __shfl(incl, lane - CW, 64);
            excl = ph >= 1 ? t : 0.0;
        }
    };
    // window sums of slab w0 -> Vdst: the window starting at w0 + i is the running sum (window w0) plus the differences
    // BEFORE i; the slab's total carries the running sum to w0 + 16.  tail: the slab reaches past w_end (uniform), its
    // windows there are zeroed
    const int dupoff = pcol < S::DUPC ? PP : PP + S::DUPC - pcol;      // duplicate column, or the row's padding slot
    const bool wave_dups = wave * CW < S::DUPC;                        // uniform: waves past the duplicated columns skip it
    // For the one-pass route (see pilot_centre_kernel): the windows 0, len, 2 len, ... tile the series, so their sums about
    // the centre add up to n * (mean - centre) less a tail of < len rows: the exact mean comes out of the window sums,
    // without a pass over x and without cancellation.  Such a window comes by once in len windows; its sum is added to
    // the column's slot of run[] (the seeds there are spent, and a column belongs to one wave, whose LDS operations
    // stay in order).  next_sel: the next such window, relative to w_begin.
    auto fresh_lane = [&]() {
        int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(l));
        return l;
    };
    if (ph == 0) run[pcol] = 0.0;
    int next_sel = (int)((len - w_begin % len) % len);
    if ((long)next_sel >= (long)nwl) next_sel = 0x7fffffff;
    auto phase_b_write = [&](int r0, double *Vdst) {                   // r0: first window of the slab, relative to w_begin
        base = runc + excl;
        double *rowp = Vdst + (ph * WPP) * LD + pcol;
        if (r0 + OM_SLAB <= nwl) {
            // (one uniform branch around the loop: left inside it, the compiler tests wave_dups once per window)
            if (wave_dups) {
#pragma unroll
                for (int k = 0; k < WPP; k++) {
                    const double v = k == 0 ? base : base + dv[k - 1];
                    rowp[k * LD] = v;
                    rowp[k * LD + dupoff] = v;
                }
            } else {
#pragma unroll
                for (int k = 0; k < WPP; k++) rowp[k * LD] = k == 0 ? base : base + dv[k - 1];
            }
        } else {                                                       // the segment's last slab (or past it)
            const int left = nwl - (r0 + ph * WPP);
            const int nv = left < 0 ? 0 : (left > WPP ? WPP : left);
#pragma unroll
            for (int k = 0; k < WPP; k++) {
                double v = k == 0 ? base : base + dv[k - 1];
                v = k < nv ? v : 0.0;
                rowp[k * LD] = v;
                rowp[k * LD + dupoff] = v;
            }
        }
        while (next_sel < r0 + OM_SLAB) {                              // uniform, once in len windows
            // read back from the slab just written (the same wave wrote it; zero past the end of the segment).  The lane
            // index is taken afresh from the hardware: nothing lane-dependent of this rare path may stay live across
            // the slab loop, which has no register to spare (a spill there reloads through the vector memory queue,
            // behind the loads for the slab after next)
            const int l2 = fresh_lane();
            if (l2 < CW) run[wave * CW + l2] += Vdst[(next_sel - r0) * LD + wave * CW + l2];
            next_sel = len >= (long)(nwl - next_sel) ? 0x7fffffff : next_sel + (int)len;
        }
        runc = __shfl(base + dv[WPP - 1], (NPART - 1) * CW + pc, 64);  // the window at r0 + 16: the next running sum
    };
    // the slab hand-over only concerns LDS: wait for this wave's LDS traffic, not for the global loads in flight
    // (__syncthreads would also drain vmcnt and expose the HBM latency of the loads for the slab after next)
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    load_slab();
    phase_a();
    phase_b_read();
    phase_b_prefix();
    phase_b_write(0, V);
    load_slab();
    __syncthreads();

    constexpr bool all_mfma = S::NACT == OM_NW;
    constexpr int NKS = OM_SLAB / 4;
    // fragments of a k-step: NFR for the circulant tiles, and for NB == 2 the two operands of the wave's delta = H tile,
    // read at a wave-dependent offset (a select between registers would put vector instructions inside the MFMA group)
    constexpr int NFX = S::NFR + (NB == 2 ? 2 : 0);
    const int xoff = NB == 2 ? 16 * S::extra_base(wave) : 0;
    auto read_frags = [&](const double *Vc, int ks, double *fr) {
#pragma unroll
        for (int i = 0; i < S::NFR; i++) fr[i] = Vc[2 * ks * LD + 16 * i];
        if constexpr (NB == 2) {
            fr[S::NFR] = Vc[2 * ks * LD + xoff];
            fr[S::NFR + 1] = Vc[2 * ks * LD + xoff + 16 * H];
        }
    };
    auto mfma_step = [&](const double *fr) {
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int dlt = 0; dlt < H; dlt++)
                acc[b * H + dlt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[b], fr[b + dlt], acc[b * H + dlt], 0, 0, 0);
        if constexpr (NB == 2)       // every wave has one
            acc[NB * H] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[S::NFR], fr[S::NFR + 1], acc[NB * H], 0, 0, 0);
        else                         // waves without a delta = H tile compute one nobody reads
            acc[NB * H] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[0], fr[H], acc[NB * H], 0, 0, 0);
    };
    int cur = 0;
    for (int r0 = 0; r0 < nwl; r0 += OM_SLAB, cur ^= 1) {
        const double *Vc = Vw + cur * (OM_SLAB * LD);
        double *Vn = V + (cur ^ 1) * (OM_SLAB * LD);
        // A slab = 4 k-steps of NS MFMAs, issued as whole groups; the producer's work for the next slab sits BETWEEN the
        // groups as clusters (a switch between the fp64 matrix and vector instruction costs ~6 cycles; the even
        // interleave of round 1 paid that 17 times a k-step), one step of its dependent chain per cluster so that every
        // LDS round trip has an MFMA group to complete in:
        //   after k-step 0: differences -> D, read-back issued, global loads of the slab after next (a slab and a half
        //                   ahead of their use)
        //   after k-step 1: prefix sums, part totals requested     after 2: base + sums -> V, next running sum
        // The fragments of k-step ks+1 are requested just before the MFMAs of ks.
        double fr[2][NFX];
        const bool do_mfma = all_mfma || mfma_wave;
        if (do_mfma) read_frags(Vc, 0, fr[0]);
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) {
            if (do_mfma && ks + 1 < NKS) read_frags(Vc, ks + 1, fr[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (do_mfma) mfma_step(fr[ks & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) {
                phase_a();
                phase_b_read();
                load_slab();                     // the slab after next: consumed by phase A of the next slab
            }
            if (ks == 1) phase_b_prefix();
            if (ks == 2) phase_b_write(r0 + OM_SLAB, Vn);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
    }
    // colpart[segment][column]: the windows at multiples of len
    {
        const int l2 = fresh_lane();
        if (l2 < CW) colpart[(size_t)blockIdx.x * PP + wave * CW + l2] = run[wave * CW + l2];
    }
    // partial[segment][wave][slot][reg][lane]
    if (mfma_wave) {
        double *out = partial + ((size_t)blockIdx.x * S::NACT + wave) * S::NS * 256;
#pragma unroll
        for (int q = 0; q < S::NS; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) out[q * 256 + r * 64 + lane] = acc[q][r];
    }
}

// one workgroup per (wave, slot) tile: sums the segments in order, scales (olbm.c:74) and mirrors (olbm.c:75).
// accumulator layout of v_mfma_f64_16x16x4: register r of lane l holds D[row = l/16 + 4r][col = l % 16]
// One-pass route (means not supplied, long series): the main kernel centres the window sums about a PROVISIONAL centre
// c (pilot_centre_kernel: a thin, evenly spread sample of the series) and collects the column sums S' of its window
// sums V' = W - len*c, and the sums over the windows 0, len, 2 len, ... alone, which tile the series and so give the
// exact mean m = c + (their sum + a tail of < len rows) / n.  With delta = m - c
// and V = W - len*m = V' - len*delta, the matrix olbm.c:52-75 accumulates is
//     sum_w V V^T = sum_w V' V'^T - len (S' delta^T + delta S'^T) + nwin len^2 delta delta^T,
// so the separate pass over x that computed the means first (a quarter of the HBM traffic of the whole call) is gone.
// The correction cancels against the main term by a factor 1 + nwin len^2 delta_c^2 / (sum_w V_c^2); the caller checks
// that factor and, should a sample ever miss the mean that badly, repeats the kernel about the exact mean.
__global__ __launch_bounds__(256) void pilot_centre_kernel(const double *__restrict__ x, long n, int runs,
                                                           double *__restrict__ centre)
{
    __shared__ double red[4];
    const double *xc = x + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (int r = threadIdx.x; r < runs; r += 256) {
        const long t0 = ((long)r * (n - 8)) / (runs - 1);
#pragma unroll
        for (int k = 0; k < 8; k++) s += xc[t0 + k];
    }
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) centre[blockIdx.x] = (((red[0] + red[1]) + red[2]) + red[3]) / (8.0 * runs);
}

// one workgroup per column: delta = mean - centre from the tiling windows and the < len rows after the last of them;
// vtot = S', the column sum of all window sums about the centre = sum_t cnt(t) (x_t - c) with cnt(t) = the number of
// windows that hold row t = min(t + 1, len, n - t, nwin): len n delta less what the rows near the two ends fall short
__global__ __launch_bounds__(256) void olbm_centre_finish_kernel(const double *__restrict__ x, long n, long len,
                                                                 const double *__restrict__ colpart, long nseg, int PP,
                                                                 const double *__restrict__ centre,
                                                                 double *__restrict__ mean, double *__restrict__ delta,
                                                                 double *__restrict__ vtot)
{
    __shared__ double red[2][4];
    const int col = blockIdx.x;
    const double c = centre[col];
    const long nwin = n - len + 1;
    double st = 0.0, sv = 0.0;
    for (long g = threadIdx.x; g < nseg; g += 256) st += colpart[(size_t)g * PP + col];
    const double *xc = x + (size_t)col * n;
    for (long t = (n / len) * len + threadIdx.x; t < n; t += 256) st += xc[t] - c;
    // rows held by fewer than len windows: the first and the last len rows (all rows if they overlap)
    const long head = len < n ? len : n, tail0 = n - len > head ? n - len : head;
    for (long i = threadIdx.x; i < head + (n - tail0); i += 256) {
        const long t = i < head ? i : tail0 + (i - head);
        long cnt = t + 1 < len ? t + 1 : len;
        cnt = n - t < cnt ? n - t : cnt;
        cnt = nwin < cnt ? nwin : cnt;
        sv += (double)(len - cnt) * (xc[t] - c);
    }
    for (int m = 32; m > 0; m >>= 1) {
        st += __shfl_xor(st, m, 64);
        sv += __shfl_xor(sv, m, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = st;
        red[1][threadIdx.x >> 6] = sv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nd = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];        // n * delta
        const double d = nd / (double)n;
        delta[col] = d;
        mean[col] = c + d;                                             // olbm.c:42
        vtot[col] = (double)len * nd - (((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]);
    }
}

template <int F>
__global__ __launch_bounds__(256) void olbm_mfma_finish_kernel(const double *__restrict__ partial, long nseg, int p,
                                                               double denom, double *__restrict__ var,
                                                               const double *__restrict__ delta,
                                                               const double *__restrict__ vtot, double len, double nwin)
{
    using S = OmShape<F>;
    const int w = blockIdx.x / S::NS, q = blockIdx.x % S::NS, e = threadIdx.x, r = e >> 6, l = e & 63;
    int b, dlt;
    if (q < S::NB * S::H) { b = q / S::H; dlt = q % S::H; }
    else {
        if (!S::has_extra(w)) return;
        b = S::extra_base(w); dlt = S::H;
    }
    const int fa = S::NB * w + b, fb = (fa + dlt) % F;
    double s = 0.0;
    for (long gseg = 0; gseg < nseg; gseg++) s += partial[(((size_t)gseg * S::NACT + w) * S::NS + q) * 256 + e];
    const int row = 16 * fa + (l >> 4) + 4 * r, col = 16 * fb + (l & 15);
    if (row < p && col < p) {
        if (delta) {
            // (row, col) and (col, row) of a diagonal tile are computed by different lanes and must agree bit for bit
            // whatever the compiler contracts: both evaluate the expression of the ordered pair (a <= b)
            const int a = row < col ? row : col, b = row < col ? col : row;
            const double da = delta[a], db = delta[b];
            s = s - len * (vtot[a] * db + da * vtot[b]) + nwin * len * len * da * db;
        }
        const double v = s / denom;
        var[row + (size_t)p * col] = v;
        var[col + (size_t)p * row] = v;
    }
}

// dcentre: what the window sums are centred about.  dmean_out == nullptr: dcentre is the exact mean (supplied, or from
// colsum_kernel).  Otherwise dcentre is provisional and the exact means go to dmean_out, delta = mean - centre to ddelta.
template <int F>
static int launch_olbm_mfma(const double *x, long n, int p, long len, long nwin, const double *dcentre, double *dmean_out,
                            double *ddelta, double *dvar, DeviceCtx *dc)
{
    using S = OmShape<F>;
    hipStream_t s = dc->stream;
    auto kern = olbm_mfma_kernel<F>;
    if (S::lds_bytes > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)S::lds_bytes));
    // segments: exactly one round of resident workgroups.  Every workgroup does the same work (a seed of
    // len loads per column, then 2 loads per window and column), so one round of equal segments finishes
    // together; more, shorter segments would only repeat the seed.
    int bpc = 1;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, reinterpret_cast<const void *>(kern), 64 * OM_NW,
                                                         S::lds_bytes));
    if (bpc < 1) bpc = 1;
    const long resident = (long)bpc * dc->num_cu;
    long seg = (nwin + resident - 1) / resident;
    if (seg < 16L * OM_SLAB) seg = 16L * OM_SLAB;
    seg = ((seg + OM_SLAB - 1) / OM_SLAB) * OM_SLAB;
    const long nseg = (nwin + seg - 1) / seg;
    DevBuf part, colpart, vtot;
    HIP_TRY(part.alloc((size_t)nseg * S::NACT * S::NS * 256 * 8));
    HIP_TRY(colpart.alloc((size_t)nseg * S::PP * 8));
    HIP_TRY(vtot.alloc((size_t)p * 8));
    timing_reset("olbm_mfma_kernel");
    timing_begin(s);
    hipLaunchKernelGGL(kern, dim3((unsigned)nseg), dim3(64 * OM_NW), S::lds_bytes, s, x, n, p, len, nwin, seg, dcentre,
                       part.as<double>(), colpart.as<double>());
    timing_end(s);
    HIP_TRY(hipGetLastError());
    if (dmean_out) {
        hipLaunchKernelGGL(olbm_centre_finish_kernel, dim3((unsigned)p), dim3(256), 0, s, x, n, len,
                           colpart.as<double>(), nseg, S::PP, dcentre, dmean_out, ddelta, vtot.as<double>());
        HIP_TRY(hipGetLastError());
    }
    const double denom = (double)nwin * (double)n * (double)len;
    hipLaunchKernelGGL(olbm_mfma_finish_kernel<F>, dim3((unsigned)(S::NACT * S::NS)), dim3(256), 0, s,
                       part.as<double>(), nseg, p, denom, dvar, dmean_out ? ddelta : (const double *)nullptr,
                       vtot.as<double>(), (double)len, (double)nwin);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));       // part is released on return
    return MCMCHIP_OK;
}

int olbm_pilot_centre(const double *x, long n, int p, double *dcentre, hipStream_t s)
{
    hipLaunchKernelGGL(pilot_centre_kernel, dim3((unsigned)p), dim3(256), 0, s, x, n, OL_PILOT_RUNS, dcentre);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

int olbm_exact_means(const double *x, long n, int p, double *dmean, hipStream_t s)
{
    const long chunk = colsum_chunk(n);
    const long nchunk = (n + chunk - 1) / chunk;
    DevBuf cpart;
    HIP_TRY(cpart.alloc((size_t)p * nchunk * 8));
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nchunk, (unsigned)p), dim3(256), 0, s, x, n, chunk, nchunk, cpart.as<double>());
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(colmean_kernel, dim3((unsigned)((p + 63) / 64)), dim3(64), 0, s, cpart.as<double>(), nchunk, n, p, dmean);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));       // cpart is released on return
    return MCMCHIP_OK;
}

int run_olbm(const double *x, long n, int p, long len, double *mean, double *var, int nocalc, DeviceCtx *dc)
{
    hipStream_t s = dc->stream;
    if (n >= (1L << 31)) return set_error(MCMCHIP_EINVAL, "olbm: n exceeds the reference's int range (olbm.c:25)");
    if (olbm_lane_applies(n, p, len)) return run_olbm_lane(x, n, p, len, mean, var, nocalc, dc);   // 3 .. 192 columns, long windows: olbm_lane.hip
    if (olbm_small_applies(n, p, len)) return run_olbm_small(x, n, p, len, mean, var, nocalc, dc); // few columns: olbm_small.hip
    if (olbm_comb_applies(n, p, len)) return run_olbm_comb(x, n, p, len, mean, var, nocalc, dc);   // long windows: olbm_comb.hip
    DevBuf dmean, dvar, part, cpart, dcentre, ddelta;
    HIP_TRY(dmean.alloc((size_t)p * 8));
    HIP_TRY(dvar.alloc((size_t)p * p * 8));
    const long nwin = n - len + 1;
    const bool mfma_route = p <= 256 && 8 * (4 * n + len + 64) < (1L << 32);   // 32-bit byte offsets in the matrix-core kernel: 3 columns + a segment + len
    // one pass over x when the means are not supplied and the series is long enough for a second pass to matter
    const bool one_pass = mfma_route && !nocalc && n >= OL_ONEPASS_MIN_N;
    if (nocalc) {
        HIP_TRY(hipMemcpyAsync(dmean.p, mean, (size_t)p * 8, hipMemcpyHostToDevice, s));
    } else if (one_pass) {
        HIP_TRY(dcentre.alloc((size_t)p * 8));
        HIP_TRY(ddelta.alloc((size_t)p * 8));
        hipLaunchKernelGGL(pilot_centre_kernel, dim3((unsigned)p), dim3(256), 0, s, x, n, OL_PILOT_RUNS,
                           dcentre.as<double>());
        HIP_TRY(hipGetLastError());
    } else {
        const long chunk = colsum_chunk(n);
        const long nchunk = (n + chunk - 1) / chunk;
        HIP_TRY(cpart.alloc((size_t)p * nchunk * 8));
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nchunk, (unsigned)p), dim3(256), 0, s, x, n, chunk, nchunk,
                           cpart.as<double>());
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(colmean_kernel, dim3((unsigned)((p + 63) / 64)), dim3(64), 0, s, cpart.as<double>(), nchunk,
                           n, p, dmean.as<double>());
        HIP_TRY(hipGetLastError());
    }
    if (mfma_route) {
        auto launch = [&](const double *centre, double *mean_out) {
            if (p <= 64) return launch_olbm_mfma<4>(x, n, p, len, nwin, centre, mean_out, ddelta.as<double>(), dvar.as<double>(), dc);
            if (p <= 128) return launch_olbm_mfma<8>(x, n, p, len, nwin, centre, mean_out, ddelta.as<double>(), dvar.as<double>(), dc);
            return launch_olbm_mfma<16>(x, n, p, len, nwin, centre, mean_out, ddelta.as<double>(), dvar.as<double>(), dc);
        };
        int rc = one_pass ? launch(dcentre.as<double>(), dmean.as<double>()) : launch(dmean.as<double>(), nullptr);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(var, dvar.p, (size_t)p * p * 8, hipMemcpyDeviceToHost, s));
        if (!nocalc) HIP_TRY(hipMemcpyAsync(mean, dmean.p, (size_t)p * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (one_pass) {
            // digits the centre correction cancelled: 1 + nwin len^2 delta_c^2 / (sum_w V_c^2), worst column
            std::vector<double> hd((size_t)p);
            HIP_TRY(hipMemcpy(hd.data(), ddelta.p, (size_t)p * 8, hipMemcpyDeviceToHost));
            bool redo = false;
            for (int c = 0; c < p; c++) {
                const double main_term = var[c + (size_t)p * c] * (double)n;          // sum_w V_c^2 / (nwin len)
                const double corr = (double)len * hd[c] * hd[c];
                // (a delta that is rounding of the centre itself, as for a constant column, cancels nothing)
                if (!(corr <= OL_ONEPASS_MAX_LOSS * main_term) && !(fabs(hd[c]) <= 1e-14 * fabs(mean[c])))
                    redo = true;                                                        // also catches NaN
            }
            if (redo) {
                rc = launch(dmean.as<double>(), nullptr);
                if (rc) return rc;
                HIP_TRY(hipMemcpyAsync(var, dvar.p, (size_t)p * p * 8, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
            }
        }
        return MCMCHIP_OK;
    }
    const int nt = (p + OL_CT - 1) / OL_CT;
    const int npairs = nt * (nt + 1) / 2;
    // segments: about 8 workgroups per CU over all tile pairs, at least 16 slabs each
    long want_seg = (8L * dc->num_cu + npairs - 1) / npairs;
    long seg = (nwin + want_seg - 1) / want_seg;
    if (seg < 16L * OL_SLAB) seg = 16L * OL_SLAB;
    seg = ((seg + OL_SLAB - 1) / OL_SLAB) * OL_SLAB;
    const long nseg = (nwin + seg - 1) / seg;
    HIP_TRY(part.alloc((size_t)nseg * npairs * OL_CT * OL_CT * 8));
    timing_reset("olbm_kernel");
    timing_begin(s);
    hipLaunchKernelGGL(olbm_kernel, dim3((unsigned)nseg, (unsigned)npairs), dim3(256), 0, s, x, n, p, len, nwin, seg,
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

