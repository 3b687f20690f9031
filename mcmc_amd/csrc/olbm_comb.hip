// olbm_comb.hip -- olbm() (reference olbm.c:22-78) on the fp64 matrix cores for long windows: x is read ONCE.
//
// olbm_mfma_kernel (diag.hip) walks a time segment window by window and needs x[t + len] and x[t] for every step
// (olbm.c:61-62); for len * p * 8 bytes beyond what a cache holds (C5: 20 MB per workgroup) those are two HBM streams,
// 2.03x the input.  Here a workgroup walks a COMB instead: m teeth, len windows apart, swept together,
//     step (r, k):  the 16 windows  base + k*len + r .. + 15            k = 0 .. m-1 (inner),  r = 0, 16, .. (outer)
// so the rows x[base + (k+1)*len + r ..] that tooth k adds are the rows tooth k+1 drops in the very next step: they are
// loaded once and stay in registers for one step -- (m+1)/m loads per element instead of 2.
//
// A tooth would need its first window sum as a seed (a direct sum of len rows, olbm.c:52-57): m seeds of len rows per
// sweep of len steps would bring the second stream back.  The seeds are not needed: with s the (unknown) window sum at
// the start of a piece (a run of windows of one tooth) and D_w the sum of the differences since,
//     sum_w (s + D_w)(s + D_w)^T = sum_w D_w D_w^T  +  cnt s s^T + s d^T + d s^T,        d = sum_w D_w,
// the matrix cores accumulate sum D D^T (which involves neither seed nor centre), the producer keeps d and the
// piece's last running sum e per column, and the pieces are chained afterwards: s_{j+1} = s_j + e_j from ONE direct
// sum s_0 per column (olbm_chain kernels), the rank-two terms being a small GEMM over the pieces
// (olbm_comb_corr_kernel).  The windows 0, len, 2 len, ... tile the series, so the exact column means come out of
// the chain as well (olbm.c:38-43 without its pass over x).
//
// Per-tooth state (running sum, d) lives in LDS ([M][columns] pairs); everything else -- tile schedule, LDS image of
// the window sums, MFMA groups with the producer's clusters between them -- is olbm_mfma_kernel's.
#include "kernels.h"
#include "olbm_shape.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

template <int F> struct CombShape {
    using S = OmShape<F>;
    static constexpr int M = F == 16 ? 5 : 8;            // teeth whose state fits beside the window-sum image
    static constexpr size_t lds_bytes = ((size_t)2 * OM_SLAB * S::LD + (size_t)S::PP * S::DLD + (size_t)M * S::PP * 2) * 8;
};

template <int F>
__global__ __launch_bounds__(64 * OM_NW) void olbm_comb_kernel(const double *__restrict__ x, long n, int p, long len,
                                                               const OlbmComb *__restrict__ combs,
                                                               const int *__restrict__ first,
                                                               const int *__restrict__ nsteps,
                                                               double *__restrict__ partial, double *__restrict__ pd,
                                                               double *__restrict__ pe, int exper)
{
    using S = OmShape<F>;
    constexpr int LD = S::LD, PP = S::PP, H = S::H, NB = S::NB, NLD = S::NLD, M = CombShape<F>::M;
    constexpr int CW = S::CW, NPART = S::NPART, WPP = S::WPP, DLD = S::DLD;
    extern __shared__ double oc_sm[];
    double *V = oc_sm;                                        // [2][OM_SLAB][LD]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    double *D = oc_sm + 2 * OM_SLAB * LD + wave * (CW * DLD);  // per wave [CW][DLD]
    double2 *st = (double2 *)(oc_sm + 2 * OM_SLAB * LD + PP * DLD);   // [M][PP] {running sum, sum of the window sums}

    om_double4 acc[S::NS];
#pragma unroll
    for (int q = 0; q < S::NS; q++) acc[q] = om_double4{0.0, 0.0, 0.0, 0.0};
    const int g = lane >> 4;
    const double *Vw = V + ((g >> 1) + 8 * (g & 1)) * LD + (lane & 15) + 16 * NB * wave;
    const bool mfma_wave = wave < S::NACT;

    // the walk: combs [first[b], first[b+1]) of this workgroup, flattened into steps
    struct Cur { int ci, k, r0, m, rlen, piece0, pstride; unsigned ebase, eoff; };
    const int ci_end = first[blockIdx.x + 1];
    const int total_steps = nsteps[blockIdx.x];
    const unsigned ulen = (unsigned)len;
    auto load_comb = [&](Cur &c) {
        const OlbmComb *cb = combs + c.ci;
        c.m = __builtin_amdgcn_readfirstlane(cb->m);
        c.rlen = __builtin_amdgcn_readfirstlane(cb->rlen);
        c.piece0 = __builtin_amdgcn_readfirstlane(cb->piece0);
        c.pstride = __builtin_amdgcn_readfirstlane(cb->pstride);
        c.ebase = (unsigned)__builtin_amdgcn_readfirstlane((int)cb->base);
        c.k = 0;
        c.r0 = 0;
        c.eoff = c.ebase;
    };
    auto advance = [&](Cur &c) {
        c.k++;
        c.eoff += ulen;
        if (c.k >= c.m) {
            c.k = 0;
            c.r0 += OM_SLAB;
            c.eoff = c.ebase + (unsigned)c.r0;
            if (c.r0 >= c.rlen) {
                c.ci++;
                if (c.ci < ci_end) load_comb(c);
                else {                       // past the end: steps that produce zeros and load what was loaded before
                    c.m = 1;
                    c.rlen = 0;
                    c.r0 = 0;
                    c.eoff = c.ebase;
                }
            }
        }
    };
    Cur L;
    L.ci = first[blockIdx.x];
    load_comb(L);

    // loads: lane = (column cs of 4, window wl of 16); one buffer descriptor per 4-column slot, built once (base = the
    // slot's first column, extent = the rest of x: the hardware range check returns 0 past the end of x); a step's
    // offset is uniform and is added to the lane's own in one vector add
    const int cs = lane >> 4, wl = lane & 15;
    const unsigned lane_lo8 = (unsigned)((size_t)cs * n + wl) * 8u, len8 = ulen * 8u;
    const long total = (long)n * p;
    __amdgpu_buffer_rsrc_t rs[NLD];
#pragma unroll
    for (int q = 0; q < NLD; q++) {
        const long fst = (long)(wave * CW + q * 4) * n;                       // uniform
        long rem = (total - fst) * 8;
        rem = rem < 0 ? 0 : (rem > 0xffffffffL ? 0xffffffffL : rem);
        rs[q] = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (fst < total ? fst : 0)), 0, (int)(unsigned)rem, 0x00020000);
    }
    double cur[NLD], prev[NLD];
    auto load_slab = [&](const Cur &c) {
        // (exper & 1: a timing experiment -- every step reads the same rows, which then come from the cache; results are wrong)
        const unsigned vo = lane_lo8 + ((exper & 1) ? 0u : c.eoff * 8u);
        if (c.k == 0) {                      // tooth 0 has no neighbour below: its dropped rows are loaded too
#pragma unroll
            for (int q = 0; q < NLD; q++) prev[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs[q], vo, 0, 0));
        }
        const unsigned vh = vo + len8;
#pragma unroll
        for (int q = 0; q < NLD; q++) cur[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs[q], vh, 0, 0));
    };

    const int pc = lane % CW, ph = lane / CW;            // phase B: column of the wave, part
    const int pcol = wave * CW + pc;                     // global column
    const int dupoff = pcol < S::DUPC ? PP : PP + S::DUPC - pcol;
    const bool wave_dups = wave * CW < S::DUPC;
    if (ph == NPART - 1)
#pragma unroll
        for (int k = 0; k < M; k++) st[k * PP + pcol] = double2{0.0, 0.0};

    // producer of one step (see olbm_mfma_kernel): A differences -> the wave's LDS image; B each (column, part) lane sums
    // its WPP consecutive differences, joins the parts and writes the window sums; the tooth's {running sum, d} pair
    // comes from and goes back to LDS
    auto phase_a = [&](const Cur &c) {
        const int nvalid = c.rlen - c.r0;                // uniform; >= 16: a full step
        if (nvalid >= OM_SLAB) {
#pragma unroll
            for (int q = 0; q < NLD; q++) D[(q * 4 + cs) * DLD + wl] = cur[q] - prev[q];
        } else {                                         // the last windows of a piece (or a step past the end): the
#pragma unroll                                           // differences beyond it count as zero
            for (int q = 0; q < NLD; q++) D[(q * 4 + cs) * DLD + wl] = wl < nvalid ? cur[q] - prev[q] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NLD; q++) prev[q] = cur[q];
    };
    double dv[WPP];
    double run = 0.0;
    auto phase_b_read = [&](const Cur &c) {
        const double *dp = D + pc * DLD + ph * WPP;
#pragma unroll
        for (int k = 0; k < WPP; k += 2) {
            const double2 t = *(const double2 *)(dp + k);
            dv[k] = t.x;
            dv[k + 1] = t.y;
        }
        run = st[c.k * PP + pcol].x;
    };
    // after the prefix step: base = the window sum the part starts from; tot = the 16 window sums of the step added up
    // (at the last part's lane)
    double base = 0.0, tot = 0.0;
    auto phase_b_prefix = [&]() {
        double excl, sd;
#pragma unroll
        for (int k = 1; k < WPP; k++) dv[k] += dv[k - 1];              // inclusive prefix of the part
        double tl = 0.0;                                               // the part's window sums less WPP * (their base)
#pragma unroll
        for (int k = 0; k < WPP - 1; k++) tl += dv[k];
        if constexpr (NPART == 2) {
            const double t1 = __shfl(dv[WPP - 1], lane - CW, 64), t2 = __shfl(tl, lane - CW, 64);
            excl = ph >= 1 ? t1 : 0.0;
            sd = ph >= 1 ? (tl + t2) + (double)WPP * t1 : tl;
        } else {
            double incl = dv[WPP - 1];
#pragma unroll
            for (int d = 1; d < NPART; d <<= 1) {
                const double t = __shfl(incl, lane - d * CW, 64);
                incl += ph >= d ? t : 0.0;
            }
            const double t = __shfl(incl, lane - CW, 64);
            excl = ph >= 1 ? t : 0.0;
            double gi = tl + (double)WPP * excl;                       // summed over the parts: at the last part's lane
#pragma unroll
            for (int d = 1; d < NPART; d <<= 1) {
                const double u = __shfl(gi, lane - d * CW, 64);
                gi += ph >= d ? u : 0.0;
            }
            sd = gi;
        }
        base = run + excl;
        tot = fma((double)OM_SLAB, run, sd);
    };
    auto phase_b_write = [&](const Cur &c, double *Vdst) {
        const int nvalid = c.rlen - c.r0;
        const double dsum = st[c.k * PP + pcol].y;                     // needed at the end of this cluster only
        double *rowp = Vdst + (ph * WPP) * LD + pcol;
        if (nvalid >= OM_SLAB) {
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
        } else {
            const int left = nvalid - ph * WPP;
#pragma unroll
            for (int k = 0; k < WPP; k++) {
                double v = k == 0 ? base : base + dv[k - 1];
                v = k < left ? v : 0.0;
                rowp[k * LD] = v;
                rowp[k * LD + dupoff] = v;
            }
        }
        if (ph == NPART - 1) {
            // the differences beyond the piece are zero, so the running sum after the step is the one after the piece,
            // and the (16 - nvalid) window sums beyond it, all equal to it, are taken out of the step's total
            double nrun = base + dv[WPP - 1];
            const int pad = nvalid >= OM_SLAB ? 0 : (nvalid > 0 ? OM_SLAB - nvalid : OM_SLAB);
            double ds = dsum + fma(-(double)pad, nrun, tot);
            if (nvalid > 0 && nvalid <= OM_SLAB) {                     // uniform: the piece ends here
                const size_t o = (size_t)(c.piece0 + c.k * c.pstride) * PP + pcol;
                pe[o] = nrun;
                pd[o] = ds;
                nrun = 0.0;
                ds = 0.0;
            }
            st[c.k * PP + pcol] = double2{nrun, ds};
        }
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    Cur P = L;
    load_slab(L);
    advance(L);
    phase_a(P);
    phase_b_read(P);
    phase_b_prefix();
    phase_b_write(P, V);
    P = L;
    load_slab(L);
    advance(L);
    __syncthreads();

    constexpr bool all_mfma = S::NACT == OM_NW;
    constexpr int NKS = OM_SLAB / 4;
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
        if constexpr (NB == 2)
            acc[NB * H] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[S::NFR], fr[S::NFR + 1], acc[NB * H], 0, 0, 0);
        else
            acc[NB * H] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[0], fr[H], acc[NB * H], 0, 0, 0);
    };
    // Where the step's barrier stands decides what the waves of a CU do to each other.  They run the same code on the
    // same clock: with one placement for all, all eight sit in their producer clusters at the same time (LDS round trips,
    // the wait for the writes, the barrier itself) and the matrix pipes idle -- a quarter of the kernel's cycles by the
    // counters.  So half of the waves (2, 3, 6, 7) take the barrier right behind their writes (cluster 2) and issue their
    // last MFMA group after it; the others issue that group first and take the barrier at the end of the step.  The
    // barrier then holds the first half one MFMA group behind the second, for good, and one half's clusters fall under
    // the other half's MFMA groups.  Which half matters (same box, ms at C5; mask of the early waves): 0xCC 113.6,
    // 0x66 113.3, 0x33 113.7, 0xC0 113.3, 0xAA 117.1, all waves alike 119.8, 0xF0 (one wave of every SIMD, w and w + 4, in
    // each half) 119.3; a third phase (barrier in the middle of the last group) did not add to it.  Either way a wave has received all fragments of the current image and written its share of the
    // next one before it arrives, and requests the next image's fragments only after it leaves: no race.
    int cbuf = 0;
    double fr[2][NFX];
    const bool do_mfma = all_mfma || mfma_wave;
    // timing experiments: exper & 2: one placement for all waves; exper >> 8: mask of the early waves
    const bool early = (exper & 2) ? true : ((exper >> 8) ? ((exper >> (8 + wave)) & 1) != 0 : (wave & 2) != 0);
    if (do_mfma) read_frags(Vw, 0, fr[0]);
    for (int s = 0; s < total_steps; s++, cbuf ^= 1) {
        const double *Vc = Vw + cbuf * (OM_SLAB * LD);
        double *Vn = V + (cbuf ^ 1) * (OM_SLAB * LD);
        Cur Pn = P;
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) {
            if (do_mfma && ks + 1 < NKS) read_frags(Vc, ks + 1, fr[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (do_mfma) mfma_step(fr[ks & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) {
                phase_a(P);                      // step s+1
                phase_b_read(P);
                Pn = L;
                load_slab(L);                    // step s+2
                advance(L);
            }
            if (ks == 1) phase_b_prefix();
            if (ks == 2) {
                phase_b_write(P, Vn);
                if (early) lds_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!early) lds_barrier();
        if (do_mfma) read_frags(Vw + (cbuf ^ 1) * (OM_SLAB * LD), 0, fr[0]);
        P = Pn;
    }
    if (mfma_wave) {
        double *out = partial + ((size_t)blockIdx.x * S::NACT + wave) * S::NS * 256;
#pragma unroll
        for (int q = 0; q < S::NS; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) out[q * 256 + r * 64 + lane] = acc[q][r];
    }
}

// ---------------------------------------------------------------------------------------------
// after the walk: chain the pieces, means, rank-two terms, scaling
// ---------------------------------------------------------------------------------------------
// s0 = sum over the first window of (x - centre) (olbm.c:52-57, once per column), tail = the same over the rows after the
// last window that starts at a multiple of len
__global__ __launch_bounds__(1024) void olbm_s0_kernel(const double *__restrict__ x, long n, long len, long ntile,
                                                       const double *__restrict__ centre, double *__restrict__ s0,
                                                       double *__restrict__ tail)
{
    // one workgroup of 16 waves per column, loads in batches of 8: with 256 threads and one load at a time a window of
    // 1e6 rows took 4 ms of a 7 ms call
    __shared__ double red[2][16];
    const int col = blockIdx.x;
    const double *xc = x + (size_t)col * n;
    const double c = centre[col];
    double a = 0.0, b = 0.0;
#pragma unroll 8
    for (long t = threadIdx.x; t < len; t += 1024) a += xc[t] - c;
#pragma unroll 8
    for (long t = ntile * len + threadIdx.x; t < n; t += 1024) b += xc[t] - c;
    for (int m = 32; m > 0; m >>= 1) {
        a += __shfl_xor(a, m, 64);
        b += __shfl_xor(b, m, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0.0, sb = 0.0;
        for (int w = 0; w < 16; w++) {
            sa += red[0][w];
            sb += red[1][w];
        }
        s0[col] = sa;
        tail[col] = sb;
    }
}

int olbm_first_window_sums(const double *x, long n, int p, long len, const double *dcentre, double *s0, double *tail, hipStream_t s)
{
    hipLaunchKernelGGL(olbm_s0_kernel, dim3((unsigned)p), dim3(1024), 0, s, x, n, len, n / len, dcentre, s0, tail);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

// pieces in time order: s_{j+1} = s_j + e_j.  Two levels: sums of chunks of pieces, then every chunk rewrites its e_j as the
// exclusive prefix E_j (window sum at the piece's start less the one at window 0) and sums the E_j of the pieces that
// start a tooth (windows at multiples of len).  thread = column
#define OC_CHUNK 64
__global__ __launch_bounds__(256) void olbm_chain_sum_kernel(const double *__restrict__ pe, int npieces, int PP,
                                                             double *__restrict__ csum)
{
    const int col = threadIdx.x, j0 = blockIdx.x * OC_CHUNK;
    if (col >= PP) return;
    const int j1 = j0 + OC_CHUNK < npieces ? j0 + OC_CHUNK : npieces;
    double s = 0.0;
#pragma unroll 16
    for (int j = j0; j < j1; j++) s += pe[(size_t)j * PP + col];
    csum[(size_t)blockIdx.x * PP + col] = s;
}

// csum[chunk][col] -> its exclusive prefix over the chunks, in place.  One workgroup: 256 / PP threads per column, each over
// a run of chunks (a prefix per chunk by every chunk's own workgroup was quadratic: 3000 chunks of pieces took 0.5 ms)
__global__ __launch_bounds__(256) void olbm_chain_scan_kernel(double *__restrict__ csum, int nchunk, int PP)
{
    __shared__ double part[256];
    const int t = threadIdx.x, col = t % PP, seg = t / PP, nseg = 256 / PP;      // threads beyond nseg * PP find no chunks
    const int per = (nchunk + nseg - 1) / nseg, c0 = seg * per, c1 = c0 + per < nchunk ? c0 + per : nchunk;
    double sm = 0.0;
#pragma unroll 16
    for (int c = c0; c < c1; c++) sm += csum[(size_t)c * PP + col];
    part[t] = sm;
    __syncthreads();
    double E = 0.0;
    for (int k = 0; k < seg; k++) E += part[k * PP + col];
#pragma unroll 16
    for (int c = c0; c < c1; c++) {
        const double v = csum[(size_t)c * PP + col];
        csum[(size_t)c * PP + col] = E;
        E += v;
    }
}

__global__ __launch_bounds__(256) void olbm_chain_prefix_kernel(double *__restrict__ pe, const unsigned char *__restrict__ ptile,
                                                                int npieces, int PP, const double *__restrict__ csum,
                                                                double *__restrict__ tsum)
{
    const int col = threadIdx.x, j0 = blockIdx.x * OC_CHUNK;
    if (col >= PP) return;
    const int j1 = j0 + OC_CHUNK < npieces ? j0 + OC_CHUNK : npieces;
    double E = csum[(size_t)blockIdx.x * PP + col], ts = 0.0;   // exclusive prefix over the chunks (olbm_chain_scan_kernel)
#pragma unroll 16
    for (int j = j0; j < j1; j++) {
        const double e = pe[(size_t)j * PP + col];
        pe[(size_t)j * PP + col] = E;
        if (ptile[j]) ts += E;
        E += e;
    }
    tsum[(size_t)blockIdx.x * PP + col] = ts;
}

// n (mean - centre) = sum of the window sums at 0, len, 2 len, ... (they tile the series) + the tail
// shift[c] = s0 - len * delta: what turns E_j into the centred window sum at the start of piece j
__global__ __launch_bounds__(256) void olbm_chain_mean_kernel(const double *__restrict__ tsum, int nchunk, int PP, int p,
                                                              const double *__restrict__ s0, const double *__restrict__ tail,
                                                              const double *__restrict__ centre, long n, long len, long ntile,
                                                              int want_mean, double *__restrict__ mean,
                                                              double *__restrict__ delta, double *__restrict__ shift)
{
    // the chunks' sums: 256 / PP threads per column over runs of chunks, then in run order (one thread per column walking
    // 1700 chunks took half a millisecond)
    __shared__ double part[256];
    {
        const int t = threadIdx.x, pc = t % PP, seg = t / PP, nseg = 256 / PP;
        const int per = (nchunk + nseg - 1) / nseg, c0 = seg * per, c1 = c0 + per < nchunk ? c0 + per : nchunk;
        double sm = 0.0;
        if (want_mean) {
#pragma unroll 16
            for (int c = c0; c < c1; c++) sm += tsum[(size_t)c * PP + pc];
        }
        part[t] = sm;
    }
    __syncthreads();
    const int col = threadIdx.x;
    if (col >= PP) return;
    if (col >= p) { shift[col] = 0.0; return; }
    double d = 0.0;
    if (want_mean) {
        double ts = 0.0;
        for (int k = 0; k < 256 / PP; k++) ts += part[k * PP + col];
        const double nd = ((double)ntile * s0[col] + ts) + tail[col];
        d = nd / (double)n;
        mean[col] = centre[col] + d;                                   // olbm.c:42
        delta[col] = d;
    }
    shift[col] = s0[col] - (double)len * d;
}

// the three chain steps for callers outside this file (olbm_lane.hip)
int olbm_chain_pieces(double *pe, const unsigned char *ptile, int npieces, int PP, double *csum, double *tsum, hipStream_t s)
{
    const int nchunk = (npieces + OC_CHUNK - 1) / OC_CHUNK;
    hipLaunchKernelGGL(olbm_chain_sum_kernel, dim3((unsigned)nchunk), dim3(256), 0, s, pe, npieces, PP, csum);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(olbm_chain_scan_kernel, dim3(1), dim3(256), 0, s, csum, nchunk, PP);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(olbm_chain_prefix_kernel, dim3((unsigned)nchunk), dim3(256), 0, s, pe, ptile, npieces, PP, csum, tsum);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}
int olbm_chain_chunks(int npieces) { return (npieces + OC_CHUNK - 1) / OC_CHUNK; }
int olbm_chain_means(const double *tsum, int nchunk, int PP, int p, const double *s0, const double *tail, const double *centre,
                     long n, long len, int want_mean, double *mean, double *delta, double *shift, hipStream_t s)
{
    hipLaunchKernelGGL(olbm_chain_mean_kernel, dim3(1), dim3(256), 0, s, tsum, nchunk, PP, p, s0, tail, centre, n, len, n / len,
                       want_mean, mean, delta, shift);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

// rank-two terms: sum_j cnt_j sig_j sig_j^T + sig_j d_j^T + d_j sig_j^T for the tile (w, q) of olbm_comb_kernel's
// schedule, in the accumulator layout of v_mfma_f64_16x16x4 (register r of lane l holds [row = l/16 + 4r][col = l%16]),
// over one split of the pieces.  (row, col) and (col, row) of a diagonal tile are computed by different lanes and must
// agree bit for bit: both evaluate the expression of the ordered pair.
template <int F>
__global__ __launch_bounds__(256) void olbm_comb_corr_kernel(const double *__restrict__ pe, const double *__restrict__ pd,
                                                             const int *__restrict__ pcnt, int npieces, int per_split,
                                                             const double *__restrict__ shift, double *__restrict__ corr)
{
    using S = OmShape<F>;
    constexpr int PP = S::PP;
    const int w = blockIdx.x / S::NS, q = blockIdx.x % S::NS, e = threadIdx.x, r = e >> 6, l = e & 63;
    int b, dlt;
    if (q < S::NB * S::H) { b = q / S::H; dlt = q % S::H; }
    else {
        if (!S::has_extra(w)) return;
        b = S::extra_base(w); dlt = S::H;
    }
    const int fa = S::NB * w + b, fb = (fa + dlt) % F;
    const int row = 16 * fa + (l >> 4) + 4 * r, col = 16 * fb + (l & 15);
    const int ia = row < col ? row : col, ib = row < col ? col : row;
    const double sha = shift[ia], shb = shift[ib];
    const int j0 = blockIdx.y * per_split, j1 = j0 + per_split < npieces ? j0 + per_split : npieces;
    double s = 0.0;
#pragma unroll 4
    for (int j = j0; j < j1; j++) {
        const double sa = sha + pe[(size_t)j * PP + ia], sb = shb + pe[(size_t)j * PP + ib];
        const double da = pd[(size_t)j * PP + ia], db = pd[(size_t)j * PP + ib];
        const double t = fma((double)pcnt[j], sa, da);
        s = fma(t, sb, s);
        s = fma(sa, db, s);
    }
    corr[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + e] = s;
}

template <int F>
__global__ __launch_bounds__(256) void olbm_comb_finish_kernel(const double *__restrict__ partial, long nseg,
                                                               const double *__restrict__ corr, int nsplit, int p,
                                                               double denom, double *__restrict__ var)
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
    double c = 0.0;
    for (int k = 0; k < nsplit; k++) c += corr[((size_t)k * gridDim.x + blockIdx.x) * 256 + e];
    const int row = 16 * fa + (l >> 4) + 4 * r, col = 16 * fb + (l & 15);
    if (row < p && col < p) {
        const double v = (s + c) / denom;                              // olbm.c:74
        var[row + (size_t)p * col] = v;
        var[col + (size_t)p * row] = v;                                // olbm.c:75
    }
}

// ---------------------------------------------------------------------------------------------
// host: the plan (combs per workgroup, pieces in time order) and the launches
// ---------------------------------------------------------------------------------------------
// teeth t = 0 .. Q-1 of len windows (tooth t starts at window t * len) and one of nwin % len; super-blocks of up to
// `teeth` consecutive full teeth are swept together; the (super-block, sweep position) axis is cut into `groups` shares
// of equal step count at multiples of 16 sweep positions
void plan_combs(long nwin, long len, int teeth, long groups, CombPlan &pl)
{
    struct SB { long tooth0; int m; long R; };
    std::vector<SB> sbs;
    const long Q = nwin / len, rem = nwin % len;
    for (long t0 = 0; t0 < Q; t0 += teeth) sbs.push_back({t0, (int)(Q - t0 < teeth ? Q - t0 : teeth), len});
    if (rem) sbs.push_back({Q, 1, rem});
    long total = 0;
    for (auto &sb : sbs) total += (long)sb.m * ((sb.R + OM_SLAB - 1) / OM_SLAB);
    if (groups > total / 64) groups = total / 64;            // at least 64 steps a workgroup
    if (groups < 1) groups = 1;
    const long target = (total + groups - 1) / groups;
    pl.combs.clear();
    pl.first.assign(1, 0);
    pl.steps.clear();
    std::vector<int> comb_sb;                                // super-block of each comb
    std::vector<int> nsplit(sbs.size(), 0);
    long acc = 0;
    for (size_t si = 0; si < sbs.size(); si++) {
        const SB &sb = sbs[si];
        const long units = (sb.R + OM_SLAB - 1) / OM_SLAB;   // sweep positions, 16 windows each
        long u = 0;
        while (u < units) {
            long take = (target - acc + sb.m - 1) / sb.m;
            if (take < 1) take = 1;
            if (take > units - u) take = units - u;
            const long r = u * OM_SLAB;
            OlbmComb c;
            c.base = sb.tooth0 * len + r;
            c.m = sb.m;
            c.rlen = (int)((r + take * OM_SLAB <= sb.R ? take * OM_SLAB : sb.R - r));
            c.piece0 = nsplit[si];                           // completed below
            c.pstride = 0;
            pl.combs.push_back(c);
            comb_sb.push_back((int)si);
            nsplit[si]++;
            acc += take * sb.m;
            u += take;
            if (acc >= target) {
                pl.first.push_back((int)pl.combs.size());
                pl.steps.push_back((int)acc);
                if (acc > pl.max_steps) pl.max_steps = acc;
                acc = 0;
            }
        }
    }
    if (acc > 0) {
        pl.first.push_back((int)pl.combs.size());
        pl.steps.push_back((int)acc);
        if (acc > pl.max_steps) pl.max_steps = acc;
    }
    // pieces in time order: super-block, tooth, split
    std::vector<int> sb_piece0(sbs.size());
    int np = 0;
    for (size_t si = 0; si < sbs.size(); si++) {
        sb_piece0[si] = np;
        np += sbs[si].m * nsplit[si];
    }
    pl.npieces = np;
    pl.pcnt.assign((size_t)np, 0);
    pl.ptile.assign((size_t)np, 0);
    for (size_t ci = 0; ci < pl.combs.size(); ci++) {
        OlbmComb &c = pl.combs[ci];
        const int si = comb_sb[ci], split = c.piece0;
        c.pstride = nsplit[si];
        c.piece0 = sb_piece0[si] + split;
        for (int k = 0; k < c.m; k++) {
            const int j = c.piece0 + k * c.pstride;
            pl.pcnt[(size_t)j] = c.rlen;
            pl.ptile[(size_t)j] = split == 0;
        }
    }
}

// Host-only check of the plan (no GPU call; tests/test_host_cpu.py): every window of [0, nwin) must belong to exactly one
// (comb, tooth), the pieces must be numbered in time order and tile [0, nwin) without gaps, and the workgroups' step
// counts must be balanced.  Returns the number of violations; reports the piece count and the step range.
int olbm_plan_violations(const CombPlan &pl, long nwin, long len, int teeth, int *npieces, long *min_steps, long *max_steps)
{
    long bad = 0;
    std::vector<unsigned char> seen((size_t)nwin, 0);
    std::vector<long> pstart((size_t)pl.npieces, -1);
    for (size_t ci = 0; ci < pl.combs.size(); ci++) {
        const OlbmComb &c = pl.combs[ci];
        if (c.m < 1 || c.m > teeth || c.rlen < 1) bad++;
        for (int k = 0; k < c.m; k++) {
            const int j = c.piece0 + k * c.pstride;
            if (j < 0 || j >= pl.npieces) { bad++; continue; }
            if (pstart[(size_t)j] != -1 || pl.pcnt[(size_t)j] != c.rlen) bad++;
            pstart[(size_t)j] = c.base + (long)k * len;
            if ((pl.ptile[(size_t)j] != 0) != (pstart[(size_t)j] % len == 0)) bad++;
            for (int r = 0; r < c.rlen; r++) {
                const long w = c.base + (long)k * len + r;
                if (w < 0 || w >= nwin || seen[(size_t)w]) bad++;
                else seen[(size_t)w] = 1;
            }
        }
    }
    for (long w = 0; w < nwin; w++) bad += !seen[(size_t)w];
    long at = 0;                                            // pieces in time order, contiguous
    for (int j = 0; j < pl.npieces; j++) {
        if (pstart[(size_t)j] != at) bad++;
        at = pstart[(size_t)j] + pl.pcnt[(size_t)j];
    }
    if (at != nwin) bad++;
    long lo = -1, hi = 0;
    if (pl.first.size() != pl.steps.size() + 1 || pl.first.back() != (int)pl.combs.size()) bad++;
    for (size_t g = 0; g < pl.steps.size(); g++) {
        long st = 0;
        for (int ci = pl.first[g]; ci < pl.first[g + 1]; ci++) st += (long)pl.combs[(size_t)ci].m * ((pl.combs[(size_t)ci].rlen + OM_SLAB - 1) / OM_SLAB);
        if (st != pl.steps[g]) bad++;
        if (lo < 0 || st < lo) lo = st;
        if (st > hi) hi = st;
    }
    if (npieces) *npieces = pl.npieces;
    if (min_steps) *min_steps = lo;
    if (max_steps) *max_steps = hi;
    return (int)(bad > 0x7fffffff ? 0x7fffffff : bad);
}

extern "C" int mcmchip_olbm_comb_plan_check(int64_t nwin, int64_t len, int32_t teeth, int64_t groups, int32_t *npieces,
                                            int64_t *min_steps, int64_t *max_steps)
{
    if (nwin < 1 || len < 1 || teeth < 1 || groups < 1) return -1;
    CombPlan pl;
    plan_combs((long)nwin, (long)len, teeth, (long)groups, pl);
    long lo = 0, hi = 0;
    const int bad = olbm_plan_violations(pl, (long)nwin, (long)len, teeth, npieces, &lo, &hi);
    if (min_steps) *min_steps = lo;
    if (max_steps) *max_steps = hi;
    return bad;
}

bool olbm_comb_applies(long n, int p, long len)
{
    static const char *env = getenv("MCMCHIP_OLBM_COMB");
    if (env && env[0] == '0') return false;
    // Measured at n = 1e8, len = 1e4 (ms, comb / two-stream kernel): p = 256: 113.6 / 128.3, p = 128: 45.2 / 49.2,
    // p = 64: 21.2 / 21.9, p = 32: 19.9 / 15.5 -- for few columns neither kernel is near the HBM rate (the producer, built
    // for 256 columns, is what takes the time) and the comb's extra bookkeeping loses.  MCMCHIP_OLBM_COMB=<min len>
    // lowers the window threshold and takes every p <= 256 (tests/fuzz_olbm_comb.py).
    long min_len = 1024;
    int min_p = 65;
    if (env && atol(env) > 1) { min_len = atol(env); min_p = 1; }
    const long nwin = n - len + 1;
    return p >= min_p && p <= 256 && len >= min_len && nwin >= 2 * len && 8 * (4 * n + len + 64) < (1L << 32);
}

// what a finish step needs from the walk (kept so that the finish can be repeated about another centre)
struct OlbmCombState {
    DevBuf all;
    double *part = nullptr, *pd = nullptr, *pe = nullptr, *small = nullptr, *corr = nullptr;
    long nseg = 0;
    int np = 0, nchunk = 0, nsplit = 0, per_split = 0;
    int *dpcnt = nullptr;
    unsigned char *dptile = nullptr;
};

// dcentre: what s0 and the tail are taken about.  dmean_out == nullptr: dcentre is the mean to use (supplied, or exact);
// otherwise dcentre is provisional, the exact means go to dmean_out and mean - centre to ddelta
template <int F>
static int finish_olbm_comb(const double *x, long n, int p, long len, const double *dcentre, double *dmean_out, double *ddelta,
                            double *dvar, DeviceCtx *dc, OlbmCombState *keep)
{
    using S = OmShape<F>;
    hipStream_t s = dc->stream;
    OlbmCombState &K = *keep;
    const long nwin = n - len + 1, ntile = n / len;
    const int ntiles = S::NACT * S::NS;
    double *csum = K.small, *tsum = csum + (size_t)K.nchunk * S::PP, *s0 = tsum + (size_t)K.nchunk * S::PP,
           *tail = s0 + S::PP, *shift = tail + S::PP;
    hipLaunchKernelGGL(olbm_s0_kernel, dim3((unsigned)p), dim3(1024), 0, s, x, n, len, ntile, dcentre, s0, tail);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(olbm_chain_mean_kernel, dim3(1), dim3(256), 0, s, tsum, K.nchunk, S::PP, p, s0, tail, dcentre, n, len,
                       ntile, dmean_out ? 1 : 0, dmean_out, ddelta, shift);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(olbm_comb_corr_kernel<F>, dim3((unsigned)ntiles, (unsigned)K.nsplit), dim3(256), 0, s, K.pe,
                       K.pd, K.dpcnt, K.np, K.per_split, shift, K.corr);
    HIP_TRY(hipGetLastError());
    const double denom = (double)nwin * (double)n * (double)len;
    hipLaunchKernelGGL(olbm_comb_finish_kernel<F>, dim3((unsigned)ntiles), dim3(256), 0, s, K.part, K.nseg,
                       K.corr, K.nsplit, p, denom, dvar);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

template <int F>
static int launch_olbm_comb(const double *x, long n, int p, long len, const double *dcentre, double *dmean_out,
                            double *ddelta, double *dvar, DeviceCtx *dc, OlbmCombState *keep)
{
    using S = OmShape<F>;
    using C = CombShape<F>;
    hipStream_t s = dc->stream;
    const long nwin = n - len + 1;
    auto kern = olbm_comb_kernel<F>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)C::lds_bytes));
    int bpc = 1;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, reinterpret_cast<const void *>(kern), 64 * OM_NW, C::lds_bytes));
    if (bpc < 1) bpc = 1;
    int teeth = C::M;
    if (const char *e = getenv("MCMCHIP_OLBM_TEETH")) {
        const int t = atoi(e);
        if (t >= 1 && t < teeth) teeth = t;
    }
    int exper = 0;
    if (const char *e = getenv("MCMCHIP_OLBM_EXPERIMENT")) exper = atoi(e);
    CombPlan pl;
    plan_combs(nwin, len, teeth, (long)bpc * dc->num_cu, pl);
    const long nseg = (long)pl.steps.size();
    const int np = pl.npieces, nchunk = (np + OC_CHUNK - 1) / OC_CHUNK;
    const int ntiles = S::NACT * S::NS;
    int nsplit = (8 * dc->num_cu + ntiles - 1) / ntiles;
    if (nsplit > (np + 15) / 16) nsplit = (np + 15) / 16;
    if (nsplit < 1) nsplit = 1;
    const int per_split = (np + nsplit - 1) / nsplit;
    nsplit = (np + per_split - 1) / per_split;

    // one allocation (carved on 256-byte boundaries) and one upload for the plan: a call is short enough for ten
    // hipMalloc / hipFree pairs and five copies to show
    OlbmCombState &K = *keep;
    const size_t sz_part = (size_t)nseg * S::NACT * S::NS * 256 * 8, sz_piece = (size_t)np * S::PP * 8,
                 sz_small = ((size_t)2 * nchunk + 4) * S::PP * 8, sz_corr = (size_t)nsplit * ntiles * 256 * 8;
    const size_t o_combs = 0, o_first = o_combs + pl.combs.size() * sizeof(OlbmComb), o_steps = o_first + pl.first.size() * 4,
                 o_pcnt = o_steps + pl.steps.size() * 4, o_ptile = o_pcnt + (size_t)np * 4, sz_plan = o_ptile + (size_t)np;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    HIP_TRY(K.all.alloc(up(sz_part) + 2 * up(sz_piece) + up(sz_small) + up(sz_corr) + up(sz_plan)));
    char *cursor = K.all.as<char>();
    auto carve = [&](size_t bytes) { char *r = cursor; cursor += up(bytes); return r; };
    K.part = (double *)carve(sz_part);
    K.pd = (double *)carve(sz_piece);
    K.pe = (double *)carve(sz_piece);
    K.small = (double *)carve(sz_small);
    K.corr = (double *)carve(sz_corr);
    char *pb = carve(sz_plan);
    std::vector<char> hplan(sz_plan);
    memcpy(hplan.data() + o_combs, pl.combs.data(), pl.combs.size() * sizeof(OlbmComb));
    memcpy(hplan.data() + o_first, pl.first.data(), pl.first.size() * 4);
    memcpy(hplan.data() + o_steps, pl.steps.data(), pl.steps.size() * 4);
    memcpy(hplan.data() + o_pcnt, pl.pcnt.data(), (size_t)np * 4);
    memcpy(hplan.data() + o_ptile, pl.ptile.data(), (size_t)np);
    HIP_TRY(hipMemcpyAsync(pb, hplan.data(), sz_plan, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));                       // hplan goes out of scope
    OlbmComb *dcombs = (OlbmComb *)(pb + o_combs);
    int *dfirst = (int *)(pb + o_first), *dsteps = (int *)(pb + o_steps), *dpcnt = (int *)(pb + o_pcnt);
    unsigned char *dptile = (unsigned char *)(pb + o_ptile);
    K.nseg = nseg;
    K.np = np;
    K.nchunk = nchunk;
    K.nsplit = nsplit;
    K.per_split = per_split;
    K.dpcnt = dpcnt;
    K.dptile = dptile;

    timing_reset("olbm_comb_kernel");
    timing_begin(s);
    hipLaunchKernelGGL(kern, dim3((unsigned)nseg), dim3(64 * OM_NW), C::lds_bytes, s, x, n, p, len, dcombs, dfirst, dsteps,
                       K.part, K.pd, K.pe, exper);
    timing_end(s);
    HIP_TRY(hipGetLastError());
    // e_j -> E_j in place (once: a repeat of the finish about another centre reuses them)
    double *csum = K.small, *tsum = csum + (size_t)nchunk * S::PP;
    if (int rc = olbm_chain_pieces(K.pe, dptile, np, S::PP, csum, tsum, s)) return rc;
    return finish_olbm_comb<F>(x, n, p, len, dcentre, dmean_out, ddelta, dvar, dc, keep);
}

int run_olbm_comb(const double *x, long n, int p, long len, double *mean, double *var, int nocalc, DeviceCtx *dc)
{
    hipStream_t s = dc->stream;
    DevBuf vecs;                                              // var | mean | centre | delta
    HIP_TRY(vecs.alloc(((size_t)p * p + 3 * (size_t)p) * 8));
    double *dvar = vecs.as<double>(), *dmean = dvar + (size_t)p * p, *dcentre = dmean + p, *ddelta = dcentre + p;
    // means not supplied: a long series gets them out of the chain (about a sampled centre), a short one from a pass
    // of its own, as in olbm.c:38-43
    const bool one_pass = !nocalc && n >= OL_ONEPASS_MIN_N;
    if (nocalc) HIP_TRY(hipMemcpyAsync(dmean, mean, (size_t)p * 8, hipMemcpyHostToDevice, s));
    else if (one_pass) { if (int rc = olbm_pilot_centre(x, n, p, dcentre, s)) return rc; }
    else if (int rc = olbm_exact_means(x, n, p, dmean, s)) return rc;
    OlbmCombState K;
    const double *centre = one_pass ? dcentre : dmean;
    double *mean_out = one_pass ? dmean : nullptr;
    int rc;
    if (p <= 64) rc = launch_olbm_comb<4>(x, n, p, len, centre, mean_out, ddelta, dvar, dc, &K);
    else if (p <= 128) rc = launch_olbm_comb<8>(x, n, p, len, centre, mean_out, ddelta, dvar, dc, &K);
    else rc = launch_olbm_comb<16>(x, n, p, len, centre, mean_out, ddelta, dvar, dc, &K);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(var, dvar, (size_t)p * p * 8, hipMemcpyDeviceToHost, s));
    if (!nocalc) HIP_TRY(hipMemcpyAsync(mean, dmean, (size_t)p * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (one_pass) {
        // the sampled centre enters only through s0 - len * delta (the matrix-core sums involve no centre): digits lost
        // there, 1 + len delta_c^2 / (n var_cc), worst column; beyond three the finish is repeated about the exact mean
        std::vector<double> hd((size_t)p);
        HIP_TRY(hipMemcpy(hd.data(), ddelta, (size_t)p * 8, hipMemcpyDeviceToHost));
        bool redo = false;
        for (int c = 0; c < p; c++) {
            const double main_term = var[c + (size_t)p * c] * (double)n;
            const double corr = (double)len * hd[c] * hd[c];
            if (!(corr <= OL_ONEPASS_MAX_LOSS * main_term) && !(fabs(hd[c]) <= 1e-14 * fabs(mean[c]))) redo = true;
        }
        if (redo) {
            if (p <= 64) rc = finish_olbm_comb<4>(x, n, p, len, dmean, nullptr, nullptr, dvar, dc, &K);
            else if (p <= 128) rc = finish_olbm_comb<8>(x, n, p, len, dmean, nullptr, nullptr, dvar, dc, &K);
            else rc = finish_olbm_comb<16>(x, n, p, len, dmean, nullptr, nullptr, dvar, dc, &K);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(var, dvar, (size_t)p * p * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
    }
    return MCMCHIP_OK;
}
