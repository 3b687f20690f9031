// diag.hip -- Monte Carlo standard error estimators: initseq() and olbm().
//
// initseq (reference initseq.c:7-128): the O(len x lags) double sum of lag products
// (initseq.c:24-47) is the hot loop; the monotone / convex post-processing (55-105) works on a
// few hundred numbers and stays on the host, in the reference's own sequential order.
//   autocov_kernel<NJ> computes NJ*16 consecutive lags of one column per work item on the fp64 matrix
//   cores (v_mfma_f64_4x4x4_f64, every multiply-add of the instruction a wanted lag product; see the
//   comment at the kernel).  Lags are produced speculatively in units of 64 because the stopping lag
//   (first negative Gamma, initseq.c:42-46) is data dependent.  Partial sums are reduced in a fixed order.
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
// initseq
// =============================================================================================
#define IS_B 16                 // lags per work unit (one 16-lag accumulator per lane)
#define IS_TILE 2048            // base positions per LDS tile
#define IS_TILES_PER_BLOCK 32
#define IS_MAXACC 48            // accumulators (16-lag blocks) per work item: at most 768 lags

// Lag products on the fp64 matrix cores.  v_mfma_f64_4x4x4_f64 is four independent 4x4x4 products
// (lane = q + 4*blk + 16*k: A_blk[i=q][k], B_blk[k][j=q]; result lane = j + 4*blk + 16*i; layout and a rate
// equal to the 16x16x4 form measured by tools/mfma64_4x4_probe.hip).  With
//     A_blk[i][k] = x[u + i + blk + 4k]          B_blk[k][j] = x[u + blk + 4k + 4j + 3 + L]
// entry (i, j) accumulates x[t] * x[t + lag] for lag = L + 4j + 3 - i (16 consecutive lags) over the 16 base
// positions t = u + i + (blk + 4k): every one of the 256 multiply-adds of the instruction is a wanted lag
// product, and stepping u by 16 tiles the series exactly once.  Both operands are ~20 consecutive doubles
// of one LDS image (conflict-free, mostly broadcast), and B(u + 16, L) = B(u, L + 16): a wave that walks u
// in order keeps the B fragments of NJ lag blocks in registers and reads one new A and one new B fragment per
// NJ MFMAs.  A workgroup stages a 2048-position tile in LDS, its 4 waves take 512 positions each, and the next
// tile's global loads are in flight while this one is multiplied.
// (initseq.c:24-47; the sums are the reference's up to summation order, and they do not depend on how the lags
// are grouped into work items.)
//
// Round 2.  (1) ONE work item per column and launch round: NACC accumulators (16 lags each, up to 768 lags) are
// worked off in pieces of 16 / 8 / 4 / 2 / 1 lag blocks over the SAME staged tile, so a column is streamed from
// HBM once per round whatever its lag count (items of 64 and 256 lags used to re-read it per item; the short
// columns, two thirds of C5, are HBM-bound).  (2) Every instruction of a wave that is not an MFMA is paid in
// matrix-pipe time on gfx950, and the first version spent about a third of its time on the 64-bit index
// arithmetic and range predicates of 18 guarded loads per thread and tile.  Now the tile is fetched through two
// buffer descriptors rebuilt per tile in scalar registers: the hardware range check returns 0 past the end of
// the series, the loads need no predicate and no per-lane address arithmetic beyond one precomputed byte
// offset, and every load has an in-bounds LDS slot (the image is a whole number of 256-thread rounds).  Only the
// first tile of a series, which starts 16 positions before element 0, takes guarded loads.  (3) When the
// lag-shifted operand overlaps the base tile ONE contiguous image x[a .. a + 2048 + lag0 + 16 NACC + 35) serves
// both operands (always so for the first lags of a column, the bulk of the work): half the loads and LDS writes.
template <int NACC> struct AcShape {
    static constexpr int NA = IS_TILE + 32, NB_ = IS_TILE + 16 * NACC + 32;
    static constexpr int SA = (NA + 255) / 256, SB = (NB_ + 255) / 256;      // rounds of 256 loads per operand
    static constexpr int NSLOT = SA + SB;
};

// the MFMAs of one piece: NJ lag blocks (any NJ <= 16) over this wave's 512 positions = 32 u-steps of the staged tile.
// One A and one B fragment read per u-step serve NJ MFMAs; the B fragments of the last NJ steps sit in a register ring
// that is back in phase after every NJ steps, so 32 steps are floor(32 / NJ) unrolled rounds plus a tail of 32 % NJ.
template <int NJ, int NSTEP> __device__ __forceinline__ void autocov_round(const double *pa, const double *pb, int s0, double *R, double *acc)
{
#pragma unroll
    for (int ss = 0; ss < NSTEP; ss++) {
        const int s_ = s0 + ss;
        const double fa = pa[16 * s_];
        R[(ss + NJ - 1) % NJ] = pb[16 * (s_ + NJ - 1)];
#pragma unroll
        for (int j = 0; j < NJ; j++)
            acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa, R[(ss + j) % NJ], acc[j], 0, 0, 0);
    }
    // keep the fragment reads of a step next to the MFMAs of the step before it: left alone, the scheduler hoists
    // all reads of the unrolled body to its top (64 VGPRs at NJ = 16, spills in the 768-lag kernel)
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int ss = 0; ss < NSTEP; ss++) {
        __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
}

template <int NJ> __device__ __forceinline__ void autocov_piece(const double *pa, const double *pb, double *acc)
{
    constexpr int STEPS = IS_TILE / 4 / 16;           // 32 u-steps per wave and tile
    double R[NJ];                                      // R[m % NJ] = B fragment at u + 16 m
#pragma unroll
    for (int m = 0; m < NJ - 1; m++) R[m] = pb[16 * m];
    for (int s0 = 0; s0 + NJ <= STEPS; s0 += NJ) autocov_round<NJ, NJ>(pa, pb, s0, R, acc);
    if constexpr (STEPS % NJ != 0) autocov_round<NJ, STEPS % NJ>(pa, pb, STEPS - STEPS % NJ, R, acc);
}

// NACC lag blocks as one piece up to 16, else as the fewest near-equal pieces (compile-time recursion: static
// register indices)
template <int NACC, int J0 = 0> __device__ __forceinline__ void autocov_pieces(const double *pa, const double *pb, double *acc)
{
    constexpr int REST = NACC - J0;
    if constexpr (REST > 0) {
        constexpr int NPIECE = (REST + 15) / 16;
        constexpr int NJ = (REST + NPIECE - 1) / NPIECE;
        autocov_piece<NJ>(pa, pb + 16 * J0, acc + J0);
        autocov_pieces<NACC, J0 + NJ>(pa, pb, acc);
    }
}

template <int NACC>
__global__ __launch_bounds__(256, 2) void autocov_kernel(const double *__restrict__ x, long len, const int *__restrict__ cols,
                                                      const long *__restrict__ lag_of, const int *__restrict__ unit_of,
                                                      long nblk, int img_doubles, double *__restrict__ partial)
{
    using S = AcShape<NACC>;
    constexpr int SA = S::SA, SB = S::SB, NSLOT = S::NSLOT;
    extern __shared__ double ac_lds[];
    double *P = ac_lds;                                   // [img_doubles] tile image(s)
    double(*red)[16 * NACC] = (double(*)[16 * NACC])(ac_lds + img_doubles);   // [4][16 NACC]
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    // work item = (column, 16 NACC lags) on grid.x, time range on grid.y
    const int col = cols[blockIdx.x];
    const long lag0 = lag_of[blockIdx.x];
    const double *xs = x + (size_t)col * len;
    double acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = 0.0;
    const int q = lane & 3, blk = (lane >> 2) & 3, k = lane >> 4;
    const int offA = q + blk + 4 * k, offB = blk + 4 * k + 4 * q;
    // one image for both operands when the shifted one ends inside the padded image
    const bool merged = lag0 + 3 + S::NB_ <= (long)NSLOT * 256;
    const long shiftB = merged ? (long)SA * 256 : lag0 + 3;         // element offset of the second descriptor from the tile base
    const double *pbase = merged ? P + (lag0 + 3) : P + SA * 256;   // where the lag-shifted operand starts in LDS
    const long lenA = merged ? lag0 + 3 + S::NB_ : S::NA;            // elements the first descriptor may deliver (from the tile base)
    const long lenB = merged ? lenA - (long)SA * 256 : S::NB_;       // ... and the second (from its own base)
    const int nslot = merged ? (int)((lenA + 255) / 256) : NSLOT;    // rounds actually needed (wave-uniform)

    double r[NSLOT];
    const unsigned lane_off = (unsigned)t * 8u;
    // descriptor over x[g .. g + cnt) of this column clipped to len: elements past the end come back as 0 from the
    // hardware range check.  Only the very first tile of a series starts before element 0 (g = -16): that one
    // takes guarded loads.
    auto fetch_seg = [&](long g, long cnt, int slot0, int nseg) {
        if (g < 0) {
#pragma unroll
            for (int i = 0; i < NSLOT; i++)
                if (i >= slot0 && i < slot0 + nseg && i < nslot) {
                    const long e = t + 256 * (i - slot0), idx = g + e;
                    r[i] = (e < cnt && idx >= 0 && idx < len) ? xs[idx] : 0.0;
                }
            return;
        }
        long n_el = cnt;
        if (g + n_el > len) n_el = len - g;
        if (n_el < 0) n_el = 0;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void *)(xs + (g < len ? g : 0)), 0, (int)(unsigned)(n_el * 8), 0x00020000);
#pragma unroll
        for (int i = 0; i < NSLOT; i++)
            if (i >= slot0 && i < slot0 + nseg && i < nslot)
                r[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off + (unsigned)(i - slot0) * 2048u, 0, 0));
    };
    auto fetch = [&](long a) {
        fetch_seg(a, lenA < (long)SA * 256 ? lenA : (long)SA * 256, 0, SA);
        fetch_seg(a + shiftB, lenB, SA, SB);
    };
    const long first = (long)blockIdx.y * IS_TILES_PER_BLOCK * IS_TILE - 16;
    fetch(first);
    for (int tile = 0; tile < IS_TILES_PER_BLOCK; tile++) {
        const long a = first + (long)tile * IS_TILE;
        if (a >= len) break;
        __syncthreads();                        // the previous tile's fragments are consumed
#pragma unroll
        for (int i = 0; i < NSLOT; i++)
            if (i < nslot) P[t + 256 * i] = r[i];
        __syncthreads();
        if (tile + 1 < IS_TILES_PER_BLOCK && a + IS_TILE < len) fetch(a + IS_TILE);
        autocov_pieces<NACC>(P + wave * (IS_TILE / 4) + offA, pbase + wave * (IS_TILE / 4) + offB, acc);
    }
    // the four blk groups of a wave, then the 4 waves in order
#pragma unroll
    for (int j = 0; j < NACC; j++) {
        double v = acc[j];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (blk == 0) red[wave][16 * j + 4 * q + 3 - k] = v;     // result lane = j' + 4*blk + 16*i: q = j', k = i
    }
    __syncthreads();
    for (int e = t; e < 16 * NACC; e += 256) {
        const double v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        const int unit = unit_of[blockIdx.x] + e / IS_B;
        partial[((size_t)unit * nblk + blockIdx.y) * IS_B + e % IS_B] = v;
    }
}

// sums the per-time-block partials of one 16-lag unit in a fixed order: 16 lanes per lag take the blocks
// b = part, part + 16, ... each, then the 16 part sums are added in order (reruns give the same bits)
__global__ __launch_bounds__(256) void autocov_reduce_kernel(const double *__restrict__ partial, long nblk, double *__restrict__ out)
{
    __shared__ double part_sum[16][IS_B];
    const int k = threadIdx.x % IS_B, part = threadIdx.x / IS_B;      // IS_B = 16: 16 parts x 16 lags
    const double *p = partial + (size_t)blockIdx.x * nblk * IS_B + k;
    double s = 0.0;
    for (long b = part; b < nblk; b += 16) s += p[b * IS_B];
    part_sum[part][k] = s;
    __syncthreads();
    if (part == 0) {
        double v = 0.0;
        for (int j = 0; j < 16; j++) v += part_sum[j][k];
        out[(size_t)blockIdx.x * IS_B + k] = v;
    }
}

// Host post-processing of one column's Gamma sequence (a few hundred numbers).  What initseq.c:49-105 returns:
//   Gamma.pos  the sequence as computed (truncated at the first negative term, which is stored as 0),
//   Gamma.dec  its greatest non-increasing minorant = running minimum,
//   Gamma.con  its greatest convex minorant: isotonic regression (pool adjacent violators) of the slopes of
//              Gamma.dec, integrated back from Gamma.dec[0],
//   var.*      2 * sum(Gamma.*) - gamma0.
// The pooled slopes are kept as a stack of {sum, count} blocks; block means are compared as sum / count and
// merged by adding sums, and the convex sequence is rebuilt by repeated addition of the block mean: the same
// floating-point operations in the same order as the reference, so results agree bit for bit with its arithmetic.
namespace {
struct SlopeBlock {
    double sum;     // sum of the slopes pooled into this block
    long count;     // how many
    double mean() const { return sum / count; }
};

struct GammaOut {
    double *pos, *dec, *con;
    double *var_pos, *var_dec, *var_con;
};

double twice_sum_minus(const double *v, long m, double g0)
{
    double s = 0.0;
    for (long j = 0; j < m; ++j) s += v[j];
    return 2.0 * s - g0;
}

void gamma_sequences(const double *G, long m, double g0, const GammaOut &o)
{
    if (m > 0) {
        // running minimum
        double low = G[0];
        for (long j = 0; j < m; ++j) {
            o.pos[j] = G[j];
            if (!(G[j] > low)) low = G[j];
            o.dec[j] = low;
        }
        // isotonic regression of the slopes dec[j] - dec[j-1], j = 1..m-1, by pooling adjacent violators
        std::vector<SlopeBlock> pool;
        pool.reserve((size_t)m);
        for (long j = 1; j < m; ++j) {
            pool.push_back(SlopeBlock{o.dec[j] - o.dec[j - 1], 1});
            while (pool.size() > 1 && pool.back().mean() < pool[pool.size() - 2].mean()) {
                const SlopeBlock top = pool.back();
                pool.pop_back();
                pool.back().sum += top.sum;
                pool.back().count += top.count;
            }
        }
        // integrate the pooled slopes back
        o.con[0] = o.dec[0];
        long j = 1;
        for (const SlopeBlock &b : pool) {
            const double slope = b.mean();
            for (long r = 0; r < b.count; ++r, ++j) o.con[j] = o.con[j - 1] + slope;
        }
    }
    *o.var_pos = twice_sum_minus(o.pos, m, g0);
    *o.var_dec = twice_sum_minus(o.dec, m, g0);
    *o.var_con = twice_sum_minus(o.con, m, g0);
}
}  // namespace

// one launch round of lag products: one work item per column (more only beyond 768 lags), grouped by accumulator
// count into launches; partial sums per time block, the fixed-order reduction over the time blocks, and the sums
// back on the host
namespace {
const int kAccSizes[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 32, 48};      // instantiated NACC
const int kNumAccSizes = (int)(sizeof kAccSizes / sizeof kAccSizes[0]);

struct LagRound {
    std::vector<int> unit_col;                                  // per 16-lag unit, in (column, lag) order
    std::vector<int> item_col[kNumAccSizes], item_unit[kNumAccSizes];
    std::vector<long> item_lag[kNumAccSizes];
    void clear()
    {
        unit_col.clear();
        for (int w = 0; w < kNumAccSizes; w++) { item_col[w].clear(); item_unit[w].clear(); item_lag[w].clear(); }
    }
    // at least nb units (16 lags each) of column c from lag0 on: one item of the next instantiated size, or several
    // beyond IS_MAXACC.  Returns the units actually scheduled.
    long add(int c, long lag0, long nb)
    {
        long done = 0;
        while (done < nb) {
            const long rest = nb - done;
            int w = kNumAccSizes - 1;
            if (rest < IS_MAXACC) { w = 0; while (kAccSizes[w] < rest) w++; }
            item_col[w].push_back(c);
            item_unit[w].push_back((int)unit_col.size());
            item_lag[w].push_back(lag0 + done * IS_B);
            for (int u = 0; u < kAccSizes[w]; u++) unit_col.push_back(c);
            done += kAccSizes[w];
        }
        return done;
    }
};

struct LagBuffers {
    DevBuf partial, sums, dcols, dlags, dunits;
    size_t cap_units = 0, cap_blocks = 0;
};

template <int NACC>
int launch_autocov(dim3 grid, hipStream_t s, const double *x, long len, const int *pc, const long *pl, const int *pu,
                   long nblk, bool all_first_lags, double *pp)
{
    using S = AcShape<NACC>;
    // LDS: the tile image -- when every item of the launch starts at lag 0 the two operands share one image of
    // 2048 + 16 NACC + 35 positions -- and the [4][16 NACC] cross-wave reduction
    const int img = all_first_lags ? ((3 + S::NB_ + 255) / 256) * 256 : S::NSLOT * 256;
    const size_t lds = ((size_t)img + 4 * 16 * NACC) * 8;
    auto kern = autocov_kernel<NACC>;
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, len, pc, pl, pu, nblk, img, pp);
    return MCMCHIP_OK;
}

// runs the round over time blocks [0, nblk_run) and leaves unit_col.size() * IS_B sums in `host`
int run_lag_round(const double *x, long len, long nblk_run, const LagRound &w, LagBuffers &B, std::vector<double> &host,
                  DeviceCtx *dc)
{
    hipStream_t s = dc->stream;
    const size_t nunit = w.unit_col.size();
    if (nunit > B.cap_units || (size_t)nblk_run > B.cap_blocks) {       // sizes follow the work list
        B.partial.release(); B.sums.release(); B.dcols.release(); B.dlags.release(); B.dunits.release();
        B.cap_units = nunit > B.cap_units ? nunit : B.cap_units;
        B.cap_blocks = (size_t)nblk_run > B.cap_blocks ? (size_t)nblk_run : B.cap_blocks;
        HIP_TRY(B.partial.alloc(B.cap_units * B.cap_blocks * IS_B * 8));
        HIP_TRY(B.sums.alloc(B.cap_units * IS_B * 8));
        HIP_TRY(B.dcols.alloc(B.cap_units * sizeof(int)));
        HIP_TRY(B.dlags.alloc(B.cap_units * sizeof(long)));
        HIP_TRY(B.dunits.alloc(B.cap_units * sizeof(int)));
    }
    // the item lists first, then the launches on two side streams: items of few lags are HBM-bound (one pass over the
    // column for 32-64 lags), items of many lags MFMA-bound; run side by side they fill both resources
    size_t off = 0;
    size_t offs[kNumAccSizes];
    for (int k = kNumAccSizes - 1; k >= 0; k--) {
        const size_t ni = w.item_col[k].size();
        offs[k] = off;
        if (!ni) continue;
        HIP_TRY(hipMemcpyAsync(B.dcols.as<int>() + off, w.item_col[k].data(), ni * sizeof(int), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(B.dlags.as<long>() + off, w.item_lag[k].data(), ni * sizeof(long), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(B.dunits.as<int>() + off, w.item_unit[k].data(), ni * sizeof(int), hipMemcpyHostToDevice, s));
        off += ni;
    }
    const bool overlap = getenv("MCMCHIP_INITSEQ_OVERLAP") != nullptr;   // measured at full C5 size: no gain (span 117.7 ms vs 111 ms serial), off by default
    if (overlap) {
        HIP_TRY(hipEventRecord(dc->fork_ev, s));
        for (int i = 0; i < 2; i++) HIP_TRY(hipStreamWaitEvent(dc->side[i], dc->fork_ev, 0));
    }
    for (int k = kNumAccSizes - 1; k >= 0; k--) {                 // the long items first
        const size_t ni = w.item_col[k].size();
        if (!ni) continue;
        hipStream_t ls = overlap ? dc->side[kAccSizes[k] >= 5 ? 0 : 1] : s;
        dim3 grid((unsigned)ni, (unsigned)nblk_run);
        const int *pc = B.dcols.as<int>() + offs[k], *pu = B.dunits.as<int>() + offs[k];
        const long *pl = B.dlags.as<long>() + offs[k];
        double *pp = B.partial.as<double>();
        bool first_lags = true;
        for (long l : w.item_lag[k]) first_lags = first_lags && l == 0;
        int rc = MCMCHIP_OK;
        timing_begin(ls);
        switch (kAccSizes[k]) {
#define AC_CASE(N) case N: rc = launch_autocov<N>(grid, ls, x, len, pc, pl, pu, nblk_run, first_lags, pp); break;
        AC_CASE(1) AC_CASE(2) AC_CASE(3) AC_CASE(4) AC_CASE(5) AC_CASE(6) AC_CASE(7) AC_CASE(8) AC_CASE(10) AC_CASE(12)
        AC_CASE(14) AC_CASE(16) AC_CASE(20) AC_CASE(24) AC_CASE(32) AC_CASE(48)
#undef AC_CASE
        default: rc = set_error(MCMCHIP_EINVAL, "initseq: no kernel for %d accumulators", kAccSizes[k]);
        }
        timing_end(ls);
        if (rc) return rc;
        HIP_TRY(hipGetLastError());
    }
    if (overlap)
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipEventRecord(dc->join_ev[i], dc->side[i]));
            HIP_TRY(hipStreamWaitEvent(s, dc->join_ev[i], 0));
        }
    const long total = (long)nunit * IS_B;
    static_assert(IS_B == 16, "autocov_reduce_kernel splits 256 threads into 16 parts x 16 lags");
    hipLaunchKernelGGL(autocov_reduce_kernel, dim3((unsigned)nunit), dim3(256), 0, s, B.partial.as<double>(), nblk_run,
                       B.sums.as<double>());
    HIP_TRY(hipGetLastError());
    host.resize((size_t)total);
    HIP_TRY(hipMemcpyAsync(host.data(), B.sums.p, (size_t)total * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return MCMCHIP_OK;
}

// Where will the stop rule fire?  Gamma_k of a geometrically mixing chain decays like r^k and the rule (first
// negative Gamma, initseq.c:42-46) fires once the sequence reaches its noise floor, about Gamma_0 / sqrt(len):
// the decay is read off the part of the sequence that is still well above the noise (>= 10 % of Gamma_0) and
// extrapolated.  Returns the expected number of lag PAIRS; only the amount of speculative work depends on it.
double predict_pairs(const double *G, long m, long len)
{
    if (m < 2 || !(G[0] > 0.0)) return 1.0;
    long j = 0;
    while (j + 1 < m && G[j + 1] > 0.1 * G[0] && G[j + 1] < G[j]) j++;
    if (j == 0) {
        if (!(G[1] > 0.0) || !(G[1] < G[0])) return 2.0;
        j = 1;
    }
    const double decay = std::log(G[0] / G[j]) / (double)j;        // per lag pair
    return 0.5 * std::log((double)len) / decay;
}
}  // namespace

#define IS_MAX_UNITS_PER_LAUNCH 64      // at most 1024 new lags per column and round
#define IS_PILOT_BLOCKS 32              // pilot pass: the first 32 * 65536 positions of every column

int run_initseq(const double *x, long len, long ncol, long cap, long *n_out, double *gamma0, double *Gpos,
                double *Gdec, double *Gcon, double *vpos, double *vdec, double *vcon, DeviceCtx *dc)
{
    const long half = len / 2;
    // tiles start at position -16 (zero padded), so the 4 result rows of an MFMA see every base position
    const long span = (long)IS_TILE * IS_TILES_PER_BLOCK;
    const long nblk = (len + 16 + span - 1) / span;
    if (nblk > 65535) return set_error(MCMCHIP_EINVAL, "initseq: series longer than 2^32 elements");
    // The stopping lag is data dependent, so lags are produced speculatively in units of 16.  Long series start
    // with a PILOT: 64 lags of every column over its first 2M positions only (~1 % of a pass), whose Gamma
    // sequence predicts where the stop rule will fire; the first full pass then asks each column for that many
    // lags plus a margin, so most columns need one pass over HBM.  A column that is still active afterwards gets
    // 20 % more lags per further round.  The sums themselves do not depend on any of this.
    const bool trace = getenv("MCMCHIP_TRACE") != nullptr;
    LagBuffers bufs;
    LagRound work;
    std::vector<std::vector<double>> Gam((size_t)ncol);   // Gamma_i per column, as far as computed
    std::vector<long> done((size_t)ncol, -1);            // output length once the stop rule fired
    std::vector<long> have((size_t)ncol, 0);             // lags computed so far
    std::vector<double> g0((size_t)ncol, 0.0);
    std::vector<double> guess((size_t)ncol, 0.0);        // predicted lag pairs
    std::vector<int> rounds((size_t)ncol, 0);            // full-length launches a column has taken part in
    std::vector<int> active;
    for (long c = 0; c < ncol; c++) active.push_back((int)c);
    std::vector<double> host;
    timing_reset("autocov_kernel");
    if (half == 0) active.clear();
    const long npilot = nblk >= 16 ? (nblk / 64 < 1 ? 1 : (nblk / 64 > IS_PILOT_BLOCKS ? IS_PILOT_BLOCKS : nblk / 64)) : 0;
    if (npilot > 0 && !active.empty()) {
        work.clear();
        for (int c : active) work.add(c, 0, 64 / IS_B);
        if (int rc = run_lag_round(x, len, npilot, work, bufs, host, dc)) return rc;
        std::vector<double> G(32);
        for (size_t a = 0; a < active.size(); a++) {
            const double *g = host.data() + a * 64;
            for (int j = 0; j < 32; j++) G[(size_t)j] = g[2 * j] + g[2 * j + 1];
            guess[(size_t)active[a]] = predict_pairs(G.data(), 32, len);
        }
        if (trace) fprintf(stderr, "[initseq] pilot: %zu columns x 64 lags over %ld of %ld time blocks\n", active.size(), npilot, nblk);
    }
    while (!active.empty()) {
        work.clear();
        for (int c : active) {
            const long h = have[(size_t)c];
            long nb;
            if (rounds[(size_t)c] == 0 && npilot == 0) {
                nb = 64 / IS_B;                                      // no pilot: 64 full-length lags are the prediction
            } else if (rounds[(size_t)c] == 0 || (rounds[(size_t)c] == 1 && npilot == 0)) {
                // The first pass sized by the prediction.  Measured at C5 size the stop comes at 0.7 ... 2 x the
                // prediction (median 1.18, upper decile 1.76: the rule fires in the noise).  A column of few lags is
                // HBM-bound and a second pass over it costs as much as the first, so it gets a generous margin; a
                // column of many lags is MFMA-bound, every extra lag costs, and a further round is cheap by comparison.
                const double p = npilot > 0 ? guess[(size_t)c] : predict_pairs(Gam[(size_t)c].data(), (long)Gam[(size_t)c].size(), len);
                const double f = p <= 16.0 ? 1.7 : (p <= 40.0 ? 1.4 : (p <= 150.0 ? 1.15 : 1.05));
                const double want = 2.0 * p * f + 4.0 - (double)h;
                nb = want > 0.0 ? (long)std::ceil(want / IS_B) : 1;
            } else {
                nb = (long)std::ceil(0.2 * (double)h / IS_B);      // overshoot of the stop rule into the noise
            }
            if (nb < 1) nb = 1;
            rounds[(size_t)c]++;
            if (nb > IS_MAX_UNITS_PER_LAUNCH) nb = IS_MAX_UNITS_PER_LAUNCH;
            const long need = (2 * half - h + IS_B - 1) / IS_B;    // never past lag 2*half
            if (nb > need) nb = need;
            work.add(c, h, nb);
        }
        const long nitem = (long)work.unit_col.size();         // units
        if (trace)
        {
            fprintf(stderr, "[initseq] round: %zu active columns, %ld units of %d lags, %ld time blocks; items x lags:", active.size(), nitem, IS_B, nblk);
            for (int k = 0; k < kNumAccSizes; k++)
                if (!work.item_col[k].empty()) fprintf(stderr, " %zu x %d", work.item_col[k].size(), 16 * kAccSizes[k]);
            fprintf(stderr, "\n");
        }
        if (int rc = run_lag_round(x, len, nblk, work, bufs, host, dc)) return rc;
        const std::vector<int> &item_col_flat = work.unit_col;
        std::vector<int> next;
        long it = 0;
        for (int c : active) {
            long nb = 0;
            while (it + nb < nitem && item_col_flat[(size_t)(it + nb)] == c) nb++;
            const double *g = host.data() + (size_t)it * IS_B;    // lags have[c] .. have[c] + nb*16
            const long lag_first = have[(size_t)c];
            bool fin = false;
            for (long q = 0; q < nb * IS_B / 2 && !fin; q++) {
                const long i = lag_first / 2 + q;
                if (i >= half) { done[(size_t)c] = half; fin = true; break; }
                const double gam1 = g[2 * q] / len, gam2 = g[2 * q + 1] / len;    // initseq.c:30,39
                if (i == 0) {
                    g0[(size_t)c] = gam1;
                    // the lag-0 sum of squares is finite iff every element is (short of overflow): this is where a
                    // series handed over as a device pointer gets the reference's finiteness check (initseq.c:15)
                    if (!std::isfinite(gam1)) {
                        int bits = 0;
                        if (int rc = device_validate(dc, x + (size_t)c * len, len, 1, 0, &bits)) return rc;
                        if (bits) return set_error(MCMCHIP_EINVAL, "all elements of argument must be finite");
                        return set_error(MCMCHIP_EDOMAIN, "initseq: the lag-0 sum of squares of column %ld overflows", (long)c);
                    }
                }
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
            have[(size_t)c] += nb * IS_B;
            it += nb;
            if (!fin) next.push_back(c);
            else if (trace && getenv("MCMCHIP_TRACE")[0] == '2')
                fprintf(stderr, "[initseq] column %d: predicted %.1f pairs, stopped at %ld, computed %ld lags in %d rounds\n",
                        c, guess[(size_t)c], done[(size_t)c], have[(size_t)c], rounds[(size_t)c]);
        }
        active.swap(next);
    }
    bool overflow = false;
    for (long c = 0; c < ncol; c++) {
        long m = half == 0 ? 0 : done[(size_t)c];
        n_out[c] = m;
        gamma0[c] = g0[(size_t)c];
        if (m > cap) { overflow = true; continue; }
        gamma_sequences(Gam[(size_t)c].data(), m, g0[(size_t)c],
                        GammaOut{Gpos + (size_t)c * cap, Gdec + (size_t)c * cap, Gcon + (size_t)c * cap, vpos + c, vdec + c, vcon + c});
    }
    if (overflow) return set_error(MCMCHIP_ECAP, "initseq: output capacity too small; required lengths are in n_out");
    return MCMCHIP_OK;
}

// raw lag sums sum_t x[t] x[t + l], l = 0 .. nlags-1, of every column over the whole array given (zero beyond its end):
// what a time-sharded initseq all-reduces (mcmc_amd/dist.py::initseq_time_sharded)
int run_lagsums(const double *x, long len, long ncol, long nlags, double *sums, DeviceCtx *dc)
{
    const long span = (long)IS_TILE * IS_TILES_PER_BLOCK;
    const long nblk = (len + 16 + span - 1) / span;
    if (nblk > 65535) return set_error(MCMCHIP_EINVAL, "initseq: series longer than 2^32 elements");
    const long units = (nlags + IS_B - 1) / IS_B;
    LagBuffers bufs;
    LagRound work;
    std::vector<long> first((size_t)ncol);
    for (long c = 0; c < ncol; c++) {
        first[(size_t)c] = (long)work.unit_col.size();
        work.add((int)c, 0, units);
    }
    std::vector<double> host;
    timing_reset("autocov_kernel");
    if (int rc = run_lag_round(x, len, nblk, work, bufs, host, dc)) return rc;
    for (long c = 0; c < ncol; c++)
        for (long l = 0; l < nlags; l++) sums[(size_t)c * nlags + l] = host[(size_t)first[(size_t)c] * IS_B + l];
    return MCMCHIP_OK;
}

// the stop rule and the three sequences from TOTAL lag sums (host only; initseq.c:24-105 on sums computed elsewhere).
// n_out[c] = -1 when column c has not reached its stopping lag within nlags: the caller comes back with more lags.
int run_initseq_from_lagsums(const double *sums, long len, long ncol, long nlags, long cap, long *n_out, double *gamma0,
                             double *Gpos, double *Gdec, double *Gcon, double *vpos, double *vdec, double *vcon)
{
    const long half = len / 2;
    bool overflow = false;
    std::vector<double> G;
    for (long c = 0; c < ncol; c++) {
        const double *g = sums + (size_t)c * nlags;
        G.clear();
        long m = -1;
        double g0 = 0.0;
        if (half == 0) m = 0;
        for (long i = 0; m < 0 && 2 * i + 1 < nlags; i++) {
            if (i >= half) { m = half; break; }
            const double gam1 = g[2 * i] / len, gam2 = g[2 * i + 1] / len;        // initseq.c:30,39
            if (i == 0) g0 = gam1;
            const double Gi = gam1 + gam2;
            if (Gi < 0.0) { G.push_back(0.0); m = i + 1; }                        // initseq.c:42-46
            else { G.push_back(Gi); if (i + 1 >= half) m = half; }
        }
        gamma0[c] = nlags > 0 ? g[0] / len : 0.0;
        n_out[c] = m;
        if (m < 0) continue;
        if (m > cap) { overflow = true; continue; }
        gamma_sequences(G.data(), m, g0, GammaOut{Gpos + (size_t)c * cap, Gdec + (size_t)c * cap, Gcon + (size_t)c * cap,
                                                   vpos + c, vdec + c, vcon + c});
    }
    if (overflow) return set_error(MCMCHIP_ECAP, "initseq: output capacity too small; required lengths are in n_out");
    return MCMCHIP_OK;
}

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
    const long chunk = 1 << 20, nchunk = (n + chunk - 1) / chunk;
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
#define OM_SLAB 16
#define OM_NW 8
#define OL_ONEPASS_MIN_N (1L << 20)      // shorter series: the means come from colsum_kernel first, as in olbm.c:38-43
#define OL_PILOT_RUNS 4096               // runs of 8 rows in the provisional centre's sample
#define OL_ONEPASS_MAX_LOSS 1.0e3        // repeat about the exact mean if the correction would cost more than 3 digits

typedef double om_double4 __attribute__((ext_vector_type(4)));

template <int F> struct OmShape {
    static constexpr int NB = F >= 8 ? F / 8 : 1;       // base fragments per wave
    static constexpr int NACT = F / NB;                 // waves that own tiles (8, or 4 at F = 4)
    static constexpr int H = F / 2;
    static constexpr int NS = NB * H + 1;               // accumulator tiles per wave (last = the delta = H tile)
    static constexpr int NFR = NB + H;                  // fragments read per k-step
    static constexpr int PP = F * 16;
    static constexpr int DUPC = (H + 1) * 16;           // columns stored twice
    // LD = 2 mod 4: the two fragment rows a 32-lane ds_read_b64 group touches are 8 rows = 32 banks apart
    // (conflict-free); the transposed ds_write_b64 of the producer is 2-way, off the critical path
    static constexpr int LD = PP + DUPC + 2;
    static constexpr int NLD = PP / (4 * OM_NW);        // 4-column load slots of a wave per slab
    static constexpr int CW = PP / OM_NW;               // columns a wave produces (contiguous): 8, 16, 32
    static constexpr int NPART = 64 / CW;               // a column's 16 windows are summed by NPART lanes ...
    static constexpr int WPP = OM_SLAB / NPART;         // ... WPP consecutive windows each
    // per-wave image of the differences x[t+len] - x[t], [column][window]; 18 doubles = 144 B per column: the 16-byte
    // reads of 8 consecutive lanes (columns) fall in 8 disjoint bank groups
    static constexpr int DLD = OM_SLAB + 2;
    // the delta = H tile of wave w: base index and whether the wave has one
    __host__ __device__ static constexpr int extra_base(int w) { return NB == 2 ? (w < NACT / 2 ? 0 : 1) : 0; }
    __host__ __device__ static constexpr bool has_extra(int w) { return NB == 2 ? true : w < NACT / 2; }
    static constexpr size_t lds_bytes = ((size_t)2 * OM_SLAB * LD + PP + (size_t)PP * DLD) * 8;
};

// row_shr:N within the 16 lanes of a DPP row, zero fill
template <int N> __device__ __forceinline__ double om_row_shr(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

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

int run_olbm(const double *x, long n, int p, long len, double *mean, double *var, int nocalc, DeviceCtx *dc)
{
    hipStream_t s = dc->stream;
    if (n >= (1L << 31)) return set_error(MCMCHIP_EINVAL, "olbm: n exceeds the reference's int range (olbm.c:25)");
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
