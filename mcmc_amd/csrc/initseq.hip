// initseq.hip -- Monte Carlo standard error estimator initseq() (reference initseq.c:7-128).
//
// The O(len x lags) double sum of lag products (initseq.c:24-47) is the hot loop; the monotone / convex
// post-processing (55-105) works on a few hundred numbers and stays on the host, in the reference's own
// sequential order.  autocov kernels compute 16 NACC consecutive lags of one column per work item on the fp64
// matrix cores (v_mfma_f64_4x4x4_f64, every multiply-add of the instruction a wanted lag product; see the comment
// at the kernel).  Lags are produced speculatively because the stopping lag (first negative Gamma,
// initseq.c:42-46) is data dependent.  Partial sums are reduced in a fixed order.
#include "kernels.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <vector>

// =============================================================================================
// initseq
// =============================================================================================
#include "initseq_kernels.h"

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

// The provisional centre of mcmchip_initseq_uncentred: the mean of 64 runs of 1024 rows spread evenly over the column (of
// the whole column up to 65 536 rows).  Any value would do -- the sums about the mean follow exactly (centre_correction)
// -- but the nearer the centre lies to the mean, the fewer digits the correction cancels: a sample over the WHOLE
// length, so that a drifting start (burn-in) does not decide it.
__global__ __launch_bounds__(256) void sample_centre_kernel(const double *__restrict__ x, long len, double *__restrict__ centre)
{
    __shared__ double red[4];
    constexpr long SEG = 1024, NSEG = 64;
    const double *xc = x + (size_t)blockIdx.x * len;
    double s = 0.0;
    long cnt = len;
    if (len <= SEG * NSEG) {
        for (long t = threadIdx.x; t < len; t += 256) s += xc[t];
    } else {
        cnt = SEG * NSEG;
        for (long j = 0; j < NSEG; j++) {
            const long start = j * (len - SEG) / (NSEG - 1);
#pragma unroll
            for (long t = threadIdx.x; t < SEG; t += 256) s += xc[start + t];
        }
    }
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) centre[blockIdx.x] = (((red[0] + red[1]) + red[2]) + red[3]) / (double)cnt;
}

// the first and the last K rows of every column side by side, out[c][0 .. K) and out[c][K .. 2K): one contiguous copy to the
// host instead of two pitched ones of ncol short rows each (which took 4 ms of the C5 call)
__global__ __launch_bounds__(256) void edge_gather_kernel(const double *__restrict__ x, long len, long K, double *__restrict__ out)
{
    const double *xc = x + (size_t)blockIdx.x * len;
    double *o = out + (size_t)blockIdx.x * 2 * K;
    for (long k = threadIdx.x; k < K; k += 256) {
        o[k] = xc[k];
        o[K + k] = xc[len - K + k];
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

// one launch round of lag products: one work item per column (more only beyond 768 lags), grouped by kernel variant
// into launches; partial sums per time block, the fixed-order reduction over the time blocks, and the sums back on
// the host
namespace {
// instantiated kernels (a unit = 16 sums = 32 lags): lag pairs from lag 0 on (initseq_first_a/b.hip; fine steps,
// because the unit count of an item is rounded up to the next one and these carry > 90 % of the work), lag pairs of
// continuation items (initseq_cont.hip), single lags of one parity (initseq_raw.hip)
enum Family { FAM_FIRST = 0, FAM_CONT = 1, FAM_RAW = 2 };
struct Variant { int nacc; int fam; };
const Variant kVariants[] = {
    {1, FAM_FIRST}, {2, FAM_FIRST}, {3, FAM_FIRST}, {4, FAM_FIRST}, {5, FAM_FIRST}, {6, FAM_FIRST}, {7, FAM_FIRST},
    {8, FAM_FIRST}, {9, FAM_FIRST}, {10, FAM_FIRST}, {11, FAM_FIRST}, {12, FAM_FIRST}, {14, FAM_FIRST}, {16, FAM_FIRST},
    {18, FAM_FIRST}, {20, FAM_FIRST}, {22, FAM_FIRST}, {24, FAM_FIRST},
    {2, FAM_CONT}, {3, FAM_CONT}, {4, FAM_CONT}, {6, FAM_CONT}, {8, FAM_CONT}, {12, FAM_CONT}, {16, FAM_CONT},
    {1, FAM_RAW}, {2, FAM_RAW}, {4, FAM_RAW}, {8, FAM_RAW}, {16, FAM_RAW}, {24, FAM_RAW}};
const int kNumVariants = (int)(sizeof kVariants / sizeof kVariants[0]);

// time blocks: a workgroup walks `tpb` tiles of IS_TILE positions; a function of the length of the series alone, so
// the summation order -- and with it every bit of the result -- does not depend on what else a call computes
struct TimeGrid {
    long tpb, nblk, span;
    explicit TimeGrid(long len)
    {
        const long ntiles = (len + IS_PAD + IS_TILE - 1) / IS_TILE;
        tpb = ntiles / 256;
        if (tpb < 1) tpb = 1;
        if (tpb > 64) tpb = 64;
        span = tpb * IS_TILE;
        nblk = (len + IS_PAD + span - 1) / span;
    }
};

struct LagRound {
    struct Item { int col, unit; long lag; };
    long nunit = 0;                                              // units (16 sums each) of the round, in order of request
    std::vector<Item> items[kNumVariants];
    void clear()
    {
        nunit = 0;
        for (int w = 0; w < kNumVariants; w++) items[w].clear();
    }
    // at least nb units of column c from lag0 on: one item of the next instantiated size of its family, or several
    // beyond the family's largest.  pairs: units of 16 Gamma sums (lags lag0 + 2e, + 2e + 1), and the item that starts
    // at lag 0 is followed by one more unit whose first entry is the sum of squares (gamma_0); else units of 16 single
    // lags lag0 + 2e.  Returns the first unit; *got = the data units scheduled (consecutive from there on, the sum of
    // squares after the first item's).
    long add(int c, long lag0, long nb, bool pairs, long *got, long *ssq_unit = nullptr)
    {
        const long first_unit = nunit;
        long done = 0;
        if (ssq_unit) *ssq_unit = -1;
        while (done < nb) {
            const long rest = nb - done, lag = lag0 + done * IS_UNIT_LAGS;
            const int fam = !pairs ? FAM_RAW : (lag == 0 ? FAM_FIRST : FAM_CONT);
            int w = -1;                                     // the family's first size that covers the rest, else its largest
            for (int v = 0; v < kNumVariants; v++)
                if (kVariants[v].fam == fam) {
                    w = v;
                    if (kVariants[v].nacc >= rest) break;
                }
            items[w].push_back(Item{c, (int)nunit, lag});
            nunit += kVariants[w].nacc;
            done += kVariants[w].nacc;
            if (fam == FAM_FIRST) {
                // the sum-of-squares unit sits right behind the item's data units: the units of a column stay
                // consecutive only if this item is the column's last of the round
                if (ssq_unit) *ssq_unit = nunit;
                nunit += 1;
                if (done < nb) {                            // more than 768 lags at once from lag 0: keep the data units apart
                    *got = done;
                    return first_unit;
                }
            }
        }
        *got = done;
        return first_unit;
    }
};

struct LagBuffers {
    DevBuf partial, sums, lists;
    size_t cap_units = 0, cap_blocks = 0;
    std::vector<char> host_lists;
};

// runs the round over time blocks [0, nblk_run) and leaves nunit * IS_B sums in `host`
// centre != nullptr: the lag pairs about the provisional centres (device, one per column)
int run_lag_round(const double *x, long len, const TimeGrid &tg, long nblk_run, const LagRound &w, LagBuffers &B,
                  std::vector<double> &host, DeviceCtx *dc, const double *centre = nullptr)
{
    hipStream_t s = dc->stream;
    const size_t nunit = (size_t)w.nunit;
    if (nunit > B.cap_units || (size_t)nblk_run > B.cap_blocks) {       // sizes follow the work list
        B.partial.release(); B.sums.release(); B.lists.release();
        B.cap_units = nunit > B.cap_units ? nunit : B.cap_units;
        B.cap_blocks = (size_t)nblk_run > B.cap_blocks ? (size_t)nblk_run : B.cap_blocks;
        HIP_TRY(B.partial.alloc(B.cap_units * B.cap_blocks * IS_B * 8));
        HIP_TRY(B.sums.alloc(B.cap_units * IS_B * 8));
        HIP_TRY(B.lists.alloc(B.cap_units * 16));
    }
    // the item lists of all launches in one upload: lags (long), then columns, then first units (int), each in launch order
    size_t nitem = 0, offs[kNumVariants];
    for (int v = 0; v < kNumVariants; v++) { offs[v] = nitem; nitem += w.items[v].size(); }
    B.host_lists.resize(nitem * 16);
    long *hl = (long *)B.host_lists.data();
    int *hc = (int *)(hl + nitem), *hu = hc + nitem;
    for (int v = 0; v < kNumVariants; v++)
        for (size_t i = 0; i < w.items[v].size(); i++) {
            hl[offs[v] + i] = w.items[v][i].lag;
            hc[offs[v] + i] = w.items[v][i].col;
            hu[offs[v] + i] = w.items[v][i].unit;
        }
    if (nitem) HIP_TRY(hipMemcpyAsync(B.lists.p, B.host_lists.data(), nitem * 16, hipMemcpyHostToDevice, s));
    const long *dl = B.lists.as<long>();
    const int *dcol = (const int *)(dl + nitem), *dunit = dcol + nitem;
    // the launches, short items first.  Side by side on two streams (short, HBM-heavy items next to long ones) they
    // take the same time as one after the other, measured twice at full C5 size (round 2: span 117.7 ms vs 111 ms;
    // round 3: 0.1026 s either way): the launches are bound by power, not by idle units (see initseq_kernels.h)
    int order[kNumVariants];
    for (int v = 0; v < kNumVariants; v++) order[v] = v;
    for (int i = 1; i < kNumVariants; i++)                                 // by accumulator count
        for (int j = i; j > 0 && kVariants[order[j]].nacc < kVariants[order[j - 1]].nacc; j--) {
            const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp;
        }
    for (int o = 0; o < kNumVariants; o++) {
        const int v = order[o];
        const size_t ni = w.items[v].size();
        if (!ni) continue;
        const int nacc = kVariants[v].nacc;
        dim3 grid((unsigned)ni, (unsigned)nblk_run);
        int rc;
        timing_begin(s);
        switch (kVariants[v].fam) {
        case FAM_FIRST:
            if (centre) {
                rc = launch_autocov_first_ca(nacc, grid, s, x, len, dcol + offs[v], dl + offs[v], dunit + offs[v], nblk_run, (int)tg.tpb, B.partial.as<double>(), centre);
                if (rc < 0) rc = launch_autocov_first_cb(nacc, grid, s, x, len, dcol + offs[v], dl + offs[v], dunit + offs[v], nblk_run, (int)tg.tpb, B.partial.as<double>(), centre);
                break;
            }
            rc = launch_autocov_first_a(nacc, grid, s, x, len, dcol + offs[v], dl + offs[v], dunit + offs[v], nblk_run, (int)tg.tpb, B.partial.as<double>());
            if (rc < 0) rc = launch_autocov_first_b(nacc, grid, s, x, len, dcol + offs[v], dl + offs[v], dunit + offs[v], nblk_run, (int)tg.tpb, B.partial.as<double>());
            break;
        case FAM_CONT:
            if (centre) {
                rc = launch_autocov_cont_c(nacc, grid, s, x, len, dcol + offs[v], dl + offs[v], dunit + offs[v], nblk_run, (int)tg.tpb, B.partial.as<double>(), centre);
                break;
            }
            rc = launch_autocov_cont(nacc, grid, s, x, len, dcol + offs[v], dl + offs[v], dunit + offs[v], nblk_run, (int)tg.tpb, B.partial.as<double>());
            break;
        default:
            if (centre) return set_error(MCMCHIP_EINVAL, "initseq: single lags are not taken about a centre");
            rc = launch_autocov_raw(nacc, grid, s, x, len, dcol + offs[v], dl + offs[v], dunit + offs[v], nblk_run, (int)tg.tpb, B.partial.as<double>());
        }
        timing_end(s);
        if (rc < 0) return set_error(MCMCHIP_EINVAL, "initseq: no kernel for %d accumulators", nacc);
        if (rc) return rc;
        HIP_TRY(hipGetLastError());
    }
    const long total = (long)nunit * IS_B;
    static_assert(IS_B == 16, "autocov_reduce_kernel splits 256 threads into 16 parts x 16 lags");
    if (nunit) {
        hipLaunchKernelGGL(autocov_reduce_kernel, dim3((unsigned)nunit), dim3(256), 0, s, B.partial.as<double>(), nblk_run,
                           B.sums.as<double>());
        HIP_TRY(hipGetLastError());
    }
    host.resize((size_t)total);
    if (total) HIP_TRY(hipMemcpyAsync(host.data(), B.sums.p, (size_t)total * 8, hipMemcpyDeviceToHost, s));
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

// From sums about a provisional centre to sums about the mean (mcmchip_initseq_uncentred).  With z[t] = x[t] - centre,
// S = sum z, delta = S / len (so mean = centre + delta) and w[t] = z[t] - delta the series initseq.c is handed:
//     sum_{t < len-k} w[t] w[t+k] = sum_{t < len-k} z[t] z[t+k] - delta (2 S - H_k - T_k) + (len - k) delta^2
//                                 = sum_{t < len-k} z[t] z[t+k] - (len + k) delta^2 + delta (H_k + T_k),
// H_k (T_k) the sum of the first (last) k values of z.  This holds H_k + T_k per column for k <= K, from the first and last
// K rows of the series copied to the host (gathered on the device for all columns, one copy; a column that goes beyond K
// lags -- rare: K starts at 2048 -- has its edges fetched again, alone).
struct EdgeSums {
    long K = 0;
    std::vector<std::vector<double>> ht;          // [column][k], k = 0 .. K (of the columns asked for last)
    std::vector<double> head, tail;
    int ensure(const double *x, long len, long ncol, const std::vector<int> &cols, long need, const std::vector<double> &cen, hipStream_t s)
    {
        if (need > len) need = len;
        if (need <= K) return MCMCHIP_OK;
        const bool all = K == 0;
        long k = K == 0 ? first_rows(len, ncol) : 2 * K;
        if (k < need) k = need;
        if (k > len) k = len;
        K = k;
        if (all) {
            ht.assign((size_t)ncol, std::vector<double>());
            head.resize((size_t)ncol * 2 * K);
            DevBuf gathered;
            HIP_TRY(gathered.alloc((size_t)ncol * 2 * K * 8));
            hipLaunchKernelGGL(edge_gather_kernel, dim3((unsigned)ncol), dim3(256), 0, s, x, len, K, gathered.as<double>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(head.data(), gathered.p, (size_t)ncol * 2 * K * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            for (long c = 0; c < ncol; c++) fill(c, head.data() + (size_t)c * 2 * K, head.data() + (size_t)c * 2 * K + K, cen[(size_t)c]);
        } else {
            head.resize((size_t)K);
            tail.resize((size_t)K);
            for (int c : cols) {
                HIP_TRY(hipMemcpyAsync(head.data(), x + (size_t)c * len, (size_t)K * 8, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipMemcpyAsync(tail.data(), x + (size_t)c * len + (len - K), (size_t)K * 8, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                fill(c, head.data(), tail.data(), cen[(size_t)c]);
            }
        }
        return MCMCHIP_OK;
    }
    // the edges of all columns as edge_gather_kernel left them ([column][2 k]: first k rows, last k rows), already on the host
    void take_all(const double *gathered, long ncol, long k, const std::vector<double> &cen)
    {
        K = k;
        ht.assign((size_t)ncol, std::vector<double>());
        for (long c = 0; c < ncol; c++) fill(c, gathered + (size_t)c * 2 * K, gathered + (size_t)c * 2 * K + K, cen[(size_t)c]);
    }
    static long first_rows(long len, long ncol)       // at most 2048 rows, and at most 2 x 256 MB of host memory however many columns
    {
        const long k = std::max(64L, std::min(2048L, (32L << 20) / ncol));
        return k < len ? k : len;
    }
    void fill(long c, const double *h, const double *t, double centre)
    {
        std::vector<double> &v = ht[(size_t)c];
        v.resize((size_t)K + 1);
        double hs = 0.0, ts = 0.0;
        v[0] = 0.0;
        for (long k = 1; k <= K; k++) {
            hs += h[k - 1] - centre;              // the kernel's z, rounded as there
            ts += t[K - k] - centre;
            v[(size_t)k] = hs + ts;
        }
    }
    // what to add to the kernel's sum for lag pair i (lags 2i and 2i + 1)
    double pair_correction(long c, long i, long len, double delta) const
    {
        const std::vector<double> &v = ht[(size_t)c];
        const long k = 2 * i;
        return delta * (v[(size_t)k] + v[(size_t)k + 1]) - ((double)(len + k) + (double)(len + k + 1)) * delta * delta;
    }
};
}  // namespace

#define IS_MAX_UNITS_PER_LAUNCH 32      // at most 1024 new lags per column and round
#define IS_PILOT_POSITIONS (1L << 21)   // pilot pass: at most the first 2M positions of every column

// about_mean: x is taken about its column means (R/initseq.R:3) without being modified or read more than once per round:
// lag sums about a provisional centre, corrected on the host (EdgeSums); the means go to mean_out (host, or NULL)
int run_initseq(const double *x, long len, long ncol, long cap, long *n_out, double *gamma0, double *Gpos,
                double *Gdec, double *Gcon, double *vpos, double *vdec, double *vcon, DeviceCtx *dc, bool about_mean,
                double *mean_out)
{
    const long half = len / 2;
    // tiles start at position -IS_PAD (zero padded), so every result row of an MFMA sees every base position
    const TimeGrid tg(len);
    const long nblk = tg.nblk;
    if (nblk > 65535) return set_error(MCMCHIP_EINVAL, "initseq: series longer than 2^32 elements");
    // The stopping lag is data dependent, so Gamma's are produced speculatively in units of 16 (32 lags).  Long series
    // start with a PILOT: 32 Gamma's of every column over its first 2M positions only (~1 % of a pass), whose sequence
    // predicts where the stop rule will fire; the first full pass then asks each column for that many plus a margin,
    // so most columns need one pass over HBM.  A column that is still active afterwards gets 20-40 % more per further
    // round.  The sums themselves do not depend on any of this.
    const bool trace = getenv("MCMCHIP_TRACE") != nullptr;
    const int PU = IS_B;                                 // Gamma's (lag pairs) per unit
    LagBuffers bufs;
    LagRound work;
    std::vector<std::vector<double>> Gam((size_t)ncol);   // Gamma_i per column, as far as computed
    std::vector<long> done((size_t)ncol, -1);            // output length once the stop rule fired
    std::vector<long> have((size_t)ncol, 0);             // lag pairs computed so far
    std::vector<double> g0((size_t)ncol, 0.0);
    std::vector<double> guess((size_t)ncol, 0.0);        // predicted lag pairs
    std::vector<int> rounds((size_t)ncol, 0);            // full-length launches a column has taken part in
    std::vector<int> active;
    std::vector<long> unit0, units, ssq_at;              // per active column of the round
    for (long c = 0; c < ncol; c++) active.push_back((int)c);
    std::vector<double> host;
    timing_reset("autocov_kernel");
    DevBuf dcentre;
    std::vector<double> cen, delta;                      // provisional centre; mean - centre once a full pass has summed the column
    EdgeSums edges;
    const double *centre = nullptr;
    if (about_mean) {
        if (ncol > 0x7fffffffL) return set_error(MCMCHIP_EINVAL, "too many columns");
        // one buffer, two small kernels, one copy, one wait: the centres, then the first / last rows of every column
        const long k0 = EdgeSums::first_rows(len, ncol);
        const size_t nd = (size_t)ncol * (1 + 2 * (size_t)k0);
        HIP_TRY(dcentre.alloc(nd * 8));
        hipLaunchKernelGGL(sample_centre_kernel, dim3((unsigned)ncol), dim3(256), 0, dc->stream, x, len, dcentre.as<double>());
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(edge_gather_kernel, dim3((unsigned)ncol), dim3(256), 0, dc->stream, x, len, k0, dcentre.as<double>() + ncol);
        HIP_TRY(hipGetLastError());
        std::vector<double> up(nd);
        HIP_TRY(hipMemcpyAsync(up.data(), dcentre.p, nd * 8, hipMemcpyDeviceToHost, dc->stream));
        HIP_TRY(hipStreamSynchronize(dc->stream));
        cen.assign(up.begin(), up.begin() + ncol);
        delta.assign((size_t)ncol, 0.0);
        edges.take_all(up.data() + ncol, ncol, k0, cen);
        centre = dcentre.as<double>();
    }
    if (half == 0) active.clear();
    long npilot = 0;                                     // time blocks of the pilot: 1/64 of a series of 1M positions and more
    if (len >= (1L << 20)) {
        const long pos = len / 64 < IS_PILOT_POSITIONS ? len / 64 : IS_PILOT_POSITIONS;
        npilot = (pos + tg.span - 1) / tg.span;
    }
    if (npilot > 0 && !active.empty()) {
        work.clear();
        unit0.assign(active.size(), 0);
        ssq_at.assign(active.size(), -1);
        for (size_t a = 0; a < active.size(); a++) {
            long got;
            unit0[a] = work.add(active[a], 0, 2, true, &got, &ssq_at[a]);
        }
        if (int rc = run_lag_round(x, len, tg, npilot, work, bufs, host, dc, centre)) return rc;
        for (size_t a = 0; a < active.size(); a++) {
            double *g = host.data() + (size_t)unit0[a] * IS_B;
            if (about_mean && ssq_at[a] >= 0) {
                // about the pilot's own mean, near enough for a prediction: every sum holds two lag products over ~np positions
                long np = npilot * tg.span - IS_PAD;
                if (np > len) np = len;
                const double sp = host[(size_t)ssq_at[a] * IS_B + 1];
                for (int q = 0; q < 2 * PU; q++) g[q] -= 2.0 * sp * sp / (double)np;
            }
            guess[(size_t)active[a]] = predict_pairs(g, 2 * PU, len);
        }
        if (trace) fprintf(stderr, "[initseq] pilot: %zu columns x 64 lags over %ld of %ld time blocks\n", active.size(), npilot, nblk);
    }
    while (!active.empty()) {
        work.clear();
        unit0.assign(active.size(), 0);
        units.assign(active.size(), 0);
        ssq_at.assign(active.size(), -1);
        // units (32 lags of one column) that cost about as much as the round itself: ~30 effective TFLOP/s on the matrix cores
        const long cheap_units = (long)(150e-6 / ((double)len * 1.1e-12 * (double)active.size()));
        for (size_t a = 0; a < active.size(); a++) {
            const int c = active[a];
            const long h = have[(size_t)c];              // pairs so far
            long nb;
            if (rounds[(size_t)c] == 0 && npilot == 0) {
                nb = 2;                                              // no pilot: 64 full-length lags are the prediction
            } else if (rounds[(size_t)c] == 0 || (rounds[(size_t)c] == 1 && npilot == 0)) {
                // The first pass sized by the prediction.  Measured at C5 size the stop comes at 0.7 ... 2 x the
                // prediction (median 1.18, upper decile 1.76: the rule fires in the noise).  A column of few lags is
                // HBM-bound and a second pass over it costs as much as the first, so it gets a generous margin; a
                // column of many lags is MFMA-bound, every extra lag costs, and a further round is cheap by comparison.
                const double p = npilot > 0 ? guess[(size_t)c] : predict_pairs(Gam[(size_t)c].data(), (long)Gam[(size_t)c].size(), len);
                const double f = p <= 16.0 ? 1.7 : (p <= 40.0 ? 1.4 : (p <= 150.0 ? 1.15 : 1.05));
                const double want = p * f + 2.0 - (double)h;
                nb = want > 0.0 ? (long)std::ceil(want / PU) : 1;
            } else {
                nb = (long)std::ceil((h <= 128 ? 0.4 : 0.2) * (double)h / PU);      // overshoot of the stop rule into the noise
                // A round costs ~0.15 ms whatever it computes (launches, the reduction over the time blocks, the copy back,
                // the host's look at the sums).  While the units asked for are cheaper than that, a column gets up to as
                // many again as it has: a slowly mixing column of 1e6 rows took 342 launches (48 ms) in steps of 20 %.
                const long dbl = (h + PU - 1) / PU;
                if (nb < cheap_units) nb = std::min(cheap_units, dbl);
                // ... and a column still going after six rounds (long memory: the rule may not fire before len / 2) grows
                // geometrically as long as a round stays within a millisecond
                if (rounds[(size_t)c] >= 6) nb = std::max(nb, std::min(dbl, 8 * cheap_units));
            }
            if (nb < 1) nb = 1;
            rounds[(size_t)c]++;
            const long max_units = std::max((long)IS_MAX_UNITS_PER_LAUNCH, std::min(8 * cheap_units, 4096L));
            if (nb > max_units) nb = max_units;
            const long need = (half - h + PU - 1) / PU;            // never past pair half - 1
            if (nb > need) nb = need;
            unit0[a] = work.add(c, 2 * h, nb, true, &units[a], &ssq_at[a]);
        }
        if (trace)
        {
            fprintf(stderr, "[initseq] round: %zu active columns, %ld units of %d lags, %ld time blocks; items x lags:", active.size(), work.nunit, IS_UNIT_LAGS, nblk);
            for (int k = 0; k < kNumVariants; k++)
                if (!work.items[k].empty()) fprintf(stderr, " %zu x %d%s", work.items[k].size(), IS_UNIT_LAGS * kVariants[k].nacc, kVariants[k].fam == FAM_FIRST ? "" : "c");
            fprintf(stderr, "\n");
        }
        const auto t_round = std::chrono::steady_clock::now();
        if (int rc = run_lag_round(x, len, tg, nblk, work, bufs, host, dc, centre)) return rc;
        if (about_mean) {
            long need = 0;
            for (size_t a = 0; a < active.size(); a++) need = std::max(need, 2 * (have[(size_t)active[a]] + units[a] * PU));
            if (int rc = edges.ensure(x, len, ncol, active, need, cen, dc->stream)) return rc;
        }
        if (trace) fprintf(stderr, "[initseq] round took %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_round).count());
        std::vector<int> next;
        for (size_t a = 0; a < active.size(); a++) {
            const int c = active[a];
            const long nb = units[a];
            const double *g = host.data() + (size_t)unit0[a] * IS_B;    // pairs have[c] .. have[c] + nb*16
            const long pair_first = have[(size_t)c];
            bool fin = false;
            if (ssq_at[a] >= 0) {
                double ssq = host[(size_t)ssq_at[a] * IS_B];
                if (about_mean) {
                    const double S = host[(size_t)ssq_at[a] * IS_B + 1];              // sum of x - centre
                    delta[(size_t)c] = S / (double)len;
                    ssq -= S * delta[(size_t)c];                                      // sum z^2 - len delta^2
                }
                const double gam0 = ssq / len;                                        // initseq.c:30 at lag 0
                g0[(size_t)c] = gam0;
                // the lag-0 sum of squares is finite iff every element is (short of overflow): this is where a
                // series handed over as a device pointer gets the reference's finiteness check (initseq.c:15)
                if (!std::isfinite(gam0)) {
                    int bits = 0;
                    if (int rc = device_validate(dc, x + (size_t)c * len, len, 1, 0, &bits)) return rc;
                    if (bits) return set_error(MCMCHIP_EINVAL, "all elements of argument must be finite");
                    return set_error(MCMCHIP_EDOMAIN, "initseq: the lag-0 sum of squares of column %ld overflows", (long)c);
                }
            }
            for (long q = 0; q < nb * PU && !fin; q++) {
                const long i = pair_first + q;
                if (i >= half) { done[(size_t)c] = half; fin = true; break; }
                const double G = (about_mean ? g[q] + edges.pair_correction(c, i, len, delta[(size_t)c]) : g[q]) / len;   // gam1 + gam2, initseq.c:30,39,41
                if (G < 0.0) {                                                    // initseq.c:42-46
                    Gam[(size_t)c].push_back(0.0);
                    done[(size_t)c] = i + 1;
                    fin = true;
                } else {
                    Gam[(size_t)c].push_back(G);
                    if (i + 1 >= half) { done[(size_t)c] = half; fin = true; }
                }
            }
            have[(size_t)c] += nb * PU;
            if (!fin) next.push_back(c);
            else if (trace && getenv("MCMCHIP_TRACE")[0] == '2')
                fprintf(stderr, "[initseq] column %d: predicted %.1f pairs, stopped at %ld, computed %ld lags in %d rounds\n",
                        c, guess[(size_t)c], done[(size_t)c], 2 * have[(size_t)c], rounds[(size_t)c]);
        }
        active.swap(next);
    }
    bool overflow = false;
    for (long c = 0; c < ncol; c++) {
        long m = half == 0 ? 0 : done[(size_t)c];
        n_out[c] = m;
        gamma0[c] = g0[(size_t)c];
        if (about_mean && mean_out) mean_out[c] = cen[(size_t)c] + delta[(size_t)c];
        if (m > cap) { overflow = true; continue; }
        gamma_sequences(Gam[(size_t)c].data(), m, g0[(size_t)c],
                        GammaOut{Gpos + (size_t)c * cap, Gdec + (size_t)c * cap, Gcon + (size_t)c * cap, vpos + c, vdec + c, vcon + c});
    }
    if (overflow) return set_error(MCMCHIP_ECAP, "initseq: output capacity too small; required lengths are in n_out");
    return MCMCHIP_OK;
}

// raw lag sums sum_t x[t] x[t + l], l = 0 .. nlags-1, of every column over the whole array given (zero beyond its end):
// what a time-sharded initseq all-reduces (mcmc_amd/dist.py::initseq_time_sharded).  Single lags: the even and the odd
// ones of a column are two work items (the kernel's units hold lags of one parity).
int run_lagsums(const double *x, long len, long ncol, long nlags, double *sums, DeviceCtx *dc)
{
    const TimeGrid tg(len);
    const long nblk = tg.nblk;
    if (nblk > 65535) return set_error(MCMCHIP_EINVAL, "initseq: series longer than 2^32 elements");
    const long n_even = (nlags + 1) / 2, n_odd = nlags / 2;
    LagBuffers bufs;
    LagRound work;
    std::vector<long> first_even((size_t)ncol, 0), first_odd((size_t)ncol, 0);
    for (long c = 0; c < ncol; c++) {
        long got;
        if (n_even) first_even[(size_t)c] = work.add((int)c, 0, (n_even + IS_B - 1) / IS_B, false, &got);
        if (n_odd) first_odd[(size_t)c] = work.add((int)c, 1, (n_odd + IS_B - 1) / IS_B, false, &got);
    }
    std::vector<double> host;
    timing_reset("autocov_kernel");
    if (int rc = run_lag_round(x, len, tg, nblk, work, bufs, host, dc)) return rc;
    for (long c = 0; c < ncol; c++)
        for (long l = 0; l < nlags; l++)
            sums[(size_t)c * nlags + l] = host[(size_t)((l & 1) ? first_odd[(size_t)c] : first_even[(size_t)c]) * IS_B + l / 2];
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

