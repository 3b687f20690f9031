// initseq_kernels.h -- the lag-product kernel of initseq() (initseq.c:24-47) on the fp64 matrix cores and its launcher;
// instantiated per accumulator count in initseq_first_a.hip / initseq_first_b.hip (items that start at lag 0),
// initseq_cont.hip (continuation items) and initseq_raw.hip (single lags for mcmchip_lagsums), driven by initseq.hip.
#pragma once
#ifndef AC_S_OCC
#define AC_S_OCC 3
#define AC_S_DEPTH 2
#endif
#include "kernels.h"
#include <type_traits>

#define IS_B 16                 // sums per work unit (one accumulator double per lane)
#define IS_UNIT_LAGS 32         // lags a unit spans: 16 Gamma's (lag pairs), or 16 lags of one parity
#define IS_TILE 2048            // base positions per LDS tile
#define IS_PAD 32               // a series is tiled from position -IS_PAD on (zeros), see below
#define IS_MAXACC 24            // accumulators per work item: at most 768 lags

// What is computed.  initseq needs Gamma_k = gamma_2k + gamma_2k+1 (initseq.c:27-41) and gamma_0, never a single
// gamma beyond lag 0:
//     len * Gamma_k = sum_t x[t] x[t+2k] + sum_t x[t] x[t+2k+1] = sum_t x[t] * y[t+2k],   y[t] = x[t] + x[t+1],
// ONE product per base position and lag PAIR instead of two (round 3; the sums differ from the reference's by the
// rounding of y, ~1e-16 relative per term, far inside the summation-order differences any blocked sum has).  gamma_0
// = sum x[t]^2 / len is accumulated on the side while a tile is staged.  With PAIR = false the second operand is x
// itself and a unit holds 16 single lags of one parity: the raw sums of mcmchip_lagsums.
//
// On the matrix cores.  v_mfma_f64_4x4x4_f64 is four independent 4x4x4 products (lane = q + 4*blk + 16*k holds
// A_blk[i=q][k] and B_blk[k][j=q]; result lane = j + 4*blk + 16*i; layout and a rate equal to the 16x16x4 form
// measured by tools/mfma64_4x4_probe.hip).  With
//     A_blk[i][k] = x[u + 8i + blk + 4k]          B_blk[k][j] = y[u + blk + 4k + 2j + 24 + L]
// entry (i, j) accumulates x[t] * y[t + lag] for lag = L + 24 + 2j - 8i (the 16 even offsets L, L+2, ... L+30: entry
// index 12 + j - 4i) over the 16 base positions t = u + 8i + (blk + 4k): every one of the 256 multiply-adds of the
// instruction is a wanted product, and stepping u by 16 tiles the series exactly once (for a fixed lag the base
// positions start 8i after u, hence the zero pad of IS_PAD >= 24 positions before element 0).  Both operands are
// ~40 consecutive doubles of an LDS image (the 32 lanes of a half wave read 32 distinct consecutive doubles, or
// broadcast: conflict-free), and B(u + 32, L) = B(u, L + 32): a wave that walks u in steps of 32 (two MFMAs per
// lag block) keeps the B fragment pairs of NJ lag blocks in a register ring and reads one new A pair and one new B
// pair per 2 NJ MFMAs.
//
// Work decomposition (round 2).  ONE work item per column and launch round: NACC accumulators (32 lags each, up to
// 768 lags) are worked off in near-equal pieces of at most 12 lag blocks over the SAME staged tile, so a column is
// streamed from HBM once per round whatever its lag count.  The tile is fetched through buffer descriptors rebuilt per
// tile in scalar registers: the hardware range check returns 0 past the end of the series, the loads need no
// predicate and no per-lane address arithmetic beyond one precomputed byte offset.
//
// Pipeline (round 3).  A workgroup stages a 2048-position tile in LDS -- an x image and a y image, y[e] = x[e] + x[e+1]
// from a second load of the same cache lines -- and its 4 waves take 512 positions each.  The images are double
// buffered: the registers holding tile k+1 are written to the other buffer and the loads of the tile(s) after are
// issued BEFORE the MFMAs of tile k, so a wave's matrix stream is interrupted once per tile by one LDS-only barrier
// (s_waitcnt lgkmcnt(0); s_barrier -- __syncthreads would also drain the loads in flight).  Items of few lags keep TWO
// tiles of loads in flight per workgroup (two register sets).  FIRST (the item starts at lag 0: every column's first
// full pass, > 90 % of the work): the two images cover the same positions and share their first load.  The time blocks
// a column is cut into follow the length of the series (1 to 64 tiles).
// (Measured at C5 with the round-2 kernel and with this pipeline around the old two-products-per-pair arithmetic, the
// time of an item was additive, 0.08-0.10 ms per pass over the column + 2.4 us per lag: the matrix pipe was 82-91 %
// busy in every launch, but the clock fell from 2.37 GHz (no HBM traffic) to 1.8 GHz at 5 TB/s -- power, not latency;
// profiles/r3_c5_initseq_pmc_two_products.txt.  Hence the halved arithmetic rather than more overlap.)
// f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>): loop indices that are constant expressions
template <int N, int I = 0, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// ... as long as f returns true
template <int N, int I = 0, class F> __device__ __forceinline__ void static_while(F &&f)
{
    if constexpr (I < N) {
        if (f(std::integral_constant<int, I>{})) static_while<N, I + 1>(f);
    }
}

template <int NACC, bool FIRST, bool PAIR> struct AcShape {
    static constexpr int NA = IS_TILE + 24, NY = IS_TILE + 32 * NACC;        // x image: indices < 2072; y image: < 2046 + 32 NACC
    static constexpr int SA = (NA + 255) / 256, SY = (NY + 255) / 256;       // rounds of 256 loads per image
    static constexpr int IMG = (SA + SY) * 256;                              // doubles per buffer: x image, then y image
    static constexpr int DEPTH = FIRST ? (NACC <= 2 ? AC_S_DEPTH : (NACC <= 4 ? 3 : (NACC <= 8 ? 2 : 1))) : (NACC <= 2 ? 2 : 1);   // tiles of loads in flight per workgroup
    // NBUF = 1: one LDS buffer and two barriers per tile, 4 workgroups per CU -- the items of <= 64 lags, which do little
    // arithmetic per byte: with two workgroups per CU their load / LDS-write / MFMA phases hardly overlap
    static constexpr int NBUF = (FIRST && PAIR && NACC <= 2) ? 1 : 2;
    static constexpr int OCC = NBUF == 1 ? AC_S_OCC : 2;                            // workgroups per CU (LDS: 36.9 KB per buffer and up)
    // registers per tile in flight: FIRST: x[e] and x[e+1] for SY rounds; else x image + y image (two loads each when PAIR)
    static constexpr int NREG = FIRST ? (PAIR ? 2 * SY : SY) : SA + (PAIR ? 2 * SY : SY);
};

typedef __attribute__((address_space(3))) const double ac_lds_cd;

// Fragment reads by hand.  Written as C++ loads the two fragments of a pair, and the fragments of neighbouring steps,
// are merged by the compiler into ds_read2_b64, which the LDS serves 16 lanes at a time: 8 cycles for two doubles per
// lane against 2 + 2 for two ds_read_b64 (MI355X_MICROARCH.md, LDS table) -- and items of few lags are bound by the
// LDS as much as by HBM (four fragment reads per two MFMAs at NJ = 1).  So: ds_read_b64 through inline asm with
// immediate offsets from two address registers per wave (x image, y image).  The compiler does not count these reads
// in lgkmcnt, so the waits are explicit: LDS reads return in order, the reads of step s + 1 are issued before the
// MFMAs of step s, and `lgkmcnt(4)` before those MFMAs therefore means "everything up to step s has arrived".  The
// registers a wait covers pass through an empty asm behind it (ac_tie): the MFMAs that consume them depend on that
// asm's output and cannot be scheduled above the wait.
template <int OFF> __device__ __forceinline__ double ac_lds_read(unsigned addr)
{
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void ac_lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }
__device__ __forceinline__ void ac_tie(double &v) { asm volatile("" : "+v"(v)); }

// the MFMAs of one piece: NJ lag blocks (from block J0 of the item on) over this wave's 512 positions = 16 double
// steps (32 positions) of the staged tile.  One A pair and one B pair read per double step serve 2 NJ MFMAs.  The
// reads run D steps ahead of the MFMAs (an LDS round trip is 100-200 cycles, a step of NJ = 1 only 32: with the reads
// one step ahead the short items, two thirds of C5, spent their time on that round trip); the A pairs of D + 1 steps
// and the B pairs of NJ + D steps sit in register rings: G[m] = the B pair at u + 32 m lives in slot m % (NJ + D).
template <int NJ> struct AcAhead { static constexpr int D = NJ >= 8 ? 1 : (NJ >= 4 ? 2 : 3); };   // 4 D <= 15: lgkmcnt has 4 bits

template <int NJ, int J0, int N>          // the reads of step N: its A pair and the B pair G[N + NJ - 1] it brings in
__device__ __forceinline__ void autocov_issue(unsigned aaddr, unsigned baddr, double (&fa)[AcAhead<NJ>::D + 1][2],
                                              double (&Rlo)[NJ + AcAhead<NJ>::D], double (&Rhi)[NJ + AcAhead<NJ>::D])
{
    constexpr int D = AcAhead<NJ>::D, M = N + NJ - 1;
    fa[N % (D + 1)][0] = ac_lds_read<256 * N>(aaddr);
    fa[N % (D + 1)][1] = ac_lds_read<256 * N + 128>(aaddr);
    Rlo[M % (NJ + D)] = ac_lds_read<256 * (M + J0)>(baddr);
    Rhi[M % (NJ + D)] = ac_lds_read<256 * (M + J0) + 128>(baddr);
}

template <int NJ, int J0, int SS>
__device__ __forceinline__ void autocov_step(unsigned aaddr, unsigned baddr, double (&fa)[AcAhead<NJ>::D + 1][2],
                                             double (&Rlo)[NJ + AcAhead<NJ>::D], double (&Rhi)[NJ + AcAhead<NJ>::D], double *acc)
{
    constexpr int STEPS = IS_TILE / 4 / 32;           // 16 double steps per wave and tile
    constexpr int D = AcAhead<NJ>::D, RING = NJ + D;
    if constexpr (SS + D < STEPS) autocov_issue<NJ, J0, SS + D>(aaddr, baddr, fa, Rlo, Rhi);
    constexpr int AHEAD = SS + D < STEPS ? D : STEPS - 1 - SS;       // steps whose reads may still be in flight
    ac_lds_wait<4 * AHEAD>();
    ac_tie(fa[SS % (D + 1)][0]);
    ac_tie(fa[SS % (D + 1)][1]);
    ac_tie(Rlo[(SS + NJ - 1) % RING]);
    ac_tie(Rhi[(SS + NJ - 1) % RING]);
    if constexpr (SS == 0) {                            // the ring's preloaded pairs G[0 .. NJ-2] are older still
#pragma unroll
        for (int m = 0; m < NJ - 1; m++) {
            ac_tie(Rlo[m]);
            ac_tie(Rhi[m]);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; j++)
        acc[J0 + j] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[SS % (D + 1)][0], Rlo[(SS + j) % RING], acc[J0 + j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; j++)
        acc[J0 + j] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[SS % (D + 1)][1], Rhi[(SS + j) % RING], acc[J0 + j], 0, 0, 0);
    if constexpr (SS + 1 < STEPS) autocov_step<NJ, J0, SS + 1>(aaddr, baddr, fa, Rlo, Rhi, acc);
}

template <int NJ, int J0, int M>
__device__ __forceinline__ void autocov_preload(unsigned baddr, double (&Rlo)[NJ + AcAhead<NJ>::D], double (&Rhi)[NJ + AcAhead<NJ>::D])
{
    if constexpr (M < NJ - 1) {
        Rlo[M] = ac_lds_read<256 * (M + J0)>(baddr);
        Rhi[M] = ac_lds_read<256 * (M + J0) + 128>(baddr);
        autocov_preload<NJ, J0, M + 1>(baddr, Rlo, Rhi);
    }
}

template <int NJ, int J0, int N>
__device__ __forceinline__ void autocov_prime(unsigned aaddr, unsigned baddr, double (&fa)[AcAhead<NJ>::D + 1][2],
                                              double (&Rlo)[NJ + AcAhead<NJ>::D], double (&Rhi)[NJ + AcAhead<NJ>::D])
{
    if constexpr (N < AcAhead<NJ>::D) {
        autocov_issue<NJ, J0, N>(aaddr, baddr, fa, Rlo, Rhi);
        autocov_prime<NJ, J0, N + 1>(aaddr, baddr, fa, Rlo, Rhi);
    }
}

template <int NJ, int J0> __device__ __forceinline__ void autocov_piece(unsigned aaddr, unsigned baddr, double *acc)
{
    constexpr int D = AcAhead<NJ>::D;
    double fa[D + 1][2], Rlo[NJ + D], Rhi[NJ + D];
    autocov_preload<NJ, J0, 0>(baddr, Rlo, Rhi);          // G[0 .. NJ-2]
    autocov_prime<NJ, J0, 0>(aaddr, baddr, fa, Rlo, Rhi);   // the reads of steps 0 .. D-1
    autocov_step<NJ, J0, 0>(aaddr, baddr, fa, Rlo, Rhi, acc);
}

// NACC lag blocks as one piece up to 12, else as the fewest near-equal pieces (compile-time recursion: static
// register indices)
template <int NACC, int J0 = 0> __device__ __forceinline__ void autocov_pieces(unsigned aaddr, unsigned baddr, double *acc)
{
    constexpr int REST = NACC - J0;
    if constexpr (REST > 0) {
        constexpr int NPIECE = (REST + 11) / 12;
        constexpr int NJ = (REST + NPIECE - 1) / NPIECE;
        autocov_piece<NJ, J0>(aaddr, baddr, acc);
        autocov_pieces<NACC, J0 + NJ>(aaddr, baddr, acc);
    }
}

// item = (column cols[], first lag lag_of[] (FIRST: 0), NACC units from unit_of[] on) on grid.x, time block on grid.y.
// partial[(unit * nblk + block) * 16 + e]: the block's share of sum_t x[t] y[t + lag0 + 32 J + 2 e], J the unit within
// the item; FIRST && PAIR: one more unit, NACC, whose entry 0 is the block's share of sum_t x[t]^2.
//
// CENTRE (round 3; mcmchip_initseq_uncentred): the series is taken about a provisional centre per column, z[t] = x[t] -
// centre[col], subtracted while a tile goes from registers to LDS (the zeros before element 0 and past the end stay zeros:
// the first and the last tiles of a series take a path with a range test per element, the tiles between them one without),
// and entry 1 of the sum-of-squares unit carries the block's share of sum_t z[t].  What the sums about the MEAN are follows
// on the host (initseq.hip, centre_correction): the series is read once, not three times (mean pass, subtract in place).
template <int NACC, bool FIRST, bool PAIR, bool CENTRE = false>
__global__ __launch_bounds__(256, (AcShape<NACC, FIRST, PAIR>::OCC)) void autocov_kernel(
    const double *__restrict__ x, long len, const int *__restrict__ cols, const long *__restrict__ lag_of,
    const int *__restrict__ unit_of, long nblk, int tiles_per_block, double *__restrict__ partial,
    const double *__restrict__ centre)
{
    static_assert(!CENTRE || PAIR, "the provisional centre serves initseq's lag pairs");
    using S = AcShape<NACC, FIRST, PAIR>;
    constexpr int SA = S::SA, SY = S::SY, NREG = S::NREG, IMG = S::IMG;
    extern __shared__ double ac_lds[];                    // [2][IMG] tile images; the cross-wave reduction reuses them
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int col = cols[blockIdx.x];
    const long lag0 = FIRST ? 0 : lag_of[blockIdx.x];
    const double *xs = x + (size_t)col * len;
    double acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = 0.0;
    double ssq = 0.0, ssum = 0.0;
    double cen = 0.0;
    if constexpr (CENTRE) cen = centre[col];
    const int q = lane & 3, blk = (lane >> 2) & 3, k = lane >> 4;
    const int offA = 8 * q + blk + 4 * k, offB = blk + 4 * k + 2 * q + 24;
    const unsigned lane_off = (unsigned)t * 8u;

    const long first = (long)blockIdx.y * tiles_per_block * IS_TILE - IS_PAD;
    long ntile = (len - first + IS_TILE - 1) / IS_TILE;         // tiles that start before the end of the series
    if (ntile > tiles_per_block) ntile = tiles_per_block;
    const long stop = first + ntile * IS_TILE;
    // one round of 256 loads per slot of x[g + shift .. ) of this column through a descriptor clipped to the series:
    // elements past the end come back as 0 from the hardware range check (a tile past the end of the block: length 0,
    // zeros and no traffic -- the loop below fetches without asking).  g >= 0.
    auto fetch_seg = [&](double *r, long g, bool live, int nseg, int stride, int shift) __attribute__((always_inline)) {
        long n_el = len - g;
        if (n_el < 0 || !live) n_el = 0;
        if (n_el > (1L << 28)) n_el = 1L << 28;            // the descriptor's length is 32 bits of bytes
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void *)(xs + (g < len ? g : 0)), 0, (int)(unsigned)(n_el * 8), 0x00020000);
        const unsigned voff = lane_off + (unsigned)shift * 8u;
#pragma unroll
        for (int i = 0; i < nseg; i++)
            r[i * stride] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (unsigned)i * 2048u, 0, 0));
    };
    // the first tile of a series starts before element 0 (the zero pad): guarded loads, prologue only
    auto fetch_seg_guarded = [&](double *r, long g, int nseg, int stride, int shift) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < nseg; i++) {
            const long idx = g + shift + t + 256 * i;
            r[i * stride] = (idx >= 0 && idx < len) ? xs[idx] : 0.0;
        }
    };
    // registers of a tile: FIRST: {x[e], x[e+1]} per round of the y image (the x image is its first SA rounds);
    // else the SA rounds of the x image, then {x[e'], x[e'+1]} per round of the y image, e' = e + lag0
    auto fetch = [&](double(&r)[NREG], long a) __attribute__((always_inline)) {
        const bool live = a < stop;
        if constexpr (FIRST) {
            fetch_seg(r, a, live, SY, PAIR ? 2 : 1, 0);
            if constexpr (PAIR) fetch_seg(r + 1, a, live, SY, 2, 1);
        } else {
            fetch_seg(r, a, live, SA, 1, 0);
            fetch_seg(r + SA, a + lag0, live, SY, PAIR ? 2 : 1, 0);
            if constexpr (PAIR) fetch_seg(r + SA + 1, a + lag0, live, SY, 2, 1);
        }
    };
    auto fetch_first_tile = [&](double(&r)[NREG], long a) __attribute__((always_inline)) {
        if (a >= 0) { fetch(r, a); return; }
        if constexpr (FIRST) {
            fetch_seg_guarded(r, a, SY, PAIR ? 2 : 1, 0);
            if constexpr (PAIR) fetch_seg_guarded(r + 1, a, SY, 2, 1);
        } else {
            fetch_seg_guarded(r, a, SA, 1, 0);
            fetch_seg_guarded(r + SA, a + lag0, SY, PAIR ? 2 : 1, 0);
            if constexpr (PAIR) fetch_seg_guarded(r + SA + 1, a + lag0, SY, 2, 1);
        }
    };
    // a = the series position of the tile's element 0 (used by CENTRE only)
    auto store = [&](const double(&r)[NREG], double *P, long a) __attribute__((always_inline)) {
        constexpr int YS = PAIR ? 2 : 1;
        constexpr int Y0 = FIRST ? 0 : SA;
        if constexpr (CENTRE) {
            // registers hold x (0 outside the series, and all of a tile past the block: `fetch`); in range -> x - cen
            const bool live = a < stop;
            const long ya = FIRST ? a : a + lag0;              // position of the y image's (second operand's) element 0
            // in-range tile indices [lo, hi) of the x image's and of the y image's loads (scalars; a tile has < 2^13 elements)
            auto clip = [](long v) __attribute__((always_inline)) { return (int)(v < 0 ? 0 : (v > (1L << 20) ? (1L << 20) : v)); };
            const int lox = clip(-a), hix = live ? clip(len - a) : 0, loy = clip(-ya), hiy = live ? clip(len - ya) : 0;
            auto body = [&](auto inner_c) __attribute__((always_inline)) {
                constexpr bool INNER = inner_c;
                // element t + e0 of an image: t against scalars
                auto zx = [&](double v, int e0) __attribute__((always_inline)) { return (INNER || (t >= lox - e0 && t < hix - e0)) ? v - cen : 0.0; };
                auto zy = [&](double v, int e0) __attribute__((always_inline)) { return (INNER || (t >= loy - e0 && t < hiy - e0)) ? v - cen : 0.0; };
                if constexpr (!FIRST) {
#pragma unroll
                    for (int i = 0; i < SA; i++) P[t + 256 * i] = zx(r[i], 256 * i);
                }
                double zt[IS_TILE / 256];
#pragma unroll
                for (int i = 0; i < SY; i++) {
                    const double v0 = zy(r[Y0 + 2 * i], 256 * i), v1 = zy(r[Y0 + 2 * i + 1], 256 * i + 1);
                    if constexpr (FIRST) {
                        if (i < SA) P[t + 256 * i] = v0;
                        if (i < IS_TILE / 256) zt[i] = v0;      // each position once
                    }
                    P[SA * 256 + t + 256 * i] = v0 + v1;
                }
                if constexpr (FIRST) {
                    // sum and sum of squares of the tile's eight values as trees: two dependent chains of eight would sit in
                    // the way of the loads' consumers
                    static_assert(IS_TILE / 256 == 8, "the trees below take eight values");
                    ssum += ((zt[0] + zt[1]) + (zt[2] + zt[3])) + ((zt[4] + zt[5]) + (zt[6] + zt[7]));
                    ssq += (__builtin_fma(zt[0], zt[0], zt[1] * zt[1]) + __builtin_fma(zt[2], zt[2], zt[3] * zt[3])) +
                           (__builtin_fma(zt[4], zt[4], zt[5] * zt[5]) + __builtin_fma(zt[6], zt[6], zt[7] * zt[7]));
                }
            };
            // every element the tile's loads touch lies inside the series (uniform over the workgroup)
            if (lox == 0 && loy == 0 && hiy >= SY * 256 + 1 && hix >= SA * 256) body(std::true_type{});
            else body(std::false_type{});
            return;
        }
#pragma unroll
        for (int i = 0; i < SA; i++) P[t + 256 * i] = r[FIRST ? YS * i : i];
#pragma unroll
        for (int i = 0; i < SY; i++) P[SA * 256 + t + 256 * i] = PAIR ? r[Y0 + 2 * i] + r[Y0 + 2 * i + 1] : r[Y0 + i];
        if constexpr (FIRST && PAIR) {
#pragma unroll
            for (int i = 0; i < IS_TILE / 256; i++) ssq = __builtin_fma(r[2 * i], r[2 * i], ssq);   // each position once
        }
    };
    auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // LDS byte addresses of this lane's fragments in a buffer: x image, y image
    auto multiply = [&](const double *P) __attribute__((always_inline)) {
        const unsigned aaddr = (unsigned)(uintptr_t)(ac_lds_cd *)(P + wave * (IS_TILE / 4) + offA);
        const unsigned baddr = (unsigned)(uintptr_t)(ac_lds_cd *)(P + SA * 256 + wave * (IS_TILE / 4) + offB);
        autocov_pieces<NACC>(aaddr, baddr, acc);
    };

    // Tile k + 1 goes from registers to the other buffer and the loads of tile k + 1 + DEPTH go out before the MFMAs of
    // tile k; DEPTH register sets (tile j travels in set j % DEPTH), two LDS buffers (tile j in buffer j & 1).  The loop
    // body is unrolled over lcm(2, DEPTH) tiles so that both are compile-time, and it is straight-line code (no `is
    // there a next tile` branches around the loads: the compiler's s_waitcnt placement gives up on counted waits
    // across such joins and drains every load in flight; tiles past the block are fetched through descriptors of
    // length 0 -- zeros, no memory traffic -- and stored, never multiplied).
    constexpr int DEPTH = S::DEPTH, U = (DEPTH % 2 == 0) ? DEPTH : 2 * DEPTH;
    double r[DEPTH][NREG];
    if constexpr (S::NBUF == 2) {
        fetch_first_tile(r[0], first);
        store(r[0], ac_lds, first);
        static_for<DEPTH>([&](auto d) __attribute__((always_inline)) { fetch(r[(d + 1) % DEPTH], first + (long)(d + 1) * IS_TILE); });
        lds_barrier();
        auto phase = [&](auto p, long tile) __attribute__((always_inline)) {   // tile + p: buffer p & 1; stages tile + p + 1, fetches tile + p + 1 + DEPTH
            constexpr int P = p, NEXT = (P + 1) % DEPTH;
            store(r[NEXT], ac_lds + ((P + 1) & 1) * IMG, first + (tile + P + 1) * IS_TILE);
            fetch(r[NEXT], first + (tile + P + 1 + DEPTH) * IS_TILE);
            __builtin_amdgcn_sched_barrier(0);             // the loads go out before the MFMAs, not after them
            multiply(ac_lds + (P & 1) * IMG);
            lds_barrier();            // everyone has read this buffer (the tile after next overwrites it) and written the other
        };
        long tile = 0;
        for (; tile + U <= ntile; tile += U) static_for<U>([&](auto p) __attribute__((always_inline)) { phase(p, tile); });
        static_while<U>([&](auto p) __attribute__((always_inline)) {   // the last tiles of the block: leaves at the first one past the end
            if (tile + p >= ntile) return false;
            phase(p, tile);
            return true;
        });
    } else {
        // one buffer: tile j (register set j % DEPTH) is staged between two barriers, its successor's loads go out, then
        // its MFMAs; the other three workgroups of the CU fill the gaps
        fetch_first_tile(r[0], first);
        static_for<DEPTH - 1>([&](auto d) __attribute__((always_inline)) { fetch(r[d + 1], first + (long)(d + 1) * IS_TILE); });
        auto phase = [&](auto p, long tile) __attribute__((always_inline)) {
            constexpr int P = p % DEPTH;
            lds_barrier();                                 // everyone has read the previous tile
            store(r[P], ac_lds, first + (tile + p) * IS_TILE);
            fetch(r[P], first + (tile + p + DEPTH) * IS_TILE);
            lds_barrier();
            multiply(ac_lds);
        };
        long tile = 0;
        for (; tile + DEPTH <= ntile; tile += DEPTH) static_for<DEPTH>([&](auto p) __attribute__((always_inline)) { phase(p, tile); });
        static_while<DEPTH>([&](auto p) __attribute__((always_inline)) {
            if (tile + p >= ntile) return false;
            phase(p, tile);
            return true;
        });
        lds_barrier();                                     // the reduction below reuses the buffer
    }
    // the four blk groups of a wave, then the 4 waves in order (the images are free: the loop ends with a barrier)
    double(*red)[16 * NACC + 16] = (double(*)[16 * NACC + 16]) ac_lds;   // [4][16 NACC + 16] <= 2 images
#pragma unroll
    for (int j = 0; j < NACC; j++) {
        double v = acc[j];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (blk == 0) red[wave][16 * j + 12 + q - 4 * k] = v;     // result lane = j' + 4*blk + 16*i: q = j', k = i
    }
    if constexpr (FIRST && PAIR) {
        // store() squared every tile it staged: the ones past the block are zeros
        for (int m = 32; m > 0; m >>= 1) ssq += __shfl_xor(ssq, m, 64);
        if constexpr (CENTRE)
            for (int m = 32; m > 0; m >>= 1) ssum += __shfl_xor(ssum, m, 64);
        if (lane < 16) red[wave][16 * NACC + lane] = lane == 0 ? ssq : (lane == 1 ? ssum : 0.0);
    }
    __syncthreads();
    constexpr int NOUT = 16 * NACC + (FIRST && PAIR ? 16 : 0);
    for (int e = t; e < NOUT; e += 256) {
        const double v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        const int unit = unit_of[blockIdx.x] + e / IS_B;
        partial[((size_t)unit * nblk + blockIdx.y) * IS_B + e % IS_B] = v;
    }
}

// launches one variant; LDS = two buffers of an x and a y image (the cross-wave reduction reuses them at the end)
template <int NACC, bool FIRST, bool PAIR, bool CENTRE = false>
int launch_autocov(dim3 grid, hipStream_t s, const double *x, long len, const int *pc, const long *pl, const int *pu,
                   long nblk, int tiles_per_block, double *pp, const double *centre = nullptr)
{
    using S = AcShape<NACC, FIRST, PAIR>;
    const size_t lds = (size_t)S::NBUF * S::IMG * 8;
    auto kern = autocov_kernel<NACC, FIRST, PAIR, CENTRE>;
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, len, pc, pl, pu, nblk, tiles_per_block, pp, centre);
    return MCMCHIP_OK;
}

// the instantiation units: MCMCHIP_OK / an error, or -1 when the unit does not hold this accumulator count
#define AUTOCOV_LAUNCH_ARGS int nacc, dim3 grid, hipStream_t s, const double *x, long len, const int *pc, const long *pl, \
                            const int *pu, long nblk, int tiles_per_block, double *pp
int launch_autocov_first_a(AUTOCOV_LAUNCH_ARGS);      // pairs, lag 0 on
int launch_autocov_first_b(AUTOCOV_LAUNCH_ARGS);
int launch_autocov_cont(AUTOCOV_LAUNCH_ARGS);         // pairs, continuation
int launch_autocov_raw(AUTOCOV_LAUNCH_ARGS);          // single lags of one parity (mcmchip_lagsums)
// ... about a provisional centre per column (mcmchip_initseq_uncentred)
#define AUTOCOV_CENTRE_ARGS AUTOCOV_LAUNCH_ARGS, const double *centre
int launch_autocov_first_ca(AUTOCOV_CENTRE_ARGS);
int launch_autocov_first_cb(AUTOCOV_CENTRE_ARGS);
int launch_autocov_cont_c(AUTOCOV_CENTRE_ARGS);
