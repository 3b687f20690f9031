// initseq_kernels.h -- the lag-product kernel of initseq() (initseq.c:24-47) on the fp64 matrix cores and its launcher;
// instantiated per accumulator count in initseq_first_a.hip / initseq_first_b.hip (items that start at lag 0) and
// initseq_cont.hip (continuation items), driven by initseq.hip.
#pragma once
#include "kernels.h"

#define IS_B 16                 // lags per work unit (one 16-lag accumulator per lane)
#define IS_TILE 2048            // base positions per LDS tile
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
// Work decomposition (round 2).  ONE work item per column and launch round: NACC accumulators (16 lags each, up to
// 768 lags) are worked off in near-equal pieces of at most 16 lag blocks over the SAME staged tile, so a column is
// streamed from HBM once per round whatever its lag count.  The tile is fetched through buffer descriptors rebuilt per
// tile in scalar registers: the hardware range check returns 0 past the end of the series, the loads need no
// predicate and no per-lane address arithmetic beyond one precomputed byte offset.  Only the first tile of a series,
// which starts 16 positions before element 0, takes guarded loads.  When the lag-shifted operand overlaps the base
// tile ONE contiguous image x[a .. a + 2048 + lag0 + 16 NACC + 35) serves both operands.
//
// Pipeline (round 3).  Round 2's loop was  barrier, registers -> LDS, barrier, issue the next tile's loads, MFMAs:
// measured at C5 a tile cost 2 T + 1.0 us (T = the MFMA time of one wave, two workgroups per CU sharing the SIMDs),
// i.e. about 2400 cycles per tile in which NEITHER resident workgroup issued an MFMA -- the two run in lockstep
// (they alternate on the matrix pipe, so they reach their staging phase together), and for the short items, whose T
// (0.4-0.9 us) is far below the HBM latency, one tile of loads in flight per workgroup is not enough either.  Now:
//   * the LDS image is double buffered: the registers holding tile k+1 are written to the other buffer and the loads
//     of the tile after are issued BEFORE the MFMAs of tile k, so a wave's matrix stream is interrupted once per tile
//     by one LDS-only barrier (s_waitcnt lgkmcnt(0); s_barrier -- __syncthreads would also drain the loads in flight);
//   * items of up to 128 lags keep TWO tiles of loads in flight per workgroup (two register sets) and run 4 (<= 64
//     lags) or 3 workgroups per CU instead of 2: 8 / 6 tiles in flight per CU instead of 2;
//   * the kernel is instantiated twice: FIRST (the item starts at lag 0: every column's first full pass, > 90 % of the
//     work) with one descriptor and the image length known at compile time (9 rounds of loads per thread up to 208 lags),
//     and the general form for continuation items (two descriptors, merged at run time when they overlap);
//   * the time blocks a column is cut into follow the length of the series (1 to 64 tiles), not a constant 32.
template <int NACC, bool FIRST> struct AcShape {
    static constexpr int NA = IS_TILE + 32, NB_ = IS_TILE + 16 * NACC + 32;
    static constexpr int SA = (NA + 255) / 256, SB = (NB_ + 255) / 256;      // rounds of 256 loads per operand
    static constexpr int NSLOT = FIRST ? (3 + NB_ + 255) / 256 : SA + SB;     // FIRST: one image x[a .. a + 3 + NB_)
    static constexpr int DEPTH = NACC <= (FIRST ? 8 : 4) ? 2 : 1;             // tiles of loads in flight per workgroup
    static constexpr int OCC = !FIRST ? (NACC <= 4 ? 3 : 2) : (NACC <= 4 ? 4 : (NACC <= 8 ? 3 : 2));   // workgroups per CU
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

template <int NACC, bool FIRST>
__global__ __launch_bounds__(256, (AcShape<NACC, FIRST>::OCC)) void autocov_kernel(
    const double *__restrict__ x, long len, const int *__restrict__ cols, const long *__restrict__ lag_of,
    const int *__restrict__ unit_of, long nblk, int tiles_per_block, int img_doubles, double *__restrict__ partial)
{
    using S = AcShape<NACC, FIRST>;
    constexpr int SA = S::SA, SB = S::SB, NSLOT = S::NSLOT;
    extern __shared__ double ac_lds[];                    // [2][img_doubles] tile images; the cross-wave reduction reuses them
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    // work item = (column, 16 NACC lags) on grid.x, time range on grid.y
    const int col = cols[blockIdx.x];
    const long lag0 = FIRST ? 0 : lag_of[blockIdx.x];
    const double *xs = x + (size_t)col * len;
    double acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = 0.0;
    const int q = lane & 3, blk = (lane >> 2) & 3, k = lane >> 4;
    const int offA = q + blk + 4 * k, offB = blk + 4 * k + 4 * q;
    // one image for both operands when the shifted one ends inside the padded image
    const bool merged = FIRST || lag0 + 3 + S::NB_ <= (long)NSLOT * 256;
    const long shiftB = merged ? (long)SA * 256 : lag0 + 3;         // element offset of the second descriptor from the tile base
    const int boff = merged ? (int)(lag0 + 3) : SA * 256;           // where the lag-shifted operand starts in the LDS image
    const long lenA = merged ? lag0 + 3 + S::NB_ : S::NA;            // elements the first descriptor may deliver (from the tile base)
    const long lenB = merged ? lenA - (long)SA * 256 : S::NB_;       // ... and the second (from its own base)
    const int nslot = FIRST ? NSLOT : (merged ? (int)((lenA + 255) / 256) : NSLOT);    // rounds actually needed (wave-uniform)
    const unsigned lane_off = (unsigned)t * 8u;

    const long first = (long)blockIdx.y * tiles_per_block * IS_TILE - 16;
    long ntile = (len - first + IS_TILE - 1) / IS_TILE;         // tiles that start before the end of the series
    if (ntile > tiles_per_block) ntile = tiles_per_block;
    const long stop = first + ntile * IS_TILE;
    // descriptor over x[g .. g + cnt) of this column clipped to len: elements past the end come back as 0 from the
    // hardware range check (a tile past the end of the series is all zeros: the loop below fetches without asking).
    // Only the very first tile of a series starts before element 0 (g = -16): its descriptor starts at element 0 and
    // the lane offsets are 128 bytes lower, so the first 16 wrap around to 4 GB and are out of range = 0 as well.
    auto fetch_seg = [&](double(&r)[NSLOT], long g, long cnt, int slot0, int nseg) {
        const long lead = g < 0 ? -g : 0;                  // elements before the start of the series
        long n_el = cnt - lead;
        if (g + cnt > len) n_el = len - (g + lead);
        if (n_el < 0) n_el = 0;
        const long g0 = g + lead;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void *)(xs + (g0 < len ? g0 : 0)), 0, (int)(unsigned)(n_el * 8), 0x00020000);
        const unsigned voff = lane_off - (unsigned)lead * 8u;
#pragma unroll
        for (int i = 0; i < NSLOT; i++)
            if (i >= slot0 && i < slot0 + nseg && i < nslot)
                r[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (unsigned)(i - slot0) * 2048u, 0, 0));
    };
    auto fetch = [&](double(&r)[NSLOT], long a) {
        const bool live = a < stop;                        // a tile of the next block: descriptors of length 0, no traffic
        if constexpr (FIRST) {
            fetch_seg(r, a, live ? (long)NSLOT * 256 : 0, 0, NSLOT);
        } else {
            fetch_seg(r, a, live ? (lenA < (long)SA * 256 ? lenA : (long)SA * 256) : 0, 0, SA);
            fetch_seg(r, a + shiftB, live ? lenB : 0, SA, SB);
        }
    };
    auto store = [&](const double(&r)[NSLOT], double *P) {
#pragma unroll
        for (int i = 0; i < NSLOT; i++)
            if (i < nslot) P[t + 256 * i] = r[i];
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto multiply = [&](const double *P) {
        autocov_pieces<NACC>(P + wave * (IS_TILE / 4) + offA, P + boff + wave * (IS_TILE / 4) + offB, acc);
    };

    double *P0 = ac_lds, *P1 = ac_lds + img_doubles;
    // The loops below are straight-line code per tile (no `is there a next tile` branches: the compiler's s_waitcnt
    // placement gives up on counted waits across such joins and drains every load in flight): tiles past the last
    // one are fetched (descriptors of length 0: zeros, no memory traffic) and stored too, and never multiplied.
    if constexpr (S::DEPTH == 1) {
        double r[NSLOT];
        fetch(r, first);
        store(r, P0);
        fetch(r, first + IS_TILE);
        lds_barrier();
        for (long tile = 0; tile < ntile; tile++) {
            double *cur = (tile & 1) ? P1 : P0, *nxt = (tile & 1) ? P0 : P1;
            store(r, nxt);                                                   // tile + 1, loaded during the tile before
            fetch(r, first + (tile + 2) * IS_TILE);
            __builtin_amdgcn_sched_barrier(0);         // the loads go out before the MFMAs, not after them
            multiply(cur);
            lds_barrier();        // everyone has read `cur` (the tile after next overwrites it) and written `nxt`
        }
    } else {
        static_assert(S::DEPTH == 2, "one or two tiles of loads in flight");
        double r0[NSLOT], r1[NSLOT];
        fetch(r0, first);
        store(r0, P0);
        fetch(r0, first + IS_TILE);                              // odd tiles travel in r0, even ones in r1
        fetch(r1, first + 2 * IS_TILE);
        lds_barrier();
        long tile = 0;
        for (; tile + 1 < ntile; tile += 2) {
            store(r0, P1);
            fetch(r0, first + (tile + 3) * IS_TILE);
            __builtin_amdgcn_sched_barrier(0);         // the loads go out before the MFMAs, not after them
            multiply(P0);
            lds_barrier();
            store(r1, P0);
            fetch(r1, first + (tile + 4) * IS_TILE);
            __builtin_amdgcn_sched_barrier(0);
            multiply(P1);
            lds_barrier();
        }
        if (tile < ntile) {
            multiply(P0);
            lds_barrier();
        }
    }
    // the four blk groups of a wave, then the 4 waves in order (the images are free: the loop ends with a barrier)
    double(*red)[16 * NACC] = (double(*)[16 * NACC])ac_lds;   // [4][16 NACC] <= 2 images
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


// launches one variant; LDS = two tile images (the [4][16 NACC] cross-wave reduction reuses them at the end).
// FIRST: every item of the launch starts at lag 0; else `img` is the image length the launch's largest lag0 needs.
template <int NACC, bool FIRST>
int launch_autocov(dim3 grid, hipStream_t s, const double *x, long len, const int *pc, const long *pl, const int *pu,
                   long nblk, int tiles_per_block, int img, double *pp)
{
    using S = AcShape<NACC, FIRST>;
    if (FIRST) img = S::NSLOT * 256;
    const size_t lds = (size_t)2 * img * 8;
    auto kern = autocov_kernel<NACC, FIRST>;
    if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, len, pc, pl, pu, nblk, tiles_per_block, img, pp);
    return MCMCHIP_OK;
}

// the instantiation units: MCMCHIP_OK / an error, or -1 when the unit does not hold this accumulator count
int launch_autocov_first_a(int nacc, dim3 grid, hipStream_t s, const double *x, long len, const int *pc, const long *pl,
                           const int *pu, long nblk, int tiles_per_block, int img, double *pp);
int launch_autocov_first_b(int nacc, dim3 grid, hipStream_t s, const double *x, long len, const int *pc, const long *pl,
                           const int *pu, long nblk, int tiles_per_block, int img, double *pp);
int launch_autocov_cont(int nacc, dim3 grid, hipStream_t s, const double *x, long len, const int *pc, const long *pl,
                        const int *pu, long nblk, int tiles_per_block, int img, double *pp);
// image length (doubles, a multiple of 256) a continuation item of `nacc` accumulators starting at lag0 needs
long autocov_cont_image(int nacc, long lag0);
