// olbm_lane.hip -- olbm() (reference olbm.c:22-78) for 3 .. 192 columns and windows of >= 256 rows: a wave walks FOUR TEETH of
// a comb with one lane per (tooth, column), and nothing is scanned.
//
// v_mfma_f64_16x16x4_f64 takes a 16 x 4 operand: lane l supplies element (column l % 16, k = l / 16).  olbm_mfma_kernel
// and olbm_comb_kernel fill k with four CONSECUTIVE windows, which is why they need a producer: window sums of 16
// consecutive windows are a prefix scan of the differences x[t + len] - x[t] (olbm.c:61-62), across lanes.  But the sum
// over windows of (window sum)(window sum)^T does not care which four windows share an instruction.  Here k is the
// TOOTH: lane (j, c) walks the windows  base + j * len + i,  i = 0, 1, 2, ...  of column c one after another, its window
// sum is a running sum in a register (D += x[t + len] - x[t]: one add per step, no scan, no cross-lane traffic), and the
// four teeth are len apart, so the row tooth j adds is the row tooth j + 1 drops: a chunk of T steps needs five rows
// per column for four teeth -- 1.25 x the input instead of 2 x (the comb of olbm_comb.hip, for free).
//
// As there, D is relative to the (unknown) window sum s at the start of a piece (a run of windows of one tooth):
//     sum_w (s + D_w)(s + D_w)^T = sum_w D_w D_w^T  +  cnt s s^T + s d^T + d s^T,        d = sum_w D_w,
// the matrix cores accumulate sum D D^T, a lane keeps d and the piece's last D (e), and the pieces are chained
// afterwards from one direct sum per column (olbm_comb.hip's chain kernels, shared); the exact means (olbm.c:38-43) come
// out of the chain too.
//
// Columns come in blocks of 16 (NB = 1 .. 12); a workgroup has one wave per block.  Wave w loads its block -- whole 128-byte
// runs of a column, coalesced, [column][time] -- keeps the NEXT chunk in registers while the current one is worked on
// (40 VGPRs in flight per lane: 10 KB per wave), takes the differences in that layout and turns them to [tooth][column]
// through LDS (odd pitch, no bank conflicts), where the blocks meet: wave w accumulates the tiles (w, w), (w, w+1), ..,
// (w, w + NB/2) of the circulant, every unordered pair of blocks once (for even NB the farthest pair is shared between
// waves w and w + NB/2, half of the steps each).  Two workgroup barriers per chunk; NB = 1 needs none.
//
// The plan deals units of the sweep round-robin to the workgroups (LanePlan below): the chip then works inside one
// stretch of every row and column at a time, which is what the DRAM pages and the address translation want -- 6.5 TB/s
// of HBM traffic at p = 16 by the counters, and no slowdown for windows of 1e5 .. 1e6 rows (contiguous shares: 2 .. 2.6 x).
#include "kernels.h"
#include "olbm_shape.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <chrono>
#include <stdio.h>
#include <vector>

namespace {

// beyond 12 blocks of 16 columns a workgroup of one wave per block leaves a wave 128 registers: the accumulators of 7+ tiles,
// the chunk in flight and the running sums do not fit (spills); 193 .. 256 columns stay with olbm_comb.hip
#define OL_LANE_MAX_P 192

// The plan in closed form (no arrays: a call of n = 1e8, p = 16 spent 0.3 - 0.7 of its 3 ms building and uploading 27 000
// units and 110 000 piece counts).  Teeth t = 0 .. Q-1 of len windows (tooth t starts at window t * len) and one of
// nwin % len; super-blocks of four consecutive full teeth (the last one: m_last of them) are swept together.  A
// super-block's sweep of len windows is cut into U units of ru windows (whole multiples of 16; the last one the rest),
// the units are numbered in time order and dealt round-robin: share g walks units g, g + nshare, ...  At any moment the
// workgroups of the chip are then inside the same stretch of nshare * ru rows of every tooth row and column: the pages
// and DRAM rows they touch are shared instead of one private set of 5 * 16 NB streams per wave (n = 1e8, len = 1e6 rows,
// 64 columns: 43 ms with contiguous shares -- address translation -- against 17 ms dealt this way; len = 1e4, 16
// columns: 3.7 against 2.6 ms).  Pieces (a unit's run of one tooth) are numbered in time order: super-block, tooth, unit.
struct LanePlan {
    long len, rem, Q;              // window length, nwin % len, nwin / len
    int nsb, m_last;               // super-blocks of full teeth, teeth of the last one
    int U, ru;                     // units per super-block, windows per unit
    int Ur, rur;                   // the same for the remainder tooth (Ur = 0: none)
    int nreg, nunit;               // units of the super-blocks, all units
    int npreg, np;                 // pieces of the super-blocks, all pieces
    int nshare;
};

__host__ __device__ inline void lane_unit(const LanePlan &P, int u, OlbmComb &c)
{
    if (u < P.nreg) {
        const int sb = u / P.U, iu = u - sb * P.U;
        c.base = (long)sb * 4 * P.len + (long)iu * P.ru;
        c.m = sb == P.nsb - 1 ? P.m_last : 4;
        const long rest = P.len - (long)iu * P.ru;
        c.rlen = (int)(rest < P.ru ? rest : P.ru);
        c.piece0 = sb * 4 * P.U + iu;
        c.pstride = P.U;
    } else {
        const int iu = u - P.nreg;
        c.base = P.Q * P.len + (long)iu * P.rur;
        c.m = 1;
        const long rest = P.rem - (long)iu * P.rur;
        c.rlen = (int)(rest < P.rur ? rest : P.rur);
        c.piece0 = P.npreg + iu;
        c.pstride = P.Ur;
    }
}
// windows of piece j and whether it starts a tooth (a window at a multiple of len)
__host__ __device__ inline void lane_piece(const LanePlan &P, int j, int &cnt, bool &tile)
{
    int iu, ru;
    long R;
    if (j < P.npreg) { iu = j % P.U; ru = P.ru; R = P.len; }
    else { iu = j - P.npreg; ru = P.rur; R = P.rem; }
    const long rest = R - (long)iu * ru;
    cnt = (int)(rest < ru ? rest : ru);
    tile = iu == 0;
}

// units of about R windows: between 6 and 12 rounds per share, the candidate whose busiest share sweeps the fewest windows
// (more rounds: more pieces to chain afterwards).  sub > 1 (olbm_few_kernel): sub consecutive units of a sweep form a
// bundle, and it is bundles that are dealt.
LanePlan make_lane_plan(long nwin, long len, long groups, int sub = 1)
{
    LanePlan P{};
    P.len = len;
    P.Q = nwin / len;
    P.rem = nwin % len;
    P.nsb = (int)((P.Q + 3) / 4);
    P.m_last = P.nsb ? (int)(P.Q - 4L * (P.nsb - 1)) : 0;
    if (groups < 1) groups = 1;
    const long total = (long)P.nsb * len + P.rem;
    auto cut = [](long rt, long R, int &U, int &ru) {         // equal units inside a sweep of R windows
        if (R <= 0) { U = 0; ru = 16; return; }
        const long nu = (R + rt - 1) / rt;
        long r = (R + nu - 1) / nu;
        r = (r + 15) / 16 * 16;
        ru = (int)r;
        U = (int)((R + r - 1) / r);
    };
    auto bundles = [sub](int U, int ru, long R) { return (long)((U + sub - 1) / sub); };      // of a sweep: its units in groups of sub
    long best_rt = 256;
    double best_cost = 1e300;
    for (int k = sub > 1 ? 4 : 6; k <= (sub > 1 ? 8 : 12); k++) {
        long rt = total / (groups * k * sub);
        rt = (rt + 15) / 16 * 16;
        if (rt < 256) rt = 256;
        int U, ru, Ur, rur;
        cut(rt, P.nsb ? len : 0, U, ru);
        cut(rt, P.rem, Ur, rur);
        const long nb = (long)P.nsb * bundles(U, ru, len) + bundles(Ur, rur, P.rem);
        const long rounds = (nb + groups - 1) / groups;
        const double cost = (double)rounds * (double)(ru > rur || !Ur ? ru : rur);      // windows the busiest share sweeps
        if (cost < best_cost * (1.0 - 1e-9)) { best_cost = cost; best_rt = rt; }
        if (rt == 256) break;
    }
    cut(best_rt, P.nsb ? len : 0, P.U, P.ru);
    cut(best_rt, P.rem, P.Ur, P.rur);
    if (P.U < 1) P.U = 1;                                     // (divisor of the decode; no super-block: nreg = 0)
    P.nreg = P.nsb ? P.nsb * P.U : 0;
    P.nunit = P.nreg + P.Ur;
    P.npreg = P.nsb ? ((P.nsb - 1) * 4 + P.m_last) * P.U : 0;
    P.np = P.npreg + P.Ur;
    P.nshare = (int)(P.nunit < groups ? P.nunit : groups);
    return P;
}

template <int NB, int TT> struct LaneShape {
    static constexpr int PP = 16 * NB;                              // padded columns
    static constexpr int T = TT;                                    // steps per chunk
    static constexpr int TP = T + 1;                                // LDS pitch of a (tooth, column) run: odd
    static constexpr int SEG = 64 / T;                              // column runs per load instruction
    static constexpr int KPR = 16 / SEG;                            // load instructions per tooth row (a wave loads 16 columns)
    static constexpr int H = NB / 2;
    static constexpr int NK = H + 1;                                // tiles of wave w: (w, (w + k) % NB), k = 0 .. H
    static constexpr int NTT = NB * NK;                             // tile slots
    static constexpr size_t lds_bytes = (size_t)NB * 4 * 16 * TP * 8;
    // even NB: the pair (w, w + H) would be met twice by the circulant; waves w and w + H share it instead -- w takes the
    // first half of every group of four (two) steps as tile (w, w + H), w + H the other half as its transpose, and the
    // finish adds them
    static constexpr bool SHARED_H = NB > 1 && NB % 2 == 0;
    __host__ __device__ static constexpr bool is_partner(int w, int k) { return SHARED_H && k == H && w >= NB / 2; }
};

// A workgroup = one share of the plan; wave w loads the 16 columns of block w (five rows of T steps for four teeth),
// the differences of all blocks meet in LDS, wave w accumulates the tiles (w, w), (w, w+1), .. (circulant: every unordered
// pair of blocks once) from running sums of its own and the next H blocks.
template <int NB, int TT>
__global__ __launch_bounds__(64 * NB, (NB == 7 || NB == 8) ? 4 : 1) void olbm_lane_kernel(const double *__restrict__ x, long n, int p, long len,
                                                            const LanePlan P, double *__restrict__ partial,
                                                            double *__restrict__ pd, double *__restrict__ pe)
{
    using S = LaneShape<NB, TT>;
    constexpr int T = S::T, TP = S::TP, SEG = S::SEG, KPR = S::KPR, PP = S::PP, NK = S::NK;
    extern __shared__ double W[];                                   // [NB][4][16][TP]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int g = blockIdx.x;
    const int j = lane >> 4, c = lane & 15;                         // the walk: tooth, column inside a block of 16
    const int li = lane % T, lseg = lane / T;                       // the loads: step inside the chunk, column run
    const bool upper = S::SHARED_H && wave >= NB / 2;               // which half of the steps of the shared k = H tile

    om_double4 acc[NK];
#pragma unroll
    for (int t = 0; t < NK; t++) acc[t] = om_double4{0.0, 0.0, 0.0, 0.0};
    double D[NK], dsum = 0.0;
#pragma unroll
    for (int q = 0; q < NK; q++) D[q] = 0.0;

    struct Cur { int ci, i0, m, rlen, piece0, pstride; long base; };
    const int ci_end = P.nunit;                                     // ci: the unit; share g walks g, g + nshare, ...
    auto load_comb = [&](Cur &u) {
        u.i0 = 0;
        if (u.ci < ci_end) {
            OlbmComb cb;
            lane_unit(P, u.ci, cb);                                 // uniform arithmetic
            u.m = cb.m;
            u.rlen = cb.rlen;
            u.piece0 = cb.piece0;
            u.pstride = cb.pstride;
            u.base = cb.base;
        } else {
            u.m = 0;                                                // past the end: loads land where they did before, nothing is kept
            u.rlen = 0;
        }
    };
    auto advance = [&](Cur &u) {
        u.i0 += T;
        if (u.i0 >= u.rlen) {
            u.ci += P.nshare;
            load_comb(u);
        }
    };

    // the lane's columns in the loads, clamped into the matrix (what a clamped load returns is replaced by zero later)
    size_t coloff[KPR];
#pragma unroll
    for (int k = 0; k < KPR; k++) {
        const int col = 16 * wave + k * SEG + lseg;
        coloff[k] = (size_t)(col < p ? col : p - 1) * (size_t)n;
    }
    double st[5][KPR];                                              // the chunk in flight: rows of tooth 0 .. 4
    auto issue = [&](const Cur &u) {
        const long tb = u.base + u.i0 + li;
#pragma unroll
        for (int jj = 0; jj < 5; jj++) {
            long t = tb + jj * len;
            t = t < n ? t : n - 1;
#pragma unroll
            for (int k = 0; k < KPR; k++) st[jj][k] = x[coloff[k] + t];
        }
    };
    double *Wc = W + (size_t)wave * 64 * TP + li;                   // commit: block = wave
    auto commit = [&](const Cur &u) {
        const long tb = u.base + u.i0 + li;
#pragma unroll
        for (int jt = 0; jt < 4; jt++) {
            const bool okd = jt < u.m && tb + (jt + 1) * len < n;   // the last window of the series adds nothing (olbm.c:60)
#pragma unroll
            for (int k = 0; k < KPR; k++) {
                const int cc = k * SEG + lseg;
                const double del = okd && 16 * wave + cc < p ? st[jt + 1][k] - st[jt][k] : 0.0;
                Wc[(jt * 16 + cc) * TP] = del;
            }
        }
    };
    auto sync = [&]() {
        if constexpr (NB == 1) {                                    // one wave: LDS operations complete in order
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else __syncthreads();
    };

    Cur cur;
    cur.ci = g;
    cur.base = 0;
    cur.m = cur.rlen = cur.piece0 = cur.pstride = 0;
    load_comb(cur);
    issue(cur);
    const double *Wr[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) Wr[k] = W + (size_t)((wave + k) % NB) * 64 * TP + (j * 16 + c) * TP;
    auto step = [&](const double (&d)[NK], bool with_h) {          // with_h: wave-uniform
#pragma unroll
        for (int k = 0; k < NK - 1; k++) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(D[0], D[k], acc[k], 0, 0, 0);
        if (!S::SHARED_H || with_h) acc[NK - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(D[0], D[NK - 1], acc[NK - 1], 0, 0, 0);
        dsum += D[0];
#pragma unroll
        for (int k = 0; k < NK; k++) D[k] += d[k];
    };
    auto walk = [&](int nst) {
        constexpr int G = NB >= 5 ? 2 : 4;                          // steps whose differences are read ahead (registers: G NK doubles)
        int i = 0;
        for (; i + G <= nst; i += G) {
            double d[G][NK];
#pragma unroll
            for (int u = 0; u < G; u++)
#pragma unroll
                for (int k = 0; k < NK; k++) d[u][k] = Wr[k][i + u];
#pragma unroll
            for (int u = 0; u < G / 2; u++) step(d[u], !upper);
#pragma unroll
            for (int u = G / 2; u < G; u++) step(d[u], upper);
        }
        for (; i < nst; i++) {                                      // the ragged end of a unit: the lower wave's
            double d[NK];
#pragma unroll
            for (int k = 0; k < NK; k++) d[k] = Wr[k][i];
            step(d, !upper);
        }
    };
    while (cur.ci < ci_end) {
        commit(cur);
        sync();
        Cur nxt = cur;
        advance(nxt);
        issue(nxt);
        int nst = cur.rlen - cur.i0;
        nst = nst < T ? nst : T;
        walk(nst);
        if (cur.i0 + T >= cur.rlen) {                               // the comb is done: its pieces' d and e, in time order
            if (j < cur.m) {
                const size_t at = (size_t)(cur.piece0 + j * cur.pstride) * PP + 16 * wave + c;
                pd[at] = dsum;
                pe[at] = D[0];
            }
            dsum = 0.0;
#pragma unroll
            for (int k = 0; k < NK; k++) D[k] = 0.0;
        }
        sync();
        cur = nxt;
    }
    double *out = partial + ((size_t)g * S::NTT + (size_t)wave * NK) * 256;
#pragma unroll
    for (int t = 0; t < NK; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) out[t * 256 + r * 64 + lane] = acc[t][r];
}

// one split of the shares' matrix-core sums plus one split of the pieces' rank-two terms, in the accumulator layout of
// v_mfma_f64_16x16x4 (register r of lane l: row l / 16 + 4 r, column l % 16); (row, col) and (col, row) of a diagonal
// tile evaluate the expression of the ordered pair and agree bit for bit
template <int NB, int TT>
__global__ __launch_bounds__(256) void olbm_lane_reduce_kernel(const double *__restrict__ partial, int nshare, int share_split,
                                                               const double *__restrict__ pe, const double *__restrict__ pd,
                                                               const int *__restrict__ pcnt, int npieces, int per_split,
                                                               const double *__restrict__ shift, double *__restrict__ out)
{
    using S = LaneShape<NB, TT>;
    constexpr int PP = S::PP, NTT = S::NTT;
    const int t = blockIdx.x, e = threadIdx.x, r = e >> 6, l = e & 63;
    const int qa = t / S::NK, k = t % S::NK, qb = (qa + k) % NB;
    const int row = 16 * qa + (l >> 4) + 4 * r, col = 16 * qb + (l & 15);
    const int ia = row < col ? row : col, ib = row < col ? col : row;
    double s = 0.0;
    const int g0 = blockIdx.y * share_split, g1 = g0 + share_split < nshare ? g0 + share_split : nshare;
#pragma unroll 8
    for (int g = g0; g < g1; g++) s += partial[((size_t)g * NTT + t) * 256 + e];
    const double sha = shift[ia], shb = shift[ib];
    const int j0 = blockIdx.y * per_split, j1 = S::is_partner(qa, k) ? 0 : (j0 + per_split < npieces ? j0 + per_split : npieces);
    double cs = 0.0;                                                // a partner slot holds matrix-core sums only
#pragma unroll 8
    for (int jp = j0; jp < j1; jp++) {
        const double sa = sha + pe[(size_t)jp * PP + ia], sb = shb + pe[(size_t)jp * PP + ib];
        const double da = pd[(size_t)jp * PP + ia], db = pd[(size_t)jp * PP + ib];
        const double tt = fma((double)pcnt[jp], sa, da);
        cs = fma(tt, sb, cs);
        cs = fma(sa, db, cs);
    }
    out[((size_t)blockIdx.y * NTT + t) * 256 + e] = s + cs;
}

template <int NB, int TT>
__global__ __launch_bounds__(256) void olbm_lane_finish_kernel(const double *__restrict__ red, int nsplit, int p, double denom,
                                                               double *__restrict__ var)
{
    using S = LaneShape<NB, TT>;
    const int t = blockIdx.x, e = threadIdx.x, r = e >> 6, l = e & 63;
    const int qa = t / S::NK, k = t % S::NK, qb = (qa + k) % NB;
    if (S::is_partner(qa, k)) return;
    double s = 0.0;
#pragma unroll 16
    for (int i = 0; i < nsplit; i++) s += red[((size_t)i * S::NTT + t) * 256 + e];
    if (S::SHARED_H && k == S::H) {                                    // the other half of the steps, transposed, in wave qb's slot
        const int ri = (l >> 4) + 4 * r, ci = l & 15;
        const int tp = qb * S::NK + S::H, ep = (ci >> 2) * 64 + (ci & 3) * 16 + ri;
        double sp = 0.0;
#pragma unroll 16
        for (int i = 0; i < nsplit; i++) sp += red[((size_t)i * S::NTT + tp) * 256 + ep];
        s += sp;
    }
    const int row = 16 * qa + (l >> 4) + 4 * r, col = 16 * qb + (l & 15);
    if (row < p && col < p) {
        const double v = s / denom;                                    // olbm.c:74
        var[row + (size_t)p * col] = v;
        var[col + (size_t)p * row] = v;                                // olbm.c:75
    }
}

// piece counts and tooth starts of the plan for the chain and the rank-two kernels
__global__ void lane_fill_pieces_kernel(const LanePlan P, int *__restrict__ pcnt, unsigned char *__restrict__ ptile)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.np) return;
    int cnt;
    bool tile;
    lane_piece(P, j, cnt, tile);
    pcnt[j] = cnt;
    ptile[j] = tile;
}

// the plan as arrays, share by share (for the host-side check only)
void lane_plan_arrays(const LanePlan &P, CombPlan &pl)
{
    pl.npieces = P.np;
    pl.pcnt.assign((size_t)P.np, 0);
    pl.ptile.assign((size_t)P.np, 0);
    for (int j = 0; j < P.np; j++) {
        int cnt;
        bool tile;
        lane_piece(P, j, cnt, tile);
        pl.pcnt[(size_t)j] = cnt;
        pl.ptile[(size_t)j] = tile;
    }
    pl.combs.clear();
    pl.first.assign(1, 0);
    pl.steps.clear();
    pl.max_steps = 0;
    for (int g = 0; g < P.nshare; g++) {
        long st = 0;
        for (int u = g; u < P.nunit; u += P.nshare) {
            OlbmComb c;
            lane_unit(P, u, c);
            pl.combs.push_back(c);
            st += (long)c.m * ((c.rlen + 15) / 16);
        }
        pl.first.push_back((int)pl.combs.size());
        pl.steps.push_back((int)st);
        if (st > pl.max_steps) pl.max_steps = st;
    }
}

// ---------------------------------------------------------------------------------------------
// p <= 8: several units side by side in one instruction
// ---------------------------------------------------------------------------------------------
// With 16 column slots per instruction and p <= 8 (<= 4) columns, half (three quarters) of every MFMA would multiply
// padding: at p <= 8 the walk above is bound by the matrix pipe, not by HBM (1.5 ms per 1e8 rows whatever p is).  Here
// the 16 slots hold SUB = 16 / CG consecutive UNITS of the plan side by side (CG = 4 or 8 columns each): lane
// (tooth j, slot v) walks unit u0 + v / CG of a bundle, column v % CG.  The diagonal CG x CG blocks of the 16 x 16
// accumulator are the units' sums (the off-diagonal blocks, products across units, are dropped); units, pieces, chain
// and rank-two terms are the plan's, unchanged.  Only the sweep's last unit can be shorter than the others of its bundle:
// the chunks in which it has ended give its lanes a zero operand.
struct FewPlan {
    LanePlan L;
    int SUB;                       // units side by side
    int B, Br;                     // bundles of a super-block's sweep, of the remainder tooth's
    int nbund, nshare;
};

// bundle b: units u0 .. u0 + nsub - 1 of one sweep, ustride windows apart, rlen steps; the sweep's last unit may be
// shorter than the others: member short_s then has only short_len windows (short_s = -1: none)
__host__ __device__ inline void few_bundle(const FewPlan &F, int b, OlbmComb &c, int &nsub, int &ustride, int &short_s, int &short_len)
{
    const LanePlan &P = F.L;
    const int nregb = P.nsb * F.B;
    const bool reg = b < nregb;
    const int sb = reg ? b / F.B : 0, ib = reg ? b - sb * F.B : b - nregb;
    const int U = reg ? P.U : P.Ur, ru = reg ? P.ru : P.rur;
    const long R = reg ? P.len : P.rem;
    const int u0 = ib * F.SUB;
    nsub = U - u0 < F.SUB ? U - u0 : F.SUB;
    const int last_len = (int)(R - (long)(U - 1) * ru);       // of the sweep's last unit (<= ru)
    const bool has_last = u0 + nsub == U;
    c.rlen = has_last && nsub == 1 ? last_len : ru;
    short_s = has_last && last_len < c.rlen ? nsub - 1 : -1;
    short_len = last_len;
    ustride = ru;
    if (reg) {
        c.base = (long)sb * 4 * P.len + (long)u0 * ru;
        c.m = sb == P.nsb - 1 ? P.m_last : 4;
        c.piece0 = sb * 4 * P.U + u0;
        c.pstride = P.U;
    } else {
        c.base = P.Q * P.len + (long)u0 * ru;
        c.m = 1;
        c.piece0 = P.npreg + u0;
        c.pstride = P.Ur;
    }
}

FewPlan make_few_plan(long nwin, long len, long groups, int sub)
{
    FewPlan F{};
    F.SUB = sub;
    F.L = make_lane_plan(nwin, len, groups, sub);
    const LanePlan &P = F.L;
    F.B = P.nsb ? (P.U + sub - 1) / sub : 1;                  // (1: divisor of the decode)
    F.Br = (P.Ur + sub - 1) / sub;
    F.nbund = (P.nsb ? P.nsb * F.B : 0) + F.Br;
    long shares = F.nbund / 3;                                // at least three bundles a share: a share more or less is then 33 % at worst
    if (shares < 1) shares = 1;
    F.nshare = (int)(shares < groups ? shares : groups);
    return F;
}

// the plan as arrays, share by share (for the host-side check only)
void few_plan_arrays(const FewPlan &F, CombPlan &pl)
{
    const LanePlan &P = F.L;
    pl.npieces = P.np;
    pl.pcnt.assign((size_t)P.np, 0);
    pl.ptile.assign((size_t)P.np, 0);
    for (int j = 0; j < P.np; j++) {
        int cnt;
        bool tile;
        lane_piece(P, j, cnt, tile);
        pl.pcnt[(size_t)j] = cnt;
        pl.ptile[(size_t)j] = tile;
    }
    pl.combs.clear();
    pl.first.assign(1, 0);
    pl.steps.clear();
    pl.max_steps = 0;
    for (int g = 0; g < F.nshare; g++) {
        long st = 0;
        for (int b = g; b < F.nbund; b += F.nshare) {
            OlbmComb c;
            int nsub, ustride, short_s, short_len;
            few_bundle(F, b, c, nsub, ustride, short_s, short_len);
            for (int q = 0; q < nsub; q++) {
                OlbmComb d = c;
                d.base += (long)q * ustride;
                d.piece0 += q;
                if (q == short_s) d.rlen = short_len;
                pl.combs.push_back(d);
                st += (long)d.m * ((d.rlen + 15) / 16);
            }
        }
        pl.first.push_back((int)pl.combs.size());
        pl.steps.push_back((int)st);
        if (st > pl.max_steps) pl.max_steps = st;
    }
}

template <int CG>
__global__ __launch_bounds__(64) void olbm_few_kernel(const double *__restrict__ x, long n, int p, long len, const FewPlan F,
                                                      double *__restrict__ partial, double *__restrict__ pd,
                                                      double *__restrict__ pe)
{
    constexpr int T = 16, TP = T + 1;
    __shared__ double W[4 * 16 * TP];                               // [tooth][slot][TP]
    const int lane = threadIdx.x, g = blockIdx.x;
    const int j = lane >> 4, v = lane & 15, su = v / CG, col = v % CG;   // the walk: tooth, slot = (unit of the bundle, column)
    const int li = lane & 15, lseg = lane >> 4;                     // the loads: step inside the chunk, slot run
    om_double4 acc = om_double4{0.0, 0.0, 0.0, 0.0};
    double D = 0.0, dsum = 0.0;

    struct Cur { int b, i0, m, rlen, nsub, piece0, pstride, ustride, short_s, short_len; long base; };
    auto load_bundle = [&](Cur &u) {
        u.i0 = 0;
        if (u.b < F.nbund) {
            OlbmComb c;
            few_bundle(F, u.b, c, u.nsub, u.ustride, u.short_s, u.short_len);   // uniform arithmetic
            u.m = c.m;
            u.rlen = c.rlen;
            u.piece0 = c.piece0;
            u.pstride = c.pstride;
            u.base = c.base;
        } else {
            u.m = 0;                                                // past the end: loads land somewhere valid, nothing is kept
            u.rlen = 0;
            u.nsub = 0;
            u.short_s = -1;
        }
    };
    auto advance = [&](Cur &u) {
        u.i0 += T;
        if (u.i0 >= u.rlen) {
            u.b += F.nshare;
            load_bundle(u);
        }
    };
    int sk[4], ck[4];
    size_t colpart[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int vk = 4 * k + lseg;
        sk[k] = vk / CG;
        ck[k] = vk % CG;
        colpart[k] = (size_t)(ck[k] < p ? ck[k] : p - 1) * (size_t)n;
    }
    double st[5][4];                                                // the chunk in flight: rows of tooth 0 .. 4
    auto issue = [&](const Cur &u) {
        const long tb = u.base + u.i0 + li;
#pragma unroll
        for (int jj = 0; jj < 5; jj++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                long t = tb + jj * len + (long)sk[k] * u.ustride;
                t = t < n ? t : n - 1;
                st[jj][k] = x[colpart[k] + t];
            }
    };
    double *Wc = W + lseg * TP + li;
    auto commit = [&](const Cur &u) {
        const long tb = u.base + u.i0 + li;
#pragma unroll
        for (int jt = 0; jt < 4; jt++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool ok = jt < u.m && sk[k] < u.nsub && ck[k] < p && tb + (jt + 1) * len + (long)sk[k] * u.ustride < n &&
                                (sk[k] != u.short_s || u.i0 + li < u.short_len);      // past the end of a shorter last unit: nothing
                Wc[(jt * 16 + 4 * k) * TP] = ok ? st[jt + 1][k] - st[jt][k] : 0.0;    // (olbm.c:60: the last window adds nothing)
            }
    };
    auto sync = [&]() {                                             // one wave: LDS operations complete in order
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    Cur cur;
    cur.b = g;
    cur.base = 0;
    cur.m = cur.rlen = cur.nsub = cur.piece0 = cur.pstride = cur.ustride = cur.short_len = 0;
    cur.short_s = -1;
    load_bundle(cur);
    issue(cur);
    const double *Wr = W + (j * 16 + v) * TP;
    while (cur.b < F.nbund) {
        commit(cur);
        sync();
        Cur nxt = cur;
        advance(nxt);
        issue(nxt);
        int nst = cur.rlen - cur.i0;
        nst = nst < T ? nst : T;
        int i = 0;
        if (cur.short_s >= 0 && cur.i0 + nst > cur.short_len) {
            // the chunk in which a shorter last unit ends (and those after it): its lanes take part with a zero operand from
            // there on -- their running sum stays what it was at the end (the piece's e), their differences are zero
            const int end = su == cur.short_s ? cur.short_len - cur.i0 : nst;
            for (; i < nst; i++) {
                const double d = Wr[i];
                const double Dl = i < end ? D : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Dl, Dl, acc, 0, 0, 0);
                dsum += Dl;
                D += d;
            }
        }
        for (; i + 4 <= nst; i += 4) {
            double d[4];
#pragma unroll
            for (int q = 0; q < 4; q++) d[q] = Wr[i + q];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D, D, acc, 0, 0, 0);
                dsum += D;
                D += d[q];
            }
        }
        for (; i < nst; i++) {
            const double d = Wr[i];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D, D, acc, 0, 0, 0);
            dsum += D;
            D += d;
        }
        if (cur.i0 + T >= cur.rlen) {                               // the bundle is done: its pieces' d and e
            if (j < cur.m && su < cur.nsub && col < p) {
                const size_t at = (size_t)(cur.piece0 + su + j * cur.pstride) * CG + col;
                pd[at] = dsum;
                pe[at] = D;
            }
            dsum = 0.0;
            D = 0.0;
        }
        sync();
        cur = nxt;
    }
    double *out = partial + (size_t)g * 256;
#pragma unroll
    for (int r = 0; r < 4; r++) out[r * 64 + lane] = acc[r];
}

// one split of the shares' accumulators (all 256 entries) and, by the first CG x CG threads, one split of the pieces'
// rank-two terms for the ordered pair of columns
template <int CG>
__global__ __launch_bounds__(256) void olbm_few_reduce_kernel(const double *__restrict__ partial, int nshare, int share_split,
                                                              const double *__restrict__ pe, const double *__restrict__ pd,
                                                              const int *__restrict__ pcnt, int npieces, int per_split,
                                                              const double *__restrict__ shift, double *__restrict__ red,
                                                              double *__restrict__ redc)
{
    const int e = threadIdx.x, b = blockIdx.x;
    double s = 0.0;
    const int g0 = b * share_split, g1 = g0 + share_split < nshare ? g0 + share_split : nshare;
#pragma unroll 8
    for (int g = g0; g < g1; g++) s += partial[(size_t)g * 256 + e];
    red[(size_t)b * 256 + e] = s;
    if (e < CG * CG) {
        const int a = e / CG, c = e % CG, ia = a < c ? a : c, ib = a < c ? c : a;
        const double sha = shift[ia], shb = shift[ib];
        const int j0 = b * per_split, j1 = j0 + per_split < npieces ? j0 + per_split : npieces;
        double cs = 0.0;
#pragma unroll 8
        for (int jp = j0; jp < j1; jp++) {
            const double sa = sha + pe[(size_t)jp * CG + ia], sb = shb + pe[(size_t)jp * CG + ib];
            const double da = pd[(size_t)jp * CG + ia], db = pd[(size_t)jp * CG + ib];
            const double tt = fma((double)pcnt[jp], sa, da);
            cs = fma(tt, sb, cs);
            cs = fma(sa, db, cs);
        }
        redc[(size_t)b * CG * CG + e] = cs;
    }
}

template <int CG>
__global__ __launch_bounds__(64) void olbm_few_finish_kernel(const double *__restrict__ red, const double *__restrict__ redc,
                                                             int nsplit, int p, double denom, double *__restrict__ var)
{
    const int e = threadIdx.x;
    if (e >= CG * CG) return;
    const int a = e / CG, c = e % CG;
    double s = 0.0;
    for (int q = 0; q < 16 / CG; q++) {                                // the units' diagonal blocks of the accumulator
        const int row = q * CG + a, colm = q * CG + c, idx = (row >> 2) * 64 + (row & 3) * 16 + colm;
#pragma unroll 8
        for (int b = 0; b < nsplit; b++) s += red[(size_t)b * 256 + idx];
    }
    double cs = 0.0;
#pragma unroll 8
    for (int b = 0; b < nsplit; b++) cs += redc[(size_t)b * CG * CG + e];
    if (a < p && c < p) var[a + (size_t)p * c] = (s + cs) / denom;     // olbm.c:74-75
}

struct LaneState {
    DevBuf all;
    double *part = nullptr, *pd = nullptr, *pe = nullptr, *small = nullptr, *red = nullptr, *redc = nullptr;
    int nshare = 0, np = 0, nchunk = 0, nsplit = 0, per_split = 0, share_split = 0;
    int *dpcnt = nullptr;
};

template <int NB, int TT>
int finish_lane(const double *x, long n, int p, long len, const double *dcentre, double *dmean_out, double *ddelta, double *dvar,
                DeviceCtx *dc, LaneState &K)
{
    using S = LaneShape<NB, TT>;
    hipStream_t s = dc->stream;
    const long nwin = n - len + 1;
    double *csum = K.small, *tsum = csum + (size_t)K.nchunk * S::PP, *s0 = tsum + (size_t)K.nchunk * S::PP, *tail = s0 + S::PP,
           *shift = tail + S::PP;
    if (int rc = olbm_first_window_sums(x, n, p, len, dcentre, s0, tail, s)) return rc;
    if (int rc = olbm_chain_means(tsum, K.nchunk, S::PP, p, s0, tail, dcentre, n, len, dmean_out ? 1 : 0, dmean_out, ddelta, shift, s))
        return rc;
    hipLaunchKernelGGL((olbm_lane_reduce_kernel<NB, TT>), dim3((unsigned)S::NTT, (unsigned)K.nsplit), dim3(256), 0, s, K.part, K.nshare,
                       K.share_split, K.pe, K.pd, K.dpcnt, K.np, K.per_split, shift, K.red);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL((olbm_lane_finish_kernel<NB, TT>), dim3((unsigned)S::NTT), dim3(256), 0, s, K.red, K.nsplit, p,
                       (double)nwin * (double)n * (double)len, dvar);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

template <int NB, int TT>
int launch_lane(const double *x, long n, int p, long len, const double *dcentre, double *dmean_out, double *ddelta, double *dvar,
                DeviceCtx *dc, LaneState &K)
{
    using S = LaneShape<NB, TT>;
    hipStream_t s = dc->stream;
    const long nwin = n - len + 1;
    auto kern = olbm_lane_kernel<NB, TT>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::lds_bytes));
    int bpc = 1;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, reinterpret_cast<const void *>(kern), 64 * NB, S::lds_bytes));
    if (bpc < 1) bpc = 1;
    if (const char *e = getenv("MCMCHIP_OLBM_LANE_GROUPS")) {       // workgroups per CU (experiments)
        const long v = atol(e);
        if (v >= 1 && v <= 64) bpc = (int)v;
    }
    static const bool dbg = getenv("MCMCHIP_OLBM_LANE_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tq[8] = {now()};
    const LanePlan P = make_lane_plan(nwin, len, (long)bpc * dc->num_cu);
    tq[1] = now();
    const int nshare = P.nshare, np = P.np, nchunk = olbm_chain_chunks(np);
    int nsplit = (4 * dc->num_cu + S::NTT - 1) / S::NTT;
    if (nsplit > (np + 15) / 16) nsplit = (np + 15) / 16;
    if (nsplit < 1) nsplit = 1;
    const int per_split = (np + nsplit - 1) / nsplit, share_split = (nshare + nsplit - 1) / nsplit;

    const size_t sz_part = (size_t)nshare * S::NTT * 256 * 8, sz_piece = (size_t)np * S::PP * 8,
                 sz_small = ((size_t)2 * nchunk + 4) * S::PP * 8, sz_red = (size_t)nsplit * S::NTT * 256 * 8;
    const size_t o_pcnt = 0, o_ptile = o_pcnt + (size_t)np * 4, sz_plan = o_ptile + (size_t)np;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    HIP_TRY(K.all.alloc(up(sz_part) + 2 * up(sz_piece) + up(sz_small) + up(sz_red) + up(sz_plan)));
    tq[2] = now();
    char *cursor = K.all.as<char>();
    auto carve = [&](size_t bytes) { char *r = cursor; cursor += up(bytes); return r; };
    K.part = (double *)carve(sz_part);
    K.pd = (double *)carve(sz_piece);
    K.pe = (double *)carve(sz_piece);
    K.small = (double *)carve(sz_small);
    K.red = (double *)carve(sz_red);
    char *pb = carve(sz_plan);
    hipLaunchKernelGGL(lane_fill_pieces_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, P, (int *)(pb + o_pcnt),
                       (unsigned char *)(pb + o_ptile));
    HIP_TRY(hipGetLastError());
    tq[3] = now();
    K.nshare = nshare;
    K.np = np;
    K.nchunk = nchunk;
    K.nsplit = nsplit;
    K.per_split = per_split;
    K.share_split = share_split;
    K.dpcnt = (int *)(pb + o_pcnt);

    timing_reset("olbm_lane_kernel");
    timing_begin(s);
    hipLaunchKernelGGL(kern, dim3((unsigned)nshare), dim3(64 * NB), S::lds_bytes, s, x, n, p, len, P, K.part, K.pd, K.pe);
    timing_end(s);
    HIP_TRY(hipGetLastError());
    if (dbg) { (void)hipStreamSynchronize(s); tq[4] = now(); }
    double *csum = K.small, *tsum = csum + (size_t)nchunk * S::PP;
    if (int rc = olbm_chain_pieces(K.pe, (const unsigned char *)(pb + o_ptile), np, S::PP, csum, tsum, s)) return rc;
    const int rcf = finish_lane<NB, TT>(x, n, p, len, dcentre, dmean_out, ddelta, dvar, dc, K);
    if (dbg) {
        (void)hipStreamSynchronize(s);
        tq[5] = now();
        fprintf(stderr, "olbm_lane: shares %d units %d pieces %d | plan %.3f alloc %.3f fill %.3f kernel %.3f post %.3f ms\n", nshare,
                P.nunit, np, tq[1] - tq[0], tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4]);
    }
    return rcf;
}

template <int CG>
int finish_few(const double *x, long n, int p, long len, const double *dcentre, double *dmean_out, double *ddelta, double *dvar,
               DeviceCtx *dc, LaneState &K)
{
    hipStream_t s = dc->stream;
    const long nwin = n - len + 1;
    double *csum = K.small, *tsum = csum + (size_t)K.nchunk * CG, *s0 = tsum + (size_t)K.nchunk * CG, *tail = s0 + CG, *shift = tail + CG;
    if (int rc = olbm_first_window_sums(x, n, p, len, dcentre, s0, tail, s)) return rc;
    if (int rc = olbm_chain_means(tsum, K.nchunk, CG, p, s0, tail, dcentre, n, len, dmean_out ? 1 : 0, dmean_out, ddelta, shift, s))
        return rc;
    hipLaunchKernelGGL(olbm_few_reduce_kernel<CG>, dim3((unsigned)K.nsplit), dim3(256), 0, s, K.part, K.nshare, K.share_split, K.pe,
                       K.pd, K.dpcnt, K.np, K.per_split, shift, K.red, K.redc);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(olbm_few_finish_kernel<CG>, dim3(1), dim3(64), 0, s, K.red, K.redc, K.nsplit, p,
                       (double)nwin * (double)n * (double)len, dvar);
    HIP_TRY(hipGetLastError());
    return MCMCHIP_OK;
}

template <int CG>
int launch_few(const double *x, long n, int p, long len, const double *dcentre, double *dmean_out, double *ddelta, double *dvar,
               DeviceCtx *dc, LaneState &K)
{
    hipStream_t s = dc->stream;
    const long nwin = n - len + 1;
    auto kern = olbm_few_kernel<CG>;
    int bpc = 1;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, reinterpret_cast<const void *>(kern), 64, 0));
    if (bpc < 1) bpc = 1;
    if (bpc > 8) bpc = 8;                                           // two waves per SIMD keep HBM busy; more shares are more pieces to chain
    if (const char *e = getenv("MCMCHIP_OLBM_LANE_GROUPS")) {       // workgroups per CU (experiments)
        const long v = atol(e);
        if (v >= 1 && v <= 64) bpc = (int)v;
    }
    const FewPlan F = make_few_plan(nwin, len, (long)bpc * dc->num_cu, 16 / CG);
    const int nshare = F.nshare, np = F.L.np, nchunk = olbm_chain_chunks(np);
    int nsplit = 2 * dc->num_cu;
    if (nsplit > (np + 15) / 16) nsplit = (np + 15) / 16;
    if (nsplit < 1) nsplit = 1;
    const int per_split = (np + nsplit - 1) / nsplit, share_split = (nshare + nsplit - 1) / nsplit;
    const size_t sz_part = (size_t)nshare * 256 * 8, sz_piece = (size_t)np * CG * 8, sz_small = ((size_t)2 * nchunk + 4) * CG * 8,
                 sz_red = (size_t)nsplit * 256 * 8, sz_redc = (size_t)nsplit * CG * CG * 8;
    const size_t o_pcnt = 0, o_ptile = o_pcnt + (size_t)np * 4, sz_plan = o_ptile + (size_t)np;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    HIP_TRY(K.all.alloc(up(sz_part) + 2 * up(sz_piece) + up(sz_small) + up(sz_red) + up(sz_redc) + up(sz_plan)));
    char *cursor = K.all.as<char>();
    auto carve = [&](size_t bytes) { char *r = cursor; cursor += up(bytes); return r; };
    K.part = (double *)carve(sz_part);
    K.pd = (double *)carve(sz_piece);
    K.pe = (double *)carve(sz_piece);
    K.small = (double *)carve(sz_small);
    K.red = (double *)carve(sz_red);
    K.redc = (double *)carve(sz_redc);
    char *pb = carve(sz_plan);
    // columns >= p of a piece's record are never written by the walk; the chain kernels read all CG
    HIP_TRY(hipMemsetAsync(K.pd, 0, 2 * up(sz_piece), s));
    hipLaunchKernelGGL(lane_fill_pieces_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, F.L, (int *)(pb + o_pcnt),
                       (unsigned char *)(pb + o_ptile));
    HIP_TRY(hipGetLastError());
    K.nshare = nshare;
    K.np = np;
    K.nchunk = nchunk;
    K.nsplit = nsplit;
    K.per_split = per_split;
    K.share_split = share_split;
    K.dpcnt = (int *)(pb + o_pcnt);
    timing_reset("olbm_few_kernel");
    timing_begin(s);
    hipLaunchKernelGGL(kern, dim3((unsigned)nshare), dim3(64), 0, s, x, n, p, len, F, K.part, K.pd, K.pe);
    timing_end(s);
    HIP_TRY(hipGetLastError());
    double *csum = K.small, *tsum = csum + (size_t)nchunk * CG;
    if (int rc = olbm_chain_pieces(K.pe, (const unsigned char *)(pb + o_ptile), np, CG, csum, tsum, s)) return rc;
    return finish_few<CG>(x, n, p, len, dcentre, dmean_out, ddelta, dvar, dc, K);
}

}  // namespace

// Host-only check of the plan (no GPU call; tests/test_host_cpu.py), as mcmchip_olbm_comb_plan_check: the arrays are made
// by the same decode functions the kernels use
extern "C" int mcmchip_olbm_lane_plan_check(int64_t nwin, int64_t len, int64_t groups, int32_t *npieces, int64_t *min_steps,
                                            int64_t *max_steps)
{
    if (nwin < 1 || len < 1 || groups < 1) return -1;
    CombPlan pl;
    lane_plan_arrays(make_lane_plan((long)nwin, (long)len, (long)groups), pl);
    long lo = 0, hi = 0;
    const int bad = olbm_plan_violations(pl, (long)nwin, (long)len, 4, npieces, &lo, &hi);
    if (min_steps) *min_steps = lo;
    if (max_steps) *max_steps = hi;
    return bad;
}

extern "C" int mcmchip_olbm_few_plan_check(int64_t nwin, int64_t len, int64_t groups, int32_t sub, int32_t *npieces,
                                           int64_t *min_steps, int64_t *max_steps)
{
    if (nwin < 1 || len < 1 || groups < 1 || (sub != 2 && sub != 4)) return -1;
    CombPlan pl;
    few_plan_arrays(make_few_plan((long)nwin, (long)len, (long)groups, sub), pl);
    long lo = 0, hi = 0;
    const int bad = olbm_plan_violations(pl, (long)nwin, (long)len, 4, npieces, &lo, &hi);
    if (min_steps) *min_steps = lo;
    if (max_steps) *max_steps = hi;
    return bad;
}

bool olbm_lane_applies(long n, int p, long len)
{
    static const char *env = getenv("MCMCHIP_OLBM_LANE");
    if (env && env[0] == '0') return false;
    // Measured at n = 1e8, len = 1e4 (ms per call, this file / what served before): p = 3: 1.3 / 1.45 (olbm_small), p = 4: 1.3 / 2.3,
    // p = 5: 1.7 / 3.0, p = 8: 1.8 / 6.0 (olbm_few_kernel); p = 9: 2.1 / 14.4 (olbm_mfma_kernel), p = 16: 2.7 / 14.5, p = 32: 5.8 / 15.6,
    // p = 64: 15.4 / 22.4, p = 65: 24.2 / 44.7 (olbm_comb_kernel), p = 128: 41.4 / 46.2, p = 129: 64.7 / 111.5, p = 192: 80.3 / 112.0;
    // p <= 2 stays with olbm_small (p = 2: 1.34 / 1.18), and so do few columns of a shorter series (n p < 4e7: n = 1e7, len = 1e3:
    // p = 3: 0.52 / 0.39, p = 4: 0.55 / 0.55, p = 8: 0.45 / 0.88; n = 2e6: p = 4: 0.43 / 0.35, p = 8: 0.38 / 0.45).
    // MCMCHIP_OLBM_LANE=<min len> lowers the window threshold and takes every p <= 192 (tests/fuzz_olbm_lane.py).
    long min_len = 256;
    int min_p = 3;
    long min_np_few = 40000000;
    if (env && atol(env) > 1) { min_len = atol(env); min_p = 1; min_np_few = 0; }
    const long nwin = n - len + 1;
    if (p <= 8 && n * p < min_np_few) return false;
    static const int max_p = getenv("MCMCHIP_OLBM_LANE_MAXP") ? atoi(getenv("MCMCHIP_OLBM_LANE_MAXP")) : OL_LANE_MAX_P;
    return p >= min_p && p <= max_p && len >= min_len && nwin >= 4 * len;
}

int run_olbm_lane(const double *x, long n, int p, long len, double *mean, double *var, int nocalc, DeviceCtx *dc)
{
    hipStream_t s = dc->stream;
    DevBuf vecs;                                              // var | mean | centre | delta
    HIP_TRY(vecs.alloc(((size_t)p * p + 3 * (size_t)p) * 8));
    double *dvar = vecs.as<double>(), *dmean = dvar + (size_t)p * p, *dcentre = dmean + p, *ddelta = dcentre + p;
    const bool one_pass = !nocalc && n >= OL_ONEPASS_MIN_N;   // as in olbm_comb.hip
    if (nocalc) HIP_TRY(hipMemcpyAsync(dmean, mean, (size_t)p * 8, hipMemcpyHostToDevice, s));
    else if (one_pass) { if (int rc = olbm_pilot_centre(x, n, p, dcentre, s)) return rc; }
    else if (int rc = olbm_exact_means(x, n, p, dmean, s)) return rc;
    LaneState K;
    const double *centre = one_pass ? dcentre : dmean;
    double *mean_out = one_pass ? dmean : nullptr;
    const int nb = (p + 15) / 16;
    static const int tt = getenv("MCMCHIP_OLBM_LANE_T") ? atoi(getenv("MCMCHIP_OLBM_LANE_T")) : 16;
    static const bool few = !(getenv("MCMCHIP_OLBM_FEW") && getenv("MCMCHIP_OLBM_FEW")[0] == '0');
    auto walk = [&](bool first_time, const double *cen, double *mo, double *dl) {
        if (few && p <= 4)
            return first_time ? launch_few<4>(x, n, p, len, cen, mo, dl, dvar, dc, K) : finish_few<4>(x, n, p, len, cen, mo, dl, dvar, dc, K);
        if (few && p <= 8)
            return first_time ? launch_few<8>(x, n, p, len, cen, mo, dl, dvar, dc, K) : finish_few<8>(x, n, p, len, cen, mo, dl, dvar, dc, K);
#define OL_CASE_T(NBV, TV)                                                                                   \
    if (nb == NBV && (tt == TV || TV == 16))                                                                 \
        return first_time ? launch_lane<NBV, TV>(x, n, p, len, cen, mo, dl, dvar, dc, K)                     \
                          : finish_lane<NBV, TV>(x, n, p, len, cen, mo, dl, dvar, dc, K);
#define OL_CASE(NBV) OL_CASE_T(NBV, 64) OL_CASE_T(NBV, 32) OL_CASE_T(NBV, 16)
        OL_CASE(1) OL_CASE(2) OL_CASE(3) OL_CASE(4)
#define OL_CASE16(NBV) OL_CASE_T(NBV, 16)
        OL_CASE16(5) OL_CASE16(6) OL_CASE16(7) OL_CASE16(8) OL_CASE16(9) OL_CASE16(10) OL_CASE16(11) OL_CASE16(12)
#undef OL_CASE16
#undef OL_CASE_T
#undef OL_CASE
        return set_error(MCMCHIP_EINVAL, "olbm_lane: p > 192");
    };
    if (int rc = walk(true, centre, mean_out, ddelta)) return rc;
    HIP_TRY(hipMemcpyAsync(var, dvar, (size_t)p * p * 8, hipMemcpyDeviceToHost, s));
    if (!nocalc) HIP_TRY(hipMemcpyAsync(mean, dmean, (size_t)p * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (one_pass) {
        // the sampled centre enters only through s0 - len * delta; beyond three digits lost there the finish (not the walk)
        // is repeated about the exact mean
        std::vector<double> hd((size_t)p);
        HIP_TRY(hipMemcpy(hd.data(), ddelta, (size_t)p * 8, hipMemcpyDeviceToHost));
        bool redo = false;
        for (int c = 0; c < p; c++) {
            const double main_term = var[c + (size_t)p * c] * (double)n;
            const double corr = (double)len * hd[c] * hd[c];
            if (!(corr <= OL_ONEPASS_MAX_LOSS * main_term) && !(fabs(hd[c]) <= 1e-14 * fabs(mean[c]))) redo = true;
        }
        if (redo) {
            if (int rc = walk(false, dmean, nullptr, nullptr)) return rc;
            HIP_TRY(hipMemcpyAsync(var, dvar, (size_t)p * p * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
    }
    return MCMCHIP_OK;
}
