// olbm_shape.h -- tile / LDS geometry shared by the two matrix-core olbm kernels (diag.hip, olbm_comb.hip)
#pragma once
#include <vector>
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

// host entry points shared between diag.hip and olbm_comb.hip
int olbm_pilot_centre(const double *x, long n, int p, double *dcentre, hipStream_t s);   // sampled provisional centre
int olbm_exact_means(const double *x, long n, int p, double *dmean, hipStream_t s);      // olbm.c:38-43 as a pass of its own
bool olbm_comb_applies(long n, int p, long len);
struct DeviceCtx;
int run_olbm_comb(const double *x, long n, int p, long len, double *mean, double *var, int nocalc, DeviceCtx *dc);
// olbm_comb.hip: s0[c] = sum over the first window of (x - centre), tail[c] = the same over the rows after the last window
// that starts at a multiple of len
int olbm_first_window_sums(const double *x, long n, int p, long len, const double *dcentre, double *s0, double *tail, hipStream_t s);
// olbm_comb.hip: the plan of a comb walk (shared with olbm_lane.hip)
struct OlbmComb {
    long base;          // first window of tooth 0
    int m;              // teeth, len windows apart
    int rlen;           // windows per tooth in this comb (the sweep)
    int piece0;         // piece of tooth k: piece0 + k * pstride (pieces are numbered in time order)
    int pstride;
};
struct CombPlan {
    std::vector<OlbmComb> combs;
    std::vector<int> first, steps, pcnt;
    std::vector<unsigned char> ptile;
    int npieces = 0;
    long max_steps = 0;
};
void plan_combs(long nwin, long len, int teeth, long groups, CombPlan &pl);
// host-only check of a plan (every window once, pieces in time order and contiguous, step counts consistent): violations
int olbm_plan_violations(const CombPlan &pl, long nwin, long len, int teeth, int *npieces, long *min_steps, long *max_steps);
// chain of the pieces: e_j -> E_j in place, sums of the E_j of the pieces that start a tooth per chunk of pieces (tsum);
// then the exact means (want_mean) and shift = s0 - len * delta
int olbm_chain_chunks(int npieces);
int olbm_chain_pieces(double *pe, const unsigned char *ptile, int npieces, int PP, double *csum, double *tsum, hipStream_t s);
int olbm_chain_means(const double *tsum, int nchunk, int PP, int p, const double *s0, const double *tail, const double *centre,
                     long n, long len, int want_mean, double *mean, double *delta, double *shift, hipStream_t s);
// olbm_lane.hip: 3 <= p <= 192 columns, long windows: a wave walks four teeth, one lane per (tooth, column)
bool olbm_lane_applies(long n, int p, long len);
int run_olbm_lane(const double *x, long n, int p, long len, double *mean, double *var, int nocalc, DeviceCtx *dc);
// olbm_small.hip: p <= 8 on the vector units
bool olbm_small_applies(long n, int p, long len);
int run_olbm_small(const double *x, long n, int p, long len, double *mean, double *var, int nocalc, DeviceCtx *dc);
