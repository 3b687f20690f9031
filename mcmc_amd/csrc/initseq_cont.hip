// initseq_cont.hip -- instantiations of autocov_kernel<NACC, false> (initseq_kernels.h), a translation unit of their own so the
// library builds in parallel
#include "initseq_kernels.h"

int launch_autocov_cont(int nacc, dim3 grid, hipStream_t s, const double *x, long len, const int *pc, const long *pl,
                      const int *pu, long nblk, int tiles_per_block, int img, double *pp)
{
    switch (nacc) {
#define AC_CASE(N) case N: return launch_autocov<N, false>(grid, s, x, len, pc, pl, pu, nblk, tiles_per_block, img, pp);
    AC_CASE(1) AC_CASE(2) AC_CASE(3) AC_CASE(4) AC_CASE(6) AC_CASE(8) AC_CASE(12) AC_CASE(16) AC_CASE(24) AC_CASE(32)
#undef AC_CASE
    default: return -1;
    }
}

long autocov_cont_image(int nacc, long lag0)
{
    const long NA = IS_TILE + 32, NB_ = IS_TILE + 16 * nacc + 32;
    const long SA = (NA + 255) / 256, SB = (NB_ + 255) / 256;
    const long merged = lag0 + 3 + NB_;
    return merged <= (SA + SB) * 256 ? (merged + 255) / 256 * 256 : (SA + SB) * 256;
}
