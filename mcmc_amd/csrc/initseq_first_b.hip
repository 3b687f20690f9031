// initseq_first_b.hip -- instantiations of autocov_kernel<NACC, true> (initseq_kernels.h), a translation unit of their own so the
// library builds in parallel
#include "initseq_kernels.h"

int launch_autocov_first_b(int nacc, dim3 grid, hipStream_t s, const double *x, long len, const int *pc, const long *pl,
                      const int *pu, long nblk, int tiles_per_block, int img, double *pp)
{
    switch (nacc) {
#define AC_CASE(N) case N: return launch_autocov<N, true>(grid, s, x, len, pc, pl, pu, nblk, tiles_per_block, img, pp);
    AC_CASE(13) AC_CASE(14) AC_CASE(16) AC_CASE(18) AC_CASE(20) AC_CASE(22) AC_CASE(24) AC_CASE(28) AC_CASE(32) AC_CASE(36) AC_CASE(40) AC_CASE(44) AC_CASE(48)
#undef AC_CASE
    default: return -1;
    }
}
