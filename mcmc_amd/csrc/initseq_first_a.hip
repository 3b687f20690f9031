// initseq_first_a.hip -- instantiations of autocov_kernel<NACC, true> (initseq_kernels.h), a translation unit of their own so the
// library builds in parallel
#include "initseq_kernels.h"

int launch_autocov_first_a(int nacc, dim3 grid, hipStream_t s, const double *x, long len, const int *pc, const long *pl,
                      const int *pu, long nblk, int tiles_per_block, int img, double *pp)
{
    switch (nacc) {
#define AC_CASE(N) case N: return launch_autocov<N, true>(grid, s, x, len, pc, pl, pu, nblk, tiles_per_block, img, pp);
    AC_CASE(1) AC_CASE(2) AC_CASE(3) AC_CASE(4) AC_CASE(5) AC_CASE(6) AC_CASE(7) AC_CASE(8) AC_CASE(9) AC_CASE(10) AC_CASE(11) AC_CASE(12)
#undef AC_CASE
    default: return -1;
    }
}
