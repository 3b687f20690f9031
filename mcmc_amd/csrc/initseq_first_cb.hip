// initseq_first_cb.hip -- instantiations of autocov_kernel<NACC, true, true, CENTRE = true> (initseq_kernels.h: lag pairs, items that start at lag 0,
// the series taken about a provisional centre: mcmchip_initseq_uncentred), a translation unit of their own so the library
// builds in parallel
#include "initseq_kernels.h"

int launch_autocov_first_cb(AUTOCOV_CENTRE_ARGS)
{
    switch (nacc) {
#define AC_CASE(N) case N: return launch_autocov<N, true, true, true>(grid, s, x, len, pc, pl, pu, nblk, tiles_per_block, pp, centre);
    AC_CASE(7) AC_CASE(8) AC_CASE(9) AC_CASE(10) AC_CASE(11) AC_CASE(12) AC_CASE(14) AC_CASE(16) AC_CASE(18) AC_CASE(20) AC_CASE(22) AC_CASE(24)
#undef AC_CASE
    default: return -1;
    }
}
