// initseq_first_ca.hip -- instantiations of autocov_kernel<NACC, true, true, CENTRE = true> (initseq_kernels.h: lag pairs, items that start at lag 0,
// the series taken about a provisional centre: mcmchip_initseq_uncentred), a translation unit of their own so the library
// builds in parallel
#include "initseq_kernels.h"

int launch_autocov_first_ca(AUTOCOV_CENTRE_ARGS)
{
    switch (nacc) {
#define AC_CASE(N) case N: return launch_autocov<N, true, true, true>(grid, s, x, len, pc, pl, pu, nblk, tiles_per_block, pp, centre);
    AC_CASE(1) AC_CASE(2) AC_CASE(3) AC_CASE(4) AC_CASE(5) AC_CASE(6)
#undef AC_CASE
    default: return -1;
    }
}
