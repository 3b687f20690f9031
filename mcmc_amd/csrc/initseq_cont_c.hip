// initseq_cont_c.hip -- instantiations of autocov_kernel<NACC, false, true, CENTRE = true> (initseq_kernels.h: lag pairs, continuation items,
// the series taken about a provisional centre: mcmchip_initseq_uncentred), a translation unit of their own so the library
// builds in parallel
#include "initseq_kernels.h"

int launch_autocov_cont_c(AUTOCOV_CENTRE_ARGS)
{
    switch (nacc) {
#define AC_CASE(N) case N: return launch_autocov<N, false, true, true>(grid, s, x, len, pc, pl, pu, nblk, tiles_per_block, pp, centre);
    AC_CASE(2) AC_CASE(3) AC_CASE(4) AC_CASE(6) AC_CASE(8) AC_CASE(12) AC_CASE(16)
#undef AC_CASE
    default: return -1;
    }
}
