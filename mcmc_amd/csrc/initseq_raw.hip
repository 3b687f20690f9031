// initseq_raw.hip -- instantiations of autocov_kernel<NACC, false, false> (initseq_kernels.h: single lags of one parity, for mcmchip_lagsums), a translation unit
// of their own so the library builds in parallel
#include "initseq_kernels.h"

int launch_autocov_raw(AUTOCOV_LAUNCH_ARGS)
{
    switch (nacc) {
#define AC_CASE(N) case N: return launch_autocov<N, false, false>(grid, s, x, len, pc, pl, pu, nblk, tiles_per_block, pp);
    AC_CASE(1) AC_CASE(2) AC_CASE(4) AC_CASE(8) AC_CASE(16) AC_CASE(24)
#undef AC_CASE
    default: return -1;
    }
}
