// host build of mcmc_amd/csrc/softplus.h for accuracy checks (no GPU needed)
#include "../mcmc_amd/csrc/softplus.h"
static SoftplusTables T; static int init = 0;
extern "C" void softplus_host(const double *a, double *out, long n)
{
    if (!init) { softplus_tables_init(&T); init = 1; }
    for (long i = 0; i < n; i++) out[i] = softplus_neg(sp_clamp_abs(a[i]), T.e);
}
extern "C" void softplus_ref(const double *a, double *out, double *unused, long n)
{
    for (long i = 0; i < n; i++) out[i] = (double)log1pl(expl(-(long double)a[i]));
}
