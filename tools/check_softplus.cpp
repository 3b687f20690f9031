// host build of mcmc_amd/csrc/softplus.h for accuracy checks (no GPU needed)
#include "../mcmc_amd/csrc/softplus.h"
static SoftplusTables T; static int init = 0;
extern "C" void softplus_host(const double *a, double *out, long n)
{
    if (!init) { softplus_tables_init(&T); init = 1; }
    for (long i = 0; i < n; i++) out[i] = softplus_half_abs(a[i], T.e);
}
// M(eta) = |eta|/2 + log1p(exp(-|eta|)) in long double
extern "C" void softplus_ref(const double *a, double *out, double *unused, long n)
{
    for (long i = 0; i < n; i++) {
        const long double x = fabsl((long double)a[i]);
        out[i] = (double)(x / 2 + log1pl(expl(-x)));
    }
}
