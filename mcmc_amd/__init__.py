"""mcmc_amd -- MI355X (gfx950) implementation of the R package `mcmc` sampler hot path.

The compute lives in mcmc_amd/csrc (hand-written HIP, built into libmcmchip.so behind the
C-ABI of include/mcmchip.h); this package is the thin host-side mirror of the reference's
R front ends (R/metrop.R, R/temper.R, R/initseq.R, R/olbm.R).
"""
_API = ("metrop", "temper", "initseq", "olbm", "logdens", "Target", "logistic_target",
        "load_library", "library_path", "McmcHipError")


def __getattr__(name):
    if name in _API:
        from . import api
        return getattr(api, name)
    raise AttributeError(name)
