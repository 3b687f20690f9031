"""mcmc_amd -- MI355X (gfx950) implementation of the R package `mcmc` sampler hot path.

The compute lives in mcmc_amd/csrc (hand-written HIP, built into libmcmchip.so behind the
C-ABI of include/mcmchip.h); this package is the thin host-side mirror of the reference's
R front ends (R/metrop.R, R/morph.metrop.R, R/morph.R, R/temper.R, R/initseq.R, R/olbm.R).
"""
_API = ("metrop", "morph_metrop", "temper", "initseq", "olbm", "logdens", "Target", "logistic_target",
        "load_library", "library_path", "McmcHipError", "tune_pseudo_prior", "logistic_models_target",
        "set_interrupt_poll", "set_launch_limit")
_MORPH = ("morph", "morph_identity", "Morph")


def __getattr__(name):
    if name in _API:
        from . import api
        return getattr(api, name)
    if name in _MORPH:
        from . import _morph as _m
        return getattr(_m, name)
    raise AttributeError(name)
