"""Does the 130 KB softplus table (one workgroup per CU) still pay when there are many chains?  Kernel time of the
stream path on the logit data for growing chain counts; run again with MCMCHIP_NO_SOFTPLUS_TABLE=1 to compare.
usage (GPU box): python tools/stream_occupancy.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mcmc_amd import abi, api
from rhelpers import load_logit, glm_binomial
api.init()
X, y = load_logit()
mle = glm_binomial(X, y)
rng = np.random.default_rng(0)
for n_rep, C in ((1, 4096), (1, 16384), (1, 65536), (1, 262144), (40, 4096), (40, 65536)):
    Xr, yr = np.tile(X, (n_rep, 1)), np.tile(y, n_rep)
    tg = api.logistic_target(Xr, yr)
    init = mle + 0.05 * rng.standard_normal((C, 5))
    api.metrop(tg, init, 2, 1, 1, scale=0.4 / np.sqrt(n_rep), seed=1, path=abi.PATH_STREAM)
    nb = 200 if n_rep == 1 else 20
    api.metrop(tg, init, nb, 1, 1, scale=0.4 / np.sqrt(n_rep), seed=1, path=abi.PATH_STREAM)
    ms = api.last_kernel_time()[1]
    print("n=%d chains=%d iterations=%d kernel %.2f ms  %.3g chain-it/s" % (100 * n_rep, C, nb, ms, C * nb / ms * 1e3), flush=True)
