"""Where does a C2 iteration go?  Kernel time of 4096 chains x 1000 iterations on the logit data with device Philox
and with injected streams (run it again with MCMCHIP_NO_SOFTPLUS_TABLE=1 for the library exp/log1p).
usage (GPU box): python tools/c2_split.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mcmc_amd import api
from rhelpers import load_logit, glm_binomial
api.init()
X, y = load_logit()
mle = glm_binomial(X, y)
rng = np.random.default_rng(0)
C, nbatch, blen, nspac = 4096, 100, 5, 2
niter = nbatch * blen * nspac
init = mle + 0.05 * rng.standard_normal((C, 5))
z, u = rng.standard_normal((C, niter, 5)), rng.random((C, niter))
tg = api.logistic_target(X, y)
for name, kw in (("philox", dict(seed=1)), ("injected", dict(z=z, u=u))):
    api.metrop(tg, init, 2, 1, 1, scale=0.4, **kw if name == "philox" else dict(z=z[:, :2], u=u[:, :2]))
    o = api.metrop(tg, init, nbatch, blen, nspac, scale=0.4, **kw)
    print(name, "kernel ms", api.last_kernel_time()[1], "per iteration us", api.last_kernel_time()[1] * 1e3 / niter, flush=True)
