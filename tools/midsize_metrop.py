"""A mid-size batched run on the lockstep matrix-core path (two launches per iteration): wall time per iteration against the
kernels' own time.  usage (GPU box): python tools/midsize_metrop.py [nobs d nchain niter]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import abi, api  # noqa: E402

nobs, d, nchain, niter = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (100_000, 16, 256, 200)
api.init()
rng = np.random.default_rng(0)
X = rng.standard_normal((nobs, d))
X[:, 0] = 1.0
y = (rng.random(nobs) < 0.5).astype(float)
tg = api.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.5)
init = np.zeros((nchain, d))
api.metrop(tg, init, 5, scale=0.3 / np.sqrt(d), seed=1)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    o = api.metrop(tg, init, niter, scale=0.3 / np.sqrt(d), seed=1)
    best = min(best, time.perf_counter() - t0)
kn, kms, nl = api.last_kernel_time()
print("metrop logistic %d x %d, %d chains x %d iterations: %.3f ms = %.1f us per iteration; %s %.1f us x %d; accept %.3f" % (
    nobs, d, nchain, niter, best * 1e3, best / niter * 1e6, kn, kms * 1e3, nl, float(np.mean(o["accept"]))))
