"""Wall time of small and mid-size calls through the numpy mirror (host arrays in, host arrays out), after one warm-up call
each: what a user of the R package's sizes pays per call.  usage (GPU box): python tools/latency_audit.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import abi, api  # noqa: E402

api.init()
rng = np.random.default_rng(0)


def timed(label, f, reps=5):
    f()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        best = min(best, time.perf_counter() - t0)
    kn, kms, nl = api.last_kernel_time()
    print("%-72s %9.3f ms   (%s %.3f ms x %d)" % (label, best * 1e3, kn, kms, nl), flush=True)


for n, kind in ((10_000, "ar1"), (100_000, "ar1"), (1_000_000, "ar1"), (1_000_000, "walk")):
    for p in (1, 3, 8):
        from scipy.signal import lfilter
        rho = [0.99, 0.9, 0.5, 0.95, 0.3, 0.98, 0.7, 0.0][:p]
        x = np.asfortranarray(np.stack([lfilter([1.0], [1.0, -r], rng.standard_normal(n)) for r in rho], axis=1))
        if kind == "walk":                           # a random walk under the noise: the stop rule fires thousands of lags out
            x = np.asfortranarray(rng.standard_normal((n, p)).cumsum(axis=0) * 0.01 + rng.standard_normal((n, p)))
        blen = max(1, int(np.sqrt(n)))
        timed("olbm n=%d p=%d len=%d, means supplied" % (n, p, blen), lambda: api.olbm(x, blen))
        timed("olbm n=%d p=%d len=%d, means computed" % (n, p, blen), lambda: api.olbm(x, blen, demean=False))
        timed("initseq n=%d (%s), %d column(s) one call each" % (n, kind, p), lambda: [api.initseq(x[:, c]) for c in range(p)])
# metrop: the logit example (tests/logit.R: 100 x 4) and a mid-size regression, one chain and 64 chains
for (nobs, d, nchain, nbatch) in ((100, 4, 1, 1000), (100, 4, 64, 1000), (10_000, 8, 64, 200), (100_000, 16, 256, 100)):
    X = rng.standard_normal((nobs, d))
    X[:, 0] = 1.0
    y = (rng.random(nobs) < 0.5).astype(float)
    tg = api.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.5)
    init = np.zeros((nchain, d))
    timed("metrop logistic %d x %d, %d chain(s) x %d iterations" % (nobs, d, nchain, nbatch),
          lambda: api.metrop(tg, init, nbatch, scale=0.3 / np.sqrt(d), seed=1), reps=3)
# temper: ONE ensemble (what the reference runs), the model-selection example of tests/temp-par.R and the witch's hat of
# tests/temp-par-witch.R, parallel and serial
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rhelpers import temp_par_setup  # noqa: E402
X, y, models, masks, betas, nb = temp_par_setup()
kw = dict(x=X, y=y, prior_c=0.5, par=abi.logistic_models_par(masks))
for parallel in (True, False):
    tg = api.Target(abi.T_LOGISTIC_MODELS, 4, **kw)
    init = betas if parallel else np.concatenate([[8.0], betas[7]])
    for nbatch in (1000, 10000):
        timed("temper %s, 8 logistic sub-models (temp-par.R), 1 ensemble x %d iterations" % ("parallel" if parallel else "serial", nbatch),
              lambda: api.temper(tg, init, nb, nbatch, 1, 1, 0.56789, parallel=parallel, seed=3), reps=3)
