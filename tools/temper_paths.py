#!/usr/bin/env python3
"""Tempering over the model-selection logistic posterior (8 sub-models of 4 coefficients, the shape of tests/temp-par.R)
on growing data sets: one wavefront per ensemble (temper_kernel) against lockstep ensembles with the log densities on the
matrix cores (temper_mfma.hip).  usage: python tools/temper_paths.py [nens] [iterations]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (before libmcmchip.so)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import abi, api  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
niter = int(sys.argv[2]) if len(sys.argv) > 2 else 200
api.init()
rng = np.random.default_rng(0)
p, k = 4, 8
masks = np.array([[1, a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], dtype=float)
nb = np.zeros((k, k), dtype=bool)
for i in range(k):
    for j in range(k):
        nb[i, j] = np.sum(masks[i] != masks[j]) == 1
for n in (4096, 16384, 32768, 65536, 262144, 1048576):
    X = rng.standard_normal((n, p)) / 2
    X[:, 0] = 1.0
    beta = np.array([0.3, 0.5, -0.4, 0.2])
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    tg = api.Target(abi.T_LOGISTIC_MODELS, p, x=X, y=y, prior_c=0.125, par=abi.logistic_models_par(masks))
    for parallel in (True, False):
        init = np.tile(beta, (E, k, 1)) if parallel else np.column_stack([np.ones(E), np.tile(beta, (E, 1))])
        row = []
        for path, name in ((abi.PATH_LDS, "lds"), (abi.PATH_TMFMA, "mfma")):
            it = niter if (path == abi.PATH_TMFMA or n <= 65536) else max(niter // 10, 10)
            api.temper(tg, init, nb, 2, 1, 1, 2.0 / np.sqrt(n), parallel=parallel, seed=1, _path=path)       # warm-up
            t0 = time.perf_counter()
            o = api.temper(tg, init, nb, it, 1, 1, 2.0 / np.sqrt(n), parallel=parallel, seed=1, _path=path)
            el = time.perf_counter() - t0
            row.append("%s %8.1f us/iteration (acceptx %.2f)" % (name, el / it * 1e6, np.nanmean(o["acceptx"])))
        print("n=%8d  %-8s %d ensembles:  %s" % (n, "parallel" if parallel else "serial", E, "   ".join(row)), flush=True)
