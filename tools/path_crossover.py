"""Where does the fp64-MFMA path overtake the stream path?  Kernel-side chain-iterations/s of both on synthetic logistic
posteriors (used to set the AUTO rule in capi.hip: n >= 4096 and >= 64 chains -> MFMA).
usage (GPU box): python tools/path_crossover.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import abi, api
api.init()
rng = np.random.default_rng(0)
for n, d, C in ((1024, 5, 4096), (4096, 5, 4096), (4096, 5, 65536), (4096, 20, 4096), (16384, 20, 4096), (16384, 20, 65536), (2048, 64, 4096)):
    X = np.column_stack([np.ones(n), rng.standard_normal((n, d - 1)) / np.sqrt(d)])
    beta = 0.5 * rng.standard_normal(d)
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    tg = api.logistic_target(X, y)
    init = beta + 0.01 * rng.standard_normal((C, d))
    row = "n=%d d=%d chains=%d:" % (n, d, C)
    for name, path in (("stream", abi.PATH_STREAM), ("mfma", abi.PATH_MFMA)):
        nb = 20
        api.metrop(tg, init, 2, 1, 1, scale=0.5 / np.sqrt(n), seed=1, path=path)
        t0 = time.perf_counter()
        api.metrop(tg, init, nb, 1, 1, scale=0.5 / np.sqrt(n), seed=1, path=path)
        el = time.perf_counter() - t0
        row += "  %s %.3g chain-it/s (call %.1f ms)" % (name, C * nb / el, el * 1e3)
    print(row, flush=True)
