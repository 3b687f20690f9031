#!/usr/bin/env python3
"""The MFMA log density at the C3 shape (n = 1e6, d = 64, 65 536 chains) for the three kinds that reach it: logistic
posterior, binomial regression (weighted epilogue), model-selection posterior (masks folded into the states).
usage: python tools/mfma_kinds.py [n] [d] [chains]"""
import os
import sys

import numpy as np
import torch  # noqa: F401  (before libmcmchip.so: torch's HIP runtime has to initialise first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcmc_amd import abi, api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
C = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
api.init()
rng = np.random.default_rng(0)
X = rng.standard_normal((n, d)) / np.sqrt(d)
X[:, 0] = 1.0
beta = 0.5 * rng.standard_normal(d)
y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
tot = rng.integers(1, 30, n) + 0.5
states = beta + 0.05 * rng.standard_normal((C, d))
masks = (rng.random((4, d)) < 0.7).astype(float)
masks[:, 0] = 1.0
st = np.empty((C, d + 1))
st[:, 0] = rng.integers(1, 5, C)
st[:, 1:] = states
for name, tg, s in (("logistic", api.logistic_target(X, y), states),
                    ("binomial (weighted)", api.Target(abi.T_BINOMIAL_LOGIT, d, x=X, y=tot * 0.4, par=tot), states),
                    ("logistic models", api.Target(abi.T_LOGISTIC_MODELS, d, x=X, y=y, prior_c=0.125, par=abi.logistic_models_par(masks)), st)):
    for rep in range(3):
        api.logdens(tg, s, path=abi.PATH_MFMA)
    kn, kms, _ = api.last_kernel_time()
    print("%-22s %s %.2f ms  %.1f TFLOP/s = %.1f %% of 78.6" % (name, kn, kms, 2.0 * n * d * C / kms / 1e9, 2.0 * n * d * C / kms / 1e9 / 0.786))
