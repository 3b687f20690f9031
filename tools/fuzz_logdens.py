"""Random shapes for the MFMA log density against the oracle and the stream kernel (run on the GPU box).
usage: python tools/fuzz_logdens.py [ncases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: tests/conftest.py)
from mcmc_amd import abi, api  # noqa: E402
import oracle  # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    api.init(0)
    worst = 0.0
    for it in range(ncases):
        d = int(rng.integers(1, 257))
        n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 3000), rng.integers(3000, 60000)]))
        C = int(rng.choice([rng.integers(1, 20), rng.integers(20, 700)]))
        X = rng.standard_normal((n, d))
        X[:, 0] = 1.0
        beta = rng.standard_normal(d) / np.sqrt(d)
        y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
        st = beta + 0.3 * rng.standard_normal((C, d))
        st[0] *= rng.choice([1.0, 30.0, 1e3])
        tg = api.logistic_target(X, y)
        ref = oracle.logdens(oracle.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.125), st)
        m = api.logdens(tg, st, path=abi.PATH_MFMA)
        s = api.logdens(tg, st, path=abi.PATH_STREAM)
        e = max(np.max(np.abs(m - ref) / np.abs(ref)), np.max(np.abs(s - ref) / np.abs(ref)))
        worst = max(worst, e)
        flag = "" if e < 1e-12 else "   <-- FAIL"
        print("n=%6d d=%3d C=%4d  rel err %.2e%s" % (n, d, C, e, flag), flush=True)
        if e >= 1e-12:
            sys.exit(1)
    print("worst %.2e over %d cases" % (worst, ncases))


if __name__ == "__main__":
    main()
