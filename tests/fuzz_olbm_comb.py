"""One-off fuzz of the comb route of olbm (olbm_comb.hip) against the oracle (not collected by pytest).  The route is
taken for windows of >= 1024 rows; MCMCHIP_OLBM_COMB=<min len> lowers that so that small series exercise every branch
of the plan (several combs per workgroup, ragged last steps, the remainder tooth, 1 .. M teeth).
usage (GPU box): MCMCHIP_OLBM_COMB=16 python tests/fuzz_olbm_comb.py [nrounds] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from mcmc_amd import api  # noqa: E402

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
minlen = int(os.environ.get("MCMCHIP_OLBM_COMB", "1024"))
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    p = int(rng.choice([1, 2, 3, 15, 16, 17, 33, 64, 65, 100, 128, 129, 200, 255, 256]))
    blen = int(rng.choice([minlen, minlen + 1, minlen + 15, 2 * minlen + 7, 5 * minlen, 37 * minlen + 3]))
    nw = int(rng.choice([2, 3, 5, 6, 7, 11, 23, 40])) * blen + int(rng.integers(0, blen))
    n = nw + blen - 1
    if n * p > 8_000_000:
        p = max(1, 8_000_000 // n)
    os.environ["MCMCHIP_OLBM_TEETH"] = str(int(rng.integers(1, 9)))
    rho = rng.uniform(-0.5, 0.99, p)
    e = rng.standard_normal((n, p))
    x = np.empty((n, p))
    x[0] = e[0]
    for t in range(1, n):
        x[t] = rho * x[t - 1] + e[t]
    x += rng.uniform(-50, 50, p)
    ref, refmean = oracle.olbm(x, blen)
    sc = np.sqrt(np.outer(np.diag(ref), np.diag(ref))) + 1e-300
    got = api.olbm(x, blen)                         # means supplied by the front end (R/olbm.R:8-14)
    err_a = np.max(np.abs(got - ref) / sc)
    got_d = api.olbm(x, blen, demean=False)         # means computed by the call
    err_b = np.max(np.abs(got_d - ref) / sc)
    # the definition in 80-bit arithmetic (first columns): whose rounding is it?
    k = min(p, 4)
    import math
    exact = np.array([math.fsum(x[:, c]) / n for c in range(k)])
    xl = x[:, :k].astype(np.longdouble) - exact.astype(np.longdouble)
    cs = np.vstack([np.zeros((1, k), dtype=np.longdouble), np.cumsum(xl, axis=0)])
    V = cs[blen:] - cs[:-blen]
    true = np.asarray((V.T @ V) / (np.longdouble(n - blen + 1) * n * blen), dtype=np.float64)
    st = np.sqrt(np.outer(np.diag(true), np.diag(true))) + 1e-300
    t_dev, t_orc = np.max(np.abs(got_d[:k, :k] - true) / st), np.max(np.abs(ref[:k, :k] - true) / st)
    sym = np.array_equal(got, got.T) and np.array_equal(got_d, got_d.T)
    flag = "" if (max(err_a, err_b) < 1e-10 + 3 * t_orc and sym) else "  <<<<<<"   # olbm.c itself is t_orc away from the 80-bit value
    bad += bool(flag)
    print("n=%d p=%d len=%d teeth=%s  means given %.2e  computed %.2e  vs 80-bit: device %.1e oracle %.1e  sym %s%s" % (n, p, blen, os.environ["MCMCHIP_OLBM_TEETH"], err_a, err_b, t_dev, t_orc, sym, flag), flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
