"""One-off fuzz of the lane-per-(tooth, column) olbm kernel (olbm_lane.hip, p <= 192, long windows) against the oracle and
an 80-bit evaluation (not collected by pytest).  usage (GPU box): MCMCHIP_OLBM_LANE=8 python tests/fuzz_olbm_lane.py
[nrounds] [seed]   (the environment value lowers the window threshold to 8 rows and takes every p <= 192)"""
import math
import os
import sys

import numpy as np

os.environ.setdefault("MCMCHIP_OLBM_LANE", "8")
os.environ.setdefault("MCMCHIP_OLBM_SMALL", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from mcmc_amd import api  # noqa: E402

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    p = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33, 40, 48, 49, 63, 64, 65, 80, 81, 96, 100, 112, 113, 128, 129, 144, 150, 160, 161, 176, 177, 192]))
    n = int(rng.choice([257, 1000, 1025, 4099, 30000, 262144, 1_048_576, 1_300_001]))
    if n * p > 12_000_000:
        n = 12_000_000 // p
    blen = int(rng.choice([8, 9, 16, 31, 33, 100, 255, 256, 1023, 1024, 5000, max(8, n // 5), max(8, n // 7), max(8, n // 40)]))
    blen = max(8, min(blen, (n + 1) // 5))           # nwin >= 4 len
    rho = rng.uniform(-0.5, 0.99, p)
    e = rng.standard_normal((n, p))
    x = np.empty((n, p))
    x[0] = e[0]
    for t in range(1, n):
        x[t] = rho * x[t - 1] + e[t]
    x += rng.uniform(-50, 50, p)
    ref, refmean = oracle.olbm(x, blen)
    sc = np.sqrt(np.outer(np.diag(ref), np.diag(ref))) + 1e-300
    got = api.olbm(x, blen)                         # means supplied by the front end
    kn = api.last_kernel_time()[0]
    got_d = api.olbm(x, blen, demean=False)         # means computed by the call
    err_a, err_b = np.max(np.abs(got - ref) / sc), np.max(np.abs(got_d - ref) / sc)
    exact = np.array([math.fsum(x[:, c]) / n for c in range(p)])
    xl = x.astype(np.longdouble) - exact.astype(np.longdouble)
    cs = np.vstack([np.zeros((1, p), dtype=np.longdouble), np.cumsum(xl, axis=0)])
    V = cs[blen:] - cs[:-blen]
    true = np.asarray((V.T @ V) / (np.longdouble(n - blen + 1) * n * blen), dtype=np.float64)
    st = np.sqrt(np.outer(np.diag(true), np.diag(true))) + 1e-300
    t_dev, t_orc = np.max(np.abs(got_d - true) / st), np.max(np.abs(ref - true) / st)
    sym = np.array_equal(got, got.T) and np.array_equal(got_d, got_d.T)
    ok = sym and kn == ("olbm_few_kernel" if p <= 8 else "olbm_lane_kernel") and max(err_a, err_b) < (1e-10 if blen <= 10_000 and 4 * blen <= n else 1e-9) + 3 * t_orc and t_dev < max(1e-12 if 4 * blen <= n else 1e-10, 3 * t_orc)
    bad += not ok
    print("n=%d p=%d len=%d  vs oracle %.1e / %.1e  vs 80-bit: device %.1e oracle %.1e  sym %s %s%s" % (n, p, blen, err_a, err_b, t_dev, t_orc, sym, kn, "" if ok else "  <<<<<<"), flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
