"""One-off fuzz of initseq / olbm on the device against the oracle (not collected by pytest).
usage (GPU box): python tests/fuzz_diag.py [nrounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from mcmc_amd import api  # noqa: E402

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    n = int(rng.choice([2, 3, 5, 63, 64, 65, 127, 129, 1000, 2047, 2048, 2049, 4097, 30000, 65520, 65536, 65553, 131100, 1_048_576, 1_200_001]))
    p = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 64, 65, 100, 129, 255, 256, 257]))
    if n * p > 6_000_000:
        p = max(1, 6_000_000 // n)
    rho = rng.uniform(-0.5, 0.995, p)
    e = rng.standard_normal((n, p))
    x = np.empty((n, p))
    x[0] = e[0]
    for t in range(1, n):
        x[t] = rho * x[t - 1] + e[t]
    x += rng.uniform(-5, 5, p)
    # olbm
    blen = int(rng.integers(1, n + 1))
    ref, _ = oracle.olbm(x, blen)
    got = api.olbm(x, blen)
    err_o = np.max(np.abs(got - ref)) / (np.max(np.abs(ref)) + 1e-18)      # len == n: the reference is exactly 0
    # no.calc.mean = FALSE: the device computes the means (n >= 2^20: the one-pass route of diag.hip, whose means are
    # closer to the truth than olbm.c's sequential sums -- with long windows that alone moves the matrix by ~1e-10)
    got_d = api.olbm(x, blen, demean=False)
    err_d = np.max(np.abs(got_d - ref)) / (np.max(np.abs(ref)) + 1e-18)
    err_o = max(err_o, err_d / (5.0 if n >= (1 << 20) else 1.0))
    # initseq on a few columns
    cols = rng.choice(p, size=min(p, 5), replace=False)
    res = api.initseq(x[:, cols])
    err_i, lenbad = 0.0, 0
    for i, c in enumerate(cols):
        r = oracle.initseq(x[:, c] - x[:, c].mean())
        g = res[i] if isinstance(res, list) else res
        if len(g["Gamma.pos"]) != len(r["Gamma.pos"]):
            lenbad += 1
            continue
        for key in ("gamma0", "Gamma.con", "var.con", "var.pos"):
            a, b = np.asarray(g[key]), np.asarray(r[key])
            # the variances are 2 sum(Gamma) - gamma0 and can cancel to nothing (n = 2: exactly 0): scale by gamma0 too
            err_i = max(err_i, float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), r["gamma0"], 1e-300)) if b.size else 0.0)
    flag = "" if (err_o < 1e-9 and err_i < 1e-8 and lenbad == 0) else "  <<<<<<"
    bad += bool(flag)
    print("n=%d p=%d len=%d olbm %.2e initseq %.2e lenbad %d%s" % (n, p, blen, err_o, err_i, lenbad, flag), flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
