"""How far olbm.c's own arithmetic and the device routes are from the true matrix on a long-window shape
(n = 1 050 000, len = n/2, an AR(1) column and a trend column), judged against an 80-bit evaluation of the definition.
Run on the GPU box: python tools/olbm_conditioning.py.  Measured (round 2): oracle 1.4e-10 (1.6e-10 with its own
sequentially summed means), device one-pass route 6e-14, device with supplied exact means 9e-14 -- which is why
tests/test_gpu_round2.py::test_olbm_one_pass_route_vs_oracle holds the device to 1e-12 against the 80-bit evaluation
and to 1e-9, not 1e-10, against the oracle where the windows are long."""
import sys, math, numpy as np
sys.path.insert(0, ".")
import torch
from scipy.signal import lfilter
from mcmc_amd import api
import oracle as oracle_mod
n, p, blen = 1_050_000, 2, 525_000
rng = np.random.default_rng(n % 1000 + p)
x = np.empty((n, p))
for c in range(p):
    x[:, c] = lfilter([1.0], [1.0, -(0.2 + 0.7 * c / p)], rng.standard_normal(n)) + (3.0 + 100.0 * (c % 3))
x[:, 1] += np.linspace(-5.0, 5.0, n)
exact = np.array([math.fsum(x[:, c]) / n for c in range(p)])
xl = x.astype(np.longdouble)
ml = np.array([np.longdouble(math.fsum(x[:, c])) / n for c in range(p)])
cs = np.vstack([np.zeros((1, p), dtype=np.longdouble), np.cumsum(xl - ml, axis=0)])
V = cs[blen:] - cs[:-blen]
nwin = n - blen + 1
true = (V.T @ V) / (np.longdouble(nwin) * n * blen)
ref, refmean = oracle_mod.olbm(x, blen)
ref_exact, _ = oracle_mod.olbm(x, blen, mean=exact)
got = api.olbm(x, blen, demean=False)
got_e = api.olbm(x, blen, mean=exact)
got_r = api.olbm(x, blen, mean=refmean)
d = np.diag(true).astype(float)
sc = np.sqrt(np.outer(d, d))
for name, m in (("oracle own mean", ref), ("oracle exact mean", ref_exact), ("gpu one-pass", got), ("gpu exact mean", got_e), ("gpu oracle mean", got_r)):
    print(name, (np.abs(m - true.astype(float)) / sc).ravel())
print("refmean-exact", refmean - exact)
