"""Regenerates tests/golden/*.npz: small input/expected-output vectors produced by the CPU oracle
(oracle/, itself pinned against the reference's printed transcripts in test_oracle_golden.py).
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from mcmc_amd import abi  # noqa: E402
from rhelpers import glm_binomial, load_logit  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def metrop_logit():
    X, y = load_logit()
    rng = np.random.default_rng(11)
    C, nbatch, blen, nspac = 8, 20, 5, 2
    niter = nbatch * blen * nspac
    init = glm_binomial(X, y) + 0.05 * rng.standard_normal((C, 5))
    z = rng.standard_normal((C, niter, 5))
    u = rng.random((C, niter))
    tg = oracle.Target(abi.T_LOGISTIC, 5, x=X, y=y, prior_c=0.125)
    o = oracle.metrop(tg, init, nbatch, blen, nspac, scale=0.4, z=z, u=u, debug=True)
    np.savez_compressed(os.path.join(OUT, "metrop_logit.npz"), init=init, z=z, u=u, scale=0.4,
                        nbatch=nbatch, blen=blen, nspac=nspac, accept=o["accept"], batch=o["batch"],
                        final=o["final"], accept_batch=o["accept_batch"], log_green=o["log_green"],
                        debug_accept=o["debug_accept"], u_trace=o["u"])


def temper_witch():
    rng = np.random.default_rng(12)
    k, p = 6, 3
    witch = 1 - (1 / 2) ** (1 / 3) * (1 / 4) ** (np.arange(6) / 3)
    i, j = np.indices((k, k))
    nb = (np.abs(i - j) == 1) | (np.abs(i - j) == 2)
    E, nbatch, blen, nspac = 4, 10, 7, 3
    niter = nbatch * blen * nspac
    unif = rng.random((E, niter, 4))
    norm = rng.standard_normal((E, niter, p))
    tg = oracle.Target(abi.T_WITCH, p, par=witch)
    op = oracle.temper(tg, np.zeros((E, k, p)), nb, nbatch, blen, nspac, 0.3456789, parallel=True,
                       unif=unif, norm=norm)
    init_s = np.tile(np.array([1.0, 0, 0, 0]), (E, 1))
    os_ = oracle.temper(tg, init_s, nb, nbatch, blen, nspac, 0.3456789, parallel=False, unif=unif, norm=norm)
    np.savez_compressed(os.path.join(OUT, "temper_witch.npz"), witch=witch, neighbors=nb, unif=unif,
                        norm=norm, nbatch=nbatch, blen=blen, nspac=nspac, scale=0.3456789,
                        par_batch=op["batch"], par_acceptx=op["acceptx"], par_accepti=op["accepti"],
                        par_final=op["final"], ser_batch=os_["batch"], ser_ibatch=os_["ibatch"],
                        ser_acceptx=os_["acceptx"], ser_accepti=os_["accepti"], ser_final=os_["final"])


def diag():
    rng = np.random.default_rng(13)
    n = 20000
    e = rng.standard_normal((n, 3))
    x = np.empty_like(e)
    rho = np.array([0.5, 0.9, 0.97])
    x[0] = e[0]
    for t in range(1, n):
        x[t] = rho * x[t - 1] + e[t]
    xs = x.astype(np.float32)     # stored compactly; the fixture IS the float32-rounded series
    x = xs.astype(np.float64)
    res = [oracle.initseq(x[:, c] - x[:, c].mean()) for c in range(3)]
    var, mean = oracle.olbm(x, 150)
    np.savez_compressed(os.path.join(OUT, "diag.npz"), x=xs, olbm_len=150, olbm_var=var, olbm_mean=mean,
                        **{"c%d_%s" % (c, k.replace(".", "_")): v for c in range(3) for k, v in res[c].items()})


if __name__ == "__main__":
    oracle.build()
    metrop_logit()
    temper_witch()
    diag()
    print("golden vectors written to", OUT)
