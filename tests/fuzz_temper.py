"""One-off fuzz of the temper kernel against the oracle under injected streams, trace included (not collected by
pytest).  usage (GPU box): python tests/fuzz_temper.py [nrounds] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from mcmc_amd import abi, api  # noqa: E402

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    k = int(rng.choice([2, 3, 6, 17, 64]))
    p = int(rng.choice([1, 2, 3, 4, 5, 9, 16]))
    E = int(rng.choice([1, 3, 8]))
    parallel = bool(rng.integers(2))
    debug = bool(rng.integers(2))
    nbatch, blen, nspac = int(rng.integers(1, 8)), int(rng.integers(1, 6)), int(rng.integers(1, 4))
    if rng.integers(3) == 0:                                # several 64-iteration windows of the run-parallel kernel
        nbatch, blen, nspac = int(rng.integers(1, 40)), int(rng.integers(1, 9)), int(rng.integers(1, 6))
    niter = nbatch * blen * nspac
    nb = np.zeros((k, k), dtype=bool)                       # random connected symmetric graph
    for i in range(1, k):
        j = int(rng.integers(0, i))
        nb[i, j] = nb[j, i] = True
    extra = rng.random((k, k)) < float(rng.choice([0.0, 0.02, 0.15]))
    nb |= np.triu(extra, 1) | np.triu(extra, 1).T
    kind = str(rng.choice(["witch", "mixture", "models"]))
    out_kind, outfun = oracle.TOUT_NONE, None
    if kind == "witch":
        par = np.sort(rng.uniform(0.05, 0.9, k))
        K, kw = abi.T_WITCH, dict(par=par)
        x0 = np.zeros(p)
        if rng.integers(2):
            out_kind, outfun = (oracle.TOUT_WITCH_PAR if parallel else oracle.TOUT_WITCH_SER), "witch.corner"
    elif kind == "mixture":
        par = np.concatenate([[float(k)], np.geomspace(1.0, 0.02, k),
                              abi.normmix_par([0.5, 0.5], np.stack([-2.0 * np.ones(p), 2.0 * np.ones(p)]), 1.0)])
        K, kw = abi.T_TEMPERED_NORMMIX, dict(par=par)
        x0 = 2.0 * np.ones(p)
    else:
        n = int(rng.choice([9, 100, 257]))
        X = np.column_stack([np.ones(n), rng.standard_normal((n, p - 1))]) if p > 1 else np.ones((n, 1))
        y = (rng.random(n) < 0.5).astype(float)
        masks = (rng.random((k, p)) < 0.6).astype(float)
        masks[:, 0] = 1
        K = abi.T_LOGISTIC_MODELS
        kw = dict(x=X, y=y, prior_c=0.125, par=abi.logistic_models_par(masks, rng.standard_normal(k), -1.5))
        x0 = 0.1 * rng.standard_normal(p)
    init = np.tile(x0, (E, k, 1)) if parallel else np.column_stack([rng.integers(1, k + 1, E).astype(float), np.tile(x0, (E, 1))])
    sk = str(rng.choice(["scalar", "vector", "matrix", "list"]))
    scale = 0.4
    if sk == "vector":
        scale = 0.4 * rng.uniform(0.5, 1.5, p)
    elif sk == "matrix":
        scale = 0.4 * (np.eye(p) + 0.1 * rng.standard_normal((p, p)))
    elif sk == "list":
        scale = [float(v) for v in rng.uniform(0.1, 0.8, k)]
    unif, norm = rng.random((E, niter, 4)), rng.standard_normal((E, niter, p))
    ref = oracle.temper(oracle.Target(K, p, **kw), init, nb, nbatch, blen, nspac, scale, parallel=parallel,
                        unif=unif, norm=norm, debug=debug, out_kind=out_kind)
    # the model-selection posterior without a trace also runs with the ensembles in lockstep and the densities on the
    # matrix cores (temper_mfma.hip): half of those cases take that path
    tpath = abi.PATH_TMFMA if (kind == "models" and not debug and rng.integers(2)) else 0
    o = api.temper(api.Target(K, p, **kw), init, nb, nbatch, blen, nspac, scale, parallel=parallel, unif=unif,
                   norm=norm, debug=debug, outfun=outfun, _path=tpath)
    ok = np.array_equal(o["final"], ref["final"]) and np.array_equal(o["acceptx"], ref["acceptx"], equal_nan=True)
    ok &= np.array_equal(o["accepti"], ref["accepti"], equal_nan=True)
    ok &= np.allclose(o["batch"], ref["batch"], rtol=1e-13, atol=1e-15)
    if not parallel:
        ok &= np.array_equal(o["ibatch"], ref["ibatch"])
    why = []
    if debug:
        for key in ("which", "unif.which", "state", "unif.hastings", "proposal", "acceptd", "norm", "unif.choose") + (("coproposal",) if parallel else ()):
            if not np.array_equal(o[key], ref[key], equal_nan=True):
                if key == "unif.hastings":
                    # a log Hastings ratio that is 0 up to rounding (a swap of two still-identical states) is < 0 on
                    # one side and >= 0 on the other when the densities differ in the last bit: the uniform is then
                    # drawn on one side only, and accepts either way
                    diff = ~((o[key] == ref[key]) | (np.isnan(o[key]) & np.isnan(ref[key])))
                    if np.all(np.abs(ref["log.hastings"][diff]) < 1e-9):
                        continue
                ok = False
                why.append(key)
        if not np.allclose(o["log.hastings"], ref["log.hastings"], rtol=0, atol=1e-10, equal_nan=True):
            ok = False
            with np.errstate(invalid="ignore"):
                why.append("log.hastings %.3g" % np.nanmax(np.abs(o["log.hastings"] - ref["log.hastings"])))
    bad += not ok
    print("%-8s k=%d p=%d E=%d %s dbg=%d nb=%d bl=%d ns=%d scale=%s out=%s %s"
          % (kind + ("/mfma" if tpath else ""), k, p, E, "par" if parallel else "ser", debug, nbatch, blen, nspac, sk, outfun, "ok" if ok else "MISMATCH %s <<<<<<" % why),
          flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
