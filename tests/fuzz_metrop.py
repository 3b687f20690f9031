"""One-off fuzz of the metrop stream path (and the MFMA path where it applies) against the oracle under injected
streams (not collected by pytest).  usage (GPU box): python tests/fuzz_metrop.py [nrounds] [seed] [only_index]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from mcmc_amd import abi, api  # noqa: E402
from mcmc_amd._morph import morph  # noqa: E402

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else None     # [only this configuration index], with details
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    d = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 64, 96, 128]))
    C = int(rng.choice([1, 3, 16, 65, 200]))
    nbatch, blen, nspac = int(rng.integers(1, 12)), int(rng.integers(1, 5)), int(rng.integers(1, 4))
    niter = nbatch * blen * nspac
    kind = str(rng.choice(["simplex", "ball", "mvnorm", "normmix", "student", "logistic", "logistic_mfma", "student_morph"]))
    kw, init, scale, path, m = {}, np.zeros((C, d)), 0.1, abi.PATH_AUTO, None
    if kind == "simplex":
        K, kw = abi.T_SIMPLEX, dict(par=[1.0])
        scale = 0.3 / d
    elif kind == "ball":
        K, scale = abi.T_BALL, 0.4 / np.sqrt(d)
    elif kind == "mvnorm":
        A = rng.standard_normal((d, d))
        K, kw = abi.T_MVNORM, dict(par=np.concatenate([rng.standard_normal(d), (A @ A.T + d * np.eye(d)).ravel()]))
        init, scale = rng.standard_normal((C, d)), 0.5 / np.sqrt(d)
    elif kind == "normmix":
        K, kw = abi.T_NORMMIX, dict(par=abi.normmix_par([0.5, 0.3, 0.2], rng.standard_normal((3, d)) * 2, [1.0, 0.5, 2.0]))
        init, scale = rng.standard_normal((C, d)), 1.0 / np.sqrt(d)
    elif kind in ("student", "student_morph"):
        K, kw = abi.T_STUDENT_T, dict(par=[float(rng.choice([1.0, 3.0, 7.5]))])
        init, scale = rng.standard_normal((C, d)), 1.5 / np.sqrt(d)
        if kind == "student_morph":
            m = morph(**[dict(b=1.0), dict(r=1.0), dict(b=0.5, r=0.5, p=3), dict(r=0.2, p=2.5, center=rng.standard_normal(d))][int(rng.integers(4))])
    else:
        n = int(rng.choice([7, 100, 513, 4096 + 37])) if kind == "logistic" else int(rng.choice([4096, 5000, 16384 + 5]))
        X = np.column_stack([np.ones(n), rng.standard_normal((n, d - 1)) / np.sqrt(d)]) if d > 1 else np.ones((n, 1))
        beta = rng.standard_normal(d) * 0.5
        y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
        K, kw = abi.T_LOGISTIC, dict(x=X, y=y, prior_c=0.125)
        init, scale = beta + 0.05 * rng.standard_normal((C, d)), 0.3 / np.sqrt(n)
        if kind == "logistic_mfma":
            path = abi.PATH_MFMA
    sk = str(rng.choice(["scalar", "vector", "matrix"]))
    if sk == "vector":
        scale = scale * rng.uniform(0.5, 1.5, d)
    elif sk == "matrix":
        scale = scale * (np.eye(d) + 0.1 * rng.standard_normal((d, d)))
    outfun = [None, "moments2", list(rng.random(d) < 0.5) if d > 1 else None][int(rng.integers(3))]
    if outfun is not None and not isinstance(outfun, str) and not any(outfun):
        outfun = None
    z, u = rng.standard_normal((C, niter, d)), rng.random((C, niter))
    if m is not None:
        kw["morph"] = m
    if ONLY is not None and it != ONLY:
        continue                            # replay one configuration: the inputs above consumed the same draws
    try:
        ref = oracle.metrop(oracle.Target(K, d, **kw), init, nbatch, blen, nspac, scale=scale, outfun=outfun, z=z, u=u)
    except oracle.OracleError as ex:           # e.g. -Inf at the initial state: the device must refuse in the same words
        try:
            api.metrop(api.Target(K, d, **kw), init, nbatch, blen, nspac, scale=scale, outfun=outfun, z=z, u=u, path=path)
            same = False
        except api.McmcHipError as ex2:
            same = str(ex2).strip() == str(ex).strip()
        bad += not same
        print("%-14s d=%d C=%d error parity: %r%s" % (kind, d, C, str(ex), "" if same else "   <<<<<<"), flush=True)
        continue
    o = api.metrop(api.Target(K, d, **kw), init, nbatch, blen, nspac, scale=scale, outfun=outfun, z=z, u=u, path=path)
    mfma = o["path"] == "mfma"
    fin = np.array_equal(o["final"], ref["final"]) if not mfma else np.allclose(o["final"], ref["final"], rtol=0, atol=0)
    acc = np.array_equal(np.atleast_1d(o["accept"]), ref["accept"])
    ob, rb = np.asarray(o["batch"]).reshape(ref["batch"].shape), ref["batch"]
    if m is None:
        be = np.max(np.abs(ob - rb) / np.maximum(np.abs(rb), 1.0))
        tol = 1e-12
    else:
        # batch means of inverse(state): a composite transformation in d = 64 reaches 1e80 and more, with
        # d log f.inv / d log |y| in the hundreds, so one term of a batch carries ~1e-13 relative error and terms of
        # opposite sign cancel; measure against the largest value the chain's batches hold
        scale_c = np.maximum(np.max(np.abs(rb).reshape(rb.shape[0], -1), axis=1), 1.0)
        be = np.max(np.abs(ob - rb).reshape(rb.shape[0], -1) / scale_c[:, None])
        tol = 1e-10
    ok = fin and acc and be < tol
    bad += not ok
    print("%-14s d=%d C=%d nb=%d bl=%d ns=%d scale=%s out=%s path=%s final %s accept %s batch %.1e%s"
          % (kind, d, C, nbatch, blen, nspac, sk, "idx" if isinstance(outfun, list) else outfun, o["path"], fin, acc, be,
             "" if ok else "   <<<<<<"), flush=True)
    if ONLY is not None:
        i = np.unravel_index(np.argmax(np.abs(ob - rb) / np.maximum(np.abs(rb), 1.0)), rb.shape)
        print("worst batch entry", i, repr(float(ob[i])), repr(float(rb[i])), "morph", None if m is None else m.__dict__)
        print("final of that chain", o["final"][i[0]][:4], "|y|", np.linalg.norm(o["final"][i[0]]))
print("bad", bad)
sys.exit(1 if bad else 0)
