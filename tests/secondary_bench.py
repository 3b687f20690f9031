#!/usr/bin/env python3
"""Secondary measurements (not the headline bench): BASELINE.json configs 2, 4 and 5, each beside the CPU oracle.
It lives under tests/ because it times the oracle next to the device path (only tests/, smoke() and bench.py's
cpu_baseline leg may use oracle/); pytest does not collect it.
Prints one JSON line per config.  usage: python tests/secondary_bench.py [c2] [c4] [c4models] [c5] [c5dev=n,p,len]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # rhelpers

from mcmc_amd import abi, api  # noqa: E402


def c2():
    """logit posterior (n=100, d=5), 4096 chains, device Philox, 10 000 iterations per chain"""
    import oracle
    from rhelpers import glm_binomial, load_logit
    X, y = load_logit()
    mle = glm_binomial(X, y)
    Cn = 4096
    init = mle + 0.05 * np.random.default_rng(42).standard_normal((Cn, 5))
    tg = api.logistic_target(X, y)
    api.metrop(tg, init, 100, scale=0.4, seed=1)
    nb = 10000
    t0 = time.perf_counter()
    o = api.metrop(tg, init, nb, scale=0.4, seed=1)
    el = time.perf_counter() - t0
    name, kms, kn = api.last_kernel_time()
    ot = oracle.Target(abi.T_LOGISTIC, 5, x=X, y=y, prior_c=0.125)
    t0 = time.perf_counter()
    oracle.metrop(ot, init[:64], 2000, scale=0.4, seed=1, nthreads=16)
    cel = time.perf_counter() - t0
    return {"config": "C2 logit 100x5, 4096 chains x %d iterations, Philox" % nb, "chain_iter_per_s": Cn * nb / el,
            "kernel": name, "kernel_ms": kms * kn, "kernel_launches": kn, "accept": float(o["accept"].mean()),
            "softplus_evals_per_s": Cn * nb * 100 / (kms * kn * 1e-3),
            "cpu_oracle_16thr_chain_iter_per_s": 64 * 2000 / cel}


def c4():
    """parallel tempering, k = 64 temperatures x 1024 ensembles (witch's hat p=3; tempered bimodal mixture p=2)"""
    import oracle
    out = []
    k, E = 64, 1024
    i, j = np.indices((k, k))
    nb = np.abs(i - j) == 1
    for name in ("witch", "mixture"):
        if name == "witch":
            p = 3
            par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.linspace(0, 5, k) / p)
            K, x0 = abi.T_WITCH, np.zeros(p)
        else:
            p = 2
            par = np.concatenate([[float(k)], np.geomspace(1.0, 0.01, k),
                                  abi.normmix_par([0.5, 0.5], [[-4.0, -4.0], [4.0, 4.0]], 1.0)])
            K, x0 = abi.T_TEMPERED_NORMMIX, np.array([4.0, 4.0])
        tg = api.Target(K, p, par=par)
        init = np.tile(x0, (E, k, 1))
        api.temper(tg, init, nb, 2, 5, 5, 0.3, parallel=True, seed=3)
        nbatch, blen, nspac = 100, 10, 20
        niter = nbatch * blen * nspac
        ot = oracle.Target(K, p, par=par)
        t0 = time.perf_counter()
        oracle.temper(ot, init[:16], nb, 20, blen, nspac, 0.3, parallel=True, seed=3, nthreads=16)
        cel = time.perf_counter() - t0
        prev = None
        for path, pname in ((abi.PATH_LDS, "lds"), (abi.PATH_RUN, "run")):
            t0 = time.perf_counter()
            o = api.temper(tg, init, nb, nbatch, blen, nspac, 0.3, parallel=True, seed=3, _path=path)
            el = time.perf_counter() - t0
            kn, kms, nl = api.last_kernel_time()
            kms *= nl                              # the run is a sequence of launches: total kernel time
            same = None if prev is None else bool(all(np.array_equal(prev[n], o[n], equal_nan=True)
                                                      for n in ("final", "batch", "acceptx", "accepti")))
            prev = o
            out.append({"config": "C4 temper parallel %s k=64 p=%d, 1024 ensembles x %d iterations, %s kernel"
                                  % (name, p, niter, pname),
                        "ensemble_iter_per_s": E * niter / el, "kernel": kn, "kernel_ms": kms,
                        "kernel_ensemble_iter_per_s": E * niter / (kms * 1e-3), "same_bits_as_lds_kernel": same,
                        "acceptx_mean": float(np.nanmean(o["acceptx"])), "accepti_mean": float(np.nanmean(o["accepti"])),
                        "cpu_oracle_16thr_ensemble_iter_per_s": 16 * 20 * blen * nspac / cel})
    return out


def c4sweep():
    """which temper kernel wins where: witch's hat p = 3, 1024 ensembles x 6000 iterations, k and nspac varied"""
    out = []
    E, p = 1024, 3
    for k in (4, 8, 16, 32, 64):
        i, j = np.indices((k, k))
        nb = np.abs(i - j) == 1
        par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.linspace(0, 5, k) / p)
        tg = api.Target(abi.T_WITCH, p, par=par)
        init = np.zeros((E, k, p))
        for nbatch, blen, nspac in ((60, 100, 1), (60, 5, 20)):
            row = {"k": k, "nspac": nspac}
            for path, pname in ((abi.PATH_LDS, "lds"), (abi.PATH_RUN, "run")):
                api.temper(tg, init, nb, nbatch, blen, nspac, 0.3, parallel=True, seed=3, _path=path)
                row[pname + "_ms"] = round(api.last_kernel_time()[1], 3)
            out.append(row)
    return out


def c4models():
    """serial + parallel tempering over the 8 sub-models of tests/temp-par.R (data foo, n = 100, 4 coefficients),
    1024 ensembles: the reference's Bayes-factor workload (vignettes/bfst.Rnw)"""
    import oracle
    from rhelpers import temp_par_setup
    X, y, models, masks, betas, nb = temp_par_setup()
    kw = dict(x=X, y=y, prior_c=0.5, par=abi.logistic_models_par(masks))
    E = 1024
    out = []
    for parallel in (True, False):
        tg = api.Target(abi.T_LOGISTIC_MODELS, 4, **kw)
        init = np.tile(betas, (E, 1, 1)) if parallel else np.tile(np.concatenate([[8.0], betas[7]]), (E, 1))
        api.temper(tg, init, nb, 2, 5, 5, 0.56789, parallel=parallel, seed=3)
        nbatch, blen, nspac = 100, 10, 5
        t0 = time.perf_counter()
        o = api.temper(tg, init, nb, nbatch, blen, nspac, 0.56789, parallel=parallel, seed=3)
        el = time.perf_counter() - t0
        kn, kms, nl = api.last_kernel_time()
        kms *= nl
        niter = nbatch * blen * nspac
        ot = oracle.Target(abi.T_LOGISTIC_MODELS, 4, **kw)
        t0 = time.perf_counter()
        oracle.temper(ot, init[:16], nb, 20, blen, nspac, 0.56789, parallel=parallel, seed=3, nthreads=16)
        cel = time.perf_counter() - t0
        out.append({"config": "temper %s, logistic sub-models k=8 p=4 n=100, 1024 ensembles x %d iterations"
                    % ("parallel" if parallel else "serial", niter),
                    "ensemble_iter_per_s": E * niter / el, "kernel": kn, "kernel_ms": kms,
                    "acceptx_mean": float(np.nanmean(o["acceptx"])), "accepti_mean": float(np.nanmean(o["accepti"])),
                    "cpu_oracle_16thr_ensemble_iter_per_s": 16 * 20 * blen * nspac / cel})
    return out


def c5(n=4_000_000, p=32, blens=(1000,)):
    """initseq + olbm on AR(1) columns resident in HBM (reduced sizes: 4e6 x 32 = 1 GB, 2e6 x 256 = 4 GB)"""
    import torch
    from scipy.signal import lfilter
    rng = np.random.default_rng(5)
    rho = 0.5 + 0.49 * np.arange(p) / (p - 1)
    x = np.empty((p, n))                       # column-major n x p
    for c in range(p):
        x[c] = lfilter([1.0], [1.0, -rho[c]], rng.standard_normal(n))
    x -= x.mean(axis=1, keepdims=True)
    api.init()
    L = api.load_library()
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x).to(dev)
    torch.cuda.synchronize()
    res = []
    cap = 8192
    n_out = np.zeros(p, dtype=np.int64)
    g0, vp, vd, vc = (np.empty(p) for _ in range(4))
    Gp, Gd, Gc = (np.empty((p, cap)) for _ in range(3))
    for rep in range(2):
        t0 = time.perf_counter()
        rc = L.mcmchip_initseq(xd.data_ptr(), n, p, 1, cap, n_out.ctypes.data, g0.ctypes.data, Gp.ctypes.data,
                               Gd.ctypes.data, Gc.ctypes.data, vp.ctypes.data, vd.ctypes.data, vc.ctypes.data)
        el = time.perf_counter() - t0
        assert rc == 0, L.mcmchip_last_error()
    kn, kms, klaunch = api.last_kernel_time()
    lags = 2 * n_out
    res.append({"config": "C5r initseq n=%d p=%d (HBM resident)" % (n, p), "seconds": el, "kernel": kn,
                "kernel_ms_avg": kms, "kernel_launches": klaunch, "max_lag_pairs": int(n_out.max()),
                "useful_lag_products_per_s": float((lags * n).sum() / el),
                "var_con_over_gamma0": [float(v) for v in (vc / g0)[:4]]})
    mean = np.zeros(p)
    var = np.empty((p, p))
    for blen in blens:
        for rep in range(2):
            t0 = time.perf_counter()
            rc = L.mcmchip_olbm(xd.data_ptr(), n, p, blen, mean.ctypes.data, var.ctypes.data, 0, 1)
            el = time.perf_counter() - t0
            assert rc == 0, L.mcmchip_last_error()
        kn, kms, _ = api.last_kernel_time()
        res.append({"config": "C5r olbm n=%d p=%d len=%d (HBM resident)" % (n, p, blen), "seconds": el,
                    "kernel": kn, "kernel_ms": kms, "algorithmic_GBps": 8.0 * n * p / (kms * 1e-3) / 1e9,
                    "var00": float(var[0, 0])})
    return res


def c5dev(n=100_000_000, p=256, blen=10000, do_initseq=True):
    """BASELINE.json config 5 at full size: p AR(1) columns of n steps generated ON THE DEVICE (column-major
    n x p, 204.8 GB at the defaults), rho_j = 0.5 + 0.49 j/(p-1).  Generation: blocks of 512 steps by a scaled
    cumulative sum, carries from the two preceding blocks (rho^1024 < 4e-5: the rest is dropped)."""
    import torch
    api.init()
    L = api.load_library()
    dev = torch.device("cuda", 0)
    B = 512
    assert n % B == 0
    x = torch.empty((p, n), dtype=torch.float64, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    s_idx = torch.arange(B, dtype=torch.float64, device=dev)
    t0 = time.perf_counter()
    for c in range(p):
        rho = 0.5 + 0.49 * c / max(p - 1, 1)
        e = torch.randn((n // B, B), dtype=torch.float64, device=dev, generator=g)
        w = rho ** s_idx                                   # rho^s
        loc = torch.cumsum(e / w, dim=1) * w               # sum_{s<=t} rho^(t-s) e_s inside a block
        last = loc[:, -1].clone()
        carry = torch.zeros_like(last)
        carry[1:] = last[:-1]
        carry[2:] += (rho ** B) * last[:-2]
        loc += carry[:, None] * (rho * w)[None, :]
        x[c] = loc.reshape(-1)
        del e, loc
    x -= x.mean(dim=1, keepdim=True)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    res = []
    cap = 8192
    n_out = np.zeros(p, dtype=np.int64)
    g0, vp, vd, vc = (np.empty(p) for _ in range(4))
    Gp, Gd, Gc = (np.empty((p, cap)) for _ in range(3))
    t0 = time.perf_counter()
    rc = 0 if not do_initseq else L.mcmchip_initseq(x.data_ptr(), n, p, 1, cap, n_out.ctypes.data, g0.ctypes.data, Gp.ctypes.data,
                           Gd.ctypes.data, Gc.ctypes.data, vp.ctypes.data, vd.ctypes.data, vc.ctypes.data)
    el = time.perf_counter() - t0
    assert rc == 0, L.mcmchip_last_error()
    kn, kms, klaunch = api.last_kernel_time()
    lags = 2 * n_out
    if do_initseq:
        res.append({"config": "C5 initseq n=%d p=%d (generated on device, %.1f s)" % (n, p, gen_s), "seconds": el,
                    "kernel": kn, "kernel_ms_avg": kms, "kernel_launches": klaunch, "max_lag_pairs": int(n_out.max()),
                    "mean_lag_pairs": float(n_out.mean()),
                    "useful_lag_products_per_s": float((lags * n).sum() / el),
                    "useful_TFLOPs": float(2.0 * (lags * n).sum() / el / 1e12),
                    "var_con_over_gamma0": [float(v) for v in (vc / g0)[[0, p // 2, p - 1]]],
                    "theory_(1+rho)/(1-rho)": [(1 + r) / (1 - r) for r in (0.5, 0.5 + 0.49 * (p // 2) / max(p - 1, 1), 0.99)]})
    mean = np.zeros(p)
    var = np.empty((p, p))
    for nocalc in (0, 1):
        t0 = time.perf_counter()
        rc = L.mcmchip_olbm(x.data_ptr(), n, p, blen, mean.ctypes.data, var.ctypes.data, nocalc, 1)
        el = time.perf_counter() - t0
        assert rc == 0, L.mcmchip_last_error()
        kn, kms, _ = api.last_kernel_time()
        nf = (p + 15) // 16
        flops = 2.0 * n * 256.0 * (136 if p > 128 else (36 if p > 64 else 10))
        res.append({"config": "C5 olbm n=%d p=%d len=%d nocalc=%d" % (n, p, blen, nocalc), "seconds": el,
                    "kernel": kn, "kernel_ms": kms, "algorithmic_GBps": 8.0 * n * p / (kms * 1e-3) / 1e9,
                    "executed_TFLOPs": flops / (kms * 1e-3) / 1e12,
                    "useful_TFLOPs": n * float(p) * (p + 1) / (kms * 1e-3) / 1e12,
                    "var_diag_over_theory": [float(var[c, c] * n / ((1 / (1 - r * r)) * (1 + r) / (1 - r))) for c, r in
                                              ((0, 0.5), (p - 1, 0.99))]})
    return res


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c4", "c5"]
    for w in which:
        if w.startswith("c5dev"):          # c5dev or c5dev=n,p,len
            a = [int(v) for v in w[6:].split(",")] if "=" in w else []
            r = c5dev(*a)
        elif w.startswith("c5olbm="):        # c5olbm=n,p,len: the olbm half of c5dev alone
            r = c5dev(*[int(v) for v in w[7:].split(",")], do_initseq=False)
        elif w.startswith("c5="):            # c5=n,p,len[,len...]
            a = [int(v) for v in w[3:].split(",")]
            r = c5(a[0], a[1], tuple(a[2:]))
        else:
            r = {"c2": c2, "c4": c4, "c4models": c4models, "c4sweep": c4sweep, "c5": c5, "c5p256": lambda: c5(2_000_000, 256)}[w]()
        for line in (r if isinstance(r, list) else [r]):
            print(json.dumps(line))
