"""bench_secondary.py -- BASELINE.json configs 2, 4 and 5 on the device, for the `secondary` block of bench.py's JSON
line (the headline fields are config 3's).  Device path only: nothing here touches oracle/ (the CPU comparisons of
these configs live in tests/secondary_bench.py).

  C2  logit posterior (n = 100, d = 5), 4096 chains x 10 000 iterations, device Philox   (metrop_stream_kernel)
  C4  parallel tempering, k = 64 temperatures x 1024 ensembles x 20 000 iterations: witch's hat p = 3 and the tempered
      bimodal mixture p = 2                                                               (temper_run_kernel)
  C5  initseq + olbm on 1e8 steps x 256 AR(1) functionals generated on the device (204.8 GB resident)
"""
import os
import time

import numpy as np

PEAK_FP64_MFMA_TFLOPS = 78.6
PEAK_HBM_TBS = 8.0


def _logit():
    root = os.path.dirname(os.path.abspath(__file__))
    raw = np.loadtxt(os.path.join(root, "tests", "golden", "logit.txt"), skiprows=1)     # y, x1..x4 (pkg/data/logit.txt)
    y = raw[:, 0].copy()
    X = np.column_stack([np.ones(raw.shape[0]), raw[:, 1:5]])
    return X, y


def c2(api):
    X, y = _logit()
    Cn, nb = 4096, 10000
    rng = np.random.default_rng(42)
    init = np.array([0.633, 0.739, 1.114, 0.478, 0.694]) + 0.05 * rng.standard_normal((Cn, 5))     # the MLE (glm) to three digits
    tg = api.logistic_target(X, y)
    api.metrop(tg, init, 100, scale=0.4, seed=1)
    t0 = time.perf_counter()
    o = api.metrop(tg, init, nb, scale=0.4, seed=1)
    el = time.perf_counter() - t0
    name, kms, kn = api.last_kernel_time()
    return {"workload": "C2: logit 100x5, 4096 chains x %d iterations, device Philox, results to pageable host memory" % nb,
            "kernel": name, "kernel_ms": kms * kn, "kernel_launches": kn,
            "chain_iter_per_s_kernel": Cn * nb / (kms * kn * 1e-3), "chain_iter_per_s_end_to_end": Cn * nb / el,
            "accept": float(o["accept"].mean()), "bound": "latency (one wave per SIMD, dependent fp64 chain)"}


def c4(api):
    from mcmc_amd import abi
    out = []
    k, E = 64, 1024
    i, j = np.indices((k, k))
    nb = np.abs(i - j) == 1
    nbatch, blen, nspac = 100, 10, 20
    niter = nbatch * blen * nspac
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
        t0 = time.perf_counter()
        o = api.temper(tg, init, nb, nbatch, blen, nspac, 0.3, parallel=True, seed=3)
        el = time.perf_counter() - t0
        kn, kms, nl = api.last_kernel_time()
        out.append({"workload": "C4: parallel tempering %s k=64 p=%d, 1024 ensembles x %d iterations" % (name, p, niter),
                    "kernel": "temper_run_kernel", "kernel_ms": kms * nl, "ensemble_iter_per_s_kernel": E * niter / (kms * nl * 1e-3),
                    "ensemble_iter_per_s_end_to_end": E * niter / el, "acceptx_mean": float(np.nanmean(o["acceptx"])),
                    "accepti_mean": float(np.nanmean(o["accepti"])), "bound": "latency (one wavefront per ensemble)"})
    return out


def ar1_columns(n, p, torch, dev, seed=5):
    """p AR(1) columns of n steps, rho_j = 0.5 + 0.49 j / (p - 1), column-major n x p in HBM: blocks
    of 512 steps by a scaled cumulative sum, carries from the two preceding blocks (rho^1024 < 4e-5: the rest is dropped)"""
    B = 512
    assert n % B == 0
    x = torch.empty((p, n), dtype=torch.float64, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    s_idx = torch.arange(B, dtype=torch.float64, device=dev)
    for c in range(p):
        rho = 0.5 + 0.49 * c / max(p - 1, 1)
        e = torch.randn((n // B, B), dtype=torch.float64, device=dev, generator=g)
        w = rho ** s_idx
        loc = torch.cumsum(e / w, dim=1) * w
        last = loc[:, -1].clone()
        carry = torch.zeros_like(last)
        carry[1:] = last[:-1]
        carry[2:] += (rho ** B) * last[:-2]
        loc += carry[:, None] * (rho * w)[None, :]
        x[c] = loc.reshape(-1)
        del e, loc
    x -= x.mean(dim=1, keepdim=True)
    torch.cuda.synchronize()
    return x


def c5(api, torch, dev, n=100_000_256, p=256, blen=10_000):
    L = api.load_library()
    t0 = time.perf_counter()
    x = ar1_columns(n, p, torch, dev)
    gen_s = time.perf_counter() - t0
    cap = 8192
    n_out = np.zeros(p, dtype=np.int64)
    g0, vp, vd, vc = (np.empty(p) for _ in range(4))
    Gp, Gd, Gc = (np.empty((p, cap)) for _ in range(3))
    secs = []
    for rep in range(2):
        t0 = time.perf_counter()
        rc = L.mcmchip_initseq(x.data_ptr(), n, p, 1, cap, n_out.ctypes.data, g0.ctypes.data, Gp.ctypes.data,
                               Gd.ctypes.data, Gc.ctypes.data, vp.ctypes.data, vd.ctypes.data, vc.ctypes.data)
        secs.append(time.perf_counter() - t0)
        if rc:
            raise RuntimeError(L.mcmchip_last_error().decode())
    el = min(secs)
    lags = 2.0 * n_out                                 # lags the reference's loop evaluates per column (initseq.c:24-47)
    # roofline-optimal time with every stopping lag known in advance and the reference's arithmetic (one multiply-add
    # per base position and lag): per column the larger of one pass over HBM and its lag products on the matrix cores
    t_opt = float(np.sum(np.maximum(8.0 * n / (PEAK_HBM_TBS * 1e12), 2.0 * n * lags / (PEAK_FP64_MFMA_TFLOPS * 1e12))))
    res = [{"workload": "C5: initseq n=%d x %d AR(1) functionals resident in HBM (generated on the device in %.1f s)" % (n, p, gen_s),
            "seconds": el, "kernel": "autocov_kernel", "max_lag_pairs": int(n_out.max()), "mean_lag_pairs": float(n_out.mean()),
            "useful_TFLOPs": float(2.0 * np.sum(lags) * n / el / 1e12),
            "roofline": {"bound": "hbm for columns under ~40 lags, mfma above (per column the larger of the two)",
                         "optimal_seconds": t_opt, "frac": t_opt / el,
                         "note": "optimal = sum over columns of max(8 n B / 8 TB/s, 2 n lags flop / 78.6 TF) with the stopping "
                                 "lags known in advance; the kernel spends one multiply-add per lag PAIR (x[t] * (x[t+2k] + "
                                 "x[t+2k+1])), i.e. half the reference's count, and computes lags speculatively"},
            "var_con_over_gamma0": [float(v) for v in (vc / g0)[[0, p // 2, p - 1]]],
            "theory_(1+rho)/(1-rho)": [(1 + r) / (1 - r) for r in (0.5, 0.5 + 0.49 * (p // 2) / max(p - 1, 1), 0.99)]}]
    # the same series handed over as NOT centred: about a provisional centre in the same passes (mcmchip_initseq_uncentred),
    # next to what that cost before (centre in place: a read pass and a read-write pass, then the call above)
    n_u = np.zeros(p, dtype=np.int64)
    g0u, vpu, vdu, vcu, mean_u = (np.empty(p) for _ in range(5))
    secs_u = []
    for rep in range(2):
        t0 = time.perf_counter()
        rc = L.mcmchip_initseq_uncentred(x.data_ptr(), n, p, 1, cap, n_u.ctypes.data, g0u.ctypes.data, Gp.ctypes.data,
                                         Gd.ctypes.data, Gc.ctypes.data, vpu.ctypes.data, vdu.ctypes.data, vcu.ctypes.data,
                                         mean_u.ctypes.data)
        secs_u.append(time.perf_counter() - t0)
        if rc:
            raise RuntimeError(L.mcmchip_last_error().decode())
    t0 = time.perf_counter()
    rc = L.mcmchip_series_centre(x.data_ptr(), n, p, None)         # the series is centred already: this subtracts ~1e-17
    centre_s = time.perf_counter() - t0
    if rc:
        raise RuntimeError(L.mcmchip_last_error().decode())
    res.append({"workload": "C5: initseq of the same series handed over as not centred (mcmchip_initseq_uncentred: lag sums about a "
                            "provisional centre, carried over to the mean on the host; the series is read once per round and not modified)",
                "seconds": min(secs_u), "kernel": "autocov_kernel",
                "before": {"centre_in_place_seconds": centre_s, "then_initseq_seconds": el, "total": centre_s + el},
                "same_stopping_lags": bool(np.array_equal(n_u, n_out)),
                "max_rel_diff_var_con": float(np.max(np.abs(vcu - vc) / np.abs(vc))),
                "max_abs_mean": float(np.max(np.abs(mean_u)))})
    mean = np.zeros(p)
    var = np.empty((p, p))
    for nocalc in (0, 1):
        t0 = time.perf_counter()
        rc = L.mcmchip_olbm(x.data_ptr(), n, p, blen, mean.ctypes.data, var.ctypes.data, nocalc, 1)
        el = time.perf_counter() - t0
        if rc:
            raise RuntimeError(L.mcmchip_last_error().decode())
        kn, kms, _ = api.last_kernel_time()
        useful = n * float(p) * (p + 1) / (kms * 1e-3) / 1e12
        res.append({"workload": "C5: olbm n=%d p=%d len=%d, means %s" % (n, p, blen, "supplied" if nocalc else "computed in the same pass"),
                    "seconds": el, "kernel": kn, "kernel_ms": kms, "useful_TFLOPs": useful,
                    "roofline": {"bound": "mfma", "achieved": useful, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": useful / PEAK_FP64_MFMA_TFLOPS,
                                 "hbm_algorithmic_TBs": 8.0 * n * p / (kms * 1e-3) / 1e12},
                    "var_diag_over_theory": [float(var[c, c] * n / ((1 / (1 - r * r)) * (1 + r) / (1 - r))) for c, r in
                                              ((0, 0.5), (p - 1, 0.99))]})
    # the usual call has a handful of functionals: the first 3 (olbm_small.hip), 8 and 32 (olbm_lane.hip) columns of the same series
    for pp in (3, 8, 32):
        m2, v2 = np.zeros(pp), np.empty((pp, pp))
        for rep in range(2):
            t0 = time.perf_counter()
            rc = L.mcmchip_olbm(x.data_ptr(), n, pp, blen, m2.ctypes.data, v2.ctypes.data, 0, 1)
            el = time.perf_counter() - t0
            if rc:
                raise RuntimeError(L.mcmchip_last_error().decode())
        kn, kms, _ = api.last_kernel_time()
        res.append({"workload": "olbm n=%d p=%d len=%d (few functionals), means computed in the same pass" % (n, pp, blen),
                    "seconds": el, "kernel": kn, "kernel_ms": kms,
                    "roofline": {"bound": "hbm", "achieved": 8.0 * n * pp / (kms * 1e-3) / 1e12, "peak": PEAK_HBM_TBS, "unit": "TB/s",
                                 "frac": 8.0 * n * pp / (kms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                                 "note": "algorithmic bytes = the series once; olbm_small reads x[t+len] and x[t], olbm_lane five rows for four teeth (1.25 x)"},
                    "var_diag_over_theory": [float(v2[0, 0] * n / ((1 / (1 - 0.25)) * 3.0))]})
    del x
    torch.cuda.empty_cache()
    return res


def temper_models(api):
    """Tempering over the 8 sub-models of tests/temp-par.R on a synthetic data set of 262 144 rows: the ensembles in
    lockstep, log densities on the matrix cores (temper_mfma.hip)."""
    from mcmc_amd import abi
    rng = np.random.default_rng(0)
    n, p, k, E, niter = 262144, 4, 8, 1024, 200
    masks = np.array([[1, a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], dtype=float)
    nb = np.array([[np.sum(masks[i] != masks[j]) == 1 for j in range(k)] for i in range(k)])
    X = rng.standard_normal((n, p)) / 2
    X[:, 0] = 1.0
    beta = np.array([0.3, 0.5, -0.4, 0.2])
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    tg = api.Target(abi.T_LOGISTIC_MODELS, p, x=X, y=y, prior_c=0.125, par=abi.logistic_models_par(masks))
    init = np.tile(beta, (E, k, 1))
    api.temper(tg, init, nb, 2, 1, 1, 2.0 / np.sqrt(n), parallel=True, seed=1)
    t0 = time.perf_counter()
    o = api.temper(tg, init, nb, niter, 1, 1, 2.0 / np.sqrt(n), parallel=True, seed=1)
    el = time.perf_counter() - t0
    kn, kms, nl = api.last_kernel_time()
    return {"workload": "parallel tempering over 8 logistic sub-models (temp-par.R) on n=%d rows, %d ensembles x %d iterations" % (n, E, niter),
            "kernel": kn, "kernel_ms_per_iteration": kms, "ms_per_iteration_end_to_end": el / niter * 1e3,
            "softplus_values_per_s": 2.0 * E * n / (kms * 1e-3), "acceptx_mean": float(np.nanmean(o["acceptx"])),
            "bound": "vector epilogue of the density kernel (d = 4: 9 instructions per softplus value on the pipe the MFMAs share)"}


def run(api, torch, dev, which=("c2", "c4", "c5", "temper_models")):
    out = {}
    for w in which:
        try:
            out[w] = {"c2": lambda: c2(api), "c4": lambda: c4(api), "c5": lambda: c5(api, torch, dev),
                      "temper_models": lambda: temper_models(api)}[w]()
        except Exception as e:                   # a secondary number never takes the headline line down
            out[w] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out
