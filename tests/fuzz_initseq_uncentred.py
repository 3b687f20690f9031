"""One-off fuzz of mcmchip_initseq_uncentred (lag sums about a provisional centre, carried over to the mean on the host)
against the oracle on x - mean(x) and against mcmchip_initseq on the centred series (not collected by pytest).
usage (GPU box): python tests/fuzz_initseq_uncentred.py [nrounds] [seed]"""
import os
import sys

import numpy as np
from scipy.signal import lfilter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from mcmc_amd import api  # noqa: E402
import torch  # noqa: E402

oracle.build()
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
KEYS = ("gamma0", "Gamma.pos", "Gamma.dec", "Gamma.con", "var.pos", "var.dec", "var.con")
bad = 0
LENS = [2, 3, 4, 5, 31, 32, 33, 100, 2015, 2016, 2017, 2047, 2048, 2049, 4063, 4064, 4065, 65535, 65536, 65537, 70_001,
        262_144, 524_288 - 32, 1_048_576, 1_300_001, 3_000_000]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    n = int(LENS[it % len(LENS)]) if it < 2 * len(LENS) else int(rng.integers(2, 400_000))
    p = int(rng.choice([1, 2, 3, 9]))
    if n * p > 6_000_000:
        p = max(1, 6_000_000 // n)
    kind = it % 4
    if kind == 2 and n > 100_000:       # a trend keeps the stop rule away for ~n / 4 lags: beyond the CPU oracle's reach
        kind = 0
    if n >= 2_000_000:                  # the CPU oracle is O(n x lags): keep the stopping lag of the longest series in the hundreds
        kind = 0
    rho = rng.uniform(-0.6, 0.9 if n >= 2_000_000 else (0.99 if n >= 100_000 else 0.95), p)
    x = np.column_stack([lfilter([1.0], [1.0, -r], rng.standard_normal(n)) for r in rho])
    off = rng.choice([0.0, 1.0, -30.0, 1e3, 1e6], p)
    x = x * rng.choice([1e-3, 1.0, 40.0], p) + off
    if kind == 1:                       # a drifting start: the first rows are far from where the chain settles
        x += 25.0 * np.exp(-np.arange(n) / max(n / 50.0, 1.0))[:, None]
    if kind == 2:                       # a slow trend over the whole series
        x += np.linspace(-2.0, 2.0, n)[:, None]
    xt = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    got = api.initseq(xt)               # resident, not centred: the new entry point
    mu = x.mean(axis=0)
    xc = x - mu
    cen = api.initseq(torch.from_numpy(np.ascontiguousarray(xc.T)).cuda(), centred=True)
    worst_o = worst_c = dev80 = ora80 = 0.0
    lens_ok = True
    for c in range(p):
        ref = oracle.initseq(xc[:, c])
        if len(got[c]["Gamma.pos"]) != len(ref["Gamma.pos"]):
            lens_ok = False
            continue
        g0 = abs(ref["gamma0"]) + 1e-300
        for key in KEYS:
            sc = max(np.max(np.abs(ref[key])), g0 if key.startswith("var") else 0.0) + 1e-300     # var.* = 2 sum - gamma0 may cancel
            worst_o = max(worst_o, float(np.max(np.abs(np.asarray(got[c][key]) - np.asarray(ref[key]))) / sc))
            if len(cen[c]["Gamma.pos"]) == len(ref["Gamma.pos"]):
                worst_c = max(worst_c, float(np.max(np.abs(np.asarray(got[c][key]) - np.asarray(cen[c][key]))) / sc))
        # a few Gamma's in 80-bit arithmetic about the 80-bit mean: who is nearer, the oracle on x - fl(mean) or the device?
        xl = x[:, c].astype(np.longdouble)
        xl = xl - xl.sum() / n
        m = len(ref["Gamma.pos"])
        for i in sorted({0, m // 3, (2 * m) // 3, max(m - 2, 0)}):
            if ref["Gamma.pos"][i] == 0.0:
                continue
            k = 2 * i
            G = (np.dot(xl[: n - k], xl[k:]) + np.dot(xl[: n - k - 1], xl[k + 1:])) / n
            dev80 = max(dev80, float(abs(got[c]["Gamma.pos"][i] - G) / g0))
            ora80 = max(ora80, float(abs(ref["Gamma.pos"][i] - G) / g0))
    ok = lens_ok and (worst_o <= 1e-10 or dev80 <= max(1e-12, 2.0 * ora80))
    flag = "" if ok else "  <-- BAD"
    bad += bool(flag)
    print(f"n={n} p={p} kind={kind} max|off|/sd={np.max(np.abs(off) / x.std(axis=0)):.1e} lags={[len(g['Gamma.pos']) for g in got]} "
          f"vs oracle {worst_o:.1e} vs centred call {worst_c:.1e}  Gamma vs 80-bit / gamma0: device {dev80:.1e} oracle {ora80:.1e} lens {lens_ok}{flag}", flush=True)
print("bad", bad)
