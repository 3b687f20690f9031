"""Small numpy restatements of the R functions the reference's test scripts call around the
sampler (glm(family=binomial), t.test()$conf.int, var), used to replay those scripts against
the oracle.  Test infrastructure only."""
import os

import numpy as np
from scipy import stats

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPS = np.finfo(float).eps


def load_logit():
    """data/logit.txt of the reference (100 rows: y x1 x2 x3 x4) -> X with intercept, y"""
    a = np.loadtxt(os.path.join(GOLDEN, "logit.txt"), skiprows=1)
    y = a[:, 0].copy()
    X = np.column_stack([np.ones(len(y)), a[:, 1:]])
    return X, y


def load_foo():
    """data/foo.txt of the reference (100 rows: x1 x2 x3 y) -> model matrix with intercept, y"""
    a = np.loadtxt(os.path.join(GOLDEN, "foo.txt"), skiprows=1)
    y = a[:, 3].copy()
    X = np.column_stack([np.ones(len(y)), a[:, :3]])
    return X, y


def temp_par_setup():
    """tests/temp-par.R:8-37: the 8 sub-models of y ~ x1 + x2 + x3, their MLEs (zero for excluded
    coefficients) and the neighbour graph (models that differ in one predictor)"""
    X, y = load_foo()
    models = np.column_stack([np.repeat([0, 1], 4), np.tile(np.repeat([0, 1], 2), 2), np.tile([0, 1], 4)])
    betas = np.zeros((8, 4))
    for i, m in enumerate(models):
        inies = np.concatenate([[True], m.astype(bool)])
        betas[i, inies] = glm_binomial(X[:, inies], y)
    neighbors = (models[:, None, :] != models[None, :, :]).sum(axis=2) == 1
    masks = np.column_stack([np.ones(8), models])
    return X, y, models, masks, betas, neighbors


def glm_binomial(X, y, epsilon=1e-8, maxit=25):
    """R's glm.fit for family = binomial(link = logit), unit weights: IRLS with R's starting
    values, link clamps and convergence rule."""
    def linkinv(eta):
        t = np.where(eta < -30, EPS, np.where(eta > 30, 1 / EPS, np.exp(eta)))
        return t / (1 + t)

    def mu_eta(eta):
        e = np.exp(eta)
        return np.where(np.abs(eta) > 30, EPS, e / (1 + e) ** 2)

    def dev(mu):
        with np.errstate(divide="ignore", invalid="ignore"):
            a = np.where(y != 0, y * np.log(y / mu), 0.0)
            b = np.where((1 - y) != 0, (1 - y) * np.log((1 - y) / (1 - mu)), 0.0)
        return float(np.sum(2 * (a + b)))

    mu = (y + 0.5) / 2
    eta = np.log(mu / (1 - mu))
    devold = dev(mu)
    coef = None
    for _ in range(maxit):
        varmu = mu * (1 - mu)
        me = mu_eta(eta)
        z = eta + (y - mu) / me
        w = np.sqrt(me * me / varmu)
        coef = np.linalg.lstsq(X * w[:, None], z * w, rcond=None)[0]
        eta = X @ coef
        mu = linkinv(eta)
        d = dev(mu)
        if abs(d - devold) / (abs(d) + 0.1) < epsilon:
            break
        devold = d
    return coef


def t_conf_int(x, level=0.95):
    x = np.asarray(x, dtype=float)
    n = x.size
    m, se = x.mean(), x.std(ddof=1) / np.sqrt(n)
    q = stats.t.ppf(0.5 + level / 2, n - 1)
    return m - q * se, m + q * se


def signif(x, digits):
    """round to `digits` significant digits like R's print"""
    x = np.asarray(x, dtype=float)
    with np.errstate(divide="ignore", invalid="ignore"):
        mag = np.where(x == 0, 0, np.floor(np.log10(np.abs(x))))
    f = 10.0 ** (digits - 1 - mag)
    return np.round(x * f) / f


def duplicated(u):
    """R's duplicated(): True for every element equal to an earlier one"""
    _, first = np.unique(u, return_index=True)
    dup = np.ones(u.size, dtype=bool)
    dup[first] = False
    return dup
