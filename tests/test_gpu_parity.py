"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI (mcmc_amd.api -> libmcmchip.so),
against the CPU oracle on the same seeded inputs, and against the committed golden vectors.

Bars (BASELINE.json north_star): accept decisions identical under injected streams, states
bitwise, log densities and batch means <= 1e-12 relative; initseq/olbm <= 1e-10 relative.
"""
import os

import numpy as np
import pytest

from mcmc_amd import abi
from rhelpers import GOLDEN, glm_binomial, load_logit

pytestmark = pytest.mark.gpu

RTOL = 1e-12
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from mcmc_amd import api as a
    a.init(0)
    return a


@pytest.fixture(scope="module")
def logit():
    X, y = load_logit()
    return X, y, glm_binomial(X, y)


def rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


def test_native_library_is_loaded(api):
    import ctypes
    L = api.load_library()
    assert isinstance(L, ctypes.CDLL) and b"gfx950" in L.mcmchip_version()
    with open("/proc/self/maps") as f:
        assert "libmcmchip.so" in f.read()


# ----------------------------------------------------------------------------- metrop, stream path
def test_metrop_golden_vectors(api, logit):
    X, y, _ = logit
    g = np.load(os.path.join(GOLDEN, "metrop_logit.npz"))
    tg = api.logistic_target(X, y)
    o = api.metrop(tg, g["init"], int(g["nbatch"]), int(g["blen"]), int(g["nspac"]), scale=float(g["scale"]),
                   z=g["z"], u=g["u"], debug=True)
    assert np.array_equal(o["debug.accept"], g["debug_accept"] != 0)
    assert np.array_equal(o["final"], g["final"])
    assert np.array_equal(o["accept"], g["accept"]) and np.array_equal(o["accept.batch"], g["accept_batch"])
    assert rel(o["batch"], g["batch"]) < RTOL
    assert np.array_equal(np.isnan(o["u"]), np.isnan(g["u_trace"]))
    fin = np.isfinite(g["log_green"])
    assert np.max(np.abs(o["log.green"][fin] - g["log_green"][fin])) < 1e-11


def test_metrop_c2_4096_chains_injected(api, oracle_mod, logit):
    # BASELINE config 2: logit posterior, 4096 chains, injected streams, niter = 1000
    X, y, mle = logit
    C, nbatch, blen, nspac, d = 4096, 100, 5, 2, 5
    niter = nbatch * blen * nspac
    init = mle + 0.05 * np.random.default_rng(42).standard_normal((C, d))
    rng = np.random.default_rng(7)
    z = rng.standard_normal((C, niter, d))
    u = rng.random((C, niter))
    ref = oracle_mod.metrop(oracle_mod.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.125), init, nbatch, blen,
                            nspac, scale=0.4, z=z, u=u, nthreads=16)
    o = api.metrop(api.logistic_target(X, y), init, nbatch, blen, nspac, scale=0.4, z=z, u=u)
    assert o["path"] == "stream"
    # identical accept decisions <=> identical accept counts per batch and bitwise equal states
    assert np.array_equal(o["accept.batch"], ref["accept_batch"])
    assert np.array_equal(o["accept"], ref["accept"])
    assert np.array_equal(o["final"], ref["final"])
    assert rel(o["batch"], ref["batch"]) < RTOL
    assert 0.15 < o["accept"].mean() < 0.35


@pytest.mark.parametrize("scale_kind", ["scalar", "vector", "matrix"])
@pytest.mark.parametrize("outfun", [None, [2, 3], [-2], [True, False, True, False, True], "moments2"])
def test_metrop_trace_scale_and_outfun(api, oracle_mod, logit, scale_kind, outfun):
    # tests/logitvec.R, logitmat.R, logitidx.R, logitnegidx.R, logitlogidx.R, logitfun.R analogues
    X, y, mle = logit
    d, C, nbatch, blen, nspac = 5, 16, 10, 5, 3
    niter = nbatch * blen * nspac
    rng = np.random.default_rng(3)
    init = mle + 0.1 * rng.standard_normal((C, d))
    z, u = rng.standard_normal((C, niter, d)), rng.random((C, niter))
    scale = {"scalar": 0.4, "vector": np.array([0.5, 0.4, 0.3, 0.35, 0.45]),
             "matrix": 0.9 * np.linalg.cholesky(np.linalg.inv(X.T @ X / 4 + np.eye(d) / 4))}[scale_kind]
    ref = oracle_mod.metrop(oracle_mod.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.125), init, nbatch, blen,
                            nspac, scale=scale, z=z, u=u, outfun=outfun, debug=True)
    o = api.metrop(api.logistic_target(X, y), init, nbatch, blen, nspac, scale=scale, z=z, u=u, outfun=outfun,
                   debug=True)
    assert np.array_equal(o["debug.accept"], ref["debug_accept"] != 0)
    assert np.array_equal(o["current"], ref["current"]) and np.array_equal(o["proposal"], ref["proposal"])
    assert np.array_equal(o["z"], ref["z"])
    assert np.array_equal(np.isnan(o["u"]), np.isnan(ref["u"]))
    assert np.array_equal(o["u"][~np.isnan(o["u"])], ref["u"][~np.isnan(ref["u"])])
    assert o["batch"].shape == ref["batch"].shape and rel(o["batch"], ref["batch"]) < RTOL
    assert np.array_equal(o["final"], ref["final"])
    # reference identities: batch == path when blen = nspac = 1 is covered below; accept bookkeeping here
    assert np.allclose(o["accept"], o["accept.batch"].mean(axis=1))


@pytest.mark.parametrize("kind", ["simplex", "ball", "mvnorm", "normmix", "naive", "binomial", "student"])
def test_metrop_other_targets(api, oracle_mod, logit, kind):
    rng = np.random.default_rng(5)
    d = 5
    kw, init, scale = {}, np.zeros((32, d)), 0.1
    if kind == "simplex":
        K, kw = abi.T_SIMPLEX, dict(par=[1.0])           # tests/accept-batch.R
    elif kind == "ball":
        K, scale = abi.T_BALL, 0.4                        # tests/circle.R: -Inf proposals draw no uniform
    elif kind == "mvnorm":
        A = rng.standard_normal((d, d))
        P = A @ A.T + d * np.eye(d)
        K, kw, scale = abi.T_MVNORM, dict(par=np.concatenate([rng.standard_normal(d), P.ravel()])), 0.5
        init = rng.standard_normal((32, d))
    elif kind == "normmix":
        par = abi.normmix_par([0.5, 0.3, 0.2], rng.standard_normal((3, d)) * 2, [1.0, 0.5, 2.0])
        K, kw, scale = abi.T_NORMMIX, dict(par=par), 1.0
        init = rng.standard_normal((32, d))
    elif kind == "binomial":
        # vignettes/morph.Rnw:361-400: 12 gender x department cells, counts + nu = 5 prior students at xi = 0.3
        M = np.column_stack([np.ones(12), rng.integers(0, 2, 12), rng.integers(0, 2, (12, 3))]).astype(float)
        tot = rng.integers(20, 900, 12).astype(float)
        K, kw, scale = abi.T_BINOMIAL_LOGIT, dict(x=M, y=np.floor(tot * rng.random(12)) + 1.5, par=tot + 5.0), 0.1
        init = 0.1 * rng.standard_normal((32, d))
    elif kind == "student":
        K, kw, scale = abi.T_STUDENT_T, dict(par=[3.0]), 1.5                    # tests/morph.metrop.R:9
        init = rng.standard_normal((32, d))
    else:
        X, y, mle = logit
        K, kw, scale = abi.T_LOGISTIC_NAIVE, dict(x=X, y=y, prior_c=0.0), 0.3   # tests/logit.R:40-48
        init = mle + 0.05 * rng.standard_normal((32, d))
    nbatch, blen, nspac = 30, 4, 2
    niter = nbatch * blen * nspac
    z, u = rng.standard_normal((32, niter, d)), rng.random((32, niter))
    ref = oracle_mod.metrop(oracle_mod.Target(K, d, **kw), init, nbatch, blen, nspac, scale=scale, z=z, u=u, debug=True)
    o = api.metrop(api.Target(K, d, **kw), init, nbatch, blen, nspac, scale=scale, z=z, u=u, debug=True)
    assert np.array_equal(o["debug.accept"], ref["debug_accept"] != 0)
    assert np.array_equal(np.isnan(o["u"]), np.isnan(ref["u"]))
    assert np.array_equal(o["final"], ref["final"])
    assert rel(o["batch"] + 1.0, ref["batch"] + 1.0) < RTOL
    lg_o, lg_r = o["log.green"], ref["log_green"]
    assert np.array_equal(np.isneginf(lg_o), np.isneginf(lg_r))
    f = np.isfinite(lg_r)
    assert np.max(np.abs(lg_o[f] - lg_r[f])) < 1e-11


def test_metrop_single_chain_shapes_and_identities(api, logit):
    # tests/logit.R:67-69,102: with blen = nspac = 1 batch is the path; result shapes as in R
    X, y, mle = logit
    o = api.metrop(api.logistic_target(X, y), mle, 200, scale=0.4, seed=99, debug=True)
    assert o["batch"].shape == (200, 5) and o["current"].shape == (200, 5) and np.ndim(o["accept"]) == 0
    assert np.all(o["batch"][:-1] == o["current"][1:]) and np.all(o["current"][0] == o["initial"])
    assert np.all(o["batch"][-1] == o["final"])
    path = np.where(o["debug.accept"][:, None], o["proposal"], o["current"])
    assert np.all(path == o["batch"])
    assert np.max(np.abs(o["proposal"] - (o["current"] + 0.4 * o["z"]))) < 1e-15
    assert list(o.keys())[:11] == ["accept", "batch", "initial", "final", "accept.batch", "current", "proposal",
                                   "log.green", "u", "z", "debug.accept"]     # metrop.c:124-128,174-179


def test_metrop_philox_trajectory_replayed_by_oracle(api, oracle_mod, logit):
    # device RNG: the z/u the device drew, fed to the oracle as injected streams, give the same chain;
    # and the device's Philox/Box-Muller agrees with the oracle's restatement of it
    X, y, mle = logit
    C, nbatch, blen, nspac = 64, 25, 2, 2
    init = mle + 0.05 * np.random.default_rng(1).standard_normal((C, 5))
    o = api.metrop(api.logistic_target(X, y), init, nbatch, blen, nspac, scale=0.4, seed=2024, debug=True,
                   chain_offset=1000)
    u_fill = np.where(np.isnan(o["u"]), 0.5, o["u"])
    ref = oracle_mod.metrop(oracle_mod.Target(abi.T_LOGISTIC, 5, x=X, y=y, prior_c=0.125), init, nbatch, blen, nspac,
                            scale=0.4, z=o["z"], u=u_fill, debug=True)
    assert np.array_equal(o["debug.accept"], ref["debug_accept"] != 0)
    assert np.array_equal(o["final"], ref["final"])
    ref2 = oracle_mod.metrop(oracle_mod.Target(abi.T_LOGISTIC, 5, x=X, y=y, prior_c=0.125), init, nbatch, blen, nspac,
                             scale=0.4, seed=2024, chain_offset=1000, debug=True)
    assert np.max(np.abs(o["z"] - ref2["z"])) < 1e-12
    assert np.array_equal(o["debug.accept"], ref2["debug_accept"] != 0)
    zz = o["z"].ravel()
    assert abs(zz.mean()) < 0.02 and abs(zz.std() - 1) < 0.02


def test_metrop_restart_equals_long_run(api, logit):
    # tests/saveseed.R:12-17: rbind(metrop(out)$batch, metrop(metrop(out))$batch) identical to
    # metrop(out, nbatch = 2 * out$nbatch)$batch
    X, y, mle = logit
    tg = api.logistic_target(X, y)
    init = mle + 0.05 * np.random.default_rng(8).standard_normal((32, 5))
    out = api.metrop(tg, init, 50, blen=3, scale=0.4, seed=5)
    a = api.metrop(out)
    b = api.metrop(a)
    long = api.metrop(out, nbatch=100)
    assert np.array_equal(np.concatenate([a["batch"], b["batch"]], axis=1), long["batch"])
    assert np.array_equal(b["final"], long["final"])
    # sharding independence: chains 16.. run as their own call with chain_offset give the same numbers
    part = api.metrop(tg, init[16:], 50, blen=3, scale=0.4, seed=5, chain_offset=16)
    assert np.array_equal(part["batch"], out["batch"][16:])


def test_metrop_errors(api, logit):
    X, y, mle = logit
    with pytest.raises(api.McmcHipError, match="log unnormalized density -Inf at initial state"):
        api.metrop(api.Target(abi.T_SIMPLEX, 3, par=[0.0]), np.array([0.5, 0.6, 0.1]), 10, seed=1)
    tg = api.logistic_target(X, y)
    with pytest.raises(api.McmcHipError, match='argument "nbatch" must be positive'):
        api.metrop(tg, mle, 0, seed=1)
    with pytest.raises(api.McmcHipError, match='all elements of "state" must be finite'):
        api.metrop(tg, np.array([0, 0, np.inf, 0, 0.0]), 5, seed=1)
    with pytest.raises(ValueError, match='length of "scale" vector not d or 1'):
        api.metrop(tg, mle, 5, scale=[1.0, 2.0], seed=1)
    with pytest.raises(ValueError, match="any\\(outfun == 0\\)"):
        api.metrop(tg, mle, 5, outfun=[0, 1], seed=1)


# ----------------------------------------------------------------------------- metrop, MFMA path
def _synthetic(n, d, seed=42):
    rng = np.random.default_rng(seed)
    X = np.empty((n, d))
    X[:, 0] = 1.0
    X[:, 1:] = rng.standard_normal((n, d - 1)) / np.sqrt(d)
    beta = 0.5 * rng.standard_normal(d)
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    return X, y, beta


@pytest.mark.parametrize("n,d,C", [(20000, 64, 256), (5003, 5, 100), (8192, 20, 300), (4099, 33, 64),
                                   (6007, 65, 70), (4099, 100, 64), (8192, 128, 300), (33, 96, 5),    # d > 64: one tile per chunk
                                   (5000, 32, 64), (3000, 48, 70), (3000, 56, 70), (7, 100, 3), (17, 70, 3),
                                   (300000, 64, 40),                 # several units per workgroup
                                   (4099, 129, 40), (8192, 192, 130), (6007, 200, 64), (8192, 256, 300), (9, 250, 3),
                                   (60000, 256, 129)])               # d > 128: ring of fragment registers, 16 chains a wave
def test_logdens_mfma_vs_oracle(api, oracle_mod, n, d, C):
    X, y, beta = _synthetic(n, d, seed=n)
    states = beta + 0.3 * np.random.default_rng(1).standard_normal((C, d))
    states[0] *= 40.0           # |eta| large: exercises the tails of the softplus
    tg = api.logistic_target(X, y)
    ref = oracle_mod.logdens(oracle_mod.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.125), states)
    got_m = api.logdens(tg, states, path=abi.PATH_MFMA)
    got_s = api.logdens(tg, states, path=abi.PATH_STREAM)
    assert rel(got_m, ref) < RTOL and rel(got_s, ref) < RTOL


@pytest.mark.parametrize("n,d,C", [(100_000, 5, 300), (120_001, 64, 130), (5003, 20, 64), (40_000, 128, 70), (30_000, 200, 40)])
def test_logdens_mfma_binomial_and_models_vs_oracle(api, oracle_mod, n, d, C):
    """The two other kinds with a design matrix on the matrix cores (same contraction, another epilogue): the binomial
    regression of vignettes/morph.Rnw:361-365, whose totals weight the softplus terms (one more operand of the running
    sum's fma; the linear part goes through the column sums), and the model-selection posterior of tests/temp-par.R:39-56,
    whose masks are folded into the states.  MFMA and stream kernels against the oracle, 1e-12."""
    rng = np.random.default_rng(n + d)
    X, y, beta = _synthetic(n, d, seed=n)
    # binomial regression: posterior totals w_l > 0 (not integers: the conjugate prior adds to them), successes in (0, w_l)
    tot = rng.integers(1, 30, n) + 0.5 + rng.random(n)
    succ = tot * (0.05 + 0.9 * rng.random(n))
    states = beta + 0.3 * rng.standard_normal((C, d))
    states[0] *= 25.0
    kw = dict(x=X, y=succ, par=tot)
    ref = oracle_mod.logdens(oracle_mod.Target(abi.T_BINOMIAL_LOGIT, d, **kw), states)
    tg = api.Target(abi.T_BINOMIAL_LOGIT, d, **kw)
    got_m = api.logdens(tg, states, path=abi.PATH_MFMA)
    got_s = api.logdens(tg, states, path=abi.PATH_STREAM)
    assert rel(got_m, ref) < RTOL and rel(got_s, ref) < RTOL
    # model selection: k random masks (intercept always in), log pseudo-priors, a constant
    k = 6
    masks = (rng.random((k, d)) < 0.6).astype(float)
    masks[:, 0] = 1.0
    masks[0] = 1.0
    par = abi.logistic_models_par(masks, log_pseudo_prior=rng.standard_normal(k), constant=-1.25)
    kw = dict(x=X, y=y, prior_c=0.125, par=par)
    st = np.empty((C, d + 1))
    st[:, 0] = rng.integers(1, k + 1, C)
    st[:, 1:] = states
    ref = oracle_mod.logdens(oracle_mod.Target(abi.T_LOGISTIC_MODELS, d, **kw), st)
    tg = api.Target(abi.T_LOGISTIC_MODELS, d, **kw)
    got_m = api.logdens(tg, st, path=abi.PATH_MFMA)
    got_s = api.logdens(tg, st, path=abi.PATH_STREAM)
    assert rel(got_m, ref) < RTOL and rel(got_s, ref) < RTOL


def test_metrop_mfma_binomial_injected_vs_oracle(api, oracle_mod):
    """metrop on the binomial regression through the MFMA path with injected draws: the oracle's decisions and finals."""
    n, d, C, nbatch, blen, nspac = 20_000, 12, 128, 5, 3, 2
    niter = nbatch * blen * nspac
    rng = np.random.default_rng(5)
    X, y, beta = _synthetic(n, d, seed=3)
    tot = rng.integers(1, 30, n) + 0.5
    succ = tot / (1 + np.exp(-X @ beta))
    kw = dict(x=X, y=succ, par=tot)
    init = beta + 0.003 * rng.standard_normal((C, d))
    z, u = rng.standard_normal((C, niter, d)), rng.random((C, niter))
    ref = oracle_mod.metrop(oracle_mod.Target(abi.T_BINOMIAL_LOGIT, d, **kw), init, nbatch, blen, nspac, scale=0.004, z=z, u=u,
                            nthreads=16)
    o = api.metrop(api.Target(abi.T_BINOMIAL_LOGIT, d, **kw), init, nbatch, blen, nspac, scale=0.004, z=z, u=u, path=abi.PATH_MFMA)
    assert o["path"] == "mfma"
    assert np.array_equal(o["accept.batch"], ref["accept_batch"])
    assert np.array_equal(o["final"], ref["final"])
    assert rel(o["batch"], ref["batch"]) < RTOL
    assert 0.05 < o["accept"].mean() < 0.95
    auto = api.metrop(api.Target(abi.T_BINOMIAL_LOGIT, d, **kw), init, nbatch, blen, nspac, scale=0.004, z=z, u=u)
    assert auto["path"] == "mfma" and np.array_equal(auto["final"], o["final"])      # n d large: AUTO takes the matrix cores


@pytest.mark.parametrize("scale_kind", ["scalar", "matrix"])
@pytest.mark.parametrize("d", [16, 80, 200])
def test_metrop_mfma_injected_vs_oracle(api, oracle_mod, scale_kind, d):
    n, C, nbatch, blen, nspac = 8192, 128, 6, 3, 2
    niter = nbatch * blen * nspac
    X, y, beta = _synthetic(n, d)
    rng = np.random.default_rng(2)
    init = beta + 0.01 * rng.standard_normal((C, d))
    z, u = rng.standard_normal((C, niter, d)), rng.random((C, niter))
    s0 = {16: 0.02, 80: 0.008, 200: 0.004}[d]
    scale = s0 if scale_kind == "scalar" else s0 * np.linalg.cholesky(np.eye(d) + 0.3)
    ref = oracle_mod.metrop(oracle_mod.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.125), init, nbatch, blen, nspac,
                            scale=scale, z=z, u=u, outfun="moments2", nthreads=16)
    o = api.metrop(api.logistic_target(X, y), init, nbatch, blen, nspac, scale=scale, z=z, u=u, outfun="moments2",
                   path=abi.PATH_MFMA)
    assert o["path"] == "mfma"
    assert np.array_equal(o["accept.batch"], ref["accept_batch"])
    assert np.array_equal(o["final"], ref["final"])
    assert rel(o["batch"], ref["batch"]) < RTOL
    assert 0.05 < o["accept"].mean() < 0.95
    s = api.metrop(api.logistic_target(X, y), init, nbatch, blen, nspac, scale=scale, z=z, u=u, outfun="moments2",
                   path=abi.PATH_STREAM)
    assert np.array_equal(s["final"], o["final"])


def test_metrop_mfma_philox_restart_and_sharding(api):
    n, d, C = 8192, 64, 256
    X, y, beta = _synthetic(n, d)
    tg = api.logistic_target(X, y)
    init = beta + 1e-3 * np.random.default_rng(3).standard_normal((C, d))
    out = api.metrop(tg, init, 4, blen=2, scale=0.01, seed=77)
    assert out["path"] == "mfma"
    both = api.metrop(out, nbatch=8)
    a = api.metrop(out)
    b = api.metrop(a)
    assert np.array_equal(np.concatenate([a["batch"], b["batch"]], axis=1), both["batch"])
    part = api.metrop(tg, init[128:], 4, blen=2, scale=0.01, seed=77, chain_offset=128)
    assert np.array_equal(part["batch"], out["batch"][128:])


def test_full_size_c3_logdens_properties(api, oracle_mod):
    # BASELINE config 3 data shape (n = 1e6, d = 64): MFMA path against the independent stream
    # kernel on all chains, against the oracle on a few
    n, d, C = 1_000_000, 64, 256
    X, y, beta = _synthetic(n, d)
    tg = api.logistic_target(X, y)
    states = beta + 1e-3 * np.random.default_rng(4).standard_normal((C, d))
    m = api.logdens(tg, states, path=abi.PATH_MFMA)
    s = api.logdens(tg, states, path=abi.PATH_STREAM)
    assert rel(m, s) < RTOL
    ref = oracle_mod.logdens(oracle_mod.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.125), states[:4])
    assert rel(m[:4], ref) < RTOL
    # permutation invariance of the observations (a checksum-of-checksums style property)
    perm = np.random.default_rng(5).permutation(n)
    m2 = api.logdens(api.logistic_target(X[perm], y[perm]), states[:64], path=abi.PATH_MFMA)
    assert rel(m2, m[:64]) < 1e-11


# ----------------------------------------------------------------------------- temper
def test_temper_golden_vectors(api):
    g = np.load(os.path.join(GOLDEN, "temper_witch.npz"))
    tg = api.Target(abi.T_WITCH, 3, par=g["witch"])
    E = g["unif"].shape[0]
    args = (int(g["nbatch"]), int(g["blen"]), int(g["nspac"]), float(g["scale"]))
    o = api.temper(tg, np.zeros((E, 6, 3)), g["neighbors"], *args, parallel=True, unif=g["unif"], norm=g["norm"])
    assert np.array_equal(o["final"], g["par_final"])
    assert np.array_equal(o["acceptx"], g["par_acceptx"], equal_nan=True)
    assert np.array_equal(o["accepti"], g["par_accepti"], equal_nan=True)
    assert rel(o["batch"] + 1, g["par_batch"] + 1) < RTOL
    init_s = np.tile(np.array([1.0, 0, 0, 0]), (E, 1))
    o = api.temper(tg, init_s, g["neighbors"], *args, parallel=False, unif=g["unif"], norm=g["norm"])
    assert np.array_equal(o["final"], g["ser_final"])
    assert np.array_equal(o["acceptx"], g["ser_acceptx"], equal_nan=True)
    assert np.array_equal(o["accepti"], g["ser_accepti"], equal_nan=True)
    assert np.array_equal(o["ibatch"], g["ser_ibatch"])
    assert rel(o["batch"] + 1, g["ser_batch"] + 1) < RTOL


@pytest.mark.parametrize("parallel", [True, False])
@pytest.mark.parametrize("target", ["witch", "mixture"])
@pytest.mark.parametrize("scale_kind", ["scalar", "list_vector", "matrix"])
def test_temper_vs_oracle(api, oracle_mod, parallel, target, scale_kind):
    rng = np.random.default_rng(21)
    k, E, nbatch, blen, nspac = 8, 48, 12, 5, 3
    niter = nbatch * blen * nspac
    if target == "witch":
        p = 3
        par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / p)
        K = abi.T_WITCH
        x0 = np.zeros(p)
    else:
        p = 2
        betas = np.geomspace(1.0, 0.05, k)
        par = np.concatenate([[float(k)], betas, abi.normmix_par([0.5, 0.5], [[-4.0, -4.0], [4.0, 4.0]], 1.0)])
        K = abi.T_TEMPERED_NORMMIX          # BASELINE config 4 (ii): bimodal normal mixture
        x0 = np.array([4.0, 4.0])
    i, j = np.indices((k, k))
    nb = np.abs(i - j) == 1
    scale = {"scalar": 0.4, "list_vector": [np.full(p, 0.2 + 0.1 * c) for c in range(k)],
             "matrix": 0.3 * np.linalg.cholesky(np.eye(p) + 0.5)}[scale_kind]
    unif, norm = rng.random((E, niter, 4)), rng.standard_normal((E, niter, p))
    if parallel:
        init = np.tile(x0, (E, k, 1))
    else:
        init = np.column_stack([rng.integers(1, k + 1, E).astype(float), np.tile(x0, (E, 1))])
    ref = oracle_mod.temper(oracle_mod.Target(K, p, par=par), init, nb, nbatch, blen, nspac, scale, parallel=parallel,
                            unif=unif, norm=norm)
    o = api.temper(api.Target(K, p, par=par), init, nb, nbatch, blen, nspac, scale, parallel=parallel,
                   unif=unif, norm=norm)
    assert np.array_equal(o["final"], ref["final"])
    assert np.array_equal(o["acceptx"], ref["acceptx"], equal_nan=True)
    assert np.array_equal(o["accepti"], ref["accepti"], equal_nan=True)
    assert rel(o["batch"] + 10, ref["batch"] + 10) < RTOL
    if not parallel:
        assert np.array_equal(o["ibatch"], ref["ibatch"])
    assert np.all(np.isnan(o["accepti"][0])[~nb])              # NA for non-neighbours (temper.c:896-897)


# (the Philox-mode comparison with the oracle, decision by decision, lives in tests/test_gpu_round2.py)


@pytest.mark.parametrize("target,p", [("witch", 1), ("witch", 3), ("witch", 4), ("mixture", 2), ("mixture", 3)])
@pytest.mark.parametrize("k,nbrs", [(64, 1), (5, 2), (37, 4)])
@pytest.mark.parametrize("scale_kind", ["scalar", "list_scalar", "list_vector", "matrix", "list_matrix"])
def test_temper_run_kernel_equals_lds_kernel(api, oracle_mod, target, p, k, nbrs, scale_kind):
    """the run-parallel kernel (MCMCHIP_PATH_RUN) and the LDS kernel give the same bits, with injected streams and with device Philox, identity and corner-indicator output,
    thinned and unthinned batches; spot check against the oracle"""
    rng = np.random.default_rng(k * 100 + p)
    E = 9
    if target == "witch":
        par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / (8.0 * p))
        K, x0 = abi.T_WITCH, np.zeros(p)
    else:
        betas = np.geomspace(1.0, 0.05, k)
        par = np.concatenate([[float(k)], betas, abi.normmix_par([0.3, 0.7], [np.full(p, -3.0), np.full(p, 3.0)], 1.0)])
        K, x0 = abi.T_TEMPERED_NORMMIX, np.full(p, 3.0)
    i, j = np.indices((k, k))
    nb = (np.abs(i - j) >= 1) & (np.abs(i - j) <= nbrs)
    scale = {"scalar": 0.4, "list_scalar": [0.2 + 0.01 * c for c in range(k)],
             "list_vector": [np.full(p, 0.2 + 0.01 * c) + 0.05 * np.arange(p) for c in range(k)],
             "matrix": 0.3 * np.linalg.cholesky(np.eye(p) + 0.5),
             "list_matrix": [(0.2 + 0.01 * c) * np.linalg.cholesky(np.eye(p) + 0.5) for c in range(k)]}[scale_kind]
    init = np.tile(x0, (E, k, 1)) + 0.01 * rng.standard_normal((E, k, p))
    tg = api.Target(K, p, par=par)
    outs = [None, "witch.corner"] if target == "witch" else [None]
    for nbatch, blen, nspac in ((7, 6, 3), (150, 1, 1), (3, 43, 1), (2, 1, 70)):
        niter = nbatch * blen * nspac
        unif, norm = rng.random((E, niter, 4)), rng.standard_normal((E, niter, p))
        for outfun in outs:
            for kw in (dict(unif=unif, norm=norm), dict(seed=77, ens_offset=3)):
                a = api.temper(tg, init, nb, nbatch, blen, nspac, scale, parallel=True, outfun=outfun, _path=abi.PATH_LDS, **kw)
                for path in (abi.PATH_RUN,):
                    b = api.temper(tg, init, nb, nbatch, blen, nspac, scale, parallel=True, outfun=outfun, _path=path, **kw)
                    for name in ("final", "batch", "acceptx", "accepti"):
                        assert np.array_equal(a[name], b[name], equal_nan=True), (name, outfun, list(kw), path, nbatch)
    ref = oracle_mod.temper(oracle_mod.Target(K, p, par=par), init, nb, nbatch, blen, nspac, scale, parallel=True,
                            unif=unif, norm=norm)
    for path in (abi.PATH_RUN,):
        o = api.temper(tg, init, nb, nbatch, blen, nspac, scale, parallel=True, unif=unif, norm=norm, _path=path)
        assert np.array_equal(o["final"], ref["final"])
        assert np.array_equal(o["acceptx"], ref["acceptx"], equal_nan=True)
        assert np.array_equal(o["accepti"], ref["accepti"], equal_nan=True)
        assert rel(o["batch"] + 10, ref["batch"] + 10) < RTOL


def test_temper_run_kernel_refuses_other_shapes(api):
    k, p = 4, 3
    par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / p)
    i, j = np.indices((k, k))
    nb = np.abs(i - j) == 1
    tg = api.Target(abi.T_WITCH, p, par=par)
    with pytest.raises(Exception, match="run-parallel tempering path"):
        api.temper(tg, np.array([1.0, 0, 0, 0]), nb, 5, 1, 1, 0.3, parallel=False, seed=1, _path=abi.PATH_RUN)
    with pytest.raises(Exception, match="run-parallel tempering path"):
        api.temper(tg, np.zeros((k, p)), nb, 5, 1, 1, 0.3, parallel=True, debug=True, seed=1, _path=abi.PATH_RUN)


# ----------------------------------------------------------------------------- initseq / olbm
def test_diag_golden_vectors(api):
    g = np.load(os.path.join(GOLDEN, "diag.npz"))
    x = g["x"].astype(np.float64)
    res = api.initseq(x)
    for c in range(3):
        for key in ("gamma0", "Gamma.pos", "Gamma.dec", "Gamma.con", "var.pos", "var.dec", "var.con"):
            want = g["c%d_%s" % (c, key.replace(".", "_"))]
            got = np.asarray(res[c][key])
            assert got.shape == want.shape, key
            assert np.max(np.abs(got - want)) <= 1e-10 * np.max(np.abs(want)) + 1e-300, key
    var = api.olbm(x, int(g["olbm_len"]))
    assert rel(var, g["olbm_var"]) < 1e-10


def test_initseq_ar1_identities_and_oracle(api, oracle_mod):
    # tests/initseq.R:6-48 at its size (n = 1e5, rho = 0.99)
    rng = np.random.default_rng(42)
    n, rho = 100000, 0.99
    e = rng.standard_normal(n)
    x = np.empty(n)
    x[0] = e[0] / np.sqrt(1 - rho * rho)
    for t in range(1, n):
        x[t] = rho * x[t - 1] + e[t]
    out = api.initseq(x)
    ref = oracle_mod.initseq(x - x.mean())
    assert len(out["Gamma.pos"]) == len(ref["Gamma.pos"])
    for key in ("gamma0", "Gamma.pos", "Gamma.dec", "Gamma.con", "var.pos", "var.dec", "var.con"):
        assert np.max(np.abs(np.asarray(out[key]) - np.asarray(ref[key]))) <= 1e-10 * np.max(np.abs(ref[key])), key
    assert np.array_equal(out["Gamma.dec"], np.minimum.accumulate(out["Gamma.pos"]))
    assert out["Gamma.pos"][-1] == 0 and out["Gamma.dec"][-1] == 0 and abs(out["Gamma.con"][-1]) < 1e-8
    assert np.min(np.diff(np.diff(out["Gamma.con"]))) > -np.sqrt(np.finfo(float).eps)
    for s in ("pos", "dec", "con"):
        assert np.isclose(2 * out["Gamma." + s].sum() - out["gamma0"], out["var." + s], rtol=1e-12)


def test_initseq_edge_cases(api, oracle_mod):
    rng = np.random.default_rng(1)
    for n in (2, 3, 17, 64, 2049, 4099):
        x = rng.standard_normal(n)
        out, ref = api.initseq(x), oracle_mod.initseq(x - x.mean())
        assert len(out["Gamma.pos"]) == len(ref["Gamma.pos"]), n
        assert np.allclose(out["Gamma.con"], ref["Gamma.con"], rtol=1e-10, atol=1e-13)
        assert np.isclose(out["var.con"], ref["var.con"], rtol=1e-10, atol=1e-13)
    # never-negative case: a linear trend keeps Gamma positive up to len/2 pairs
    x = np.arange(40.0)
    out, ref = api.initseq(x), oracle_mod.initseq(x - x.mean())
    assert len(out["Gamma.pos"]) == len(ref["Gamma.pos"])
    with pytest.raises(ValueError):
        api.initseq(np.array([1.0, np.nan, 2.0]))


@pytest.mark.parametrize("n,p,blen", [(20000, 3, 25), (5000, 40, 500), (1000, 1, 1), (300, 17, 300), (70000, 16, 1000),
                                       (9000, 100, 77), (6000, 150, 64), (5000, 200, 33), (4100, 256, 100),
                                       (2500, 300, 50), (37, 65, 5)])
def test_olbm_vs_oracle(api, oracle_mod, n, p, blen):
    rng = np.random.default_rng(n + p)
    e = rng.standard_normal((n, p)) + 3.0
    x = np.cumsum(e * 0.1, axis=0) * 0.01 + e     # mildly autocorrelated, non-zero mean
    ref, refmean = oracle_mod.olbm(x, blen)
    got = api.olbm(x, blen)
    tol = 1e-10 * np.max(np.abs(ref)) + 1e-24      # len == n: a single window, var is exactly 0 in the reference
    assert np.max(np.abs(got - ref)) <= tol
    assert np.array_equal(got, got.T)
    got2 = api.olbm(x, blen, demean=False)        # R/olbm.R quirk: still demeans
    assert np.max(np.abs(got2 - ref)) <= tol
    with pytest.raises(ValueError, match="batch.length must be <= nrow"):
        api.olbm(x, n + 1)


# ----------------------------------------------------------------------------- temper debug trace
def _lud_witch(par, p, comp, theta):
    if np.any(np.abs(theta) > 1.0):
        return -np.inf
    bnd = par[int(comp) - 1]
    if np.all(np.abs(theta) > bnd):
        return -(p + 1) * np.log(2) - p * np.log(1 - bnd)
    return -(p + 1) * np.log(2) - np.log1p(-(1 - bnd) ** p)


_WITCH = _lud_witch


def _lud_models(X, y, masks, comp, theta):
    # tests/temp-par.R:39-56
    eta = X @ (theta * masks[int(comp) - 1])
    logp = np.where(eta < 0, eta - np.log1p(np.exp(eta)), -np.log1p(np.exp(-eta)))
    logq = np.where(eta < 0, -np.log1p(np.exp(eta)), -eta - np.log1p(np.exp(-eta)))
    return logp[y == 1].sum() + logq[y == 0].sum() - np.sum(theta ** 2) / 2


@pytest.mark.parametrize("parallel", [True, False])
@pytest.mark.parametrize("target", ["witch", "models"])
def test_temper_debug_trace_checks(api, oracle_mod, parallel, target):
    """The 14 checks of vignettes/debug.Rnw:250-267 as run by tests/temp-par.R:68-218 and
    tests/temp-ser.R:53-291, on the DEVICE trace, plus bitwise equality with the oracle's trace.
    "witch": the witch's hat of temp-par-witch.R; "models": the very configuration of temp-par.R / temp-ser.R
    (data foo, 8 sub-models, glm starting points, scale 0.56789) for 8 ensembles."""
    E = 8
    nbatch, blen, nspac = 20, 10, 5
    if target == "witch":
        k, p, scale = 6, 3, 0.3456789
        par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / p)
        i, j = np.indices((k, k))
        nb = (np.abs(i - j) == 1) | (np.abs(i - j) == 2)
        kw = dict(par=par)
        K = abi.T_WITCH
        init = np.zeros((E, k, p)) if parallel else np.column_stack([np.ones(E), np.zeros((E, p))])
        _lud_witch_ = lambda comp, th: _WITCH(par, p, comp, th)
        ltol = 1e-13
    else:
        from rhelpers import temp_par_setup
        X, y, models, masks, betas, nb = temp_par_setup()
        k, p, scale = 8, 4, 0.56789
        kw = dict(x=X, y=y, prior_c=0.5, par=abi.logistic_models_par(masks))
        K = abi.T_LOGISTIC_MODELS
        # temp-par.R:63: initial = betas; temp-ser.R: initial = c(nrow(models), betas of the full model)
        init = np.tile(betas, (E, 1, 1)) if parallel else np.tile(np.concatenate([[8.0], betas[7]]), (E, 1))
        _lud_witch_ = lambda comp, th: _lud_models(X, y, masks, comp, th)
        ltol = 1e-10
    niter = nbatch * blen * nspac
    rng = np.random.default_rng(77)
    unif, norm = rng.random((E, niter, 4)), rng.standard_normal((E, niter, p))
    o = api.temper(api.Target(K, p, **kw), init, nb, nbatch, blen, nspac, scale, parallel=parallel,
                   unif=unif, norm=norm, debug=True)
    ref = oracle_mod.temper(oracle_mod.Target(K, p, **kw), init, nb, nbatch, blen, nspac, scale,
                            parallel=parallel, unif=unif, norm=norm, debug=True)
    if target == "models":
        # data-carrying density: device exp/log1p differ from glibc's in the last bits
        assert np.array_equal(o["acceptd"], ref["acceptd"]) and np.array_equal(o["state"], ref["state"])
        assert np.allclose(o["log.hastings"], ref["log.hastings"], rtol=0, atol=1e-11, equal_nan=True)
        ref = dict(ref)
        ref["log.hastings"] = o["log.hastings"]
    keys = ["which", "unif.which", "state", "log.hastings", "unif.hastings", "proposal", "acceptd", "norm",
            "unif.choose"] + (["coproposal"] if parallel else [])
    for key in keys:
        assert np.array_equal(o[key], ref[key], equal_nan=True), key
    assert list(o.keys())[-len(keys):] == keys                       # slot order of temper.c:282-464
    nn = nb.sum(axis=1)
    for e in range(E):
        which, prop = o["which"][e], o["proposal"][e]
        lh, uh, acc = o["log.hastings"][e], o["unif.hastings"][e], o["acceptd"][e]
        state, nrm, uc = o["state"][e], o["norm"][e], o["unif.choose"][e]
        # 1. decision about within-component or jump/swap
        assert np.array_equal(o["unif.which"][e] < 0.5, which)
        # 5./6. hastings rejection decided correctly
        assert np.array_equal(lh >= 0, np.isnan(uh))
        assert np.all((lh < 0) | acc)
        with np.errstate(invalid="ignore"):
            assert np.array_equal(acc, (lh >= 0) | (uh < np.exp(lh)))
        # the uniforms/normals are the injected ones, consumed in the reference's order (temp-par.R:164-187)
        assert np.array_equal(o["unif.which"][e], unif[e, :, 0])
        assert np.array_equal(nrm[which], norm[e][which]) and np.all(np.isnan(nrm[~which]))
        assert np.array_equal(uh[lh < 0], unif[e, lh < 0, 3])
        after = np.concatenate([state[1:], o["final"][e][None]], axis=0)
        if parallel:
            cop = o["coproposal"][e]
            pi, ci = prop[:, 0].astype(int), cop[:, 0].astype(int)
            assert np.array_equal(which, pi == ci)                                   # 2.
            for t in range(niter):                                                   # 3. provenance
                assert np.array_equal(cop[t, 1:], state[t, ci[t] - 1])
                if not which[t]:
                    assert np.array_equal(prop[t, 1:], state[t, pi[t] - 1])
            # 4. hastings ratio
            f = np.array([_lud_witch_(prop[t, 0], prop[t, 1:]) for t in range(niter)])
            fc = np.array([_lud_witch_(cop[t, 0], cop[t, 1:]) for t in range(niter)])
            fs = np.array([_lud_witch_(cop[t, 0], prop[t, 1:]) for t in range(niter)])
            fcs = np.array([_lud_witch_(prop[t, 0], cop[t, 1:]) for t in range(niter)])
            with np.errstate(invalid="ignore"):
                want = np.where(which, f - fc, fs + fcs - f - fc)
            assert np.allclose(want, lh, rtol=ltol, atol=ltol)
            # 7. acceptance carried out
            mine = state.copy()
            for t in range(niter):
                if acc[t]:
                    if which[t]:
                        mine[t, pi[t] - 1] = prop[t, 1:]
                    else:
                        mine[t, pi[t] - 1] = cop[t, 1:]
                        mine[t, ci[t] - 1] = prop[t, 1:]
            assert np.array_equal(after, mine)
            # 8./9. within-component proposal and component choice
            assert np.max(np.abs(prop[which, 1:] - (cop[which, 1:] + scale * nrm[which]))) < 1e-15
            assert np.array_equal(np.floor(k * uc[which, 0]) + 1, cop[which, 0]) and np.all(np.isnan(uc[which, 1]))
            # 10. swap partner selection
            mi = (np.floor(k * uc[~which, 0]) + 1).astype(int)
            mk = (np.floor(nn[mi - 1] * uc[~which, 1]) + 1).astype(int)
            mj = np.array([np.nonzero(nb[a - 1])[0][b - 1] + 1 for a, b in zip(mi, mk)])
            assert np.array_equal(cop[~which, 0], mi) and np.array_equal(prop[~which, 0], mj)
            assert np.array_equal(uc[:, 0], unif[e, :, 1]) and np.array_equal(uc[~which, 1], unif[e, ~which, 2])
            # 12. batch means
            kept = after[nspac - 1::nspac].reshape(nbatch, blen, k, p).mean(axis=1)
            assert np.allclose(kept, o["batch"][e], rtol=1e-14, atol=1e-15)
            # 13./14. acceptance rates
            ax = np.array([acc[which & (ci == c + 1)].mean() if np.any(which & (ci == c + 1)) else np.nan
                           for c in range(k)])
            assert np.array_equal(ax, o["acceptx"][e], equal_nan=True)
            ai = np.full((k, k), np.nan)
            for a in range(k):
                for b in range(k):
                    sel = ~which & (ci == a + 1) & (pi == b + 1)
                    if nb[a, b] and np.any(sel):
                        ai[a, b] = acc[sel].mean()
            assert np.array_equal(ai, o["accepti"][e], equal_nan=True)
        else:
            si, pi = state[:, 0].astype(int), prop[:, 0].astype(int)
            assert np.array_equal(which, pi == si)                                   # temp-ser.R
            f = np.array([_lud_witch_(state[t, 0], state[t, 1:]) for t in range(niter)])
            g = np.array([_lud_witch_(prop[t, 0], prop[t, 1:]) for t in range(niter)])
            with np.errstate(invalid="ignore"):
                want = g - f - np.log(nn[pi - 1]) + np.log(nn[si - 1])               # temp-ser-witch.R:95-101
            assert np.allclose(want, lh, rtol=ltol, atol=ltol)
            mine = np.where(acc[:, None], prop, state)
            assert np.array_equal(after, mine)
            assert np.max(np.abs(prop[which, 1:] - (state[which, 1:] + scale * nrm[which]))) < 1e-15
            assert np.all(np.isnan(uc[which])) and np.array_equal(uc[~which], unif[e, ~which, 1])
            mk = (np.floor(nn[si[~which] - 1] * uc[~which]) + 1).astype(int)
            mj = np.array([np.nonzero(nb[a - 1])[0][b - 1] + 1 for a, b in zip(si[~which], mk)])
            assert np.array_equal(prop[~which, 0], mj)
            kept = after[nspac - 1::nspac]
            assert np.allclose(kept[:, 1:].reshape(nbatch, blen, p).mean(axis=1), o["batch"][e], rtol=1e-14, atol=1e-15)
            ib = np.stack([(kept[:, 0] == c + 1).reshape(nbatch, blen).mean(axis=1) for c in range(k)], axis=1)
            assert np.array_equal(ib, o["ibatch"][e])                                # temp-ser.R:152-161


# ----------------------------------------------------------------------------- morph.metrop on the device
MORPHS = [dict(b=1.0), dict(r=1.0, p=3), dict(b=0.7, r=0.5, p=3), dict(r=0.3, p=2.2),
          dict(b=1.0, center=[0.5, -1.0, 2.0]), dict(center=[1.0, 2.0, 3.0])]


@pytest.mark.parametrize("kw", MORPHS)
def test_morphed_logdens_vs_oracle(api, oracle_mod, kw):
    # what logh() sees under morph.metrop: lud(inverse(y)) + log.jacobian(y), R/morph.R:166-179
    from mcmc_amd import morph
    rng = np.random.default_rng(11)
    m = morph(**kw)
    d = 3
    states = np.concatenate([rng.standard_normal((200, d)) * 1.5, np.zeros((1, d)), rng.standard_normal((55, d)) * 0.01])
    for K, par in ((abi.T_STUDENT_T, [3.0]), (abi.T_MVNORM, np.concatenate([[0.3, -0.2, 0.1], np.eye(d).ravel() * 2]))):
        got = api.logdens(api.Target(K, d, par=par, morph=m), states, path=abi.PATH_STREAM)
        want = oracle_mod.logdens(oracle_mod.Target(K, d, par=par, morph=m), states)
        assert np.all(np.isfinite(want))
        assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) < 1e-12
    # unmorphed Student t against scipy's dt
    from scipy.stats import t as student
    got = api.logdens(api.Target(abi.T_STUDENT_T, d, par=[3.0]), states, path=abi.PATH_STREAM)
    assert np.max(np.abs(got - student.logpdf(states, 3.0).sum(axis=1))) < 1e-12


@pytest.mark.parametrize("kw", MORPHS)
@pytest.mark.parametrize("outfun", [None, "moments", [True, False, True]])
def test_morphed_metrop_trajectory_vs_oracle(api, oracle_mod, kw, outfun):
    # the chain runs on the morphed state; outfun sees inverse(state) (R/morph.R:95-106)
    from mcmc_amd import morph
    rng = np.random.default_rng(12)
    m = morph(**kw)
    d, C, nbatch, blen, nspac = 3, 64, 25, 3, 2
    niter = nbatch * blen * nspac
    init = rng.standard_normal((C, d))
    z, u = rng.standard_normal((C, niter, d)), rng.random((C, niter))
    of = "moments2" if outfun == "moments" else outfun
    ref = oracle_mod.metrop(oracle_mod.Target(abi.T_STUDENT_T, d, par=[3.0], morph=m), init, nbatch, blen, nspac,
                            scale=0.8, outfun=of, z=z, u=u, debug=True)
    o = api.metrop(api.Target(abi.T_STUDENT_T, d, par=[3.0], morph=m), init, nbatch, blen, nspac, scale=0.8,
                   outfun=of, z=z, u=u, debug=True)
    assert np.array_equal(o["debug.accept"], ref["debug_accept"] != 0)
    assert np.array_equal(np.isnan(o["u"]), np.isnan(ref["u"]))
    assert np.array_equal(o["final"], ref["final"])
    assert o["batch"].shape == ref["batch"].shape
    assert np.max(np.abs(o["batch"] - ref["batch"]) / np.maximum(np.abs(ref["batch"]), 1.0)) < 1e-12
    assert np.max(np.abs(o["log.green"] - ref["log_green"])) < 1e-11
    assert 0.05 < o["accept"].mean() < 0.95


@pytest.mark.parametrize("kw", [dict(b=0.4), dict(r=0.5, p=3), dict(b=0.3, r=1.0, p=3, center=True), dict(center=True)])
@pytest.mark.parametrize("kind", ["logistic", "binomial"])
def test_morphed_mfma_path_vs_oracle(api, oracle_mod, kw, kind):
    """morph.metrop on the matrix cores: the chain runs on the morphed state, the contraction on inverse(state), the
    log.jacobian is added by the update kernel (morph.R:166-179).  The worked example of vignettes/morph.Rnw:361-365 is the
    binomial regression; log densities at 1e-12, and an injected-draw run with the oracle's decisions and finals."""
    from mcmc_amd import morph
    rng = np.random.default_rng(21)
    n, d, C, nbatch, blen, nspac = 6000, 7, 96, 6, 3, 2
    niter = nbatch * blen * nspac
    X, y, beta = _synthetic(n, d, seed=4)
    kw = dict(kw)
    if kw.pop("center", False):
        kw["center"] = list(np.round(beta, 2))
    m = morph(**kw)
    if kind == "logistic":
        K, tkw = abi.T_LOGISTIC, dict(x=X, y=y, prior_c=0.125)
    else:
        tot = rng.integers(1, 30, n) + 0.5
        K, tkw = abi.T_BINOMIAL_LOGIT, dict(x=X, y=tot / (1 + np.exp(-X @ beta)), par=tot)
    states = np.concatenate([0.8 * rng.standard_normal((C - 1, d)), np.zeros((1, d))])      # morphed states, the origin among them
    want = oracle_mod.logdens(oracle_mod.Target(K, d, morph=m, **tkw), states)
    tg = api.Target(K, d, morph=m, **tkw)
    got_m = api.logdens(tg, states, path=abi.PATH_MFMA)
    got_s = api.logdens(tg, states, path=abi.PATH_STREAM)
    assert np.all(np.isfinite(want))
    assert rel(got_m, want) < RTOL and rel(got_s, want) < RTOL
    init = m.transform(beta + 0.02 * rng.standard_normal((C, d))) if hasattr(m, "transform") else states
    z, u = rng.standard_normal((C, niter, d)), rng.random((C, niter))
    sc = 0.02
    ref = oracle_mod.metrop(oracle_mod.Target(K, d, morph=m, **tkw), init, nbatch, blen, nspac, scale=sc, outfun="moments2",
                            z=z, u=u, nthreads=16)
    o = api.metrop(tg, init, nbatch, blen, nspac, scale=sc, outfun="moments2", z=z, u=u, path=abi.PATH_MFMA)
    assert o["path"] == "mfma"
    assert np.array_equal(o["accept.batch"], ref["accept_batch"])
    assert np.array_equal(o["final"], ref["final"])
    assert np.max(np.abs(o["batch"] - ref["batch"]) / np.maximum(np.abs(ref["batch"]), 1.0)) < 1e-12
    assert 0.02 < o["accept"].mean() < 0.98


def test_morph_metrop_api_and_restart(api):
    # tests/morph.metrop.R:8-18 (runs; the transformation can be changed), tests/saveseedmorph.R (restart
    # determinism), R/morph.metrop.R:42-49 (nspac is not forwarded on the first call), 55-62 (.morph.unmorph)
    import mcmc_amd
    from mcmc_amd import morph
    tg = api.Target(abi.T_STUDENT_T, 1, par=[3.0])
    obj = mcmc_amd.morph_metrop(tg, [100.0], 100, morph=morph(b=3), seed=42)
    assert set(obj) >= {"accept", "batch", "initial", "final", "accept.batch", "morph", "morph.final", "outfun"}
    assert obj["final"] == pytest.approx(obj["morph"].inverse(obj["morph.final"]))
    obj = mcmc_amd.morph_metrop(obj, morph=morph(b=1))
    assert obj["morph"].b == 1.0 and obj["batch"].shape == (100, 1)
    tg3 = api.Target(abi.T_STUDENT_T, 3, par=[3.0])
    obj = mcmc_amd.morph_metrop(tg3, np.full(3, 100.0), 100, morph=morph(p=3, b=1), seed=1)
    obj = mcmc_amd.morph_metrop(obj, morph=morph(r=1, p=3, b=1))
    assert obj["batch"].shape == (100, 3)

    # saveseedmorph.R with the simplex target of the reference's test
    h = api.Target(abi.T_SIMPLEX, 5, par=[1.0])
    out = mcmc_amd.morph_metrop(h, np.zeros(5), 100, blen=17, nspac=3, scale=0.1, seed=42)
    assert out["nspac"] == 1 and out["blen"] == 17            # the quirk: nspac dropped on the first call
    out1 = mcmc_amd.morph_metrop(out)
    out2 = mcmc_amd.morph_metrop(out1)
    out3 = mcmc_amd.morph_metrop(out, nbatch=2 * out["nbatch"])
    assert np.array_equal(np.concatenate([out1["batch"], out2["batch"]]), out3["batch"])
    out = mcmc_amd.morph_metrop(out, morph=morph(p=2.2, r=0.3))
    out1 = mcmc_amd.morph_metrop(out)
    out2 = mcmc_amd.morph_metrop(out1)
    out3 = mcmc_amd.morph_metrop(out, nbatch=2 * out["nbatch"])
    assert np.array_equal(np.concatenate([out1["batch"], out2["batch"]]), out3["batch"])
    assert np.all(out3["batch"] >= 0) and np.all(out3["batch"].sum(axis=1) <= 1 + 1e-12)   # outfun sees inverse(state)


def test_morph_metrop_samples_the_right_distribution(api):
    # morphed chains on a heavy-tailed target: the unmorphed draws must follow t_3 (vignettes/morph.Rnw's point)
    import mcmc_amd
    from mcmc_amd import morph
    from scipy.stats import t as student
    tg = api.Target(abi.T_STUDENT_T, 1, par=[3.0])
    out = mcmc_amd.morph_metrop(tg, np.zeros((2048, 1)), 400, blen=1, scale=2.0, morph=morph(b=1), seed=7)
    draws = out["batch"][:, 100:, 0].ravel()
    q = np.quantile(draws, [0.05, 0.25, 0.5, 0.75, 0.95])
    assert np.max(np.abs(q - student.ppf([0.05, 0.25, 0.5, 0.75, 0.95], 3.0))) < 0.05


def test_morph_descriptor_errors(api):
    from mcmc_amd import morph
    with pytest.raises(ValueError, match="metrop targets only"):
        api.Target(abi.T_WITCH, 2, par=[0.5, 0.25], morph=morph(b=1))
    X = np.ones((8, 2))
    y = np.zeros(8)
    # a morphed descriptor without a design matrix has no MFMA path (a morphed logistic one does since round 3)
    tg = api.Target(abi.T_STUDENT_T, 2, par=[3.0], morph=morph(b=1))
    with pytest.raises(api.McmcHipError, match="MFMA"):
        api.logdens(tg, np.zeros((4, 2)), path=abi.PATH_MFMA)


def test_logistic_models_logdens(api, oracle_mod):
    # tests/temp-par.R:58-60 (finite at the sub-model MLEs) through mcmchip_logdens; then a Philox serial-tempering
    # run replayed by the oracle (the Bayes-factor workload of vignettes/bfst.Rnw: ibatch = time spent per model)
    from rhelpers import temp_par_setup
    X, y, models, masks, betas, nb = temp_par_setup()
    lpp = np.array([0.0, 0.4, 0.1, 0.9, -0.2, 0.3, 0.5, 1.1])
    kw = dict(x=X, y=y, prior_c=0.125, par=abi.logistic_models_par(masks, lpp, -4 * np.log(2 * np.sqrt(2 * np.pi))))
    states = np.column_stack([np.arange(1, 9.0), betas])
    got = api.logdens(api.Target(abi.T_LOGISTIC_MODELS, 4, **kw), states)
    want = oracle_mod.logdens(oracle_mod.Target(abi.T_LOGISTIC_MODELS, 4, **kw), states)
    assert np.all(np.isfinite(got)) and np.max(np.abs(got - want)) < 1e-11
    E = 16
    init = np.tile(np.concatenate([[8.0], betas[7]]), (E, 1))
    # (the Philox serial-tempering run against the oracle: tests/test_gpu_round2.py::test_serial_tempering_philox_decisions_match_oracle)
    with pytest.raises(api.McmcHipError, match="components"):
        api.temper(api.Target(abi.T_LOGISTIC_MODELS, 4, **kw), init[:, :5], nb[:4, :4], 5, 1, 1, 0.5, parallel=False, seed=1)


def test_olbm_time_sharding_arithmetic_on_device(api, oracle_mod):
    # mcmc_amd/dist.py::olbm_sharded with the device entry point, ranks emulated in turn (no process group):
    # blocks with a len-1 halo, global means given (nocalc contract), un-normalised partial sums added
    from mcmc_amd import dist as mdist
    rng = np.random.default_rng(21)
    n, p, blen, ws = 30011, 20, 97, 3
    x = np.cumsum(rng.standard_normal((n, p)) * 0.05, axis=0) + rng.standard_normal((n, p)) + 1.0
    mean = x.mean(axis=0)
    total = np.zeros((p, p))
    rows = 0
    for r in range(ws):
        lo, hi, own = mdist.olbm_time_shard(n, blen, r, ws)
        rows += own
        blk = x[lo:hi]
        nwin = blk.shape[0] - blen + 1
        total += api.olbm(blk, blen, mean=mean) * (float(nwin) * blk.shape[0] * blen)
    assert rows == n
    got = total / (float(n - blen + 1) * n * blen)
    ref, _ = oracle_mod.olbm(x, blen)
    assert np.max(np.abs(got - ref)) <= 1e-10 * np.max(np.abs(ref))


def _inject_from_trace(ref, parallel, p):
    """random numbers of an oracle run (driven by the emulated R stream) laid out as the device's injected slots
    unif[iter][4] = {which, choose1, choose2, hastings}, norm[iter][p]; slots the run did not draw are never read"""
    T = ref["unif.which"].shape[1]
    unif = np.full((1, T, 4), 0.5)
    unif[0, :, 0] = ref["unif.which"][0]
    uc = ref["unif.choose"][0]
    if parallel:
        unif[0, :, 1] = np.where(np.isnan(uc[:, 0]), 0.5, uc[:, 0])
        unif[0, :, 2] = np.where(np.isnan(uc[:, 1]), 0.5, uc[:, 1])
    else:
        unif[0, :, 1] = np.where(np.isnan(uc), 0.5, uc)
    uh = ref["unif.hastings"][0]
    unif[0, :, 3] = np.where(np.isnan(uh), 0.5, uh)
    norm = np.where(np.isnan(ref["norm"][0]), 0.0, ref["norm"][0]).reshape(1, T, p)
    return unif, norm


def test_temper_device_reproduces_reference_transcripts(api, oracle_mod):
    """tests/temp-par-witch.Rout.save and tests/temp-ser-witch.Rout.save on the DEVICE: the oracle, driven by the
    emulated R stream (set.seed(42)), records which uniforms/normals each iteration drew; the device replays them
    as injected streams, with the reference's outfun (corner indicator) where the script uses one, and must print
    the transcript's numbers."""
    from test_oracle_golden import WITCH, _witch_neighbors
    nb = _witch_neighbors()
    sc = 0.3456789
    ot, dt = oracle_mod.Target(abi.T_WITCH, 3, par=WITCH), api.Target(abi.T_WITCH, 3, par=WITCH)
    # ---- parallel (temp-par-witch.R:42-75)
    rs = oracle_mod.RStream(42)
    runs = [(50, 13, None), (50, 13, "witch.corner"), (50, 103, "witch.corner")]
    state = np.zeros((6, 3))
    outs = []
    for nbatch, blen, outfun in runs:
        ref = oracle_mod.temper(ot, state, nb, nbatch, blen, 7, sc, parallel=True, rstream=rs, debug=True,
                                out_kind=oracle_mod.TOUT_WITCH_PAR if outfun else oracle_mod.TOUT_NONE)
        unif, norm = _inject_from_trace(ref, True, 3)
        o = api.temper(dt, state[None], nb, nbatch, blen, 7, sc, parallel=True, unif=unif, norm=norm, outfun=outfun)
        assert np.array_equal(o["final"][0], ref["final"][0])
        assert np.allclose(o["batch"][0], ref["batch"][0], rtol=1e-14, atol=1e-15)
        state = ref["final"][0]
        outs.append(o)
    assert list(np.round(outs[0]["acceptx"][0], 4)) == [0.6336, 0.4974, 0.3245, 0.6022, 0.6130, 0.5914]
    b = outs[1]["batch"][0]
    assert b.shape == (50, 6)
    assert np.allclose(b.mean(axis=0), [0.54923, 0.40923, 0.39538, 0.09692, 0.12923, 0.60000], atol=5e-6)
    assert np.allclose(b.std(axis=0, ddof=1) / np.sqrt(50), [0.03482, 0.04817, 0.05464, 0.02856, 0.02113, 0.05131], atol=5e-6)
    b = outs[2]["batch"][0]
    assert np.allclose(b.mean(axis=0), [0.5231, 0.5361, 0.4905, 0.5652, 0.4056, 0.2450], atol=5e-5)
    # ---- serial (temp-ser-witch.R:36-78)
    rs = oracle_mod.RStream(42)
    state = np.array([1.0, 0, 0, 0])
    ref = oracle_mod.temper(ot, state, nb, 50, 13, 7, sc, parallel=False, rstream=rs, debug=True)
    unif, norm = _inject_from_trace(ref, False, 3)
    o = api.temper(dt, state[None], nb, 50, 13, 7, sc, parallel=False, unif=unif, norm=norm)
    assert np.allclose(o["acceptx"][0], [0.6388889, 0.4385246, 0.3631714, 0.4885246, 0.4709677, 0.4735516], atol=5e-8)
    assert np.allclose(o["ibatch"][0].mean(axis=0), [0.1830769, 0.2153846, 0.1630769, 0.1369231, 0.1353846, 0.1661538], atol=5e-8)
    ref2 = oracle_mod.temper(ot, ref["final"][0], nb, 50, 103, 7, sc, parallel=False, rstream=rs, debug=True,
                             out_kind=oracle_mod.TOUT_WITCH_SER)
    unif, norm = _inject_from_trace(ref2, False, 3)
    o2 = api.temper(dt, ref["final"][0][None], nb, 50, 103, 7, sc, parallel=False, unif=unif, norm=norm,
                    outfun="witch.corner")
    b = o2["batch"][0]
    assert b.shape == (50, 12)
    eta, noo = b[:, :6].mean(axis=0), b[:, 6:].mean(axis=0)
    assert np.allclose(eta, [0.06660194, 0.06388350, 0.05766990, 0.06563107, 0.10368932, 0.22912621], atol=5e-9)
    assert np.allclose(noo, [0.1365049, 0.1258252, 0.1293204, 0.1370874, 0.1716505, 0.2996117], atol=5e-8)
    assert np.allclose(eta / noo, [0.4879090, 0.5077160, 0.4459459, 0.4787535, 0.6040724, 0.7647440], atol=5e-8)
    with pytest.raises(api.McmcHipError, match="witch"):
        api.temper(api.Target(abi.T_TEMPERED_NORMMIX, 2, par=np.concatenate([[2.0], [1.0, 0.5], abi.normmix_par([1.0], [[0.0, 0.0]], 1.0)])),
                   np.zeros((2, 2)), np.array([[0, 1], [1, 0]], dtype=bool), 3, 1, 1, 0.5, parallel=True, seed=1, outfun="witch.corner")


def _replay_metrop(api, oracle_mod, dtg, otg, x0, nbatch, blen, scale, rs):
    """one metrop call of a reference script: the oracle consumes the emulated R stream and records the normals and
    uniforms it drew (debug trace); the device replays them as injected streams"""
    ref = oracle_mod.metrop(otg, x0, nbatch, blen, scale=scale, rstream=rs, debug=True)
    z = ref["z"]                                           # (1, niter, d)
    u = np.where(np.isnan(ref["u"]), 0.5, ref["u"])        # undrawn slots are never read
    o = api.metrop(dtg, np.asarray(x0, dtype=np.float64)[None], nbatch, blen, scale=scale, z=z, u=u)
    assert np.array_equal(o["final"][0], ref["final"][0])
    return o, ref


def test_metrop_device_reproduces_reference_transcripts(api, oracle_mod):
    """tests/logit.Rout.save (with its olbm matrix), tests/accept-batch.Rout.save and tests/circle.Rout.save printed
    by the DEVICE path: random numbers as R would have drawn them (emulated stream, replayed as injected)."""
    from rhelpers import signif, t_conf_int
    # ---- tests/logit.R (Marsaglia-Multicarry; data simulated in the script; naive log likelihood)
    rs = oracle_mod.RStream(42, "Marsaglia-Multicarry")
    n, rho = 100, 0.5
    x1 = rs.rnorm(n)
    x2 = rho * x1 + np.sqrt(1 - rho ** 2) * rs.rnorm(n)
    eta = 0.25 + 1 * x1 + 0.5 * x2
    y = (rs.runif(n) < 1 / (1 + np.exp(-eta))).astype(float)
    X = np.column_stack([np.ones(n), x1, x2])
    coef = glm_binomial(X, y)
    kw = dict(x=X, y=y, prior_c=0.0)
    dtg, otg = api.Target(abi.T_LOGISTIC_NAIVE, 3, **kw), oracle_mod.Target(abi.T_LOGISTIC_NAIVE, 3, **kw)
    assert round(-api.logdens(dtg, coef[None])[0], 1) == 49.6                     # logit.Rout.save:83
    x = coef
    for sc, expect in [(0.01, 0.982), (0.1, 0.795), (0.5, 0.264)]:               # :104-112
        o, ref = _replay_metrop(api, oracle_mod, dtg, otg, x, 1000, 1, sc, rs)
        assert o["accept"][0] == expect
        x = ref["final"][0]
    b = o["batch"][0]
    assert list(np.round(b.mean(axis=0), 4)) == [0.0608, 1.4230, 0.5263]         # :115
    var = api.olbm(b, 25)                                                        # :121-125, on the device
    assert np.allclose(signif(var, 3), [[4.54e-04, 9.47e-05, -1.92e-05], [9.47e-05, 1.84e-03, -6.45e-04],
                                        [-1.92e-05, -6.45e-04, 9.09e-04]], rtol=1e-12)
    # ---- tests/accept-batch.R
    h_d, h_o = api.Target(abi.T_SIMPLEX, 5, par=[1.0]), oracle_mod.Target(abi.T_SIMPLEX, 5, par=[1.0])
    o, _ = _replay_metrop(api, oracle_mod, h_d, h_o, np.zeros(5), 100, 100, 0.1, oracle_mod.RStream(42))
    assert round(o["accept"][0], 4) == 0.2257                                    # accept-batch.Rout.save:33
    lo, hi = t_conf_int(o["accept.batch"][0])
    assert (round(lo, 4), round(hi, 4)) == (0.2124, 0.2390)                      # :39-42
    # ---- tests/circle.R (Marsaglia-Multicarry)
    rs = oracle_mod.RStream(42, "Marsaglia-Multicarry")
    c_d, c_o = api.Target(abi.T_BALL, 5), oracle_mod.Target(abi.T_BALL, 5)
    x = np.zeros(5)
    for sc, expect in [(0.01, 0.979), (0.1, 0.72), (0.5, 0.16), (0.4, 0.228)]:   # circle.Rout.save:24-50
        o, ref = _replay_metrop(api, oracle_mod, c_d, c_o, x, 1000, 1, sc, rs)
        assert o["accept"][0] == expect
        x = ref["final"][0]


# ----------------------------------------------------------------------------- the C-ABI from plain C
def test_c_demo_matches_python_api(api, tmp_path):
    """examples/capi_demo.c (gcc, only include/mcmchip.h) drives metrop / initseq / olbm through the C-ABI; the Python
    mirror with the same inputs and Philox seed must give the same bits (same library underneath)"""
    import subprocess
    exe = os.path.join(ROOT, "examples", "capi_demo")
    # a build artefact like the .so (__graft_entry__.build() makes it, it travels with the tree): its absence is a
    # failure, not a skip -- this is the only test that drives the boundary from plain C
    assert os.path.exists(exe), "examples/capi_demo is not built (make -C examples)"
    rng = np.random.default_rng(5)
    n, d, C, nbatch, blen, seed = 300, 4, 6, 60, 7, 20240611
    X = np.column_stack([np.ones(n), rng.standard_normal((n, d - 1))])
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ np.array([0.3, -0.5, 0.2, 0.1])))).astype(float)
    data = tmp_path / "data.bin"
    with open(data, "wb") as f:
        np.ascontiguousarray(X).tofile(f)
        y.tofile(f)
    out = subprocess.run([exe, str(data), str(n), str(d), str(C), str(nbatch), str(blen), str(seed)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = {}
    for line in out.stdout.splitlines():
        name, _, rest = line.partition(":")
        got[name] = rest.strip()
    vec = lambda k: np.array([float(v) for v in got[k].split()])
    assert got["error"] == 'argument "nbatch" must be positive'                 # metrop.c:60-61
    o = api.metrop(api.logistic_target(X, y), np.zeros((C, d)), nbatch, blen, 1, scale=0.4, seed=seed, path=abi.PATH_STREAM)
    assert np.array_equal(vec("accept"), o["accept"])
    assert np.array_equal(vec("final"), o["final"].ravel())
    b0 = o["batch"][0]                                                           # nbatch x d
    r = api.initseq(b0[:, 0])                  # centres with numpy's mean, the demo with a running sum: ulps apart
    assert int(got["initseq.m"]) == len(r["Gamma.pos"])
    assert np.allclose(vec("gamma0"), [r["gamma0"]], rtol=1e-10, atol=0)
    assert np.allclose(vec("Gamma.pos"), r["Gamma.pos"], rtol=1e-10, atol=1e-18)
    assert np.allclose(vec("Gamma.con"), r["Gamma.con"], rtol=1e-10, atol=1e-18)
    assert np.allclose(vec("var.con"), [r["var.con"]], rtol=1e-10, atol=0)
    v = api.olbm(b0, nbatch // 5, demean=False)                                  # both let the device compute the means
    assert np.array_equal(vec("olbm.var"), np.asarray(v).ravel(order="F"))
    assert np.allclose(vec("olbm.mean"), b0.mean(axis=0), rtol=1e-13, atol=0)
