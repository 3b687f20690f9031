"""Pin the CPU oracle against the reference's OWN printed goldens.

Each test replays one of the reference's R transcripts (the script text is visible in the
transcript itself) against oracle/ driven by the emulated R random stream (oracle/rrng.c) and
compares with the numbers R printed.  Sources, all under /root/reference:
  devel/bug/foo.Rout                      simplex + vignette logistic posterior (the north-star C1 config)
  package/mcmc/tests/accept-batch.Rout.save
  package/mcmc/tests/circle.Rout.save      (Marsaglia-Multicarry)
  package/mcmc/tests/logit.Rout.save       (Marsaglia-Multicarry; glm start; the only olbm golden)
  package/mcmc/tests/temp-par-witch.Rout.save, temp-ser-witch.Rout.save
  package/mcmc/tests/initseq.R             (identities; no numeric golden exists)
"""
import numpy as np
import pytest

from mcmc_amd import abi
from rhelpers import duplicated, glm_binomial, load_logit, signif, t_conf_int


def test_r_rng_emulation(oracle_mod):
    # documented R outputs: set.seed(42); runif(3) / rnorm(3) under the default kinds
    assert np.allclose(oracle_mod.RStream(42).runif(3), [0.9148060, 0.9370754, 0.2861395], atol=5e-8)
    assert np.allclose(oracle_mod.RStream(42).rnorm(3), [1.37095845, -0.56469817, 0.36312841], atol=5e-9)
    # AS 241 against an independent inverse normal cdf
    from scipy.special import ndtri
    ps = np.concatenate([np.linspace(1e-10, 1 - 1e-10, 4001), 10.0 ** -np.arange(1.0, 10.5, 0.5)])
    q = np.array([oracle_mod.qnorm(p) for p in ps])
    assert np.max(np.abs(q - ndtri(ps)) / (1 + np.abs(q))) < 1e-14
    # far tail (p < exp(-25), reached with probability 3e-11 per draw, never in any golden):
    # the restated AS 241 branch is only checked loosely, see the note in oracle/rrng.c
    ps = 10.0 ** -np.arange(12.0, 18)
    q = np.array([oracle_mod.qnorm(p) for p in ps])
    assert np.max(np.abs(q - ndtri(ps))) < 1e-6


def test_accept_batch_rout(oracle_mod):
    # tests/accept-batch.Rout.save:21-42
    h = oracle_mod.Target(abi.T_SIMPLEX, 5, par=[1.0])
    out = oracle_mod.metrop(h, np.zeros(5), 100, 100, scale=0.1, rstream=oracle_mod.RStream(42), debug=True)
    assert np.isclose(out["accept"][0], out["accept_batch"][0].mean())
    bar = out["debug_accept"][0].reshape(100, 100).mean(axis=1)
    assert np.allclose(out["accept_batch"][0], bar)
    assert round(out["accept"][0], 4) == 0.2257
    lo, hi = t_conf_int(out["accept_batch"][0])
    assert (round(lo, 4), round(hi, 4)) == (0.2124, 0.2390)


def test_foo_rout_simplex_and_vignette_logit(oracle_mod):
    # devel/bug/foo.Rout:26-176 -- one R session, one stream
    rs = oracle_mod.RStream(42)
    h = oracle_mod.Target(abi.T_SIMPLEX, 5, par=[0.0])
    o = oracle_mod.metrop(h, np.zeros(5), 1000, rstream=rs)
    assert o["accept"][0] == 0.0                                           # foo.Rout:30-31
    o = oracle_mod.metrop(h, o["final"], 1000, scale=0.1, rstream=rs)
    assert np.allclose(t_conf_int(o["accept_batch"][0]), (0.1952394, 0.2467606), atol=5e-8)  # :34-35
    o = oracle_mod.metrop(h, o["final"], 1000, scale=0.1, rstream=rs, debug=True)
    u = o["u"][0][~np.isnan(o["u"][0])]
    assert not duplicated(u).any() and not duplicated(o["z"][0].ravel()).any()   # :49-58
    o = oracle_mod.metrop(h, o["final"], 1000, 1000, scale=0.1, rstream=rs, debug=True)
    u = o["u"][0][~np.isnan(o["u"][0])]
    dup = duplicated(u)
    assert np.allclose(u[dup], [0.6841277, 0.8322727, 0.5598184, 0.1526256], atol=5e-8)   # :67-68
    assert list(np.nonzero(dup)[0] + 1) == [120082, 125717, 158400, 194958]              # :69-70

    X, y = load_logit()
    beta_init = glm_binomial(X, y)
    tg = oracle_mod.Target(abi.T_LOGISTIC, 5, x=X, y=y, prior_c=0.125)
    o = oracle_mod.metrop(tg, beta_init, 1000, rstream=rs)
    assert o["accept"][0] == 0.019                                          # :100-102
    for sc, expect in [(0.1, 0.725), (0.3, 0.34), (0.5, 0.153)]:            # :104-112
        o = oracle_mod.metrop(tg, o["final"], 1000, scale=sc, rstream=rs)
        assert o["accept"][0] == expect
    o = oracle_mod.metrop(tg, o["final"], 1000, scale=0.4, rstream=rs)
    assert np.allclose(t_conf_int(o["accept_batch"][0]), (0.2211882, 0.2748118), atol=5e-8)  # :113-115
    o = oracle_mod.metrop(tg, o["final"], 1000, scale=0.4, rstream=rs, debug=True)
    o = oracle_mod.metrop(tg, o["final"], 1000, 1000, scale=0.4, rstream=rs, debug=True)
    u = o["u"][0][~np.isnan(o["u"][0])]
    dup = duplicated(u)
    assert dup.sum() == 85                                                  # :148-172
    assert np.allclose(u[dup][:12], [0.70289390, 0.65096803, 0.63641961, 0.82967802, 0.38743379,
                                     0.98920203, 0.71896786, 0.20370840, 0.69353093, 0.23638367,
                                     0.05904343, 0.10012407], atol=5e-9)
    pos = np.nonzero(dup)[0] + 1
    assert list(pos[:10]) == [99993, 118147, 119756, 132096, 136010, 171510, 178591, 196729, 204156, 216334]
    assert list(pos[-5:]) == [860891, 871389, 875918, 876623, 878759]


def test_circle_rout(oracle_mod):
    # tests/circle.Rout.save:24-50 (Marsaglia-Multicarry)
    rs = oracle_mod.RStream(42, "Marsaglia-Multicarry")
    tg = oracle_mod.Target(abi.T_BALL, 5)
    x = np.zeros(5)
    for sc, expect in [(0.01, 0.979), (0.1, 0.72), (0.5, 0.16), (0.4, 0.228)]:
        o = oracle_mod.metrop(tg, x, 1000, scale=sc, rstream=rs)
        x = o["final"]
        assert o["accept"][0] == expect
    # trace identities of circle.R:32-70 on the debug run
    rs2 = rs.clone()
    o = oracle_mod.metrop(tg, x, 100, scale=0.4, rstream=rs, debug=True)
    cur, prop, uu, lg = o["current"][0], o["proposal"][0], o["u"][0], o["log_green"][0]
    assert np.all(o["batch"][0][:-1] == cur[1:]) and np.all(cur[0] == o["initial"][0])
    assert np.all(o["batch"][0][-1] == o["final"][0])
    my_u = np.full(100, np.nan)
    for i in range(100):
        assert np.max(np.abs(prop[i] - (cur[i] + 0.4 * rs2.rnorm(5)))) < 1e-15
        if not np.isnan(uu[i]):
            my_u[i] = rs2.runif(1)[0]
    assert np.all(uu[~np.isnan(uu)] == my_u[~np.isnan(my_u)])
    lc, lp = oracle_mod.logdens(tg, cur), oracle_mod.logdens(tg, prop)
    assert np.all(np.isnan(uu) == ((lp == -np.inf) | (lp > lc)))     # -Inf draws NO uniform (metrop.c:226)
    acc = (lp > -np.inf) & (np.isnan(my_u) | (my_u < np.exp(lp - lc)))
    assert np.all(acc == (o["debug_accept"][0] != 0))
    assert acc.sum() == round(100 * o["accept"][0])


def test_logit_rout_with_olbm(oracle_mod):
    # tests/logit.Rout.save:23-125 (Marsaglia-Multicarry, data simulated in the script, naive logl)
    rs = oracle_mod.RStream(42, "Marsaglia-Multicarry")
    n, rho = 100, 0.5
    x1 = rs.rnorm(n)
    x2 = rho * x1 + np.sqrt(1 - rho ** 2) * rs.rnorm(n)
    eta = 0.25 + 1 * x1 + 0.5 * x2
    p = 1 / (1 + np.exp(-eta))
    y = (rs.runif(n) < p).astype(float)
    X = np.column_stack([np.ones(n), x1, x2])
    coef = glm_binomial(X, y)
    assert list(np.round(coef, 4)) == [0.0599, 1.3682, 0.4760]      # logit.Rout.save:52-54
    tg = oracle_mod.Target(abi.T_LOGISTIC_NAIVE, 3, x=X, y=y, prior_c=0.0)
    assert round(-oracle_mod.logdens(tg, coef)[0], 1) == 49.6        # :83
    o = oracle_mod.metrop(tg, coef, 1000, scale=0.01, rstream=rs)
    assert o["accept"][0] == 0.982                                   # :104
    o = oracle_mod.metrop(tg, o["final"], 1000, scale=0.1, rstream=rs)
    assert o["accept"][0] == 0.795                                   # :108
    o = oracle_mod.metrop(tg, o["final"], 1000, scale=0.5, rstream=rs)
    assert o["accept"][0] == 0.264                                   # :112
    b = o["batch"][0]
    assert list(np.round(b.mean(axis=0), 4)) == [0.0608, 1.4230, 0.5263]        # :115
    v = np.cov(b.T)
    assert np.allclose(signif(v, 4), [[0.06755, -0.0108, 0.00989], [-0.01080, 0.1758, -0.06155],
                                      [0.00989, -0.06155, 0.10483]], rtol=2e-3)  # :117-120 (R pads columns)
    var, _ = oracle_mod.olbm(b, 25)                                  # :121-125, the reference's only olbm golden
    assert np.allclose(signif(var, 3), [[4.54e-04, 9.47e-05, -1.92e-05], [9.47e-05, 1.84e-03, -6.45e-04],
                                        [-1.92e-05, -6.45e-04, 9.09e-04]], rtol=1e-12)
    # R/olbm.R:8-14 quirk: demean=TRUE passes the mean with nocalc=TRUE; same answer
    var2, _ = oracle_mod.olbm(b, 25, mean=b.mean(axis=0))
    assert np.allclose(var, var2, rtol=1e-9)


WITCH = 1 - (1 / 2) ** (1 / 3) * (1 / 4) ** (np.arange(6) / 3)


def _witch_neighbors():
    k = 6
    i, j = np.indices((k, k))
    return (np.abs(i - j) == 1) | (np.abs(i - j) == 2)


def test_temp_par_witch_rout(oracle_mod):
    # tests/temp-par-witch.Rout.save:25-114
    rs = oracle_mod.RStream(42)
    tg = oracle_mod.Target(abi.T_WITCH, 3, par=WITCH)
    nb = _witch_neighbors()
    o = oracle_mod.temper(tg, np.zeros((6, 3)), nb, 50, 13, 7, 0.3456789, parallel=True, rstream=rs)
    assert list(np.round(o["acceptx"][0], 4)) == [0.6336, 0.4974, 0.3245, 0.6022, 0.6130, 0.5914]
    expect = np.array([[np.nan, 0.7051, 0.5497, np.nan, np.nan, np.nan],
                       [0.7523, np.nan, 0.5547, 0.6288, np.nan, np.nan],
                       [0.5794, 0.5865, np.nan, 0.5309, 0.5476, np.nan],
                       [np.nan, 0.6667, 0.5506, np.nan, 0.8272, 0.6837],
                       [np.nan, np.nan, 0.5439, 0.8926, np.nan, 0.8374],
                       [np.nan, np.nan, np.nan, 0.8391, 0.9023, np.nan]])
    assert np.array_equal(np.isnan(o["accepti"][0]), np.isnan(expect))
    assert np.allclose(np.round(o["accepti"][0], 4), expect, equal_nan=True, atol=1e-12)
    o2 = oracle_mod.temper(tg, o["final"][0], nb, 50, 13, 7, 0.3456789, parallel=True, rstream=rs,
                           out_kind=oracle_mod.TOUT_WITCH_PAR)
    b = o2["batch"][0]
    assert np.allclose(b.mean(axis=0), [0.54923, 0.40923, 0.39538, 0.09692, 0.12923, 0.60000], atol=5e-6)  # R prints 5 decimals here
    assert np.allclose(b.std(axis=0, ddof=1) / np.sqrt(50),
                       [0.03482, 0.04817, 0.05464, 0.02856, 0.02113, 0.05131], atol=5e-6)
    o3 = oracle_mod.temper(tg, o2["final"][0], nb, 50, 103, 7, 0.3456789, parallel=True, rstream=rs,
                           out_kind=oracle_mod.TOUT_WITCH_PAR)
    b = o3["batch"][0]
    assert np.allclose(b.mean(axis=0), [0.5231, 0.5361, 0.4905, 0.5652, 0.4056, 0.2450], atol=5e-5)
    assert np.allclose(b.std(axis=0, ddof=1) / np.sqrt(50),
                       [0.01390, 0.02213, 0.03961, 0.04909, 0.05107, 0.05108], atol=5e-6)


def test_temp_ser_witch_rout(oracle_mod):
    # tests/temp-ser-witch.Rout.save:21-127
    rs = oracle_mod.RStream(42)
    tg = oracle_mod.Target(abi.T_WITCH, 3, par=WITCH)
    nb = _witch_neighbors()
    o = oracle_mod.temper(tg, np.array([1.0, 0, 0, 0]), nb, 50, 13, 7, 0.3456789, parallel=False, rstream=rs)
    assert np.allclose(o["acceptx"][0], [0.6388889, 0.4385246, 0.3631714, 0.4885246, 0.4709677, 0.4735516], atol=5e-8)
    expect = np.array([[np.nan, 0.5071770, 0.2727273, np.nan, np.nan, np.nan],
                       [0.7070064, np.nan, 0.4355828, 0.4186047, np.nan, np.nan],
                       [0.5816327, 0.8039216, np.nan, 0.5888889, 0.5662651, np.nan],
                       [np.nan, 0.7415730, 0.8571429, np.nan, 0.7857143, 0.6626506],
                       [np.nan, np.nan, 0.5204082, 0.6516854, np.nan, 0.8378378],
                       [np.nan, np.nan, np.nan, 0.3515152, 0.5056818, np.nan]])
    assert np.allclose(o["accepti"][0], expect, equal_nan=True, atol=5e-8)
    assert np.allclose(o["ibatch"][0].mean(axis=0),
                       [0.1830769, 0.2153846, 0.1630769, 0.1369231, 0.1353846, 0.1661538], atol=5e-8)
    o = oracle_mod.temper(tg, o["final"][0], nb, 50, 103, 7, 0.3456789, parallel=False, rstream=rs,
                          out_kind=oracle_mod.TOUT_WITCH_SER)
    b = o["batch"][0]
    eta, noo = b[:, :6].mean(axis=0), b[:, 6:].mean(axis=0)
    assert np.allclose(eta, [0.06660194, 0.06388350, 0.05766990, 0.06563107, 0.10368932, 0.22912621], atol=5e-9)
    assert np.allclose(noo, [0.1365049, 0.1258252, 0.1293204, 0.1370874, 0.1716505, 0.2996117], atol=5e-8)
    assert np.allclose(eta / noo, [0.4879090, 0.5077160, 0.4459459, 0.4787535, 0.6040724, 0.7647440], atol=5e-8)


def test_initseq_identities(oracle_mod):
    # tests/initseq.R:6-48 with an AR(1) series generated here (arima.sim itself is not emulated)
    rng = np.random.default_rng(42)
    n, rho = 100000, 0.99
    e = rng.standard_normal(n + 2000)
    x = np.empty_like(e)
    x[0] = e[0]
    for t in range(1, e.size):
        x[t] = rho * x[t - 1] + e[t]
    x = x[2000:]
    xc = x - x.mean()
    out = oracle_mod.initseq(xc)
    lagmax = 1999
    # acf(type = "covariance"): sum_{t} xc[t] xc[t+k] / n via FFT
    f = np.fft.rfft(xc, 2 * n)
    gamma = np.fft.irfft(f * np.conj(f))[: lagmax + 1] / n
    Gamma = gamma[0::2] + gamma[1::2]
    k = np.nonzero(Gamma < 0)[0][0]
    Gamma = Gamma[: k + 1].copy()
    Gamma[k] = 0
    assert np.isclose(gamma[0], out["gamma0"], rtol=1e-10)
    assert len(out["Gamma.pos"]) == len(Gamma)
    assert np.allclose(out["Gamma.pos"], Gamma, rtol=1e-8, atol=1e-8)
    assert np.array_equal(out["Gamma.dec"], np.minimum.accumulate(out["Gamma.pos"]))
    # pava(diff(Gamma.dec)) via an independent isotonic regression
    from sklearn.isotonic import isotonic_regression
    d = np.diff(out["Gamma.dec"])
    con = out["Gamma.dec"][0] + np.concatenate([[0], np.cumsum(isotonic_regression(d))])
    assert np.allclose(out["Gamma.con"], con, rtol=1e-9, atol=1e-9)
    assert np.isclose(0, np.min(out["Gamma.pos"] - out["Gamma.dec"]))
    assert np.max(np.diff(out["Gamma.dec"])) < np.sqrt(np.finfo(float).eps)
    assert np.min(out["Gamma.dec"] - out["Gamma.con"]) > -1e-8
    assert np.min(np.diff(np.diff(out["Gamma.con"]))) > -np.sqrt(np.finfo(float).eps)
    for s in ("pos", "dec", "con"):
        assert np.isclose(2 * out["Gamma." + s].sum() - out["gamma0"], out["var." + s], rtol=1e-12)
    assert out["Gamma.pos"][-1] == 0 and out["Gamma.dec"][-1] == 0
    assert abs(out["Gamma.con"][-1]) < 1e-8


def test_metrop_batching_identities(oracle_mod):
    # tests/logitsubbat.R:46-104: blen = 5, nspac = 3; batch row = mean over blen of every nspac-th state
    X, y = load_logit()
    tg = oracle_mod.Target(abi.T_LOGISTIC, 5, x=X, y=y, prior_c=0.125)
    S = 0.1 * np.linalg.cholesky(np.linalg.inv(X.T @ X / 4 + np.eye(5) / 4))
    for scale in (0.4, np.full(5, 0.3), S):
        o = oracle_mod.metrop(tg, glm_binomial(X, y), 20, 5, 3, scale=scale, rstream=oracle_mod.RStream(7),
                              debug=True, outfun=[2, 3])
        cur, prop, z = o["current"][0], o["proposal"][0], o["z"][0]
        sc = np.asarray(scale)
        want = cur + (z @ sc.T if sc.ndim == 2 else sc * z)
        assert np.max(np.abs(prop - want)) < 1e-15                       # logitmat.R:80 / logitvec.R
        acc = o["debug_accept"][0] != 0
        nxt = np.where(acc[:, None], prop, cur)
        assert np.all(nxt[:-1] == cur[1:]) and np.all(nxt[-1] == o["final"][0])
        kept = nxt[2::3]                                                 # state after every nspac-th iteration
        assert np.allclose(o["batch"][0], kept.reshape(20, 5, 5).mean(axis=1)[:, [1, 2]], rtol=1e-15)
        assert np.allclose(o["accept_batch"][0], acc.reshape(20, 15).mean(axis=1))


def test_temp_par_setup_matches_transcript(oracle_mod):
    """tests/temp-par.Rout.save:37-41 prints the glm fit of the full model on data(foo); tests/temp-par.R:58-60
    requires the density to be finite at every sub-model's MLE.  Pins the foo fixture, the IRLS restatement and
    the LOGISTIC_MODELS descriptor of the oracle (against a direct numpy evaluation of temp-par.R:39-56)."""
    from mcmc_amd import abi
    from rhelpers import temp_par_setup
    X, y, models, masks, betas, nb = temp_par_setup()
    assert np.allclose(betas[7], [0.5772, 0.3362, 0.8475, 1.5143], atol=5e-5)
    assert nb.sum() == 8 * 3 and np.array_equal(nb, nb.T)
    t = oracle_mod.Target(abi.T_LOGISTIC_MODELS, 4, x=X, y=y, prior_c=0.5, par=abi.logistic_models_par(masks))
    got = oracle_mod.logdens(t, np.column_stack([np.arange(1, 9.0), betas]))
    assert np.all(np.isfinite(got))
    for i in range(8):
        eta = X @ (betas[i] * masks[i])
        logp = np.where(eta < 0, eta - np.log1p(np.exp(eta)), -np.log1p(np.exp(-eta)))
        logq = np.where(eta < 0, -np.log1p(np.exp(eta)), -eta - np.log1p(np.exp(-eta)))
        want = logp[y == 1].sum() + logq[y == 0].sum() - np.sum(betas[i] ** 2) / 2
        assert got[i] == pytest.approx(want, rel=1e-13)
    # bfst.Rnw:472-489 variant: N(0, 2^2) prior with its normalising constant and a log pseudo-prior
    lpp = np.linspace(-1, 1, 8)
    cst = -4 * np.log(2 * np.sqrt(2 * np.pi))
    t2 = oracle_mod.Target(abi.T_LOGISTIC_MODELS, 4, x=X, y=y, prior_c=0.125,
                           par=abi.logistic_models_par(masks, lpp, cst))
    got2 = oracle_mod.logdens(t2, np.column_stack([np.arange(1, 9.0), betas]))
    from scipy.stats import norm
    for i in range(8):
        want = got[i] + np.sum(betas[i] ** 2) / 2 + norm.logpdf(betas[i], 0, 2).sum() + lpp[i]
        assert got2[i] == pytest.approx(want, rel=1e-13)
