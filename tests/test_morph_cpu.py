"""morph (R/morph.R): the host-side transformation object and the oracle's C restatement, checked with the
reference's own tests -- tests/morph.R, tests/morphtoo.R, tests/isotropic.R -- restated line by line.
No GPU needed."""
import math

import numpy as np
import pytest

from mcmc_amd import abi
from mcmc_amd._morph import Morph, morph, morph_identity


def all_equal(a, b, tol=1.5e-8):
    """R's all.equal for numerics: mean relative difference below tolerance"""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    den = np.mean(np.abs(a))
    diff = np.mean(np.abs(a - b))
    return (diff / den if den > 0 else diff) < tol


def dnorm_log(x, mean=0.0):
    return -0.5 * math.log(2 * math.pi) - 0.5 * (x - mean) ** 2


def scal(f):
    return lambda v: float(np.asarray(f(v)).ravel()[0])


# ----------------------------------------------------------------------------- tests/morph.R
@pytest.mark.parametrize("m", [morph(), morph_identity()])
def test_morph_identity(m):
    # morph.R:6-23
    assert np.array_equal(m.transform(np.arange(1, 11)), np.arange(1.0, 11))
    assert np.array_equal(m.inverse(np.arange(1, 11)), np.arange(1.0, 11))
    x = np.linspace(-1, 1, 15)
    lud = m.lud(lambda v: dnorm_log(float(v[0])))
    assert all_equal([lud([v]) for v in x], [dnorm_log(v) for v in x])
    assert all_equal(m.outfun(lambda v: v)(x), x)
    # morph.R:25-42: outfun NULL / positions / logical
    x = np.arange(1.0, 21)
    assert np.array_equal(m.outfun(None)(x), x)
    assert np.array_equal(m.outfun(np.array([6, 8]))(x), x[[5, 7]])
    i = np.zeros(20, dtype=bool)
    i[[0, 2, 4]] = True
    assert np.array_equal(m.outfun(i)(x), x[i])


def test_morph_center():
    # morph.R:47-68
    x = np.linspace(-1, 1, 15)
    mc = morph(center=2)
    assert all_equal([scal(mc.transform)([v]) for v in x], x - 2)
    assert all_equal([scal(mc.inverse)([v]) for v in x], x + 2)
    lud = mc.lud(lambda v: dnorm_log(float(v[0])))
    assert all_equal([lud([v]) for v in x], [dnorm_log(v, mean=-2) for v in x])
    center = np.arange(1.0, 5)
    mc = morph(center=center)
    z = np.zeros(4)
    assert np.array_equal(mc.transform(z), z - center)
    assert np.array_equal(mc.inverse(z), z + center)
    f = lambda v: float(np.prod([dnorm_log(t) for t in v]))      # the reference's (odd) prod of logs
    assert mc.lud(f)(z) == pytest.approx(f(z + center), rel=1e-15)


def test_morph_r_and_round_trips():
    # morph.R:69-89: inside radius r the density is untouched, outside it is not
    mr = morph(r=1)
    lud = mr.lud(lambda v: dnorm_log(float(v[0])))
    x = np.linspace(-1, 1, 20)
    assert all_equal([lud([v]) for v in x], [dnorm_log(v) for v in x])
    x = np.linspace(1.1, 2, 10)
    assert all(lud([v]) != dnorm_log(v) for v in x)
    mr = morph(r=0.3, p=2.2)
    x = np.linspace(0, 0.3, 20)
    assert all_equal(x, [scal(mr.inverse)([v]) for v in x])
    # morph.R:119-133
    x = np.linspace(0, 10, 100)
    for kw in (dict(b=3), dict(p=3), dict(p=3, b=1)):
        f = morph(**kw)
        assert all_equal(x, [scal(f.inverse)(f.transform([v])) for v in x]), kw
    f = morph(p=3, r=10)
    v = np.arange(-10.0, 11)
    assert all_equal(v, [scal(f.transform)([t]) for t in v])
    assert sorted(vars(morph_identity())) == sorted(vars(morph(b=1)))     # morph.R:45: same members


def test_morph_argument_rules():
    # morph.R:51,75-76
    with pytest.raises(ValueError, match="b > 0"):
        morph(b=0)
    with pytest.raises(ValueError, match="p > 2"):
        morph(p=2)
    with pytest.raises(ValueError, match="r >= 0"):
        morph(r=-1)
    assert morph(p=3).r == 0.0 and morph(r=1).p == 3.0                   # morph.R:73-74


# ----------------------------------------------------------------------------- tests/morphtoo.R
CASES = [(dict(b=0.5), 1e-14, 1e-14), (dict(r=5), 1e-14, 1e-12), (dict(b=0.5, r=5), 1e-14, 1e-12),
         (dict(r=5, p=2.2), 1e-14, 1e-14)]


def closed_form(kw, x):
    b, r, p = kw.get("b"), kw.get("r"), kw.get("p", 3)
    y = x.copy()
    if r is not None:
        y = np.where(y < r, y, y + np.abs(y - r) ** p)
    if b is not None:
        e = math.exp(1)
        y = np.where(y > 1 / b, np.exp(b * y) - e / 3, (y * b) ** 3 * e / 6 + y * b * e / 2)
    return y


@pytest.mark.parametrize("kw,tol_inv,tol_rt", CASES)
def test_morphtoo_host(kw, tol_inv, tol_rt):
    x = np.linspace(0, 10, 10001)
    m = morph(**kw)
    y = np.array([scal(m.inverse)([v]) for v in x])
    assert all_equal(y, closed_form(kw, x), tol_inv)
    z = np.array([scal(m.transform)([v]) for v in y])
    assert all_equal(z, x, tol_rt)


@pytest.mark.parametrize("kw,tol_inv,tol_rt", CASES)
def test_morphtoo_oracle(oracle_mod, kw, tol_inv, tol_rt):
    x = np.linspace(0, 10, 10001)
    t = oracle_mod.Target(abi.T_STUDENT_T, 1, par=[3.0], morph=morph(**kw))
    y = np.array([oracle_mod.morph_inverse(t, [v])[0] for v in x])
    assert all_equal(y, closed_form(kw, x), tol_inv)
    z = np.array([oracle_mod.morph_transform(t, [v])[0] for v in y])
    assert all_equal(z, x, tol_rt)


# ----------------------------------------------------------------------------- tests/isotropic.R
def test_isotropic_pieces():
    m = Morph()
    for k in (1, 2, 17, 1000):
        assert m.f(np.ones(k)).shape == (k,)                   # length preserving
    assert np.array_equal(m.f(np.arange(1.0, 11)), np.arange(1.0, 11))
    assert m.log_jacobian(np.arange(1.0, 11)) == 0
    for mm in (morph(b=2), morph(r=1, p=3), morph(b=1, r=0.5)):
        assert np.array_equal(mm.f_inv(np.zeros(4)), np.zeros(4))       # no divide by zero at the origin
        assert np.array_equal(mm.f(np.zeros(1)), np.zeros(1))
        assert math.isfinite(mm.log_jacobian(np.zeros(4)))
    # isotropic(f)(x) = f(|x|) x/|x|: direction preserved, norm mapped by the scalar map
    mm = morph(b=0.7, r=1.5, p=3)
    v = np.array([1.0, -2.0, 0.5])
    w = mm.f_inv(v)
    nv = np.linalg.norm(v)
    assert np.allclose(w / np.linalg.norm(w), v / nv, rtol=1e-15)
    assert np.linalg.norm(w) == pytest.approx(mm.f_inv_scalar(nv), rel=1e-15)


# ----------------------------------------------------------------------------- host object vs oracle
@pytest.mark.parametrize("kw", [dict(b=0.5), dict(r=1.0), dict(b=2.0, r=0.5, p=3), dict(r=0.3, p=2.2),
                                dict(b=1.0, center=[0.5, -1.0, 2.0]), dict(center=[1.0, 2.0, 3.0])])
def test_host_morph_matches_oracle(oracle_mod, kw):
    rng = np.random.default_rng(3)
    m = morph(**kw)
    t = oracle_mod.Target(abi.T_STUDENT_T, 3, par=[3.0], morph=m)
    for scale_ in (0.01, 0.5, 1.5):
        for _ in range(20):
            v = scale_ * rng.standard_normal(3)
            assert np.allclose(oracle_mod.morph_inverse(t, v), m.inverse(v), rtol=1e-14, atol=1e-300)
            assert np.allclose(oracle_mod.morph_transform(t, v), m.transform(v), rtol=1e-13, atol=1e-300)
            assert oracle_mod.morph_logjac(t, v) == pytest.approx(m.log_jacobian(v), rel=1e-13, abs=1e-15)
    # the induced log density (morph.R:166-179) against the host object's lud()
    base = oracle_mod.Target(abi.T_STUDENT_T, 3, par=[3.0])
    lud = m.lud(lambda s: oracle_mod.logdens(base, s)[0])
    for _ in range(20):
        v = rng.standard_normal(3)
        assert oracle_mod.logdens(t, v)[0] == pytest.approx(lud(v), rel=1e-13)


def test_log_jacobian_is_the_log_determinant():
    # numerical check of morph.R:16-28 against a finite-difference Jacobian of inverse()
    m = morph(b=0.8, r=0.7, p=3)
    v = np.array([0.9, -0.4, 1.3])
    h = 1e-6
    J = np.empty((3, 3))
    for j in range(3):
        e = np.zeros(3)
        e[j] = h
        J[:, j] = (m.inverse(v + e) - m.inverse(v - e)) / (2 * h)
    assert math.log(abs(np.linalg.det(J))) == pytest.approx(m.log_jacobian(v), rel=1e-7)
