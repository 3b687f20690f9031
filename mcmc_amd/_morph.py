"""morph -- host side of the variable transformation of morph.metrop() (reference R/morph.R).

The transformation object mirrors the list `morph()` returns in the reference (`transform`, `inverse`,
`log.jacobian`, `lud`, `outfun`, `center`, `f`, `f.inv`).  On the device only `inverse` and `log.jacobian`
run in the hot loop (they are part of the descriptor, include/mcmchip.h); `transform` runs once at set-up and
`inverse` once at the end, here, in numpy.
"""
import math

import numpy as np

MORPH_EXPO, MORPH_SUBEXPO, MORPH_CENTER = 1, 2, 4
_E = math.exp(1.0)


def _exp(x):
    """exp with R's overflow behaviour (Inf, no exception)"""
    try:
        return math.exp(x)
    except OverflowError:
        return math.inf


def _pow(x, y):
    """x^y with R's overflow behaviour"""
    try:
        return x ** y
    except OverflowError:
        return math.inf


def _newton_raphson(f, df, x, x0):
    """R/morph.R:35-47: iterate while the error is >= sqrt(eps), then one more step"""
    err = f(x0) - x
    lim = math.sqrt(np.finfo(float).eps)
    while err >= lim:
        x0 = x0 - err / df(x0)
        err = f(x0) - x
    return x0 - err / df(x0)


class Morph:
    """morph(b, r, p, center) of R/morph.R:121-182"""

    def __init__(self, b=None, r=None, p=None, center=None):
        self.identity = b is None and r is None and p is None and center is None     # morph.R:122-123
        self.use_subexpo = b is not None                                             # morph.R:125
        self.use_expo = not (r is None and p is None)                                # morph.R:126
        self.b = 1.0 if b is None else float(b)
        self.p = 3.0 if p is None else float(p)                                      # morph.R:73
        self.r = 0.0 if r is None else float(r)                                      # morph.R:74
        if self.use_subexpo and not self.b > 0:
            raise ValueError("b > 0 is not TRUE")                                    # morph.R:51
        if self.use_expo and not self.p > 2:
            raise ValueError("p > 2 is not TRUE")                                    # morph.R:75
        if self.use_expo and not self.r >= 0:
            raise ValueError("r >= 0 is not TRUE")                                   # morph.R:76
        self.center = 0.0 if center is None else np.asarray(center, dtype=np.float64)
        self.has_center = center is not None and bool(np.any(self.center != 0))

    # ---- scalar pieces -------------------------------------------------------------------------
    def _expo_finv(self, x):                       # morph.R:77
        return x if x <= self.r else x + _pow(x - self.r, self.p)

    def _expo_dfinv(self, x):                      # morph.R:78
        return 1.0 if x <= self.r else 1.0 + self.p * _pow(x - self.r, self.p - 1.0)

    def _expo_f(self, x):                          # morph.R:79-91
        r, p = self.r, self.p
        if x < r:
            return x
        if p == 3:
            n = math.sqrt((27 * r - 27 * x) ** 2 + 108) + 27 * (r - x)
            return r + (2 / n) ** (1 / 3) - (n / 2) ** (1 / 3) / 3
        return _newton_raphson(self._expo_finv, self._expo_dfinv, x, r + x ** (1 / p))

    def _sub_finv(self, x):                        # morph.R:53-55
        b = self.b
        return _exp(b * x) - _E / 3 if x > 1 / b else (x * b) ** 3 * _E / 6 + x * b * _E / 2

    def _sub_dfinv(self, x):                       # morph.R:56-58
        b = self.b
        return b * _exp(b * x) if x > 1 / b else b * (x * b) ** 2 * _E / 2 + b * _E / 2

    def _sub_f(self, x):                           # morph.R:59-68
        b = self.b
        if x > 2 * _E / 3:
            return math.log(x + _E / 3) / b
        poly_inv = math.exp(1 / 3) * (math.sqrt(b ** 12 * (9 * x * x + math.exp(2))) - 3 * b ** 6 * x) ** (-1 / 3)
        return poly_inv * b - 1 / (poly_inv * b ** 3)

    def f_scalar(self, x):                         # morph.R:128-152
        if self.use_expo and self.use_subexpo:
            return self._expo_f(self._sub_f(x))
        if self.use_expo:
            return self._expo_f(x)
        if self.use_subexpo:
            return self._sub_f(x)
        return x

    def f_inv_scalar(self, x):
        if self.use_expo and self.use_subexpo:
            return self._sub_finv(self._expo_finv(x))
        if self.use_expo:
            return self._expo_finv(x)
        if self.use_subexpo:
            return self._sub_finv(x)
        return x

    def d_f_inv_scalar(self, x):
        if self.use_expo and self.use_subexpo:
            return self._expo_dfinv(x) * self._sub_dfinv(self._expo_finv(x))
        if self.use_expo:
            return self._expo_dfinv(x)
        if self.use_subexpo:
            return self._sub_dfinv(x)
        return 1.0

    # ---- isotropic vector maps (morph.R:5-28,154-162) ---------------------------------------------
    def _iso(self, g, x):
        x = np.atleast_1d(np.asarray(x, dtype=np.float64))
        if not (self.use_expo or self.use_subexpo):
            return x.copy()
        nrm = math.sqrt(float(np.sum(x * x)))
        if nrm == 0:
            return np.zeros_like(x)
        return g(nrm) * x / nrm

    def f(self, x):
        return self._iso(self.f_scalar, x)

    def f_inv(self, x):
        return self._iso(self.f_inv_scalar, x)

    def transform(self, state):                    # morph.R:161
        return self.f(np.atleast_1d(np.asarray(state, dtype=np.float64)) - self.center)

    def inverse(self, state):                      # morph.R:162
        return self.f_inv(state) + self.center

    def log_jacobian(self, state):                 # morph.R:16-28
        if not (self.use_expo or self.use_subexpo):
            return 0.0
        x = np.atleast_1d(np.asarray(state, dtype=np.float64))
        nrm = math.sqrt(float(np.sum(x * x)))
        k = x.size
        if nrm == 0:
            return k * math.log(self.d_f_inv_scalar(nrm))
        return math.log(self.d_f_inv_scalar(nrm)) + (k - 1) * (math.log(self.f_inv_scalar(nrm)) - math.log(nrm))

    def lud(self, fn):                             # morph.R:166-179
        def morphed(state, *args, **kw):
            foo = fn(self.inverse(state), *args, **kw)
            if np.size(foo) != 1:
                raise ValueError("log unnormalized density function returned vector not scalar")
            foo = float(foo)
            if math.isnan(foo):
                raise ValueError("log unnormalized density function returned NA or NaN")
            if foo == -math.inf:
                return foo
            if not math.isfinite(foo):
                raise ValueError("log unnormalized density function returned +Inf")
            return foo + self.log_jacobian(state)
        return morphed

    def outfun(self, f):                           # morph.R:95-106
        if f is None:
            return self.inverse
        if callable(f):
            return lambda state, *a, **kw: f(self.inverse(state), *a, **kw)
        idx = np.asarray(f)
        if idx.dtype == bool:
            return lambda state: self.inverse(state)[idx]
        return lambda state: self.inverse(state)[idx - 1]      # 1-based positions, as R indexes

    # ---- descriptor fields -----------------------------------------------------------------------
    def flags(self):
        return ((MORPH_EXPO if self.use_expo else 0) | (MORPH_SUBEXPO if self.use_subexpo else 0)
                | (MORPH_CENTER if self.has_center else 0))

    def center_vector(self, d):
        return np.ascontiguousarray(np.broadcast_to(self.center, (d,)), dtype=np.float64)


def morph(b=None, r=None, p=None, center=None):
    return Morph(b, r, p, center)


def morph_identity():
    return Morph()
