"""rglue/glue_core.c (the R-independent part of the R glue) through ctypes: argument rules and error texts of the
reference's C (metrop.c:333-365,416-503; temper.c:29-64,915-953; getScalarInteger.c), result-slot names and order,
and the descriptor marshalling (must produce what mcmc_amd.abi produces, which the GPU tests run on the device)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mcmc_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Fields(C.Structure):
    _fields_ = [("kind", C.c_char_p), ("d", C.c_int),
                ("x", C.c_void_p), ("x_nrow", C.c_int), ("x_ncol", C.c_int),
                ("y", C.c_void_p), ("y_len", C.c_int), ("prior_sd", C.c_double),
                ("mu", C.c_void_p), ("precision", C.c_void_p),
                ("nmix", C.c_int), ("weights", C.c_void_p), ("means", C.c_void_p), ("sds", C.c_void_p),
                ("bounds", C.c_void_p), ("betas", C.c_void_p), ("ncomp", C.c_int),
                ("models", C.c_void_p), ("log_pseudo_prior", C.c_void_p), ("constant", C.c_double),
                ("df", C.c_double), ("totals", C.c_void_p), ("simplex_value", C.c_double),
                ("has_b", C.c_int), ("has_r", C.c_int), ("has_center", C.c_int),
                ("morph_b", C.c_double), ("morph_r", C.c_double), ("morph_p", C.c_double),
                ("morph_center", C.c_void_p), ("morph_center_len", C.c_int)]


class Descriptor(C.Structure):
    _fields_ = [("desc", abi.TargetDesc), ("x_rowmajor", C.c_void_p), ("par", C.c_void_p), ("center", C.c_void_p)]


@pytest.fixture(scope="module")
def G():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "rglue")])
    L = C.CDLL(os.path.join(ROOT, "rglue", "libglue_core.so"))
    i, p, d, l = C.c_int, C.c_void_p, C.c_double, C.c_long
    L.glue_scalar_integer.argtypes = [i, i, d, C.c_char_p, i, p, p, i]
    L.glue_scalar_logical.argtypes = [i, i, i, C.c_char_p, p, p, i]
    L.glue_metrop_scale.argtypes = [i, i, i, i, i, i, p, p, i]
    L.glue_temper_scale.argtypes = [i, p, i, i, i, i, i, i, p, p, i]
    L.glue_metrop_outfun.argtypes = [i, p, p, i, i, p, p, p, p, i]
    L.glue_neighbors.argtypes = [i, i, i, i, p, p, p, i]
    L.glue_temper_initial.argtypes = [i, p, l, i, i, i, i, i, p, p, i]
    L.glue_make_descriptor.argtypes = [p, p, p, i]
    L.glue_free_descriptor.argtypes = [p]
    L.glue_metrop_result_names.argtypes = [i, p]
    L.glue_temper_result_names.argtypes = [i, i, p]
    L.glue_initseq_result_names.argtypes = [p]
    return L


def _err():
    return C.create_string_buffer(256)


def test_scalar_argument_rules(G):
    out, e = C.c_int(), _err()
    assert G.glue_scalar_integer(1, 1, C.c_double(7.9), b"nbatch", 1, C.byref(out), e, 256) == 0 and out.value == 7
    assert G.glue_scalar_integer(0, 1, C.c_double(1.0), b"nbatch", 1, C.byref(out), e, 256) == 1
    assert e.value == b'argument "nbatch" must be numeric'
    assert G.glue_scalar_integer(1, 2, C.c_double(1.0), b"blen", 1, C.byref(out), e, 256) == 1
    assert e.value == b'argument "blen" must be scalar'
    assert G.glue_scalar_integer(1, 1, C.c_double(0.0), b"nspac", 1, C.byref(out), e, 256) == 1
    assert e.value == b'argument "nspac" must be positive'
    assert G.glue_scalar_integer(1, 1, C.c_double(float("nan")), b"nbatch", 1, C.byref(out), e, 256) == 1
    assert G.glue_scalar_logical(0, 1, 1, b"debug", C.byref(out), e, 256) == 1 and e.value == b'argument "debug" must be logical'


def test_metrop_scale_classification(G):
    k, e = C.c_int(), _err()
    f = G.glue_metrop_scale
    assert f(1, 1, 0, 0, 0, 5, C.byref(k), e, 256) == 0 and k.value == abi.SCALE_CONSTANT
    assert f(1, 5, 0, 0, 0, 5, C.byref(k), e, 256) == 0 and k.value == abi.SCALE_DIAGONAL
    assert f(1, 25, 1, 5, 5, 5, C.byref(k), e, 256) == 0 and k.value == abi.SCALE_FULL
    assert f(1, 1, 0, 0, 0, 1, C.byref(k), e, 256) == 0 and k.value == abi.SCALE_DIAGONAL     # LENGTH == d is tested first
    assert f(1, 6, 1, 2, 3, 5, C.byref(k), e, 256) == 1 and e.value == b'dimensions of "scale" matrix not d by d'
    assert f(1, 3, 0, 0, 0, 5, C.byref(k), e, 256) == 1 and e.value == b'length of "scale" vector not d or 1'
    assert f(0, 1, 0, 0, 0, 5, C.byref(k), e, 256) == 1 and e.value == b'argument "scale" must be numeric'
    # and the Python mirror classifies the same way
    assert abi.classify_scale(0.4, 5)[0] == abi.SCALE_CONSTANT and abi.classify_scale(np.ones(5), 5)[0] == abi.SCALE_DIAGONAL


def test_temper_scale_rules(G):
    k, e = C.c_int(), _err()
    f = G.glue_temper_scale
    v = np.array([0.5, 0.25, 0.125, 1.0])
    assert f(1, v.ctypes.data, 1, 0, 0, 0, -1, 3, C.byref(k), e, 256) == 0 and k.value == abi.SCALE_CONSTANT
    assert f(1, v.ctypes.data, 3, 0, 0, 0, 2, 3, C.byref(k), e, 256) == 0 and k.value == abi.SCALE_DIAGONAL
    assert f(1, v.ctypes.data, 4, 1, 2, 2, -1, 2, C.byref(k), e, 256) == 0 and k.value == abi.SCALE_FULL
    assert f(0, v.ctypes.data, 1, 0, 0, 0, -1, 3, C.byref(k), e, 256) == 1 and e.value == b"scale not type double"
    assert f(0, v.ctypes.data, 1, 0, 0, 0, 4, 3, C.byref(k), e, 256) == 1 and e.value == b"component 5 of scale not type double"
    bad = np.array([1.0, np.inf])
    assert f(1, bad.ctypes.data, 2, 0, 0, 0, -1, 2, C.byref(k), e, 256) == 1 and e.value == b"scale has non-finite element"
    assert f(1, v.ctypes.data, 4, 1, 2, 2, 1, 3, C.byref(k), e, 256) == 1 and e.value == b"component 2 of scale matrix with wrong row dim"
    assert f(1, v.ctypes.data, 4, 1, 3, 2, -1, 3, C.byref(k), e, 256) == 1 and e.value == b"scale matrix with wrong col dim"
    assert f(1, v.ctypes.data, 2, 0, 0, 0, -1, 3, C.byref(k), e, 256) == 1
    assert e.value == b"scale not matrix, scalar, or vector of length k"


@pytest.mark.parametrize("outfun", [None, [2, 3], [-2], [-1, -5], [True, False, True, False, True], [4], "moments2"])
def test_outfun_masks_match_python_mirror(G, outfun):
    d = 5
    kind, nout, e = C.c_int(), C.c_int(), _err()
    mask = np.full(d, -7, dtype=np.int32)
    if outfun is None:
        rc = G.glue_metrop_outfun(0, None, None, 0, d, C.byref(kind), mask.ctypes.data, C.byref(nout), e, 256)
    elif outfun == "moments2":
        rc = G.glue_metrop_outfun(5, None, None, 0, d, C.byref(kind), mask.ctypes.data, C.byref(nout), e, 256)
    elif isinstance(outfun[0], bool):
        lg = np.array(outfun, dtype=np.int32)
        rc = G.glue_metrop_outfun(1, lg.ctypes.data, None, len(lg), d, C.byref(kind), mask.ctypes.data, C.byref(nout), e, 256)
    else:
        nm = np.array(outfun, dtype=np.float64)
        rc = G.glue_metrop_outfun(2, None, nm.ctypes.data, len(nm), d, C.byref(kind), mask.ctypes.data, C.byref(nout), e, 256)
    assert rc == 0, e.value
    want_kind, want_mask, want_nout = abi.resolve_outfun(outfun, d)
    assert (kind.value, nout.value) == (want_kind, want_nout)
    if want_mask is not None:
        assert np.array_equal(mask, want_mask)


def test_outfun_errors_are_the_references(G):
    d = 5
    kind, nout, e = C.c_int(), C.c_int(), _err()
    mask = np.zeros(d, dtype=np.int32)

    def num(vals):
        a = np.array(vals, dtype=np.float64)
        rc = G.glue_metrop_outfun(2, None, a.ctypes.data, len(a), d, C.byref(kind), mask.ctypes.data, C.byref(nout), e, 256)
        return rc, e.value
    assert num([0, 1]) == (1, b"is.numeric(outfun) & any(outfun == 0)")
    assert num([1.5]) == (1, b"is.numeric(outfun) & any(outfun != as.integer(outfun))")
    assert num([6]) == (1, b"is.numeric(outfun) & any(abs(outfun) > length(initial)")
    assert num([1, -2]) == (1, b"is.numeric(outfun) & any(outfun > 0) & any(outfun < 0)")
    assert num([float("nan")]) == (1, b"NA or NaN index for outfun")
    # beyond int: no undefined double -> int conversion, the error x86-64 R ends in (metrop.c:457-463)
    assert num([1e10]) == (1, b"is.numeric(outfun) & any(outfun != as.integer(outfun))")
    assert num([-3e9]) == (1, b"is.numeric(outfun) & any(outfun != as.integer(outfun))")
    assert num([float("inf")]) == (1, b"-Inf or Inf index for outfun")
    lg = np.ones(4, dtype=np.int32)
    assert G.glue_metrop_outfun(1, lg.ctypes.data, None, 4, d, C.byref(kind), mask.ctypes.data, C.byref(nout), e, 256) == 1
    assert e.value == b"is.logical(outfun) & (length(outfun) != length(initial))"
    assert G.glue_metrop_outfun(4, None, None, 0, d, C.byref(kind), mask.ctypes.data, C.byref(nout), e, 256) == 1
    assert e.value == b"outfun must be NULL, a function, a numeric vector, or a logical vector"
    assert G.glue_metrop_outfun(3, None, None, 0, d, C.byref(kind), mask.ctypes.data, C.byref(nout), e, 256) == 1
    assert b"cannot run on the device" in e.value


def test_result_slot_names_and_order(G):
    names = (C.c_char_p * 16)()
    n = G.glue_metrop_result_names(0, names)
    assert [names[i].decode() for i in range(n)] == ["accept", "batch", "initial", "final", "accept.batch"]      # metrop.c:124-128
    n = G.glue_metrop_result_names(1, names)
    assert [names[i].decode() for i in range(n)] == ["accept", "batch", "initial", "final", "accept.batch", "current",
                                                     "proposal", "log.green", "u", "z", "debug.accept"]         # 174-179
    n = G.glue_temper_result_names(1, 0, names)
    assert [names[i].decode() for i in range(n)] == ["batch", "acceptx", "accepti", "initial", "final"]          # temper.c:196-263
    n = G.glue_temper_result_names(0, 0, names)
    assert [names[i].decode() for i in range(n)][-1] == "ibatch" and n == 6
    n = G.glue_temper_result_names(1, 1, names)
    assert n == 15 and [names[i].decode() for i in range(5, n)] == [
        "which", "unif.which", "state", "log.hastings", "unif.hastings", "proposal", "acceptd", "norm", "unif.choose",
        "coproposal"]                                                                                            # temp-par.Rout.save:110-116
    assert G.glue_temper_result_names(0, 1, names) == 15
    n = G.glue_initseq_result_names(names)
    assert [names[i].decode() for i in range(n)] == ["gamma0", "Gamma.pos", "Gamma.dec", "Gamma.con", "var.pos", "var.dec", "var.con"]


def test_neighbors_and_initial_rules(G):
    e = _err()
    k = 3
    nb = np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]], dtype=np.int32)
    out = np.zeros(9, dtype=np.int32)
    assert G.glue_neighbors(1, 1, k, k, nb.ctypes.data, out.ctypes.data, e, 256) == 0 and np.array_equal(out, nb.ravel())
    assert G.glue_neighbors(0, 1, k, k, nb.ctypes.data, out.ctypes.data, e, 256) == 1 and e.value == b'argument "neighbors" must be logical'
    assert G.glue_neighbors(1, 1, 2, 3, nb.ctypes.data, out.ctypes.data, e, 256) == 1
    assert e.value == b'argument "neighbors" must have same row and column dimension'
    bad = nb.copy()
    bad[0, 2] = 1
    assert G.glue_neighbors(1, 1, k, k, bad.ctypes.data, out.ctypes.data, e, 256) == 1 and e.value == b'argument "neighbors" must be symmetric matrix'
    one = np.zeros(1, dtype=np.int32)
    assert G.glue_neighbors(1, 1, 1, 1, one.ctypes.data, out.ctypes.data, e, 256) == 1 and e.value == b"must have at least two components"
    nx = C.c_int()
    ser = np.array([2.0, 0.1, 0.2])
    assert G.glue_temper_initial(1, ser.ctypes.data, C.c_long(3), 0, 0, 0, 0, k, C.byref(nx), e, 256) == 0 and nx.value == 2
    ser[0] = 2.5
    assert G.glue_temper_initial(1, ser.ctypes.data, C.c_long(3), 0, 0, 0, 0, k, C.byref(nx), e, 256) == 1
    assert e.value == b'1st elem of argument "initial" must be integer in serial case'
    ser[0] = 4
    assert G.glue_temper_initial(1, ser.ctypes.data, C.c_long(3), 0, 0, 0, 0, k, C.byref(nx), e, 256) == 1
    assert e.value == b'1st elem of argument "initial" must be in 1, ..., k in serial case'
    par = np.zeros(6)
    assert G.glue_temper_initial(1, par.ctypes.data, C.c_long(6), 1, 3, 2, 1, k, C.byref(nx), e, 256) == 0 and nx.value == 2
    assert G.glue_temper_initial(1, par.ctypes.data, C.c_long(6), 1, 2, 3, 1, k, C.byref(nx), e, 256) == 1
    assert e.value == b'row dim of args "initial" and "neighbors" must be same in parallel case'
    assert G.glue_temper_initial(1, par.ctypes.data, C.c_long(6), 0, 0, 0, 1, k, C.byref(nx), e, 256) == 1
    assert e.value == b'argument "initial" must be matrix in parallel case'


def _make(G, **kw):
    f = Fields()
    keep = []
    for key, val in kw.items():
        if isinstance(val, np.ndarray):
            keep.append(val)
            setattr(f, key, val.ctypes.data)
        else:
            setattr(f, key, val)
    g, e = Descriptor(), _err()
    rc = G.glue_make_descriptor(C.byref(f), C.byref(g), e, 256)
    return rc, g, e.value, keep


def _par(g):
    return np.ctypeslib.as_array(C.cast(g.desc.par, C.POINTER(C.c_double)), shape=(g.desc.npar,)).copy()


def test_descriptor_marshalling_matches_python_mirror(G):
    rng = np.random.default_rng(0)
    n, d = 7, 3
    X = rng.standard_normal((n, d))
    y = (rng.random(n) < 0.5).astype(float)
    Xcol = np.asfortranarray(X)                      # what R holds
    rc, g, msg, keep = _make(G, kind=b"logistic", x=np.ravel(Xcol, order="K").copy(), x_nrow=n, x_ncol=d, y=y, y_len=n, prior_sd=2.0)
    assert rc == 0, msg
    assert (g.desc.kind, g.desc.d, g.desc.n, g.desc.prior_c) == (abi.T_LOGISTIC, d, n, 0.125)
    xr = np.ctypeslib.as_array(C.cast(g.x_rowmajor, C.POINTER(C.c_double)), shape=(n, d))
    assert np.array_equal(xr, X)                     # row-major, as the C-ABI wants it
    G.glue_free_descriptor(C.byref(g))
    # normal mixture: the parameter block of abi.normmix_par
    means = rng.standard_normal((3, d))
    w, s = np.array([0.5, 0.3, 0.2]), np.array([1.0, 0.5, 2.0])
    rc, g, msg, keep = _make(G, kind=b"normmix", d=d, nmix=3, weights=w, means=np.ravel(np.asfortranarray(means), order="K").copy(), sds=s)
    assert rc == 0, msg
    assert np.allclose(_par(g), abi.normmix_par(w, means, s), rtol=1e-15, atol=0)
    G.glue_free_descriptor(C.byref(g))
    # tempered mixture and the logistic sub-models block of abi.logistic_models_par
    betas = np.array([1.0, 0.5, 0.1])
    rc, g, msg, keep = _make(G, kind=b"tempered.normmix", d=d, nmix=3, weights=w, means=np.ravel(np.asfortranarray(means), order="K").copy(),
                             sds=s, betas=betas, ncomp=3)
    assert rc == 0, msg
    assert np.allclose(_par(g), np.concatenate([[3.0], betas, abi.normmix_par(w, means, s)]), rtol=1e-15, atol=0)
    G.glue_free_descriptor(C.byref(g))
    masks = np.array([[1, 0, 0], [1, 1, 0], [1, 0, 1], [1, 1, 1]], dtype=np.int32)
    lpp = np.array([0.0, 0.4, 0.1, 0.9])
    rc, g, msg, keep = _make(G, kind=b"logistic.models", x=np.ravel(Xcol, order="K").copy(), x_nrow=n, x_ncol=d, y=y, y_len=n,
                             prior_sd=2.0, models=np.ravel(np.asfortranarray(masks), order="K").copy(), ncomp=4,
                             log_pseudo_prior=lpp, constant=-1.5)
    assert rc == 0, msg
    assert np.array_equal(_par(g), abi.logistic_models_par(masks, lpp, -1.5))
    G.glue_free_descriptor(C.byref(g))
    # morph flags and argument rules (R/morph.R:51,75-76)
    rc, g, msg, keep = _make(G, kind=b"student.t", d=2, df=3.0, has_b=1, morph_b=0.7, has_r=1, morph_r=0.5, morph_p=3.0,
                             has_center=1, morph_center=np.array([1.5]), morph_center_len=1)
    assert rc == 0 and g.desc.morph_flags == 7 and (g.desc.morph_b, g.desc.morph_r, g.desc.morph_p) == (0.7, 0.5, 3.0)
    assert list(np.ctypeslib.as_array(C.cast(g.center, C.POINTER(C.c_double)), shape=(2,))) == [1.5, 1.5]
    G.glue_free_descriptor(C.byref(g))
    assert _make(G, kind=b"student.t", d=2, df=3.0, has_b=1, morph_b=-1.0)[2] == b"b > 0 is not TRUE"
    assert _make(G, kind=b"student.t", d=2, df=3.0, has_r=1, morph_r=0.5, morph_p=2.0)[2] == b"p > 2 is not TRUE"
    assert _make(G, kind=b"witch", d=2, bounds=np.array([0.5, 0.25]), ncomp=2, has_b=1, morph_b=1.0)[2] == b"morph applies to metrop targets only"
    assert _make(G, kind=b"nonsense", d=2)[2] == b'unknown descriptor kind "nonsense"'
    assert _make(G, kind=b"witch", d=2, bounds=np.array([0.5, 1.25]), ncomp=2)[2] == b"witch.which[2] must be in (0, 1)"


def test_glue_sources_are_complete():
    """every identifier the SEXP wrappers use from this repository is declared: the glue_* functions in glue_core.h /
    glue_r.h, the mcmchip_* entry points in include/mcmchip.h (R's own API aside, which is absent here)"""
    import re
    decl = ""
    for h in ("rglue/glue_core.h", "rglue/glue_r.h", "include/mcmchip.h"):
        decl += open(os.path.join(ROOT, h)).read()
    for src in ("glue_r.c", "metrop_descriptor.c", "temper_descriptor.c", "diag_descriptor.c"):
        text = open(os.path.join(ROOT, "rglue", src)).read()
        assert "..." not in re.sub(r"/\*.*?\*/", "", text, flags=re.S), src          # no elided steps
        for name in set(re.findall(r"\b((?:glue|mcmchip)_[a-z_0-9]+)\s*\(", text)):
            assert re.search(r"\b%s\s*\(" % name, decl), (src, name)
        for name in set(re.findall(r"\b(MCMCHIP_[A-Z_0-9]+|GLUE_[A-Z_0-9]+)\b", text)):
            assert name in decl, (src, name)
