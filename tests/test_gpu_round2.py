"""GPU tests added in round 2 (-m gpu): regression tests for the advisor's findings, the configuration bench.py
times, C5 beyond 2^31 elements, statistical equivalence under the device RNG, chunked persistent kernels."""
import os

import numpy as np
import pytest

from mcmc_amd import abi
from rhelpers import glm_binomial, load_logit

pytestmark = pytest.mark.gpu

RTOL = 1e-12


@pytest.fixture(scope="module")
def api():
    from mcmc_amd import api as a
    a.init(0)
    return a


def rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


# ----------------------------------------------------------------------------- advisor findings (round 1)
@pytest.mark.parametrize("parallel", [True, False])
@pytest.mark.parametrize("k", [4, 9, 12])
def test_temper_true_diagonal_neighbors(api, oracle_mod, parallel, k):
    """temper.c:38-41 only asks for a symmetric matrix: a TRUE diagonal is legal and gives rows of degree up to k
    (a swap of a component with itself always accepts).  Both kernels against the oracle."""
    rng = np.random.default_rng(k)
    p, E, nbatch, blen, nspac = 3, 12, 9, 4, 2
    niter = nbatch * blen * nspac
    par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / p)
    nb = np.ones((k, k), dtype=bool) if k <= 9 else (np.abs(np.subtract.outer(np.arange(k), np.arange(k))) <= 1)
    unif, norm = rng.random((E, niter, 4)), rng.standard_normal((E, niter, p))
    init = np.zeros((E, k, p)) if parallel else np.column_stack([rng.integers(1, k + 1, E).astype(float), np.zeros((E, p))])
    ref = oracle_mod.temper(oracle_mod.Target(abi.T_WITCH, p, par=par), init, nb, nbatch, blen, nspac, 0.4,
                            parallel=parallel, unif=unif, norm=norm)
    paths = [abi.PATH_LDS] + ([abi.PATH_RUN] if parallel and nb.sum(axis=1).max() <= 8 else [])
    for path in paths:
        o = api.temper(api.Target(abi.T_WITCH, p, par=par), init, nb, nbatch, blen, nspac, 0.4, parallel=parallel,
                       unif=unif, norm=norm, _path=path)
        assert np.array_equal(o["final"], ref["final"])
        assert np.array_equal(o["acceptx"], ref["acceptx"], equal_nan=True)
        assert np.array_equal(o["accepti"], ref["accepti"], equal_nan=True)
        assert rel(o["batch"] + 10, ref["batch"] + 10) < RTOL
    if parallel:
        assert np.all(np.nan_to_num(np.diagonal(ref["accepti"], axis1=1, axis2=2), nan=1.0) == 1.0)   # self swaps always accept


def test_temper_component_without_neighbors_is_refused(api):
    k, p = 4, 2
    nb = np.zeros((k, k), dtype=bool)
    nb[0, 1] = nb[1, 0] = nb[1, 2] = nb[2, 1] = True          # component 4 has no neighbour
    par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / p)
    with pytest.raises(api.McmcHipError, match="has no neighbors"):
        api.temper(api.Target(abi.T_WITCH, p, par=par), np.zeros((k, p)), nb, 5, 1, 1, 0.3, parallel=True, seed=1)


@pytest.mark.parametrize("path", [abi.PATH_STREAM, abi.PATH_MFMA])
def test_metrop_empty_outfun(api, oracle_mod, path):
    """an all-FALSE logical outfun is legal (metrop.c:435-444): the batch is nbatch x 0 and nothing is copied back"""
    rng = np.random.default_rng(8)
    n, d, C, nbatch = 2048, 8, 64, 6
    X = rng.standard_normal((n, d)) / 3
    y = (rng.random(n) < 0.5).astype(float)
    init = 0.05 * rng.standard_normal((C, d))
    z, u = rng.standard_normal((C, nbatch, d)), rng.random((C, nbatch))
    guard = np.full(16, 7.25)
    o = api.metrop(api.logistic_target(X, y), init, nbatch, scale=0.05, outfun=np.zeros(d, dtype=bool), z=z, u=u, path=path)
    ref = oracle_mod.metrop(oracle_mod.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.125), init, nbatch, scale=0.05, z=z, u=u)
    assert o["batch"].shape == (C, nbatch, 0)
    assert np.array_equal(o["final"], ref["final"]) and np.array_equal(o["accept"], ref["accept"])
    assert np.all(guard == 7.25)


def test_descriptor_parameter_block_lengths(api):
    d = 3
    with pytest.raises(api.McmcHipError, match="parameter block"):
        api.Target(abi.T_MVNORM, d, par=np.zeros(d + d * d - 1))
    with pytest.raises(api.McmcHipError, match="parameter block"):
        api.Target(abi.T_NORMMIX, d, par=abi.normmix_par([0.5, 0.5], np.zeros((2, d)), 1.0)[:-1])
    with pytest.raises(api.McmcHipError, match="parameter block"):
        api.Target(abi.T_NORMMIX, d, par=np.concatenate([[2.5], np.zeros(2 * (2 + d))]))
    good = np.concatenate([[3.0], [1.0, 0.5, 0.2], abi.normmix_par([1.0], np.zeros((1, d)), 1.0)])
    api.Target(abi.T_TEMPERED_NORMMIX, d, par=good)
    with pytest.raises(api.McmcHipError, match="parameter block"):
        api.Target(abi.T_TEMPERED_NORMMIX, d, par=good[:-2])


def test_device_pointer_inputs_are_validated(api):
    """device_ptrs = 1: the reference's finiteness / component-range checks run on the device"""
    import ctypes as C
    import torch
    L = api.load_library()
    dev = torch.device("cuda", 0)
    X, y = load_logit()
    tg = api.logistic_target(X, y)
    Cn, d, nbatch = 8, 5, 4
    init = torch.zeros((Cn, d), dtype=torch.float64, device=dev)
    init[3, 2] = float("inf")
    outs = [torch.empty(s, dtype=torch.float64, device=dev) for s in ((Cn,), (Cn, nbatch, d), (Cn, d), (Cn, nbatch))]
    scale = np.array([0.4])
    args = abi.MetropArgs(Cn, nbatch, 1, 1, init.data_ptr(), abi.SCALE_CONSTANT, scale.ctypes.data, abi.OUT_IDENTITY, None,
                          abi.RNG_PHILOX, None, None, 1, 0, 0, 0, 1, abi.PATH_STREAM)
    out = abi.MetropOut(*[t.data_ptr() for t in outs], None, None, None, None, None, None, 0, 0, 0)
    assert L.mcmchip_metrop(tg.handle, C.addressof(args), C.addressof(out)) == abi.EINVAL
    assert b'all elements of "state" must be finite' in L.mcmchip_last_error()
    init[3, 2] = 0.0
    assert L.mcmchip_metrop(tg.handle, C.addressof(args), C.addressof(out)) == 0
    # serial tempering: component index out of range / not an integer / non-finite
    k, p, E = 4, 2, 6
    par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / p)
    wt = api.Target(abi.T_WITCH, p, par=par)
    nb = np.ascontiguousarray((np.abs(np.subtract.outer(np.arange(k), np.arange(k))) == 1).astype(np.int32))
    for bad, msg in ((9.0, b"must be in 1, ..., k"), (1.5, b"must be integer"), (float("nan"), b"all elements finite")):
        ini = torch.zeros((E, 1 + p), dtype=torch.float64, device=dev)
        ini[:, 0] = 1.0
        ini[4, 0] = bad
        b_, ib, ax, ai, fin = (torch.empty(s, dtype=torch.float64, device=dev)
                               for s in ((E, p, 3), (E, k, 3), (E, k), (E, k, k), (E, 1 + p)))
        ta = abi.TemperArgs(E, k, 0, 3, 1, 1, nb.ctypes.data, ini.data_ptr(), abi.SCALE_CONSTANT, scale.ctypes.data,
                            abi.RNG_PHILOX, None, None, 1, 0, 0, 1, 0, 0, 0)
        to = abi.TemperOut(b_.data_ptr(), ib.data_ptr(), ax.data_ptr(), ai.data_ptr(), fin.data_ptr(), 0,
                           *([None] * 10))
        assert L.mcmchip_temper(wt.handle, C.addressof(ta), C.addressof(to)) == abi.EINVAL
        assert msg in L.mcmchip_last_error(), L.mcmchip_last_error()
    # initseq: a NaN in a device-resident series
    x = torch.randn(5000, dtype=torch.float64, device=dev)
    x[1234] = float("nan")
    n_out = np.zeros(1, dtype=np.int64)
    g0, vp, vd, vc = (np.empty(1) for _ in range(4))
    Gp, Gd, Gc = (np.empty((1, 64)) for _ in range(3))
    rc = L.mcmchip_initseq(x.data_ptr(), 5000, 1, 1, 64, n_out.ctypes.data, g0.ctypes.data, Gp.ctypes.data, Gd.ctypes.data,
                           Gc.ctypes.data, vp.ctypes.data, vd.ctypes.data, vc.ctypes.data)
    assert rc == abi.EINVAL and b"all elements of argument must be finite" in L.mcmchip_last_error()


@pytest.mark.parametrize("Cn,nbatch,blen,nspac,thread_counts", [
    (4099, 523, 1, 1, (None, "1", "0")),        # 85.75 MB of batch means: 5 whole 16 MB chunks and a ragged one
    # the last chunk is 16 * 4096 * 242 + 8 bytes: with 16 copy threads floor(len / T) is a whole number of pages and
    # T slices of that size stop 8 bytes short of the chunk (the last double was never copied before the slice size
    # became ceil(len / T))
    (4191, 595, 1, 1, ("16", "12", "32")),
    # batches of 3 x 2 iterations: the 256-iteration calibration launch and the launches after it end inside a batch, and the
    # pieces that go home under the next launch (metrop_stream.hip) are whole batches only; "piecewise off": the copy behind the run
    (4099, 523, 3, 2, (None, "0", "piecewise off")),
    # launches capped at 40 iterations: thirteen pieces after the first launch, one launch queued ahead of each copy -- more
    # pieces than the run keeps events for, so the events are reused; capped at 240: four pieces of 71 iterations, three queued ahead
    (4099, 523, 1, 1, ("limit 40", "limit 240")),
])
def test_large_host_results_take_the_pipelined_copy_and_match_device_results(api, monkeypatch, Cn, nbatch, blen, nspac, thread_counts):
    """Results of 64 MB and more go back to pageable host memory through hostcopy.hip (pinned ring + host threads), the
    batch means of a run of several launches piece by piece, each piece under the launch after it.
    The same Philox run with device_ptrs = 1 leaves them in HBM: both must hold the same bits, also when the last chunk
    is ragged, with one copy thread, with more threads than the default 8, and with the plain copy
    (MCMCHIP_COPY_THREADS = 0)."""
    import ctypes as C
    import torch
    L = api.load_library()
    dev = torch.device("cuda", 0)
    X, y = load_logit()
    tg = api.logistic_target(X, y)
    d = 5
    rng = np.random.default_rng(3)
    init_h = 0.1 * rng.standard_normal((Cn, d))
    scale = np.array([0.4])
    shapes = ((Cn,), (Cn, nbatch, d), (Cn, d), (Cn, nbatch))
    init_d = torch.from_numpy(init_h).to(dev)
    outs_d = [torch.empty(s_, dtype=torch.float64, device=dev) for s_ in shapes]
    args = abi.MetropArgs(Cn, nbatch, blen, nspac, init_d.data_ptr(), abi.SCALE_CONSTANT, scale.ctypes.data, abi.OUT_IDENTITY, None,
                          abi.RNG_PHILOX, None, None, 7, 0, 0, 0, 1, abi.PATH_STREAM)
    out = abi.MetropOut(*[t.data_ptr() for t in outs_d], None, None, None, None, None, None, 0, 0, 0)
    assert L.mcmchip_metrop(tg.handle, C.addressof(args), C.addressof(out)) == 0, L.mcmchip_last_error()
    want = [t.cpu().numpy() for t in outs_d]
    assert want[1].nbytes >= 64 << 20
    for threads in thread_counts:
        monkeypatch.delenv("MCMCHIP_NO_PIECEWISE_COPY", raising=False)
        if threads is None:
            monkeypatch.delenv("MCMCHIP_COPY_THREADS", raising=False)
        elif threads == "piecewise off":
            monkeypatch.delenv("MCMCHIP_COPY_THREADS", raising=False)
            monkeypatch.setenv("MCMCHIP_NO_PIECEWISE_COPY", "1")
        elif threads.startswith("limit "):
            monkeypatch.delenv("MCMCHIP_COPY_THREADS", raising=False)
            api.set_launch_limit(max_iters=int(threads.split()[1]))
        else:
            monkeypatch.setenv("MCMCHIP_COPY_THREADS", threads)
        outs_h = [np.full(s_, np.nan) for s_ in shapes]
        args_h = abi.MetropArgs(Cn, nbatch, blen, nspac, init_h.ctypes.data, abi.SCALE_CONSTANT, scale.ctypes.data, abi.OUT_IDENTITY,
                                None, abi.RNG_PHILOX, None, None, 7, 0, 0, 0, 0, abi.PATH_STREAM)
        out_h = abi.MetropOut(*[a_.ctypes.data for a_ in outs_h], None, None, None, None, None, None, 0, 0, 0)
        try:
            assert L.mcmchip_metrop(tg.handle, C.addressof(args_h), C.addressof(out_h)) == 0, L.mcmchip_last_error()
            if threads == "limit 40":
                assert api.last_kernel_time()[2] == 14        # 40 iterations, twelve launches of 40, one of 3
            if threads == "limit 240":
                assert api.last_kernel_time()[2] == 5         # 240 iterations, three launches of 71, one of 70
        finally:
            api.set_launch_limit()
        for got, ref in zip(outs_h, want):
            assert np.array_equal(got, ref), threads


def test_series_upload_and_device_transpose(api, oracle_mod, monkeypatch):
    """mcmchip_series_upload: a host matrix into HBM once, in the diagnostics' column-major layout, transposed on the
    device when the host holds it in C order (ragged 64 x 64 tiles included).  initseq / lagsums / olbm on the resident
    series give the bits of the host-pointer calls, and large C-ordered numpy arrays take that route by themselves."""
    import ctypes as C
    from scipy.signal import lfilter
    L = api.load_library()
    assert L.mcmchip_series_free(None) == 0
    rng = np.random.default_rng(21)
    for n, p in ((1000, 3), (4097, 65), (70, 130), (200_003, 9)):
        x = lfilter([1.0], [1.0, -0.6], rng.standard_normal((n, p)), axis=0) + rng.uniform(-2, 2, p)
        xc = x - x.mean(axis=0)
        blen = max(1, n // 40)
        mu = x.mean(axis=0)         # one set of means for every route (numpy sums C and Fortran order differently)
        host = api.olbm(np.asfortranarray(x), blen, mean=mu)    # column-major host array: used in place, no upload
        for arr in (x, np.asfortranarray(x)):                   # C order (device transpose) and Fortran order
            ds = api.upload_series(arr)
            assert (ds.length, ds.ncol) == (n, p)
            assert np.array_equal(api.olbm(ds, blen, mean=mu), host)
            v, m = api.olbm_device(ds, blen)
            ref, refmean = oracle_mod.olbm(x, blen)
            assert np.max(np.abs(v - ref)) <= 1e-10 * np.max(np.abs(ref)) and np.allclose(m, refmean, rtol=1e-12)
            ds.free()
            ds.free()                                           # idempotent
        ds = api.upload_series(xc)
        a, b = api.lagsums(ds, 16), api.lagsums(np.asfortranarray(xc), 16)
        assert np.array_equal(a, b)
        ra, rb = api.initseq(ds, centred=True), api.initseq(np.asfortranarray(xc), centred=True)
        for ca, cb in zip(ra, rb):
            for key in ca:
                assert np.array_equal(ca[key], cb[key])
        # not declared centred: taken about its means in the same passes (mcmchip_initseq_uncentred); here they are ~0
        for ca, cu in zip(ra, api.initseq(ds)):
            assert len(ca["Gamma.pos"]) == len(cu["Gamma.pos"])
            for key in ca:
                assert np.max(np.abs(np.asarray(ca[key]) - np.asarray(cu[key]))) <= 1e-10 * np.max(np.abs(ca[key]))
    # a C-ordered array above the threshold goes through the upload by itself: same bits as the in-place column-major call
    n, p = 300_000, 7
    x = rng.standard_normal((n, p)).cumsum(axis=0) * 1e-3 + rng.standard_normal((n, p))
    assert x.nbytes >= api._UPLOAD_MIN_BYTES and not x.flags.f_contiguous
    mu = x.mean(axis=0)
    assert np.array_equal(api.olbm(x, 500, mean=mu), api.olbm(np.asfortranarray(x), 500, mean=mu))
    assert np.max(np.abs(api.olbm(x, 500) - api.olbm(x, 500, mean=mu))) == 0.0       # demean = TRUE computes those means
    xc = x - mu
    r1, r2 = api.initseq(xc, centred=True), api.initseq(np.asfortranarray(xc), centred=True)
    for ca, cb in zip(r1, r2):
        for key in ca:
            assert np.array_equal(ca[key], cb[key])
    # initseq on a large array: uploaded raw, centred and checked on the device (mcmchip_series_centre).  AR(1) columns:
    # the stopping lag stays in the hundreds, so the CPU oracle finishes in seconds
    y = np.column_stack([lfilter([1.0], [1.0, -rho], rng.standard_normal(n)) for rho in np.linspace(0.1, 0.9, p)]) + 3.0
    assert y.nbytes >= api._UPLOAD_MIN_BYTES and not y.flags.f_contiguous
    r3 = api.initseq(y)
    for c in range(p):
        ref = oracle_mod.initseq(y[:, c] - y[:, c].mean())
        assert len(r3[c]["Gamma.pos"]) == len(ref["Gamma.pos"])
        for key in ("gamma0", "Gamma.pos", "Gamma.dec", "Gamma.con", "var.pos", "var.dec", "var.con"):
            assert np.max(np.abs(np.asarray(r3[c][key]) - np.asarray(ref[key]))) <= 1e-10 * np.max(np.abs(ref[key])), (c, key)
    ds = api.upload_series(y)
    got_mu = ds.centre()
    assert np.allclose(got_mu, y.mean(axis=0), rtol=1e-13, atol=1e-15)
    assert np.allclose(ds.centre(), 0.0, atol=1e-15)             # centred now
    ds.free()
    ybad = y.copy()
    ybad[123_456, 3] = np.inf
    with pytest.raises(ValueError, match="is.finite"):
        api.initseq(ybad)
    v = lfilter([1.0], [1.0, -0.7], rng.standard_normal(1_100_000)) + 5.0      # one long series (8.8 MB)
    rv, ref = api.initseq(v), oracle_mod.initseq(v - v.mean())
    assert len(rv["Gamma.pos"]) == len(ref["Gamma.pos"])
    assert abs(rv["var.con"] - ref["var.con"]) <= 1e-10 * abs(ref["var.con"])
    # 64 MB and more go up through the pinned ring and host threads (hostcopy.hip): same bits with 8 threads, 1 thread and
    # the plain copy, ragged last chunk included; as a host pointer (InBuf), as an upload, and as a design matrix
    nb, pb = 1_200_007, 8
    big = np.asfortranarray(rng.standard_normal((nb, pb)) + 1.0)
    assert big.nbytes >= 64 << 20
    mub = big.mean(axis=0)
    Xb = rng.standard_normal((1_100_003, 8)); Xb[:, 0] = 1.0
    yb = (rng.random(Xb.shape[0]) < 0.5).astype(float)
    states = 0.1 * rng.standard_normal((70, 8))
    want = None
    for threads in (None, "1", "0"):
        if threads is None:
            monkeypatch.delenv("MCMCHIP_COPY_THREADS", raising=False)
        else:
            monkeypatch.setenv("MCMCHIP_COPY_THREADS", threads)
        ds = api.upload_series(big)
        got = (api.olbm(big, 777, mean=mub), api.olbm(ds, 777, mean=mub), api.lagsums(np.ascontiguousarray(big), 8),
               api.logdens(api.logistic_target(Xb, yb), states))
        ds.free()
        if want is None:
            want = got
        for g_, w_ in zip(got, want):
            assert np.array_equal(g_, w_), threads
    assert np.array_equal(want[0], want[1])
    monkeypatch.delenv("MCMCHIP_COPY_THREADS", raising=False)
    # bad arguments
    out = C.c_void_p()
    assert L.mcmchip_series_centre(None, 10, 2, None) == abi.EINVAL
    assert L.mcmchip_series_upload(None, 10, 2, 0, C.byref(out)) == abi.EINVAL
    assert L.mcmchip_series_upload(x.ctypes.data, 0, 2, 0, C.byref(out)) == abi.EINVAL


# ----------------------------------------------------------------------------- the configuration bench.py times (C3)
def _synthetic(n, d, seed=42):
    rng = np.random.default_rng(seed)
    X = np.empty((n, d))
    X[:, 0] = 1.0
    X[:, 1:] = rng.standard_normal((n, d - 1)) / np.sqrt(d)
    beta = 0.5 * rng.standard_normal(d)
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    return X, y, beta


@pytest.fixture(scope="module")
def c3(api, oracle_mod):
    """BASELINE config 3 data: n = 1e6, d = 64 (bench.py's synthetic()), one descriptor for all C3 tests"""
    n, d = 1_000_000, 64
    X, y, beta = _synthetic(n, d)
    return dict(n=n, d=d, X=X, y=y, beta=beta, tg=api.logistic_target(X, y),
                otg=oracle_mod.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.125))


@pytest.mark.parametrize("C", [65536, 8192])
def test_c3_bench_shape_logdens(api, oracle_mod, c3, C):
    """mcmchip_logdens(path = MFMA) at the launch shape bench.py times (65 536 chains: 256 chain blocks, no
    observation split) and at the per-GPU shape of an 8-way strong split (8192 chains: 32 chain blocks x 32 splits):
    64 sampled chains against the oracle, ALL chains against a second plan (the same states in 256-chain calls:
    one chain block, 256 observation splits)"""
    d = c3["d"]
    states = c3["beta"] + 1e-3 * np.random.default_rng(C).standard_normal((C, d))
    got = api.logdens(c3["tg"], states, path=abi.PATH_MFMA)
    pick = np.random.default_rng(1).choice(C, 64, replace=False)
    ref = oracle_mod.logdens(c3["otg"], states[pick])
    assert rel(got[pick], ref) < RTOL
    other = np.concatenate([api.logdens(c3["tg"], states[i:i + 256], path=abi.PATH_MFMA) for i in range(0, C, 256)])
    assert rel(got, other) < RTOL


@pytest.mark.parametrize("C", [65536, 8192])
def test_c3_bench_shape_metrop_injected(api, oracle_mod, c3, C):
    """three lockstep iterations of mcmchip_metrop (MFMA path, injected streams) at bench.py's launch shape and at
    the 8-way strong-split shape; 32 sampled chains replayed by the oracle: identical accept decisions, bitwise
    equal final states, batch means within 1e-12 (metrop.c:211-271)"""
    d = c3["d"]
    rng = np.random.default_rng(7 + C)
    nbatch = 3
    init = c3["beta"] + 1e-3 * rng.standard_normal((C, d))
    z = rng.standard_normal((C, nbatch, d))
    u = rng.random((C, nbatch))
    o = api.metrop(c3["tg"], init, nbatch, scale=4.0e-3, z=z, u=u, path=abi.PATH_MFMA)
    assert o["path"] == "mfma"
    pick = np.sort(np.random.default_rng(2).choice(C, 32, replace=False))
    ref = oracle_mod.metrop(c3["otg"], init[pick], nbatch, scale=4.0e-3, z=z[pick], u=u[pick], nthreads=16)
    assert np.array_equal(o["accept.batch"][pick], ref["accept_batch"])       # blen = nspac = 1: the decisions
    assert np.array_equal(o["final"][pick], ref["final"])
    assert rel(o["batch"][pick], ref["batch"]) < RTOL
    acc = o["accept"].mean()
    assert 0.1 < acc < 0.5, acc
    # every chain, not only the sampled ones: states are either the proposal or the previous state, bit for bit
    prop = init + 4.0e-3 * z[:, 0]
    first = o["batch"][:, 0]
    moved = o["accept.batch"][:, 0] == 1.0
    assert np.array_equal(first[moved], prop[moved]) and np.array_equal(first[~moved], init[~moved])


# ----------------------------------------------------------------------------- C5 beyond 2^31 elements
def test_c5_offsets_beyond_2_31(api, oracle_mod):
    """initseq and olbm on a device-resident n x p series with n * p = 3.07e9 > 2^31 elements (24.6 GB): columns 0
    and 255 hold the same AR(1) series, so every result for them must agree bit for bit (a 32-bit offset anywhere
    would break that); column 0 against the oracle at that n (1e-10)."""
    import torch
    from scipy.signal import lfilter
    L = api.load_library()
    dev = torch.device("cuda", 0)
    n, p, blen = 12_000_000, 256, 1000
    rng = np.random.default_rng(5)
    s0 = lfilter([1.0], [1.0, -0.9], rng.standard_normal(n))
    s0 -= s0.mean()
    x = torch.empty((p, n), dtype=torch.float64, device=dev)       # column-major n x p
    col = torch.from_numpy(s0).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    for c in range(p):
        if c in (0, p - 1):
            x[c] = col
        else:                                                       # a share of the common series plus white noise
            x[c] = (c / p) * col + torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            x[c] -= x[c].mean()
    torch.cuda.synchronize()
    cap = 4096
    n_out = np.zeros(p, dtype=np.int64)
    g0, vp, vd, vc = (np.empty(p) for _ in range(4))
    Gp, Gd, Gc = (np.empty((p, cap)) for _ in range(3))
    rc = L.mcmchip_initseq(x.data_ptr(), n, p, 1, cap, n_out.ctypes.data, g0.ctypes.data, Gp.ctypes.data, Gd.ctypes.data,
                           Gc.ctypes.data, vp.ctypes.data, vd.ctypes.data, vc.ctypes.data)
    assert rc == 0, L.mcmchip_last_error()
    m = int(n_out[0])
    assert n_out[p - 1] == m and m > 10
    for arr in (g0, vp, vd, vc):
        assert arr[0] == arr[p - 1]
    for arr in (Gp, Gd, Gc):
        assert np.array_equal(arr[0, :m], arr[p - 1, :m])
    ref = oracle_mod.initseq(s0)
    assert len(ref["Gamma.pos"]) == m
    for key, got in (("gamma0", g0[0]), ("Gamma.pos", Gp[0, :m]), ("Gamma.dec", Gd[0, :m]), ("Gamma.con", Gc[0, :m]),
                     ("var.pos", vp[0]), ("var.dec", vd[0]), ("var.con", vc[0])):
        assert np.max(np.abs(np.asarray(got) - np.asarray(ref[key]))) <= 1e-10 * np.max(np.abs(ref[key])), key
    # olbm over all 256 columns: the two copies give the same row/column of var and the same mean, bit for bit
    mean = np.zeros(p)
    var = np.empty((p, p))
    rc = L.mcmchip_olbm(x.data_ptr(), n, p, blen, mean.ctypes.data, var.ctypes.data, 0, 1)
    assert rc == 0, L.mcmchip_last_error()
    assert mean[0] == mean[p - 1]
    assert var[0, 0] == var[p - 1, p - 1] == var[0, p - 1] == var[p - 1, 0]
    assert np.array_equal(var[0, 1:p - 1], var[p - 1, 1:p - 1])
    assert np.array_equal(var, var.T)
    # the first two columns by a second call (p = 2), against the oracle and against the 256-column call
    c1 = x[1].cpu().numpy()
    mean2 = np.zeros(2)
    var2 = np.empty((2, 2))
    rc = L.mcmchip_olbm(x.data_ptr(), n, 2, blen, mean2.ctypes.data, var2.ctypes.data, 0, 1)
    assert rc == 0, L.mcmchip_last_error()
    ref2, refmean = oracle_mod.olbm(np.column_stack([s0, c1]), blen)
    scale_ = np.sqrt(np.outer(np.diag(ref2), np.diag(ref2)))
    assert np.max(np.abs(var2 - ref2) / scale_) < 1e-10
    assert np.max(np.abs(var[:2, :2] - ref2) / scale_) < 1e-10
    assert np.allclose(mean2, refmean, rtol=0, atol=1e-12)
    # the last two columns start beyond element 2^31: same numbers as a call that sees them as columns 0 and 1
    var3 = np.empty((2, 2))
    mean3 = np.zeros(2)
    rc = L.mcmchip_olbm(x[p - 2:].data_ptr(), n, 2, blen, mean3.ctypes.data, var3.ctypes.data, 0, 1)
    assert rc == 0, L.mcmchip_last_error()
    assert np.max(np.abs(var3 - var[p - 2:, p - 2:]) / np.sqrt(np.outer(np.diag(var3), np.diag(var3)))) < 1e-10
    assert var3[1, 1] == var2[0, 0]
    del x
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------- statistical equivalence (N1)
def test_foo_rout_logistic_section_on_device(api, oracle_mod):
    """devel/bug/foo.Rout:100-117 (the vignette posterior on the `logit` data, one R session under set.seed(42))
    printed by the DEVICE: the oracle on the emulated R stream records the normals/uniforms each call drew, the
    device replays them as injected streams."""
    from rhelpers import t_conf_int
    rs = oracle_mod.RStream(42)
    h = oracle_mod.Target(abi.T_SIMPLEX, 5, par=[0.0])
    o = oracle_mod.metrop(h, np.zeros(5), 1000, rstream=rs)                      # foo.Rout:26-70: the simplex runs
    o = oracle_mod.metrop(h, o["final"], 1000, scale=0.1, rstream=rs)            # that precede the logistic section
    o = oracle_mod.metrop(h, o["final"], 1000, scale=0.1, rstream=rs)            # in the same session
    o = oracle_mod.metrop(h, o["final"], 1000, 1000, scale=0.1, rstream=rs)
    X, y = load_logit()
    x = glm_binomial(X, y)
    otg, dtg = oracle_mod.Target(abi.T_LOGISTIC, 5, x=X, y=y, prior_c=0.125), api.logistic_target(X, y)

    def replay(x0, scale):
        ref = oracle_mod.metrop(otg, x0, 1000, scale=scale, rstream=rs, debug=True)
        u = np.where(np.isnan(ref["u"]), 0.5, ref["u"])
        dev = api.metrop(dtg, np.asarray(x0, dtype=np.float64)[None], 1000, scale=scale, z=ref["z"], u=u)
        assert np.array_equal(dev["final"][0], ref["final"][0])
        return dev, ref["final"][0]

    for sc, expect in [(1.0, 0.019), (0.1, 0.725), (0.3, 0.34), (0.5, 0.153)]:   # foo.Rout:100-112
        dev, x = replay(x, sc)
        assert dev["accept"][0] == expect
    dev, x = replay(x, 0.4)
    assert np.allclose(t_conf_int(dev["accept.batch"][0]), (0.2211882, 0.2748118), atol=5e-8)   # foo.Rout:113-117


def test_c1_device_philox_statistically_equivalent(api, oracle_mod):
    """C1 (vignette demo: `logit` data, scale 0.4, 1e4 iterations per chain) under the DEVICE RNG, 256 chains with
    fixed seeds, against an independent long run of the oracle under its own seed:
      - the acceptance rate lies inside the reference's own 95 % interval (foo.Rout:115),
      - every coordinate's mean agrees within 4 Monte Carlo standard errors,
      - the variance of the chain means (batch means with one batch per chain: independent and near normal, so the
        ratio follows an F law) lies inside the central 99.9 % interval of F(255, 63); the pooled variances of the
        100-iteration batch means agree within 15 %,
      - initseq on the device's output gives a variance estimate consistent with the batch-means one."""
    from scipy import stats
    X, y = load_logit()
    mle = glm_binomial(X, y)
    d, blen, nbatch = 5, 100, 100
    Cd, Co = 256, 64
    dev = api.metrop(api.logistic_target(X, y), np.tile(mle, (Cd, 1)), nbatch, blen, scale=0.4, seed=20240611)
    orc = oracle_mod.metrop(oracle_mod.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=0.125), np.tile(mle, (Co, 1)),
                            nbatch, blen, scale=0.4, seed=977, nthreads=16)
    acc = dev["accept"].mean()
    assert 0.2211882 < acc < 0.2748118, acc
    assert 0.2211882 < orc["accept"].mean() < 0.2748118
    # means: chains are independent, so the standard error of the grand mean comes from the chain means
    md, mo = dev["batch"].mean(axis=1), orc["batch"].mean(axis=1)          # (chains, d)
    se = np.sqrt(md.var(axis=0, ddof=1) / Cd + mo.var(axis=0, ddof=1) / Co)
    zscore = (md.mean(axis=0) - mo.mean(axis=0)) / se
    assert np.max(np.abs(zscore)) < 4.0, zscore
    # variances: of the chain means (exact F law), and of the 100-iteration batch means pooled within chains
    lo, hi = stats.f.ppf([0.0005, 0.9995], Cd - 1, Co - 1)
    ratio = md.var(axis=0, ddof=1) / mo.var(axis=0, ddof=1)
    assert np.all((ratio > lo) & (ratio < hi)), (ratio, lo, hi)
    vd = dev["batch"].var(axis=1, ddof=1).mean(axis=0)
    vo = orc["batch"].var(axis=1, ddof=1).mean(axis=0)
    assert np.all(np.abs(vd / vo - 1.0) < 0.15), vd / vo
    # initseq on the device's path of batch means (chain 0..7, every coordinate): sigma^2 estimates of the same
    # functional by the two routes agree within sampling error
    paths = api.metrop(api.logistic_target(X, y), np.tile(mle, (8, 1)), 10000, 1, scale=0.4, seed=5)["batch"]
    for c in range(8):
        res = api.initseq(paths[c])
        for j in range(d):
            bm = paths[c][:, j].reshape(100, 100).mean(axis=1).var(ddof=1) * 100       # batch-means estimate
            assert 0.4 < res[j]["var.con"] / bm < 2.5, (c, j, res[j]["var.con"], bm)
            assert res[j]["var.con"] > res[j]["gamma0"]                                   # positive autocorrelation


# ----------------------------------------------------------------------------- Philox-mode tempering, decision by decision
def _first_divergence(o, ref):
    """per ensemble: index of the first iteration whose decision record differs (or -1)"""
    E, T = o["acceptd"].shape
    out = np.full(E, -1)
    for e in range(E):
        diff = (o["acceptd"][e] != ref["acceptd"][e]) | (o["which"][e] != ref["which"][e]) | \
               (o["proposal"][e][:, 0] != ref["proposal"][e][:, 0])
        if diff.any():
            out[e] = int(np.argmax(diff))
    return out


def _assert_near_tie(o, ref, e, t, bounds=None):
    """a diverging decision must be a genuine rounding tie: u within rounding of exp(log hastings) on one side, or a
    coordinate of the proposal within rounding of a discontinuity of the density"""
    tie = False
    for r in (o, ref):
        lh, uh = r["log.hastings"][e][t], r["unif.hastings"][e][t]
        if np.isfinite(lh) and not np.isnan(uh):
            tie |= abs(uh - np.exp(lh)) < 1e-9
    if bounds is not None:
        th = np.abs(np.concatenate([o["proposal"][e][t, 1:], ref["proposal"][e][t, 1:]]))
        tie |= np.min(np.abs(th[:, None] - np.concatenate([bounds, [1.0]])[None, :])) < 1e-9
    assert tie, (e, t)


def test_temper_philox_decisions_match_oracle(api, oracle_mod):
    """device Philox against the oracle's restatement of the same counters: the uniforms are bit-equal, the normals
    differ in the last bits (ocml vs glibc log / sincos), so states agree to rounding and a decision can only flip
    at a rounding tie.  The witch's hat is piecewise constant: no flips expected; a handful tolerated if each is a tie."""
    k, p, E = 6, 3, 32
    par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / p)
    i, j = np.indices((k, k))
    nb = (np.abs(i - j) == 1) | (np.abs(i - j) == 2)
    args = (nb, 20, 13, 7, 0.3456789)
    tg, otg = api.Target(abi.T_WITCH, p, par=par), oracle_mod.Target(abi.T_WITCH, p, par=par)
    o = api.temper(tg, np.zeros((E, k, p)), *args, parallel=True, seed=11, ens_offset=5, debug=True)
    ref = oracle_mod.temper(otg, np.zeros((E, k, p)), *args, parallel=True, seed=11, ens_offset=5, debug=True)
    assert np.array_equal(o["unif.which"], ref["unif.which"])               # the same counters, the same bits
    div = _first_divergence(o, ref)
    assert (div >= 0).sum() <= 2, div
    for e in np.nonzero(div >= 0)[0]:
        _assert_near_tie(o, ref, e, div[e], bounds=par)
    same = div < 0
    assert np.max(np.abs(o["final"][same] - ref["final"][same])) < 1e-9
    assert np.array_equal(o["acceptx"][same], ref["acceptx"][same], equal_nan=True)
    assert np.array_equal(o["accepti"][same], ref["accepti"][same], equal_nan=True)
    assert np.allclose(o["batch"][same], ref["batch"][same], rtol=0, atol=1e-9)
    # the run-parallel kernel draws the same numbers
    r = api.temper(tg, np.zeros((E, k, p)), *args, parallel=True, seed=11, ens_offset=5, _path=abi.PATH_RUN)
    assert np.array_equal(r["final"], o["final"]) and np.array_equal(r["batch"], o["batch"])
    # restart: a 40-batch run from the start equals the 20-batch run followed by its continuation
    o20 = api.temper(tg, np.zeros((E, k, p)), *args, parallel=True, seed=11, ens_offset=5)
    nxt = api.temper(o20)
    full = api.temper(tg, np.zeros((E, k, p)), nb, 40, 13, 7, 0.3456789, parallel=True, seed=11, ens_offset=5)
    assert np.array_equal(np.concatenate([o20["batch"], nxt["batch"]], axis=1), full["batch"])
    assert np.array_equal(nxt["final"], full["final"])


def test_serial_tempering_philox_decisions_match_oracle(api, oracle_mod):
    """the Bayes-factor workload (vignettes/bfst.Rnw): serial tempering over the 8 logistic sub-models, device Philox
    against the oracle's restatement, decision by decision from the debug traces"""
    from rhelpers import temp_par_setup
    X, y, models, masks, betas, nb = temp_par_setup()
    lpp = np.array([0.0, 0.4, 0.1, 0.9, -0.2, 0.3, 0.5, 1.1])
    kw = dict(x=X, y=y, prior_c=0.125, par=abi.logistic_models_par(masks, lpp, -4 * np.log(2 * np.sqrt(2 * np.pi))))
    E = 16
    init = np.tile(np.concatenate([[8.0], betas[7]]), (E, 1))
    o = api.temper(api.Target(abi.T_LOGISTIC_MODELS, 4, **kw), init, nb, 30, 10, 3, 0.5, parallel=False, seed=9, debug=True)
    ref = oracle_mod.temper(oracle_mod.Target(abi.T_LOGISTIC_MODELS, 4, **kw), init, nb, 30, 10, 3, 0.5,
                            parallel=False, seed=9, debug=True)
    div = _first_divergence(o, ref)
    assert (div >= 0).sum() <= 1, div
    for e in np.nonzero(div >= 0)[0]:
        _assert_near_tie(o, ref, e, div[e])
    same = div < 0
    assert np.max(np.abs(o["final"][same] - ref["final"][same])) < 1e-9
    assert np.array_equal(o["ibatch"][same], ref["ibatch"][same])
    assert np.allclose(o["batch"][same], ref["batch"][same], rtol=1e-9, atol=1e-11)
    assert np.allclose(o["ibatch"].sum(axis=2), 1.0)


# ----------------------------------------------------------------------------- chunked persistent kernels
@pytest.fixture
def launch_limit(api):
    yield api.set_launch_limit
    api.set_launch_limit(0, 0.25)
    api.set_interrupt_poll(None)


@pytest.mark.parametrize("rng", ["philox", "injected"])
def test_metrop_stream_chunked_equals_one_launch(api, launch_limit, rng):
    """the persistent stream kernel cut into launches of 7 / 64 / 100 iterations (batches of 5 x 3 iterations straddle
    the cuts, the Philox look-ahead blocks too) gives the bits of a single launch: batch means, acceptance rates,
    final state, the debug trace; morphed target and matrix scale included"""
    from mcmc_amd import morph
    X, y = load_logit()
    mle = glm_binomial(X, y)
    rng_ = np.random.default_rng(31)
    C, nbatch, blen, nspac, d = 48, 11, 5, 3, 5
    niter = nbatch * blen * nspac
    init = mle + 0.05 * rng_.standard_normal((C, d))
    kw = dict(seed=4242, chain_offset=7) if rng == "philox" else dict(z=rng_.standard_normal((C, niter, d)), u=rng_.random((C, niter)))
    scale = 0.3 * np.linalg.cholesky(np.eye(d) + 0.2)
    cases = [(api.logistic_target(X, y), init, scale, "moments2"),
             (api.Target(abi.T_STUDENT_T, d, par=[3.0], morph=morph(b=0.7, r=0.5, p=3)), rng_.standard_normal((C, d)), 0.8, None)]
    for tg, ini, sc, outfun in cases:
        launch_limit(0)
        one = api.metrop(tg, ini, nbatch, blen, nspac, scale=sc, outfun=outfun, debug=True, path=abi.PATH_STREAM, **kw)
        assert api.last_kernel_time()[2] == 1
        for cut in (7, 64, 100):
            launch_limit(cut)
            o = api.metrop(tg, ini, nbatch, blen, nspac, scale=sc, outfun=outfun, debug=True, path=abi.PATH_STREAM, **kw)
            assert api.last_kernel_time()[2] == -(-niter // cut)
            for key in ("batch", "accept", "accept.batch", "final", "current", "proposal", "z", "log.green", "u", "debug.accept"):
                assert np.array_equal(o[key], one[key], equal_nan=True), (key, cut)


@pytest.mark.parametrize("parallel,path", [(True, abi.PATH_LDS), (True, abi.PATH_RUN), (False, abi.PATH_LDS)])
@pytest.mark.parametrize("rng", ["philox", "injected"])
def test_temper_chunked_equals_one_launch(api, launch_limit, parallel, path, rng):
    """both tempering kernels cut into launches of 64 and 192 iterations (whole 64-iteration windows; batches of
    13 x 7 = 91 iterations straddle the cuts): same bits as a single launch"""
    k, p, E = 8, 3, 10
    par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / p)
    i, j = np.indices((k, k))
    nb = (np.abs(i - j) == 1) | (np.abs(i - j) == 2)
    nbatch, blen, nspac = 6, 13, 7
    niter = nbatch * blen * nspac
    rng_ = np.random.default_rng(5)
    kw = dict(seed=99, ens_offset=3) if rng == "philox" else dict(unif=rng_.random((E, niter, 4)), norm=rng_.standard_normal((E, niter, p)))
    init = 0.01 * rng_.standard_normal((E, k, p)) if parallel else np.column_stack([rng_.integers(1, k + 1, E).astype(float), np.zeros((E, p))])
    tg = api.Target(abi.T_WITCH, p, par=par)
    for outfun in (None, "witch.corner"):
        launch_limit(0)
        one = api.temper(tg, init, nb, nbatch, blen, nspac, 0.3456789, parallel=parallel, outfun=outfun, _path=path, **kw)
        for cut in (64, 192):
            launch_limit(cut)
            o = api.temper(tg, init, nb, nbatch, blen, nspac, 0.3456789, parallel=parallel, outfun=outfun, _path=path, **kw)
            assert api.last_kernel_time()[2] == -(-niter // cut)
            for key in ("batch", "acceptx", "accepti", "final") + (() if parallel else ("ibatch",)):
                assert np.array_equal(o[key], one[key], equal_nan=True), (key, cut, outfun)
    if path == abi.PATH_LDS:                      # the debug trace across launches
        launch_limit(0)
        one = api.temper(tg, init, nb, nbatch, blen, nspac, 0.3456789, parallel=parallel, debug=True, **kw)
        launch_limit(128)
        o = api.temper(tg, init, nb, nbatch, blen, nspac, 0.3456789, parallel=parallel, debug=True, **kw)
        for key in ("state", "proposal", "log.hastings", "unif.hastings", "acceptd", "which", "norm", "unif.choose"):
            assert np.array_equal(o[key], one[key], equal_nan=True), key


def test_interrupt_poll_between_launches(api, launch_limit):
    """the host poll point (R_CheckUserInterrupt's place, metrop.c:269 / temper.c:848): the callback runs between
    launches and a true return abandons the call with MCMCHIP_EINTR; the library stays usable"""
    X, y = load_logit()
    mle = glm_binomial(X, y)
    tg = api.logistic_target(X, y)
    calls = []
    api.set_interrupt_poll(lambda: calls.append(1) or len(calls) >= 3)
    launch_limit(10)
    with pytest.raises(api.McmcHipError, match="interrupted") as ei:
        api.metrop(tg, np.tile(mle, (8, 1)), 100, scale=0.4, seed=1, path=abi.PATH_STREAM)
    assert ei.value.code == abi.EINTR and len(calls) == 3
    # the MFMA path polls too (per group of iterations)
    n, d = 4096, 16
    rng = np.random.default_rng(3)
    Xs = rng.standard_normal((n, d)) / 4
    ys = (rng.random(n) < 0.5).astype(float)
    calls.clear()
    launch_limit(10, 1e-5)                       # a 10 microsecond budget: a poll after every iteration
    with pytest.raises(api.McmcHipError, match="interrupted"):
        api.metrop(api.logistic_target(Xs, ys), np.zeros((64, d)), 50, scale=0.02, seed=1, path=abi.PATH_MFMA)
    assert len(calls) == 3
    launch_limit(10, 0.25)
    # tempering
    k, p = 4, 2
    par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / p)
    nb = np.abs(np.subtract.outer(np.arange(k), np.arange(k))) == 1
    calls.clear()
    launch_limit(64)
    with pytest.raises(api.McmcHipError, match="interrupted"):
        api.temper(api.Target(abi.T_WITCH, p, par=par), np.zeros((4, k, p)), nb, 100, 5, 1, 0.3, parallel=True, seed=1)
    api.set_interrupt_poll(None)
    launch_limit(0)
    o = api.metrop(tg, np.tile(mle, (8, 1)), 100, scale=0.4, seed=1, path=abi.PATH_STREAM)
    assert o["batch"].shape == (8, 100, 5)


def test_time_based_launch_budget(api, launch_limit):
    """with a tiny time budget the run is cut adaptively (calibration launch of 256 iterations, then by measured
    rate) and still gives the bits of one launch"""
    X, y = load_logit()
    mle = glm_binomial(X, y)
    tg = api.logistic_target(X, y)
    init = np.tile(mle, (256, 1))
    launch_limit(0, 10.0)
    one = api.metrop(tg, init, 3000, scale=0.4, seed=8, path=abi.PATH_STREAM)
    n_one = api.last_kernel_time()[2]
    launch_limit(0, 1e-3)                        # 1 ms per launch
    o = api.metrop(tg, init, 3000, scale=0.4, seed=8, path=abi.PATH_STREAM)
    n_cut = api.last_kernel_time()[2]
    launch_limit(0, 0.25)
    assert n_one <= 2 and n_cut > n_one
    assert np.array_equal(o["batch"], one["batch"]) and np.array_equal(o["final"], one["final"])


# ----------------------------------------------------------------------------- pseudo-prior tuning (bfst.Rnw:497-559)
def test_tune_pseudo_prior_flattens_occupancy(api):
    """serial tempering over the 8 sub-models of y ~ x1 + x2 + x3 on the reference's `foo` data with the vignette's
    normalized N(0, 2^2) prior: with log pseudo-priors zero the chain sits in a few models; the adjustment loop of
    bfst.Rnw:516-550 must make the occupation fractions flat (max / min < 2) and its Bayes factors must agree with a
    second, longer run at the tuned pseudo-prior"""
    from rhelpers import temp_par_setup
    X, y, models, masks, betas, nb = temp_par_setup()
    E = 64
    init = np.tile(np.concatenate([[8.0], betas[7]]), (E, 1))
    res = api.tune_pseudo_prior(X, y, models, init, nb, nbatch=100, blen=50, scale=0.5, seed=20240612)
    ibar0, ibar1 = res["ibar.save"][0], res["ibar.save"][-1]
    assert ibar0.max() / max(ibar0.min(), 1e-300) > 2.0          # untuned: unbalanced (the vignette's "what happened")
    assert res["rounds"] >= 2 and res["rounds"] <= 30
    assert ibar1.min() > 0 and ibar1.max() / ibar1.min() < 2.0
    assert res["log.pseudo.prior"].min() == 0.0
    # Bayes factors (bfst.Rnw:610): stable under a longer run at the tuned pseudo-prior
    qux = res["log.pseudo.prior.run"]              # what the last run sampled with ("log.pseudo.prior" is one update on)
    assert np.array_equal(res["log.pseudo.prior"], res["qux.save"][-1]) and len(res["qux.save"]) == res["rounds"]
    bf1 = (qux - np.log(ibar1)) / np.log(10)
    tg = api.logistic_models_target(X, y, models, log_pseudo_prior=qux)
    o = api.temper(tg, res["out"]["final"], nb, 400, 50, 1, 0.5, parallel=False, seed=7)
    ibar2 = o["ibatch"].reshape(-1, 8).mean(axis=0)
    bf2 = (qux - np.log(ibar2)) / np.log(10)
    assert np.max(np.abs((bf1 - bf1.min()) - (bf2 - bf2.min()))) < 0.15


# ----------------------------------------------------------------------------- temper output functionals
@pytest.mark.parametrize("parallel", [True, False])
@pytest.mark.parametrize("outfun", ["moments2", "index", "negindex", "logical"])
def test_temper_index_and_moment_outfuns(api, oracle_mod, parallel, outfun):
    """index-vector and c(v, v^2) output functionals for temper (metrop's rules, metrop.c:435-497, applied to the
    identity output of temper.c:851-864) against the oracle, and against the identity run they are functions of"""
    rng = np.random.default_rng(17)
    k, p, E, nbatch, blen, nspac = 5, 2, 7, 8, 4, 3
    niter = nbatch * blen * nspac
    betas = np.geomspace(1.0, 0.1, k)
    par = np.concatenate([[float(k)], betas, abi.normmix_par([0.5, 0.5], [[-2.0, -2.0], [2.0, 2.0]], 1.0)])
    nb = np.abs(np.subtract.outer(np.arange(k), np.arange(k))) == 1
    nid = k * p if parallel else p
    of = {"moments2": "moments2", "index": [1, nid], "negindex": [-1], "logical": (np.arange(nid) % 2 == 0)}[outfun]
    unif, norm = rng.random((E, niter, 4)), rng.standard_normal((E, niter, p))
    init = np.tile([2.0, 2.0], (E, k, 1)) if parallel else np.column_stack([rng.integers(1, k + 1, E).astype(float), np.tile([2.0, 2.0], (E, 1))])
    tg, otg = api.Target(abi.T_TEMPERED_NORMMIX, p, par=par), oracle_mod.Target(abi.T_TEMPERED_NORMMIX, p, par=par)
    o = api.temper(tg, init, nb, nbatch, blen, nspac, 0.7, parallel=parallel, unif=unif, norm=norm, outfun=of)
    ident = api.temper(tg, init, nb, nbatch, blen, nspac, 0.7, parallel=parallel, unif=unif, norm=norm)
    kind, mask, nout = abi.resolve_outfun(of, nid)
    okind = oracle_mod.TOUT_MOMENTS2 if outfun == "moments2" else oracle_mod.TOUT_INDEX
    ref = oracle_mod.temper(otg, init, nb, nbatch, blen, nspac, 0.7, parallel=parallel, unif=unif, norm=norm,
                            out_kind=okind, out_index=mask)
    assert o["batch"].shape == (E, nbatch, nout) == ref["batch"].shape
    assert np.array_equal(o["final"], ref["final"]) and np.array_equal(o["final"], ident["final"])
    assert rel(o["batch"] + 10, ref["batch"] + 10) < RTOL
    # the identity output in the state's own (column-major) order
    v = ident["batch"].transpose(0, 1, 3, 2).reshape(E, nbatch, nid) if parallel else ident["batch"]
    if outfun == "moments2":
        assert np.array_equal(o["batch"][:, :, :nid], v)
        assert np.all(o["batch"][:, :, nid:] >= v ** 2 - 1e-12)            # mean of squares >= square of mean
    else:
        assert np.array_equal(o["batch"], v[:, :, mask.astype(bool)])
    # a restart keeps the functional (R/temper.R:22-30)
    o2 = api.temper(o, unif=unif, norm=norm)
    assert o2["batch"].shape == o["batch"].shape


# ----------------------------------------------------------------------------- multi-GPU drivers on device-resident data
INITSEQ_KEYS = ("gamma0", "Gamma.pos", "Gamma.dec", "Gamma.con", "var.pos", "var.dec", "var.con")


def _initseq_uncentred_raw(api, ptr, n, p, device_ptrs, cap=8192):
    """mcmchip_initseq_uncentred through the C-ABI: (rc, n_out, gamma0, Gamma.pos, Gamma.dec, Gamma.con, var.*, mean)"""
    L = api.load_library()
    n_out = np.zeros(p, dtype=np.int64)
    g0, vp, vd, vc, mean = (np.empty(p) for _ in range(5))
    Gp, Gd, Gc = (np.empty((p, cap)) for _ in range(3))
    rc = L.mcmchip_initseq_uncentred(ptr, n, p, device_ptrs, cap, n_out.ctypes.data, g0.ctypes.data, Gp.ctypes.data,
                                     Gd.ctypes.data, Gc.ctypes.data, vp.ctypes.data, vd.ctypes.data, vc.ctypes.data,
                                     mean.ctypes.data)
    return rc, n_out, g0, Gp, Gd, Gc, vp, vd, vc, mean


@pytest.mark.parametrize("n,rhos,kind", [
    (1000, (0.3, -0.4, 0.9), "offset"),
    (2015, (0.5, 0.8), "offset"),                 # around the 2048-position tiles: the first and last tile test every element's range
    (2049, (0.5, 0.8), "drift"),
    (4065, (0.0, 0.95), "offset"),
    (65_536, (0.6, 0.97), "offset"),              # the provisional centre is the mean of the whole column up to here ...
    (65_537, (0.6, 0.97), "drift"),               # ... and of 64 runs of 1024 rows beyond
    (300_000, (0.2, 0.9, 0.99), "drift"),         # a start far from where the chain settles: the sampled centre is well off the mean
    (200_000, (0.999,), "offset"),                # > 2048 lags: the edge sums are fetched again for the column that goes on
    (30_000, (0.5, 0.9), "trend+offset"),          # a slow trend keeps the stop rule away for thousands of lags
    (1_048_576 + 31, (0.5, 0.9), "offset"),
])
def test_initseq_uncentred_vs_oracle(api, oracle_mod, n, rhos, kind):
    """mcmchip_initseq_uncentred (lag sums about a provisional centre per column, carried over to the mean on the host; the
    series is not modified) against the oracle on x - mean(x), R/initseq.R:3-4 + initseq.c, to the 1e-10 of the other initseq
    comparisons; the column means it returns; host and device pointers give the same bits."""
    import torch
    from scipy.signal import lfilter
    rng = np.random.default_rng(n)
    p = len(rhos)
    x = np.column_stack([lfilter([1.0], [1.0, -r], rng.standard_normal(n)) for r in rhos])
    x = x * rng.choice([0.01, 1.0, 30.0], p) + rng.choice([-300.0, 7.0, 1000.0], p) * x.std(axis=0)      # means 7 .. 1000 sd away
    if kind == "drift":
        x += 25.0 * x.std(axis=0) * np.exp(-np.arange(n) / (n / 50.0))[:, None]
    if kind == "trend+offset":
        x[:, 0] += np.linspace(-0.2, 0.2, n) * x[:, 0].std()
    xt = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    keep = xt.clone()
    got = api.initseq(xt)                                    # resident and not declared centred -> the new entry point
    assert torch.equal(xt, keep)                             # inputs are never modified
    mu = x.mean(axis=0)
    for c in range(p):
        ref = oracle_mod.initseq(x[:, c] - mu[c])
        assert len(got[c]["Gamma.pos"]) == len(ref["Gamma.pos"]), (c, len(got[c]["Gamma.pos"]), len(ref["Gamma.pos"]))
        for key in INITSEQ_KEYS:
            assert np.max(np.abs(np.asarray(got[c][key]) - np.asarray(ref[key]))) <= 1e-10 * np.max(np.abs(ref[key])), (c, key)
    cap = max(len(g["Gamma.pos"]) for g in got)
    rc, n_out, g0, Gp, Gd, Gc, vp, vd, vc, mean = _initseq_uncentred_raw(api, xt.data_ptr(), n, p, 1, cap)
    assert rc == 0 and np.allclose(mean, mu, rtol=1e-13, atol=0.0)
    xf = np.asfortranarray(x)                                # the same through a host pointer (staged through HBM): same bits
    rc, n_out2, g02, Gp2, Gd2, Gc2, vp2, vd2, vc2, mean2 = _initseq_uncentred_raw(api, xf.ctypes.data, n, p, 0, cap)
    assert rc == 0 and np.array_equal(n_out, n_out2) and np.array_equal(vc, vc2) and np.array_equal(mean, mean2)
    for c in range(p):
        assert np.array_equal(Gc[c, :n_out[c]], Gc2[c, :n_out2[c]]) and np.array_equal(Gc[c, :n_out[c]], got[c]["Gamma.con"])


def test_initseq_uncentred_many_columns_errors_and_capacity(api, oracle_mod):
    """300 short columns in one call (stopping lags from 1 to dozens, every column its own centre); the capacity protocol and
    the finiteness check of mcmchip_initseq (initseq.c:15) carry over; agreement with mcmchip_initseq on the centred series."""
    import torch
    from scipy.signal import lfilter
    rng = np.random.default_rng(77)
    n, p = 6000, 300
    rho = rng.uniform(-0.7, 0.97, p)
    x = np.column_stack([lfilter([1.0], [1.0, -r], rng.standard_normal(n)) for r in rho]) + rng.uniform(-100, 100, p)
    xt = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    got = api.initseq(xt)
    cen = api.initseq(torch.from_numpy(np.ascontiguousarray((x - x.mean(axis=0)).T)).cuda(), centred=True)
    for c in range(p):
        assert len(got[c]["Gamma.pos"]) == len(cen[c]["Gamma.pos"])
        for key in INITSEQ_KEYS:
            assert np.max(np.abs(np.asarray(got[c][key]) - np.asarray(cen[c][key]))) <= 1e-11 * np.max(np.abs(cen[c][key])), (c, key)
    for c in (0, 17, 299):
        ref = oracle_mod.initseq(x[:, c] - x[:, c].mean())
        assert abs(got[c]["var.con"] - ref["var.con"]) <= 1e-10 * abs(ref["var.con"])
    # capacity too small: MCMCHIP_ECAP and the lengths needed, as mcmchip_initseq reports them
    rc, n_out, *_ = _initseq_uncentred_raw(api, xt.data_ptr(), n, p, 1, cap=1)
    assert rc == abi.ECAP and np.array_equal(n_out, [len(g["Gamma.pos"]) for g in got])
    # a non-finite element, resident (found on the device) and in host memory (found before the upload)
    L = api.load_library()
    xt[5, 4321] = float("inf")
    rc, *_ = _initseq_uncentred_raw(api, xt.data_ptr(), n, p, 1)
    assert rc == abi.EINVAL and b"all elements of argument must be finite" in L.mcmchip_last_error()
    xbad = np.asfortranarray(x)
    xbad[100, 200] = np.nan
    rc, *_ = _initseq_uncentred_raw(api, xbad.ctypes.data, n, p, 0)
    assert rc == abi.EINVAL and b"all elements of argument must be finite" in L.mcmchip_last_error()
    # tiny series: n = 1 (no lag pair), 2, 3
    for nn in (1, 2, 3, 5):
        v = rng.standard_normal((nn, 2)) + 10.0
        vt = torch.from_numpy(np.ascontiguousarray(v.T)).cuda()
        a, b = api.initseq(vt), api.initseq(torch.from_numpy(np.ascontiguousarray((v - v.mean(axis=0)).T)).cuda(), centred=True)
        for c in range(2):
            assert len(a[c]["Gamma.pos"]) == len(b[c]["Gamma.pos"])
            assert np.allclose(a[c]["gamma0"], b[c]["gamma0"], rtol=1e-12, atol=1e-15)
            assert np.allclose(a[c]["Gamma.pos"], b[c]["Gamma.pos"], rtol=1e-10, atol=1e-14)


def test_time_sharded_initseq_and_olbm_device_route(api, oracle_mod):
    """mcmc_amd/dist.py with device-resident shards (CUDA tensors, nothing but p and p x p / ncol x nlags doubles on the
    host): ranks emulated in turn without a process group, the all-reduce replaced by a sum over the emulated ranks.
    initseq sharded by time (raw lag sums of the block minus those of its halo) and olbm sharded by time, against the
    oracle on the whole series."""
    import torch
    from mcmc_amd import dist as mdist
    from scipy.signal import lfilter
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    n, ws, halo = 200_000, 3, 2048
    x = np.column_stack([lfilter([1.0], [1.0, -rho], rng.standard_normal(n)) for rho in (0.3, 0.8, 0.95)]) + 0.7
    p = x.shape[1]
    mean = x.mean(axis=0)
    # ---- initseq: what each emulated rank contributes per round
    nl, total = 64, None
    while True:
        total = np.zeros((p, nl))
        for r in range(ws):
            lo, hi, own = mdist.initseq_time_shard(n, halo, r, ws)
            blk = torch.from_numpy(np.ascontiguousarray((x[lo:hi] - mean).T)).to(dev)        # (ncol, rows), centred
            part = api.lagsums(blk, nl)
            if hi - lo > own:
                part = part - api.lagsums(blk[:, own:].contiguous(), nl)
            total += part
        res = api.initseq_from_lagsums(total, n)
        if all(r_ is not None for r_ in res):
            break
        nl *= 4
        assert nl <= halo
    for c in range(p):
        ref = oracle_mod.initseq(x[:, c] - mean[c])
        assert len(res[c]["Gamma.pos"]) == len(ref["Gamma.pos"])
        for key in ("gamma0", "Gamma.pos", "Gamma.dec", "Gamma.con", "var.pos", "var.dec", "var.con"):
            assert np.max(np.abs(np.asarray(res[c][key]) - np.asarray(ref[key]))) <= 1e-10 * np.max(np.abs(ref[key])), (c, key)
    # the single-GPU entry point gives the same numbers (raw sums are the same kernel)
    whole = api.initseq(x)
    for c in range(p):
        assert np.allclose(whole[c]["Gamma.con"], res[c]["Gamma.con"], rtol=1e-9, atol=1e-12)
    # ---- the driver itself with one rank (no process group) on a device tensor
    xd = torch.from_numpy(np.ascontiguousarray(x.T)).to(dev)
    one = mdist.initseq_time_sharded(api.lagsums, api.initseq_from_lagsums, xd, n, halo)
    for c in range(p):
        assert np.allclose(one[c]["Gamma.con"], whole[c]["Gamma.con"], rtol=1e-9, atol=1e-12)
    # ---- olbm by time on device tensors
    blen = 500
    tot = np.zeros((p, p))
    for r in range(ws):
        lo, hi, own = mdist.olbm_time_shard(n, blen, r, ws)
        blk = torch.from_numpy(np.ascontiguousarray(x[lo:hi].T)).to(dev)
        v, _ = api.olbm_device(blk, blen, mean=mean)
        tot += v * (float(hi - lo - blen + 1) * (hi - lo) * blen)
    got = tot / (float(n - blen + 1) * n * blen)
    ref, _ = oracle_mod.olbm(x, blen)
    assert np.max(np.abs(got - ref)) <= 1e-10 * np.max(np.abs(ref))
    v1, m1 = mdist.olbm_sharded(lambda xb, l, m: api.olbm_device(xb, l, mean=m)[0], xd, n, blen)
    assert np.max(np.abs(v1 - ref)) <= 1e-10 * np.max(np.abs(ref)) and np.allclose(m1, mean, rtol=1e-13)


@pytest.mark.parametrize("n,p,blen", [(40_000, 3, 1024), (150_000, 200, 5000), (66_000, 256, 1027), (1_200_000, 17, 3000),
                                       (30_000, 64, 1500), (50_000, 129, 2048), (13_000, 256, 4000), (3_100_000, 2, 1_000_000),
                                       (30_000, 193, 1500)])
def test_olbm_comb_route_vs_oracle(api, oracle_mod, n, p, blen):
    """Windows of >= 1024 rows of more than 192 columns take olbm_comb.hip (fewer: olbm_lane.hip, olbm_small.hip, checked by name below) (x read once: teeth one window length apart swept together, window sums
    relative to the start of each piece on the matrix cores, pieces chained and rank-two terms added afterwards, exact
    means out of the chain).  Against the oracle at 1e-10 with the means supplied (the R front end's default) and
    computed by the call, and against an 80-bit evaluation of the definition at 1e-12; shapes with ragged last steps
    (len not a multiple of 16), a remainder tooth, fewer teeth than a comb holds, one and several workgroups."""
    import math
    from scipy.signal import lfilter
    rng = np.random.default_rng(n % 1000 + p)
    x = np.empty((n, p))
    for c in range(p):
        x[:, c] = lfilter([1.0], [1.0, -(0.1 + 0.85 * c / p)], rng.standard_normal(n)) + (3.0 + 50.0 * (c % 3))
    if p >= 2:
        x[:, 1] += np.linspace(-5.0, 5.0, n)
    ref, refmean = oracle_mod.olbm(x, blen)
    sc = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
    tol = 1e-10 if blen <= 10_000 and 4 * blen <= n else 1e-9     # what olbm.c's own arithmetic is good for (see the one-pass test)
    got = api.olbm(x, blen)                          # means supplied
    assert np.all(np.abs(got - ref) <= tol * sc)
    assert np.array_equal(got, got.T)
    got2 = api.olbm(x, blen, demean=False)           # means computed by the call
    assert np.all(np.abs(got2 - ref) <= tol * sc)
    assert np.array_equal(got2, got2.T)
    k = min(p, 4)
    exact = np.array([math.fsum(x[:, c]) / n for c in range(k)])
    xl = x[:, :k].astype(np.longdouble) - exact.astype(np.longdouble)
    cs = np.vstack([np.zeros((1, k), dtype=np.longdouble), np.cumsum(xl, axis=0)])
    V = cs[blen:] - cs[:-blen]
    true = np.asarray((V.T @ V) / (np.longdouble(n - blen + 1) * n * blen), dtype=np.float64)
    st = np.sqrt(np.outer(np.diag(true), np.diag(true)))
    assert np.all(np.abs(got2[:k, :k] - true) <= 1e-12 * st)
    import torch
    xd = torch.from_numpy(np.ascontiguousarray(x.T)).to("cuda:0")
    v3, m3 = api.olbm_device(xd, blen)
    assert np.array_equal(v3, got2)
    assert np.all(np.abs(m3[:k] - exact) <= 1e-14 * np.abs(exact))
    assert api.last_kernel_time()[0] == _olbm_route(n, p, blen)


def _olbm_route(n, p, blen):
    """Which kernel diag.hip::run_olbm hands a shape to (olbm_small_applies / olbm_lane_applies / olbm_comb_applies)."""
    nwin = n - blen + 1
    lane = 3 <= p <= 192 and blen >= 256 and nwin >= 4 * blen and (p > 8 or n * p >= 40_000_000)
    if lane:                             # a lane per (tooth, column), no scan (olbm_lane.hip); p <= 8: 2 or 4 units side by side
        return "olbm_few_kernel" if p <= 8 else "olbm_lane_kernel"
    if p <= 8:
        return "olbm_small_kernel"       # the streaming kernel on the vector units (olbm_small.hip)
    if p > 64 and p <= 256 and blen >= 1024 and nwin >= 2 * blen:
        return "olbm_comb_kernel"        # x read once, window sums by a scan, 16 consecutive windows per instruction
    return "olbm_mfma_kernel"            # the two-stream matrix-core kernel


@pytest.mark.parametrize("n,p,blen", [(300_000, 9, 256), (200_000, 16, 1000), (120_000, 31, 5000), (90_000, 48, 3001),
                                       (64_000, 64, 12_800), (70_001, 33, 777), (40_000, 130, 1024), (30_000, 192, 700),
                                       (25_000, 100, 333), (20_000, 160, 2500),
                                       (5_000_001, 8, 2000), (10_000_000, 4, 2500), (13_400_000, 3, 700), (8_000_000, 5, 333),
                                       (6_000_000, 7, 40_000)])
def test_olbm_lane_route_vs_oracle(api, oracle_mod, n, p, blen):
    """3 .. 192 columns (up to 8 only from n p = 4e7 on, with 2 or 4 units of the plan side by side in an instruction:
    olbm_few_kernel) and windows of >= 256 rows take olbm_lane.hip: a wave walks four teeth one window length apart with
    one lane per (tooth, column), window sums are running sums in registers (no scan), five rows feed four teeth, units of
    the sweep are dealt round-robin to the workgroups, pieces chained and rank-two terms added afterwards, exact means out
    of the chain.  Against the oracle at 1e-10 (1e-9 where olbm.c's own arithmetic is good for no more: windows beyond
    10 000 rows or a quarter of the series) with the means supplied and computed by the call, against an 80-bit
    evaluation of the definition at 1e-12; block counts 1 .. 12 (odd ones: every pair of blocks with one wave; even ones:
    the farthest pair shared between two waves), a remainder tooth, a last
    super-block of fewer than four teeth, ragged units."""
    import math
    from scipy.signal import lfilter
    rng = np.random.default_rng(n % 1000 + p)
    x = np.empty((n, p))
    for c in range(p):
        x[:, c] = lfilter([1.0], [1.0, -(0.1 + 0.85 * c / p)], rng.standard_normal(n)) + (3.0 + 50.0 * (c % 3))
    x[:, 1] += np.linspace(-5.0, 5.0, n)
    ref, refmean = oracle_mod.olbm(x, blen)
    sc = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
    tol = 1e-10 if blen <= 10_000 and 4 * blen <= n else 1e-9
    got = api.olbm(x, blen)                          # means supplied
    assert api.last_kernel_time()[0] == _olbm_route(n, p, blen) and _olbm_route(n, p, blen) in ("olbm_lane_kernel", "olbm_few_kernel")
    assert np.all(np.abs(got - ref) <= tol * sc)
    assert np.array_equal(got, got.T)
    got2 = api.olbm(x, blen, demean=False)           # means computed by the call
    assert np.all(np.abs(got2 - ref) <= tol * sc)
    assert np.array_equal(got2, got2.T)
    k = min(p, 4)
    exact = np.array([math.fsum(x[:, c]) / n for c in range(k)])
    xl = x[:, :k].astype(np.longdouble) - exact.astype(np.longdouble)
    cs = np.vstack([np.zeros((1, k), dtype=np.longdouble), np.cumsum(xl, axis=0)])
    V = cs[blen:] - cs[:-blen]
    true = np.asarray((V.T @ V) / (np.longdouble(n - blen + 1) * n * blen), dtype=np.float64)
    st = np.sqrt(np.outer(np.diag(true), np.diag(true)))
    assert np.all(np.abs(got2[:k, :k] - true) <= 1e-12 * st)
    import torch
    xd = torch.from_numpy(np.ascontiguousarray(x.T)).to("cuda:0")
    v3, m3 = api.olbm_device(xd, blen)
    assert np.array_equal(v3, got2)
    assert np.all(np.abs(m3[:k] - exact) <= 1e-14 * np.abs(exact))


def test_olbm_lane_series_beyond_the_32_bit_offsets(api, oracle_mod):
    """olbm_lane.hip addresses with 64 bits: a resident series of 1.4e8 rows (beyond the 1.34e8 rows the other matrix-core
    kernels' 32-bit byte offsets reach; they hand such a series to the fp64-FMA kernel) x 6 columns, windows of 10 000 rows.
    Columns 0 and 5 hold the same AR(1) series: their rows of the result agree (a 32-bit offset anywhere would break that); the diagonal against the AR(1)
    theory; the whole matrix against the time-sharded evaluation (two shards of 7e7 rows, each inside the limit) at 1e-10."""
    import torch
    from mcmc_amd import dist as mdist
    dev = torch.device("cuda", 0)
    B, n, p, blen = 512, 140_000_256, 6, 10_000
    assert n % B == 0 and 8 * (4 * n + blen + 64) >= 1 << 32
    x = torch.empty((p, n), dtype=torch.float64, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    s_idx = torch.arange(B, dtype=torch.float64, device=dev)
    rhos = [0.9, 0.5, 0.7, 0.0, 0.3, 0.9]
    for c in range(p - 1):
        rho = rhos[c]
        e = torch.randn((n // B, B), dtype=torch.float64, device=dev, generator=g)
        if rho == 0.0:
            x[c] = e.reshape(-1)
            continue
        w = rho ** s_idx
        loc = torch.cumsum(e / w, dim=1) * w
        last = loc[:, -1].clone()
        carry = torch.zeros_like(last)
        carry[1:] = last[:-1]
        loc += carry[:, None] * (rho * w)[None, :]          # rho^512 = 0 in double precision for these rho
        x[c] = loc.reshape(-1)
        del e, loc
    x[p - 1] = x[0]
    x += 2.5
    torch.cuda.synchronize()
    var, mean = api.olbm_device(x, blen)
    assert api.last_kernel_time()[0] == "olbm_few_kernel"      # p <= 8: two units of the plan side by side in an instruction
    # (the rank-two terms evaluate the expression of the ORDERED pair of columns, so that var is symmetric to the bit; two
    # equal columns therefore agree to rounding, not to the bit)
    sc = np.sqrt(np.outer(np.diag(var), np.diag(var)))
    assert np.all(np.abs(var[0] - var[p - 1]) <= 1e-13 * sc[0]) and var[0, 0] == var[p - 1, p - 1] and mean[0] == mean[p - 1]
    assert np.array_equal(var, var.T)
    for c in range(p):
        theory = (1.0 / (1.0 - rhos[c] ** 2)) * (1.0 + rhos[c]) / (1.0 - rhos[c]) / n
        assert abs(var[c, c] / theory - 1.0) < 0.08, (c, var[c, c] / theory)
    tot = np.zeros((p, p))
    for r in range(2):                                       # what mcmc_amd.dist.olbm_sharded does on two ranks
        lo, hi, own = mdist.olbm_time_shard(n, blen, r, 2)
        assert 8 * (4 * (hi - lo) + blen + 64) < 1 << 32
        v, _ = api.olbm_device(x[:, lo:hi].contiguous(), blen, mean=mean)
        tot += v * (float(hi - lo - blen + 1) * (hi - lo) * blen)
    v2 = tot / (float(n - blen + 1) * n * blen)
    sc = np.sqrt(np.outer(np.diag(var), np.diag(var)))
    assert np.all(np.abs(v2 - var) <= 1e-10 * sc)
    del x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,p,blen", [(1_200_000, 5, 1000), (1_048_576, 70, 64), (1_100_003, 3, 1), (1_300_000, 130, 4097),
                                       (1_050_000, 2, 525_000), (1_048_581, 2, 1_048_581)])
def test_olbm_one_pass_route_vs_oracle(api, oracle_mod, n, p, blen):
    """Series of >= 2^20 rows without supplied means take the one-pass route (diag.hip: window sums about a provisional
    centre, exact means from the windows that tile the series, rank-two correction of the matrix).  Against the oracle,
    which computes the means first (olbm.c:38-43), on columns chosen to stress the correction: a large offset, a
    trend, and a column whose spikes sit exactly on the rows the provisional centre samples (the sample misses the mean
    by almost the spike height, which has to trigger the repeat about the exact mean)."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(n % 1000 + p)
    x = np.empty((n, p))
    for c in range(p):
        x[:, c] = lfilter([1.0], [1.0, -(0.2 + 0.7 * c / p)], rng.standard_normal(n)) + (3.0 + 100.0 * (c % 3))
    if p >= 2:
        x[:, 1] += np.linspace(-5.0, 5.0, n)                           # trend: the mean is not the mean of any stretch
    if p >= 5:
        x[:, 4] = 7.25                                                  # a constant column: variance exactly 0
    if p >= 3:
        runs = 4096                                                     # OL_PILOT_RUNS in diag.hip
        t0 = (np.arange(runs, dtype=np.int64) * (n - 8)) // (runs - 1)
        x[:, 2] = 1e-3 * rng.standard_normal(n)
        for k in range(8):
            x[t0 + k, 2] += 1.0
    import math
    ref, refmean = oracle_mod.olbm(x, blen)
    got = api.olbm(x, blen, demean=False)          # no.calc.mean = FALSE: the device computes the means (R/olbm.R:8-14)
    # What the reference's arithmetic is good for depends on the shape: olbm.c:38-43 adds the n terms of a mean one after
    # the other (~sqrt(n) ulp), a window sum moves by len times that error, and olbm.c:52-62 carries one running sum
    # through all n - len updates.  With len = n/2 that leaves olbm.c itself 1.4e-10 from the true matrix
    # (tools/olbm_conditioning.py: oracle 1.4e-10, device 6e-14 against an 80-bit evaluation).  So: the device against the
    # 80-bit evaluation of the definition at 1e-12 (first five columns), and against the oracle at 1e-10, or 1e-9 where
    # the windows are long.
    k = min(p, 5)
    exact = np.array([math.fsum(x[:, c]) / n for c in range(k)])
    xl = x[:, :k].astype(np.longdouble) - exact.astype(np.longdouble)
    cs = np.vstack([np.zeros((1, k), dtype=np.longdouble), np.cumsum(xl, axis=0)])
    V = cs[blen:] - cs[:-blen]
    true = np.asarray((V.T @ V) / (np.longdouble(n - blen + 1) * n * blen), dtype=np.float64)
    del xl, cs, V
    st = np.sqrt(np.outer(np.diag(true), np.diag(true)))
    assert np.all(np.abs(got[:k, :k] - true) <= 1e-12 * st + 1e-24)
    scale_ = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
    tol = 1e-10 if blen <= 10_000 else 1e-9
    assert np.all(np.abs(got - ref) <= tol * scale_ + 1e-24)        # len == n: a single window, var is rounding only
    assert np.array_equal(got, got.T)
    got2 = api.olbm(x, blen, mean=refmean)         # one kernel, no correction
    # (a single window about the oracle's sequentially summed mean: the window sum is n times that mean's error, ~5e-6)
    assert np.all(np.abs(got2 - ref) <= tol * scale_ + 1e-20)
    # the means the one-pass route returns: centre + delta, good to an ulp of the centre, which for the spike column
    # (centre ~ 1, mean ~ 0.03) is 4e-15 of the mean
    import torch
    xd = torch.from_numpy(np.ascontiguousarray(x.T)).to("cuda:0")
    v3, m3 = api.olbm_device(xd, blen)
    assert np.array_equal(v3, got)
    # (a few ulp of the CENTRE: for the spike column, centre ~ 1 and mean ~ 0.03, one ulp of the centre is 7e-15 of the mean)
    assert np.allclose(m3[:k], exact, rtol=6e-14, atol=1e-17), (m3[:k] - exact) / exact
    assert np.allclose(m3, refmean, rtol=1e-11, atol=1e-15)


def test_mfma_logdens_bits_do_not_depend_on_the_call_shape(api):
    """ADVICE (round 1): the observation split followed the chain count of the call, so resharding moved log densities
    in the last bits.  The partial sums are now per unit of observations (a function of the data alone): the same state
    has the same log density, bit for bit, in a call of 3, 64, 300 or 4096 chains, and a sharded Philox run equals the
    unsharded one on the MFMA path."""
    from mcmc_amd import abi
    rng = np.random.default_rng(12)
    n, d = 70_000, 20
    X = rng.standard_normal((n, d)); X[:, 0] = 1.0
    beta = 0.3 * rng.standard_normal(d)
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    tg = api.logistic_target(X, y)
    states = beta + 0.05 * rng.standard_normal((4096, d))
    full = api.logdens(tg, states, path=abi.PATH_MFMA)
    for C in (3, 64, 300, 1025):
        part = api.logdens(tg, states[:C], path=abi.PATH_MFMA)
        assert np.array_equal(part, full[:C]), C
    tail = api.logdens(tg, states[4000:], path=abi.PATH_MFMA)
    assert np.array_equal(tail, full[4000:])
    # a Philox run sharded 3 ways (ragged) against the unsharded run, MFMA path pinned
    init = states[:700]
    whole = api.metrop(tg, init, 5, blen=2, scale=0.02, seed=5, path=abi.PATH_MFMA)
    for lo, hi in ((0, 256), (256, 300), (300, 700)):
        sh = api.metrop(tg, init[lo:hi], 5, blen=2, scale=0.02, seed=5, chain_offset=lo, path=abi.PATH_MFMA)
        assert np.array_equal(sh["batch"], whole["batch"][lo:hi]) and np.array_equal(sh["final"], whole["final"][lo:hi])


def test_bench_two_rank_strong_split_rehearsal():
    """bench.py launched as the driver launches it for N > 1 (torch.distributed.run, one process per rank), with two ranks
    sharing this box's one GPU (`--rehearse-on-one-gpu`: cuda:0 for every rank, gloo on host copies for the collectives):
    the strong split of the job's 65 536 chains, the global chain ids of the Philox streams and initial states, the
    summary all-gather and the max-over-ranks timing all run; the job accepts exactly as many proposals as the one-rank run
    (the chains do not know how they were sharded)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--rehearse-on-one-gpu"] + common,
                         capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert two.returncode == 0, two.stderr[-2000:]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True,
                         timeout=200, env=env, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    r2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    r1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert r2["n_gpus"] == 2 and r2["scaling"] == "strong" and r2["config"]["chains_per_gpu"] == 32768
    assert r2["config"]["chains_total"] == r1["config"]["chains_total"] == 65536
    # the same accept decisions chain by chain; the job's rate is a sum of per-chain rates (thirds at 3 steps) taken per
    # rank and then over ranks, so only its rounding may differ
    acc2, acc1 = r2["config"]["accept_rate"] * 65536 * 3, r1["config"]["accept_rate"] * 65536 * 3
    assert abs(acc2 - acc1) < 1e-6 and abs(acc1 - round(acc1)) < 1e-6
    assert r2["value"] > 0 and r2["roofline"]["frac"] > 0       # two ranks share one GPU here: no rate is asserted


# ----------------------------------------------------------------------------- cross-chain diagnostics
@pytest.mark.parametrize("nchain,nbatch,nout", [(4096, 10, 5), (333, 40, 64), (7, 25, 130), (1, 12, 3), (50, 1, 4),
                                                (2, 2, 1), (70000, 3, 2)])
def test_chain_summary_vs_oracle(api, oracle_mod, nchain, nbatch, nout):
    """mcmchip_chain_summary against the numpy restatement of its definitions: batch means with a level far above their
    spread (1 +- 1e-3: raw sums of squares would lose 6 digits, the pairwise moments must not), from host memory and
    from HBM, per-chain outputs included; one chain / one batch give NaN where 0 / 0 says so."""
    import torch
    rng = np.random.default_rng(nchain * 131 + nbatch)
    level = 1.0 + 0.2 * rng.standard_normal(nout)
    chain_eff = 3e-4 * rng.standard_normal((nchain, 1, nout))
    b = level + chain_eff + 1e-3 * rng.standard_normal((nchain, nbatch, nout))
    ref = oracle_mod.chain_summary(b)
    got_h = api.chain_summary(b, per_chain=True)
    got_d = api.chain_summary(torch.from_numpy(b).to("cuda:0"), per_chain=True)
    for got in (got_h, got_d):
        assert got["nchain"] == nchain and got["nbatch"] == nbatch
        for key in ("grand_mean", "between", "within", "mcse_between", "mcse_within", "rhat", "chain_mean", "chain_var"):
            g = got[key].cpu().numpy() if hasattr(got[key], "cpu") else got[key]
            r = ref[key]
            assert np.array_equal(np.isnan(g), np.isnan(r)), key
            ok = ~np.isnan(r)
            # variances of a 3e-4 .. 1e-3 spread around 1: a chain mean is good to eps * level = 2e-16, so a variance of
            # the means is conditioned to eps * level / spread ~ 1e-12 whoever computes it (the oracle's numpy means
            # included); 1e-11 holds both sides to that, where raw sums of squares would sit at eps * (level / spread)^2 ~ 1e-9
            tol = 1e-11 if key in ("between", "mcse_between", "rhat") else 1e-12
            assert np.all(np.abs(g[ok] - r[ok]) <= tol * np.abs(r[ok]) + 1e-300), (key, np.max(np.abs(g[ok] - r[ok]) / np.abs(r[ok])))
    for key in ("grand_mean", "between", "within", "stats"):
        assert np.array_equal(got_h[key], got_d[key], equal_nan=True), key          # same kernel, same bits
    # shards merged in rank order give the whole: ragged split into three
    if nchain >= 3:
        cuts = [0, nchain // 3, nchain // 3 + 1 + nchain // 2 if nchain > 5 else 2, nchain]
        cuts = sorted(set(min(c, nchain) for c in cuts))
        parts = [api.chain_summary(b[lo:hi])["stats"] for lo, hi in zip(cuts[:-1], cuts[1:])]
        whole = api.chain_summary_combine(np.stack(parts), nbatch)
        assert whole["nchain"] == nchain
        for key in ("grand_mean", "between", "within", "mcse_between", "mcse_within", "rhat"):
            r = ref[key]
            ok = ~np.isnan(r)
            assert np.all(np.abs(whole[key][ok] - r[ok]) <= 1e-11 * np.abs(r[ok]) + 1e-300), key


def test_chain_summary_pooled_initseq_and_a_real_run(api, oracle_mod):
    """mcse_initseq: R/initseq.R on every chain's batch means (the device's own initseq on nchain * nout centred
    columns), var.con pooled, against the oracle's initseq chain by chain; then the summary of an actual batched run:
    the three standard errors of the grand mean agree with each other and the chains agree (rhat ~ 1)."""
    rng = np.random.default_rng(11)
    nchain, nbatch, nout = 6, 400, 3
    e = rng.standard_normal((nchain, nbatch, nout))
    b = np.empty_like(e)
    b[:, 0] = e[:, 0]
    for t in range(1, nbatch):
        b[:, t] = 0.6 * b[:, t - 1] + e[:, t]                # autocorrelated batch means: initseq has something to find
    b += 5.0
    ref = oracle_mod.chain_summary(b, with_initseq=True)
    got = api.chain_summary(b, initseq=True)
    assert np.all(np.abs(got["mcse_initseq"] - ref["mcse_initseq"]) <= 1e-10 * ref["mcse_initseq"])
    assert np.all(got["mcse_initseq"] > 1.5 * got["mcse_within"])      # rho = 0.6: batch means understate, initseq does not
    with pytest.raises(Exception, match="few long chains"):
        api.chain_summary(np.zeros((70000, 4, 2)), initseq=True)
    # a run: 512 chains on the logit posterior from the same start, device Philox
    X, y = load_logit()
    mle = glm_binomial(X, y)
    out = api.metrop(api.logistic_target(X, y), np.tile(mle, (512, 1)), 50, 100, 1, scale=0.4, seed=5)
    s = api.chain_summary(out["batch"])
    assert s["nchain"] == 512 and s["nbatch"] == 50
    assert np.all(np.abs(s["rhat"] - 1.0) < 0.05)
    assert np.all(np.abs(s["mcse_between"] / s["mcse_within"] - 1.0) < 0.25)
    assert np.all(np.abs(s["grand_mean"] - mle) < 0.2)        # posterior means sit near the MLE under the weak prior


# ----------------------------------------------------------------------------- one call, several devices
def test_metrop_and_temper_multi_device_list(api, oracle_mod):
    """mcmchip_metrop_multi / mcmchip_temper_multi: one call, the chains (ensembles) split over a device list, one host
    thread per entry.  The test box has one GPU, so the list is [0, 0] / [0, 0, 0]: blocks side by side on one device,
    each worker with a context of its own (streams, flag words, MFMA workspace).  MFMA path and temper: bit for bit the
    single call, Philox and injected streams, ragged blocks, more devices than chains; stream path: same decisions."""
    rng = np.random.default_rng(21)
    n, d, Cn = 4096, 16, 301                                  # ragged over 2 and 3 blocks
    X = rng.standard_normal((n, d)) / 4
    X[:, 0] = 1.0
    y = (rng.random(n) < 0.5).astype(float)
    tg = api.logistic_target(X, y)
    init = 0.05 * rng.standard_normal((Cn, d))
    one = api.metrop(tg, init, 4, 3, 2, scale=0.02, seed=99, path=abi.PATH_MFMA)
    for devs in ([0, 0], [0, 0, 0], [0]):
        many = api.metrop(tg, init, 4, 3, 2, scale=0.02, seed=99, path=abi.PATH_MFMA, devices=devs)
        for key in ("accept", "batch", "final", "accept.batch"):
            assert np.array_equal(one[key], many[key]), (devs, key)
    # AUTO is resolved on the whole call (here: the MFMA kernel), not per block
    auto1 = api.metrop(tg, init, 4, 3, 2, scale=0.02, seed=99)
    auto2 = api.metrop(tg, init, 4, 3, 2, scale=0.02, seed=99, devices=[0, 0, 0, 0, 0])
    assert np.array_equal(auto1["batch"], auto2["batch"])
    # injected streams and the debug trace (stream path), small posterior; more devices than chains
    Xl, yl = load_logit()
    tl = api.logistic_target(Xl, yl)
    i5 = 0.1 * rng.standard_normal((5, 5))
    niter = 6
    z, u = rng.standard_normal((5, niter, 5)), rng.random((5, niter))
    ref = oracle_mod.metrop(oracle_mod.Target(abi.T_LOGISTIC, 5, x=Xl, y=yl, prior_c=0.125), i5, 3, 2, 1, scale=0.4, z=z, u=u,
                            debug=True)
    got = api.metrop(tl, i5, 3, 2, 1, scale=0.4, z=z, u=u, debug=True, devices=[0] * 7)
    assert np.array_equal(got["final"], ref["final"]) and np.array_equal(got["batch"], ref["batch"])
    assert np.array_equal(got["proposal"], ref["proposal"]) and np.array_equal(got["debug.accept"], ref["debug_accept"])
    # an error in one block is the call's error: chain 3 starts outside the unit ball
    ball = api.Target(abi.T_BALL, 2)
    bad = np.zeros((6, 2))
    bad[3] = [2.0, 0.0]
    with pytest.raises(Exception, match="-Inf at initial state"):
        api.metrop(ball, bad, 5, scale=0.1, seed=1, devices=[0, 0])
    # temper: ensembles split, both kernels' shapes
    k, p, E = 6, 2, 11
    par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.arange(k) / p)
    wt = api.Target(abi.T_WITCH, p, par=par)
    nb = np.abs(np.subtract.outer(np.arange(k), np.arange(k))) == 1
    t1 = api.temper(wt, np.zeros((E, k, p)), nb, 5, 4, 3, 0.3, parallel=True, seed=8)
    t2 = api.temper(wt, np.zeros((E, k, p)), nb, 5, 4, 3, 0.3, parallel=True, seed=8, devices=[0, 0, 0])
    for key in ("batch", "acceptx", "accepti", "final"):
        assert np.array_equal(t1[key], t2[key], equal_nan=True), key
    s0 = np.tile(np.concatenate([[1.0], np.zeros(p)]), (E, 1))
    s1 = api.temper(wt, s0, nb, 5, 4, 3, 0.3, parallel=False, seed=8, debug=True)
    s2 = api.temper(wt, s0, nb, 5, 4, 3, 0.3, parallel=False, seed=8, debug=True, devices=[0, 0])
    for key in ("batch", "ibatch", "acceptx", "accepti", "final", "state", "proposal", "unif.choose", "log.hastings"):
        assert np.array_equal(s1[key], s2[key], equal_nan=True), key
    # the interrupt callback runs on the calling thread and stops every block
    calls = []
    api.set_interrupt_poll(lambda: calls.append(1) or len(calls) > 3)
    api.load_library().mcmchip_set_launch_limit(64, 0.0)
    try:
        with pytest.raises(Exception, match="interrupted"):
            api.metrop(tl, np.tile(i5, (200, 1)), 20000, scale=0.4, seed=3, devices=[0, 0])
    finally:
        api.set_interrupt_poll(None)
        api.load_library().mcmchip_set_launch_limit(0, 0.0)
    # bad lists
    with pytest.raises(Exception, match="device"):
        api.metrop(tl, i5, 3, scale=0.4, seed=1, devices=[0, 99])
    with pytest.raises(Exception, match="device list"):
        api.metrop(tl, i5, 3, scale=0.4, seed=1, devices=[])


# ----------------------------------------------------------------------------- BASELINE configs 4 and 5 at full size
@pytest.mark.parametrize("target", ["witch", "mixture"])
def test_c4_full_size_64_temperatures_1024_ensembles_vs_oracle(api, oracle_mod, target):
    """BASELINE.json config 4 at its size: k = 64 temperatures (tri-diagonal neighbours), 1024 ensembles, parallel
    tempering; 600 iterations with injected streams, EVERY ensemble against the oracle (16 threads): final states and
    the acceptance tables bit for bit, batch means to 1e-12; the run kernel (the one C4 takes) and the LDS kernel."""
    rng = np.random.default_rng(64)
    k, E, nbatch, blen, nspac = 64, 1024, 10, 6, 10
    niter = nbatch * blen * nspac
    if target == "witch":
        p = 3
        par = 1 - 0.5 ** (1 / p) * 0.25 ** (np.linspace(0, 5, k) / p)
        K, x0 = abi.T_WITCH, np.zeros(p)
    else:
        p = 2
        par = np.concatenate([[float(k)], np.geomspace(1.0, 0.01, k), abi.normmix_par([0.5, 0.5], [[-4.0, -4.0], [4.0, 4.0]], 1.0)])
        K, x0 = abi.T_TEMPERED_NORMMIX, np.array([4.0, 4.0])
    i, j = np.indices((k, k))
    nb = np.abs(i - j) == 1
    unif, norm = rng.random((E, niter, 4)), rng.standard_normal((E, niter, p))
    init = np.tile(x0, (E, k, 1))
    ref = oracle_mod.temper(oracle_mod.Target(K, p, par=par), init, nb, nbatch, blen, nspac, 0.3, parallel=True,
                            unif=unif, norm=norm, nthreads=16)
    for path in (abi.PATH_RUN, abi.PATH_LDS):
        o = api.temper(api.Target(K, p, par=par), init, nb, nbatch, blen, nspac, 0.3, parallel=True, unif=unif, norm=norm,
                       _path=path)
        assert np.array_equal(o["final"], ref["final"]), path
        assert np.array_equal(o["acceptx"], ref["acceptx"], equal_nan=True), path
        assert np.array_equal(o["accepti"], ref["accepti"], equal_nan=True), path
        assert rel(o["batch"] + 10, ref["batch"] + 10) < 1e-12, path
    assert np.nanmean(ref["acceptx"]) > 0.05 and np.nanmean(ref["accepti"]) > 0.05        # the chains move and swap


def test_c5_full_size_by_properties(api):
    """BASELINE.json config 5 at its size: 1e8 steps x 256 AR(1) functionals, 204.8 GB resident (generated on the device,
    bench_secondary.ar1_columns).  The oracle cannot follow here (it covers n <= 1e7 elsewhere); size-independent
    properties instead:
      initseq  var.con / gamma0 against the AR(1) value (1 + rho) / (1 - rho); duplicated columns give the same bits; the
               identities of tests/initseq.R (cummin, last element 0, variance identity) on every column; a time-sharded
               evaluation (two halves + halo, raw lag sums added) equals the unsharded one to 1e-10
      olbm     diagonal against n var = gamma0 (1 + rho) / (1 - rho) within the estimator's own error; symmetric; the means
               route and the supplied-means route agree to 1e-10."""
    import torch
    import bench_secondary
    from mcmc_amd import dist as mdist
    dev = torch.device("cuda", 0)
    n, p, blen = 100_000_256, 256, 10_000
    x = bench_secondary.ar1_columns(n, p, torch, dev)
    x[7] = x[200]                                           # duplicated columns (a long one over a short one)
    x[255] = x[254]
    rho = 0.5 + 0.49 * np.arange(p) / (p - 1)
    rho[7], rho[255] = rho[200], rho[254]
    res = api.initseq(x, centred=True)
    g0 = np.array([r["gamma0"] for r in res])
    vc = np.array([r["var.con"] for r in res])
    theory = (1 + rho) / (1 - rho)
    # var.con / gamma0 estimates (1 + rho) / (1 - rho) with relative error ~ sqrt(2 (1 + rho) / ((1 - rho) n)) x a few
    tol = 6 * np.sqrt(2 * theory / n) + 0.01
    assert np.all(np.abs(vc / g0 / theory - 1) < tol), np.max(np.abs(vc / g0 / theory - 1) / tol)
    assert np.all(np.abs(g0 * (1 - rho * rho) - 1) < 6 * np.sqrt(2 * theory / n) + 0.005)      # gamma0 = 1 / (1 - rho^2)
    for a, b in ((7, 200), (255, 254)):
        for key in ("gamma0", "Gamma.pos", "Gamma.con", "var.con"):
            assert np.array_equal(res[a][key], res[b][key]), (a, key)
    for r in res:
        assert np.array_equal(r["Gamma.dec"], np.minimum.accumulate(r["Gamma.pos"]))
        assert r["Gamma.pos"][-1] == 0 and r["Gamma.dec"][-1] == 0
        assert np.isclose(2 * r["Gamma.con"].sum() - r["gamma0"], r["var.con"], rtol=1e-12)
    # sharded by time: two ranks emulated in turn on views of the resident series (no copy of 100 GB: the second rank's
    # block is a strided view made contiguous column by column inside lagsums' contract -> use 8 columns)
    cols = [0, 100, 200, 254, 7, 31, 128, 255]
    halo = 2048
    total = None
    for r in range(2):
        lo, hi, own = mdist.initseq_time_shard(n, halo, r, 2)
        blk = torch.stack([x[c, lo:hi] for c in cols]).contiguous()          # 8 x (n/2 + halo): 3.2 GB
        part = api.lagsums(blk, halo)
        if hi - lo > own:
            part = part - api.lagsums(blk[:, own:].contiguous(), halo)
        total = part if total is None else total + part
        del blk
    sh = api.initseq_from_lagsums(total, n)
    for i_, c in enumerate(cols):
        assert sh[i_] is not None and len(sh[i_]["Gamma.pos"]) == len(res[c]["Gamma.pos"]), c
        for key in ("gamma0", "Gamma.pos", "Gamma.con", "var.con"):
            a, b = np.asarray(sh[i_][key]), np.asarray(res[c][key])
            assert np.max(np.abs(a - b)) <= 1e-10 * np.max(np.abs(b)), (c, key)
    # olbm
    var, mean = api.olbm_device(x, blen)
    assert np.array_equal(var, var.T)
    var2, _ = api.olbm_device(x, blen, mean=mean)
    assert np.max(np.abs(var - var2)) <= 1e-10 * np.max(np.abs(var))
    # olbm estimates sigma^2 / n with sigma^2 = gamma0 (1 + rho) / (1 - rho), biased down by ~ (1 + rho) / ((1 - rho) len)-ish
    # for len = 1e4 and noisy to ~ sqrt(len / n) = 1 %
    s2 = theory / (1 - rho * rho)
    ratio = np.diag(var) * n / s2
    assert np.all(np.abs(ratio - 1) < 0.08), (ratio.min(), ratio.max())
    assert np.all(np.abs(mean) < 6 * np.sqrt(s2 / n) + 1e-9)                  # the series was centred on the device
    assert abs(var[7, 7] - var[200, 200]) <= 1e-12 * var[200, 200] and abs(var[7, 200] - var[200, 200]) <= 1e-12 * var[200, 200]
    del x
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------- +Inf / NaN in the middle of a run
def test_posinf_and_nan_mid_run_are_the_references_errors(api):
    """logh() refuses +Inf and NA/NaN wherever they turn up (metrop.c:317-320; temper.c:971-974 for the density,
    578-580 for the Hastings ratio).  Here the initial state is fine and a later proposal is not; the device finishes
    the launch and then reports, in the reference's words, on every path: stream kernel, MFMA path, both temper kernels."""
    z2 = np.zeros((3, 2))
    # stream: a "normal" with precision -1e300 I is +Inf a step of 1e5 away; with precision diag(1.7e308, -1.7e308) it is
    # Inf - Inf as soon as both coordinates exceed 1 (the device reports after the launch: were only one term to
    # overflow in some later proposal, +Inf would be raised as well and named first)
    for P, text in ((-1e300 * np.eye(2), r"logh: func returned \+Inf"), (np.diag([1.7e308, -1.7e308]), "logh: func returned NA or NaN")):
        tg = api.Target(abi.T_MVNORM, 2, par=np.concatenate([np.zeros(2), P.ravel()]))
        assert np.all(api.logdens(tg, z2) == 0.0)
        for init in (z2, z2[0]):
            with pytest.raises(Exception, match=text):
                api.metrop(tg, init, 20, scale=1e5, seed=4, path=abi.PATH_STREAM)
    # MFMA: the logistic posterior with a prior "precision" of the wrong sign: - prior_c sum(beta^2) with prior_c = -1e308
    # is +Inf as soon as the chain has left the origin; prior_c = 0 and steps of 1e200 make the same term 0 * Inf
    rng = np.random.default_rng(8)
    n, d = 4096, 8
    X = rng.standard_normal((n, d))
    y = (rng.random(n) < 0.5).astype(float)
    for prior_c, scale, text in ((-1e308, 5.0, r"logh: func returned \+Inf"), (0.0, 1e200, "logh: func returned NA or NaN")):
        tg = api.Target(abi.T_LOGISTIC, d, x=X, y=y, prior_c=prior_c)
        assert np.all(np.isfinite(api.logdens(tg, np.zeros((2, d)))))
        for path in (abi.PATH_MFMA, abi.PATH_STREAM):
            with pytest.raises(Exception, match=text):
                api.metrop(tg, np.zeros((128, d)), 6, scale=scale, seed=4, path=path)
    # temper, LDS kernel: the model-selection posterior with the same priors
    from rhelpers import temp_par_setup
    Xf, yf, models, masks, betas, nb = temp_par_setup()
    for prior_c, scale, text in ((-1e308, 5.0, r"log unnormalized density function returned \+Inf"),
                                 (0.0, 1e200, "log unnormalized density function returned NA or NaN")):
        tg = api.Target(abi.T_LOGISTIC_MODELS, 4, x=Xf, y=yf, prior_c=prior_c, par=abi.logistic_models_par(masks))
        for parallel in (True, False):
            init = np.zeros((5, 8, 4)) if parallel else np.tile(np.concatenate([[8.0], np.zeros(4)]), (5, 1))
            with pytest.raises(Exception, match=text):
                api.temper(tg, init, nb, 10, 4, 1, scale, parallel=parallel, seed=4)
    # temper, run kernel (parallel, closed-form descriptor): h(i, x) = beta_i (10 - |x|^2) with beta_i = 5e307 overflows
    # to +Inf inside |x|^2 < 6.4 ...
    k = 4
    i_, j_ = np.indices((k, k))
    nbr = np.abs(i_ - j_) == 1
    mix = np.array([1.0, 10.0, 1.0, 0.0, 0.0])                        # K = 1: a = 10, b = 1, mean 0
    tg = api.Target(abi.T_TEMPERED_NORMMIX, 2, par=np.concatenate([[float(k)], np.full(k, 5e307), mix]))
    init = np.tile([np.sqrt(12.0), 0.0], (4, k, 1))
    for path in (abi.PATH_RUN, abi.PATH_LDS):
        with pytest.raises(Exception, match=r"returned \+Inf"):
            api.temper(tg, init, nbr, 20, 5, 1, 1.5, parallel=True, seed=4, _path=path)
    # ... and an inverse temperature of 0 turns the -Inf the mixture takes once a step of 1e160 has overflowed |x|^2
    # into 0 * -Inf
    mix2 = np.array([1.0, 0.0, 1.0, 0.0, 0.0])
    tg = api.Target(abi.T_TEMPERED_NORMMIX, 2, par=np.concatenate([[float(k)], [1.0, 0.5, 0.2, 0.0], mix2]))
    init = np.tile([1.0, 0.0], (4, k, 1))
    for path in (abi.PATH_RUN, abi.PATH_LDS):
        with pytest.raises(Exception, match="returned NA or NaN"):
            api.temper(tg, init, nbr, 20, 5, 1, 1e160, parallel=True, seed=4, _path=path)

@pytest.mark.parametrize("parallel", [True, False])
@pytest.mark.parametrize("outfun", [None, "moments2"])
def test_temper_mfma_path_vs_oracle_and_lds_kernel(api, oracle_mod, parallel, outfun):
    """temper_mfma.hip: tempering over the model-selection posterior with the ensembles in lockstep and the log densities
    on the matrix cores (one launch of the density kernel over 2 nens states per iteration).  Injected draws: the
    oracle's acceptance tables, batch means and final states; device Philox: the LDS kernel's.  A random neighbour graph,
    a list of scales, n not a multiple of anything."""
    rng = np.random.default_rng(31 + int(parallel))
    n, p, k, E = 6007, 5, 6, 40
    nbatch, blen, nspac = 6, 4, 3
    niter = nbatch * blen * nspac
    X = rng.standard_normal((n, p)) / np.sqrt(p)
    X[:, 0] = 1.0
    beta = 0.5 * rng.standard_normal(p)
    y = (rng.random(n) < 1 / (1 + np.exp(-X @ beta))).astype(float)
    masks = (rng.random((k, p)) < 0.6).astype(float)
    masks[:, 0] = 1.0
    masks[0] = 1.0
    nb = np.zeros((k, k), dtype=bool)
    for i in range(k - 1):
        nb[i, i + 1] = nb[i + 1, i] = True
    nb[0, 3] = nb[3, 0] = True
    par = abi.logistic_models_par(masks, log_pseudo_prior=0.3 * rng.standard_normal(k), constant=-2.0)
    kw = dict(x=X, y=y, prior_c=0.125, par=par)
    scale = [0.02 * (1 + i) * np.ones(p) for i in range(k)]            # one diagonal scale per component (temper.c:998-999)
    if parallel:
        init = beta + 0.05 * rng.standard_normal((E, k, p))
    else:
        init = np.column_stack([rng.integers(1, k + 1, E).astype(float), beta + 0.05 * rng.standard_normal((E, p))])
    unif, norm = rng.random((E, niter, 4)), rng.standard_normal((E, niter, p))
    okind = oracle_mod.TOUT_MOMENTS2 if outfun == "moments2" else oracle_mod.TOUT_NONE
    ref = oracle_mod.temper(oracle_mod.Target(abi.T_LOGISTIC_MODELS, p, **kw), init, nb, nbatch, blen, nspac, scale,
                            parallel=parallel, unif=unif, norm=norm, out_kind=okind, nthreads=16)
    tg = api.Target(abi.T_LOGISTIC_MODELS, p, **kw)
    o = api.temper(tg, init, nb, nbatch, blen, nspac, scale, parallel=parallel, unif=unif, norm=norm, outfun=outfun,
                   _path=abi.PATH_TMFMA)
    assert api.last_kernel_time()[0] == "logdens_mfma_kernel"
    assert np.array_equal(o["final"], ref["final"])
    assert np.array_equal(np.isnan(o["accepti"]), np.isnan(ref["accepti"]))
    assert np.array_equal(np.nan_to_num(o["accepti"]), np.nan_to_num(ref["accepti"]))
    assert np.array_equal(np.nan_to_num(o["acceptx"]), np.nan_to_num(ref["acceptx"]))
    assert np.max(np.abs(o["batch"] - ref["batch"])) <= 1e-12 * np.max(np.abs(ref["batch"]))
    if not parallel:
        assert np.array_equal(o["ibatch"], ref["ibatch"])
    assert 0.05 < np.nanmean(o["acceptx"]) < 0.98
    # device Philox: same draws in the same order as the one-wavefront-per-ensemble kernel
    a = api.temper(tg, init, nb, nbatch, blen, nspac, scale, parallel=parallel, seed=77, outfun=outfun, _path=abi.PATH_TMFMA)
    b = api.temper(tg, init, nb, nbatch, blen, nspac, scale, parallel=parallel, seed=77, outfun=outfun, _path=abi.PATH_LDS)
    assert np.array_equal(a["final"], b["final"])
    assert np.array_equal(np.nan_to_num(a["accepti"]), np.nan_to_num(b["accepti"]))
    assert np.max(np.abs(a["batch"] - b["batch"])) <= 1e-12 * np.max(np.abs(b["batch"]))
    # a restart continues the stream (tests/saveseed.R contract): two halves == one run
    h1 = api.temper(tg, init, nb, nbatch // 2, blen, nspac, scale, parallel=parallel, seed=77, outfun=outfun, _path=abi.PATH_TMFMA)
    h2 = api.temper(h1, _path=abi.PATH_TMFMA)
    assert np.array_equal(h2["final"], a["final"])
