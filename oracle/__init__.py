"""oracle -- TEST INFRASTRUCTURE: python wrapper of the CPU restatement (oracle.c, rrng.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
It is the checker, never the product: mcmc_amd never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from mcmc_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RNG_RSTREAM = 3
TOUT_NONE, TOUT_WITCH_PAR, TOUT_WITCH_SER, TOUT_INDEX, TOUT_MOMENTS2 = 0, 1, 2, 3, 4


def build(force=False):
    """compile liboracle.so with gcc (idempotent)"""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "rrng.c", "oracle.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "mcmchip.h"))
    stale = force or not os.path.exists(so) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_r_new.restype = C.c_void_p
        L.orc_r_new.argtypes = [C.c_int, C.c_uint32]
        L.orc_r_free.argtypes = [C.c_void_p]
        L.orc_r_clone.restype = C.c_void_p
        L.orc_r_clone.argtypes = [C.c_void_p]
        L.orc_r_count.restype = C.c_uint64
        L.orc_r_count.argtypes = [C.c_void_p]
        L.orc_r_unif.restype = C.c_double
        L.orc_r_unif.argtypes = [C.c_void_p]
        L.orc_r_norm.restype = C.c_double
        L.orc_r_norm.argtypes = [C.c_void_p]
        L.orc_r_qnorm.restype = C.c_double
        L.orc_r_qnorm.argtypes = [C.c_double]
        L.orc_r_norm_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_r_unif_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_philox_normals.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_philox_uniform.restype = C.c_double
        L.orc_philox_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_logdens.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_metrop.restype = C.c_int
        L.orc_metrop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_char_p, C.c_int]
        L.orc_temper.restype = C.c_int
        L.orc_temper.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_char_p, C.c_int]
        L.orc_initseq.restype = C.c_int64
        L.orc_initseq.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 7
        L.orc_olbm.restype = C.c_int
        L.orc_olbm.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        for nm in ("orc_morph_f", "orc_morph_finv", "orc_morph_dfinv"):
            getattr(L, nm).restype = C.c_double
            getattr(L, nm).argtypes = [C.c_void_p, C.c_double]
        L.orc_morph_inverse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_morph_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_morph_logjac.restype = C.c_double
        L.orc_morph_logjac.argtypes = [C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class RStream:
    """Emulated R RNG stream (set.seed(seed) under RNGkind(kind), normal.kind = Inversion)."""

    KINDS = {"Mersenne-Twister": 0, "Marsaglia-Multicarry": 1}

    def __init__(self, seed=42, kind="Mersenne-Twister", _handle=None):
        self.h = _handle if _handle is not None else lib().orc_r_new(self.KINDS[kind], seed)

    def clone(self):
        return RStream(_handle=lib().orc_r_clone(self.h))

    def rnorm(self, n):
        out = np.empty(int(n))
        lib().orc_r_norm_fill(self.h, out.ctypes.data, int(n))
        return out

    def runif(self, n):
        out = np.empty(int(n))
        lib().orc_r_unif_fill(self.h, out.ctypes.data, int(n))
        return out

    @property
    def count(self):
        return int(lib().orc_r_count(self.h))

    def __del__(self):
        try:
            lib().orc_r_free(self.h)
        except Exception:
            pass


def qnorm(p):
    return lib().orc_r_qnorm(float(p))


Target = abi.HostTarget


def logdens(target, states):
    s = np.ascontiguousarray(states, dtype=np.float64)
    if s.ndim == 1:
        s = s[None]
    out = np.empty(s.shape[0])
    lib().orc_logdens(C.addressof(target.desc), s.ctypes.data, s.shape[0], out.ctypes.data)
    return out


def morph_scalar(target, which, x):
    """f / f.inv / d.f.inv of the target's morph (R/morph.R:49-93,133-152) at a scalar"""
    fn = {"f": lib().orc_morph_f, "f.inv": lib().orc_morph_finv, "d.f.inv": lib().orc_morph_dfinv}[which]
    return fn(C.addressof(target.desc), float(x))


def morph_inverse(target, state):
    y = np.ascontiguousarray(state, dtype=np.float64)
    out = np.empty_like(y)
    lib().orc_morph_inverse(C.addressof(target.desc), y.ctypes.data, out.ctypes.data)
    return out


def morph_transform(target, state):
    x = np.ascontiguousarray(state, dtype=np.float64)
    out = np.empty_like(x)
    lib().orc_morph_transform(C.addressof(target.desc), x.ctypes.data, out.ctypes.data)
    return out


def morph_logjac(target, state):
    y = np.ascontiguousarray(state, dtype=np.float64)
    return lib().orc_morph_logjac(C.addressof(target.desc), y.ctypes.data)


def metrop(target, initial, nbatch, blen=1, nspac=1, scale=1.0, outfun=None, debug=False,
           rstream=None, z=None, u=None, seed=None, chain_offset=0, iter_offset=0, nthreads=1):
    """oracle metrop: rstream -> emulated R stream; z,u -> injected; seed -> Philox."""
    if rstream is not None:
        mode = RNG_RSTREAM
    elif z is not None:
        mode = abi.RNG_INJECTED
    else:
        mode = abi.RNG_PHILOX
    call = abi.MetropCall(target.d, initial, nbatch, blen, nspac, scale, outfun,
                          rng_mode=abi.RNG_INJECTED if mode == abi.RNG_INJECTED else abi.RNG_PHILOX,
                          z=z, u=u, seed=seed or 0, chain_offset=chain_offset,
                          iter_offset=iter_offset, debug=debug)
    call.args.rng_mode = mode
    err = C.create_string_buffer(256)
    rc = lib().orc_metrop(C.addressof(target.desc), C.addressof(call.args), C.addressof(call.out),
                          rstream.h if rstream is not None else None, int(nthreads), err, 256)
    if rc:
        raise OracleError(rc, err.value.decode())
    return call.result()


def temper(target, initial, neighbors, nbatch, blen=1, nspac=1, scale=1.0, parallel=False,
           rstream=None, unif=None, norm=None, seed=None, ens_offset=0, iter_offset=0,
           out_kind=TOUT_NONE, nthreads=1, debug=False, out_index=None):
    if rstream is not None:
        mode = RNG_RSTREAM
    elif unif is not None:
        mode = abi.RNG_INJECTED
    else:
        mode = abi.RNG_PHILOX
    k = np.asarray(neighbors).shape[0]
    nout = {TOUT_NONE: None, TOUT_WITCH_PAR: k, TOUT_WITCH_SER: 2 * k, TOUT_INDEX: None, TOUT_MOMENTS2: None}[out_kind]
    abi_kind = {TOUT_INDEX: abi.TOUT_INDEX, TOUT_MOMENTS2: abi.TOUT_MOMENTS2}.get(out_kind, 0)
    call = abi.TemperCall(target.d, initial, neighbors, nbatch, blen, nspac, scale, parallel,
                          rng_mode=abi.RNG_PHILOX, unif=unif, norm=norm, seed=seed or 0,
                          ens_offset=ens_offset, iter_offset=iter_offset, nout=nout, debug=debug,
                          out_kind=abi_kind, out_index=out_index)
    call.args.rng_mode = mode
    err = C.create_string_buffer(256)
    rc = lib().orc_temper(C.addressof(target.desc), C.addressof(call.args), C.addressof(call.out),
                          rstream.h if rstream is not None else None, int(out_kind), int(nthreads),
                          err, 256)
    if rc:
        raise OracleError(rc, err.value.decode())
    return call.result()


def initseq(x):
    """x: 1-d series, already centred (R/initseq.R:4 centres before the .Call)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    cap = max(x.size // 2, 1)
    g0, vp, vd, vc = (C.c_double() for _ in range(4))
    Gp, Gd, Gc = np.empty(cap), np.empty(cap), np.empty(cap)
    m = lib().orc_initseq(x.ctypes.data, x.size, cap, C.addressof(g0), Gp.ctypes.data, Gd.ctypes.data,
                          Gc.ctypes.data, C.addressof(vp), C.addressof(vd), C.addressof(vc))
    return {"gamma0": g0.value, "Gamma.pos": Gp[:m].copy(), "Gamma.dec": Gd[:m].copy(),
            "Gamma.con": Gc[:m].copy(), "var.pos": vp.value, "var.dec": vd.value, "var.con": vc.value}


def olbm(x, batch_length, mean=None):
    """x: n x p (numpy row-major is converted to the reference's column-major). Returns (var, mean).
    mean given -> nocalc = TRUE (olbm.c:37)"""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    n, p = x.shape
    xc = np.ascontiguousarray(x.T)  # column-major n x p
    nocalc = mean is not None
    m = np.zeros(p) if mean is None else np.ascontiguousarray(mean, dtype=np.float64).copy()
    var = np.empty((p, p))
    rc = lib().orc_olbm(xc.ctypes.data, n, p, int(batch_length), m.ctypes.data, var.ctypes.data, int(nocalc))
    if rc:
        raise OracleError(rc, "len > n\n")
    return var, m


def chain_summary(batch, with_initseq=False):
    """Cross-chain reduction of batch means (nchain x nbatch x nout), the definitions of include/mcmchip.h restated with
    numpy (two-pass moments; the reference has no counterpart: its users do colMeans(out$batch) and
    apply(out$batch, 2, sd) / sqrt(out$nbatch) for ONE chain, vignettes/demo.Rnw:210-222).  with_initseq: var.con of
    orc_initseq (initseq.c) on every chain's centred batch means, pooled."""
    b = np.asarray(batch, dtype=np.float64)
    nchain, nbatch, nout = b.shape
    m = b.mean(axis=1)                                   # nchain x nout
    with np.errstate(invalid="ignore", divide="ignore"):
        s2 = ((b - m[:, None, :]) ** 2).sum(axis=1) / (nbatch - 1)
        gm = m.mean(axis=0)
        between = ((m - gm) ** 2).sum(axis=0) / (nchain - 1)
        within = s2.mean(axis=0)
        out = {"grand_mean": gm, "between": between, "within": within, "mcse_between": np.sqrt(between / nchain),
               "mcse_within": np.sqrt(within / (nbatch * nchain)),
               "rhat": np.sqrt(((nbatch - 1) / nbatch * within + between) / within), "chain_mean": m, "chain_var": s2}
    if with_initseq:
        vc = np.array([[initseq(b[c, :, j] - m[c, j])["var.con"] for j in range(nout)] for c in range(nchain)])
        out["mcse_initseq"] = np.sqrt(vc.mean(axis=0) / (nbatch * nchain))
    return out
