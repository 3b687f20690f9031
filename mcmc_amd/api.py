"""Host-side mirror of the reference's R front ends for the hot path.

    metrop()   R/metrop.R:2-67      temper()  R/temper.R:2-75
    initseq()  R/initseq.R:1-5      olbm()    R/olbm.R:2-24

Same argument names, meaning and error behaviour; `obj` is a log-density *descriptor*
(`Target`) instead of an R closure, or the result of a previous run (restart semantics of
R/metrop.R:6-27).  All compute goes through the C-ABI of include/mcmchip.h into
libmcmchip.so (hand-written HIP).  There is NO CPU fallback: if the library is missing or no
GPU is present the calls raise.
"""
import ctypes as C
import os
import time

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_INIT_DEVICE = None


class McmcHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def library_path():
    return os.path.join(_HERE, "libmcmchip.so")


def load_library():
    """dlopen libmcmchip.so and declare the prototypes of include/mcmchip.h (no GPU needed)"""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise McmcHipError(abi.EHIP, "libmcmchip.so is not built: run `make -C mcmc_amd/csrc` "
                           "(there is no CPU fallback)")
    L = C.CDLL(path)
    L.mcmchip_init.argtypes = [C.c_int]
    L.mcmchip_init.restype = C.c_int
    L.mcmchip_last_error.restype = C.c_char_p
    L.mcmchip_version.restype = C.c_char_p
    L.mcmchip_target_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.mcmchip_target_create.restype = C.c_int
    L.mcmchip_target_destroy.argtypes = [C.c_void_p]
    L.mcmchip_target_destroy.restype = None
    L.mcmchip_logdens.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
    L.mcmchip_logdens.restype = C.c_int
    L.mcmchip_logdens_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.mcmchip_logdens_noalloc.restype = C.c_int
    L.mcmchip_metrop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mcmchip_metrop.restype = C.c_int
    L.mcmchip_temper.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.mcmchip_temper.restype = C.c_int
    L.mcmchip_metrop_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.mcmchip_metrop_multi.restype = C.c_int
    L.mcmchip_temper_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.mcmchip_temper_multi.restype = C.c_int
    L.mcmchip_initseq.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64] + [C.c_void_p] * 8
    L.mcmchip_initseq.restype = C.c_int
    L.mcmchip_initseq_uncentred.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64] + [C.c_void_p] * 9
    L.mcmchip_initseq_uncentred.restype = C.c_int
    L.mcmchip_chain_summary.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int, C.c_void_p]
    L.mcmchip_chain_summary.restype = C.c_int
    L.mcmchip_chain_summary_combine.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.mcmchip_chain_summary_combine.restype = C.c_int
    L.mcmchip_lagsums.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    L.mcmchip_lagsums.restype = C.c_int
    L.mcmchip_initseq_from_lagsums.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64] + [C.c_void_p] * 8
    L.mcmchip_initseq_from_lagsums.restype = C.c_int
    L.mcmchip_olbm.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                               C.c_int32, C.c_int]
    L.mcmchip_olbm.restype = C.c_int
    L.mcmchip_series_upload.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
    L.mcmchip_series_upload.restype = C.c_int
    L.mcmchip_series_free.argtypes = [C.c_void_p]
    L.mcmchip_series_free.restype = C.c_int
    L.mcmchip_series_centre.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.mcmchip_series_centre.restype = C.c_int
    L.mcmchip_last_kernel_time.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]
    L.mcmchip_last_kernel_time.restype = C.c_int
    L.mcmchip_sync.restype = C.c_int
    L.mcmchip_set_interrupt_poll.argtypes = [C.c_void_p, C.c_void_p]
    L.mcmchip_set_interrupt_poll.restype = C.c_int
    L.mcmchip_set_launch_limit.argtypes = [C.c_int64, C.c_double]
    L.mcmchip_set_launch_limit.restype = C.c_int
    _LIB = L
    return L


def _check(rc):
    if rc != abi.OK:
        raise McmcHipError(rc, load_library().mcmchip_last_error().decode())


def init(device=None):
    """select the HIP device (default: LOCAL_RANK, else 0)"""
    global _INIT_DEVICE
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if _INIT_DEVICE is None else _INIT_DEVICE
    if _INIT_DEVICE != device:
        _check(load_library().mcmchip_init(int(device)))
        _INIT_DEVICE = device
    return device


_POLL_KEEPALIVE = None


def set_interrupt_poll(fn=None):
    """fn() is called on the host between two kernel launches of a sampler call (the place of
    R_CheckUserInterrupt, metrop.c:269 / temper.c:848); a true return abandons the call (McmcHipError, code
    EINTR).  None removes the callback."""
    global _POLL_KEEPALIVE
    if fn is None:
        cb = None
        _check(load_library().mcmchip_set_interrupt_poll(None, None))
    else:
        cb = C.CFUNCTYPE(C.c_int, C.c_void_p)(lambda _ctx: 1 if fn() else 0)
        _check(load_library().mcmchip_set_interrupt_poll(C.cast(cb, C.c_void_p), None))
    _POLL_KEEPALIVE = cb


def set_launch_limit(max_iters=0, budget_seconds=0.0):
    """iterations per kernel launch of the persistent sampler kernels (0: time-based, about budget_seconds each)"""
    _check(load_library().mcmchip_set_launch_limit(int(max_iters), float(budget_seconds)))


def last_kernel_time():
    """(name, average ms per launch, launches) of the dominant kernel of the last call"""
    ms, n = C.c_double(), C.c_int64()
    name = C.create_string_buffer(64)
    _check(load_library().mcmchip_last_kernel_time(C.byref(ms), C.byref(n), name, 64))
    return name.value.decode(), ms.value, n.value


class Target(abi.HostTarget):
    """Log unnormalised density descriptor, resident in HBM (replaces `obj`/`lud`)."""

    def __init__(self, kind, d, x=None, y=None, prior_c=0.0, par=None, device=None, morph=None):
        super().__init__(kind, d, x=x, y=y, prior_c=prior_c, par=par, morph=morph)
        self._device = device
        init(device)
        h = C.c_void_p()
        _check(load_library().mcmchip_target_create(C.addressof(self.desc), 0, C.byref(h)))
        self.handle = h

    def with_morph(self, morph):
        """the same descriptor seen through a variable transformation (morph$lud(obj), R/morph.R:166-179)"""
        return Target(self.kind, self.d, x=self.x, y=self.y, prior_c=self.prior_c, par=self.par,
                      device=self._device, morph=morph)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load_library().mcmchip_target_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def logistic_target(x, y, prior_sd=2.0, device=None):
    """the vignette's posterior (demo.Rnw:59-65): iid N(0, prior_sd^2) prior on the coefficients"""
    x = np.asarray(x, dtype=np.float64)
    return Target(abi.T_LOGISTIC, x.shape[1], x=x, y=y, prior_c=1.0 / (2.0 * prior_sd ** 2), device=device)


def logdens(target, states, path=abi.PATH_AUTO):
    s = np.ascontiguousarray(states, dtype=np.float64)
    if s.ndim == 1:
        s = s[None]
    out = np.empty(s.shape[0])
    _check(load_library().mcmchip_logdens(target.handle, s.ctypes.data, s.shape[0], int(path), 0,
                                          out.ctypes.data))
    return out


# ---------------------------------------------------------------------------------------------
def metrop(obj, initial=None, nbatch=None, blen=None, nspac=None, scale=None, outfun=None, debug=None,
           seed=None, z=None, u=None, path=abi.PATH_AUTO, chain_offset=0, _iter_offset=0, devices=None):
    """metrop(obj, initial, nbatch, blen = 1, nspac = 1, scale = 1, outfun, debug = FALSE)

    obj     a Target, or the dict returned by a previous call (then missing arguments default
            to the previous ones, the run starts at obj["final"] and continues the same random
            stream: R/metrop.R:6-27, tests/saveseed.R)
    initial d-vector (one chain, result shapes as in R) or nchain x d matrix (batched extension:
            results gain a leading chain axis)
    seed    key of the device Philox stream (the role of .Random.seed); z, u: injected streams
            (parity mode)
    devices list of HIP device indices: the chains are split over them inside this one call
            (mcmchip_metrop_multi: one host thread per entry, descriptor replicated, same results)
    Returns a dict with the reference's names: accept, batch, initial, final, accept.batch
    [, current, proposal, log.green, u, z, debug.accept], initial.seed, final.seed, time, lud,
    nbatch, blen, nspac, scale, outfun, debug.
    """
    if isinstance(obj, dict):
        prev = obj
        target = prev["lud"]
        initial = prev["final"] if initial is None else initial
        nbatch = prev["nbatch"] if nbatch is None else nbatch
        # missing(blen) etc. take the value stored in the object (R/metrop.R:9-15)
        blen = prev["blen"] if blen is None else blen
        nspac = prev["nspac"] if nspac is None else nspac
        scale = prev["scale"] if scale is None else scale
        debug = prev["debug"] if debug is None else debug
        outfun = prev["outfun"] if outfun is None else outfun
        seed = prev["final.seed"][0] if seed is None else seed
        _iter_offset = prev["final.seed"][1]
        chain_offset = prev["final.seed"][2]
    else:
        target = obj
    blen = 1 if blen is None else blen          # defaults of R/metrop.R:2-3
    nspac = 1 if nspac is None else nspac
    scale = 1 if scale is None else scale
    debug = False if debug is None else debug
    if not isinstance(target, Target):
        raise TypeError('argument "obj" must be a Target (log-density descriptor) or a previous result')
    if initial is None or nbatch is None:
        raise ValueError('argument "initial"/"nbatch" missing')
    single = np.ndim(initial) == 1
    if z is not None or u is not None:
        mode = abi.RNG_INJECTED
    else:
        mode = abi.RNG_PHILOX
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
    call = abi.MetropCall(target.d, initial, nbatch, blen, nspac, scale, outfun, rng_mode=mode, z=z, u=u,
                          seed=seed or 0, chain_offset=chain_offset, iter_offset=_iter_offset,
                          debug=debug, path=path)
    t0 = time.perf_counter()
    if devices is None:
        _check(load_library().mcmchip_metrop(target.handle, C.addressof(call.args), C.addressof(call.out)))
    else:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        _check(load_library().mcmchip_metrop_multi(target.handle, C.addressof(call.args), C.addressof(call.out),
                                                   devs.ctypes.data, devs.size))
    elapsed = time.perf_counter() - t0
    r = call.result()

    def sq(a):
        return a[0] if single else a

    out = {"accept": sq(r["accept"]), "batch": sq(r["batch"]), "initial": sq(r["initial"]),
           "final": sq(r["final"]), "accept.batch": sq(r["accept_batch"])}
    if debug:
        out.update({"current": sq(r["current"]), "proposal": sq(r["proposal"]),
                    "log.green": sq(r["log_green"]), "u": sq(r["u"]), "z": sq(r["z"]),
                    "debug.accept": sq(r["debug_accept"]).astype(bool)})
    out.update({"initial.seed": (seed, _iter_offset, chain_offset),
                "final.seed": (seed, r["iter_offset"], chain_offset),
                "time": elapsed, "lud": target, "nbatch": int(nbatch), "blen": int(blen),
                "nspac": int(nspac), "scale": scale})
    if outfun is not None:
        out["outfun"] = outfun
    else:
        out["outfun"] = None
    out["debug"] = bool(debug)
    out["path"] = {abi.PATH_STREAM: "stream", abi.PATH_MFMA: "mfma"}.get(r["path_used"], "?")
    return out


def morph_metrop(obj, initial=None, nbatch=None, blen=None, nspac=None, scale=None, outfun=None, debug=None,
                 morph=None, seed=None, z=None, u=None):
    """morph.metrop(obj, initial, nbatch, blen = 1, nspac = 1, scale = 1, outfun, debug = FALSE, morph)
    (R/morph.metrop.R): metrop on the density induced by the change of variable `morph`
    (mcmc_amd.morph(b, r, p, center)); the chain runs on the morphed state on the device, the descriptor
    evaluates lud(inverse(y)) + log.jacobian(y) and the output functional sees inverse(y).

    obj   an (unmorphed) Target, or the dict returned by a previous morph_metrop call.
    Returns metrop's dict plus "morph", "morph.final" (the last morphed state) with "final" = inverse of it
    and "outfun" = the caller's outfun (R/morph.metrop.R:55-62).

    Reference behaviour kept: the first call does not forward `nspac` (R/morph.metrop.R:42-49), so it runs
    with nspac = 1 whatever is passed; a restart takes missing arguments from the object (17-21) and starts
    at obj["morph.final"] (10), or, when a new `morph` is given, at morph.transform(obj["final"]) (14).
    Deliberate deviation: with a new `morph` on restart the reference keeps sampling obj$lud, the density
    morphed by the FIRST transformation (metrop.metropolis uses obj$lud), while outfun and final use the new
    one; here the density is rebuilt with the new transformation as well.
    """
    from ._morph import Morph
    if isinstance(obj, dict):
        prev = dict(obj)
        if "morph" not in prev:
            raise TypeError('argument "obj" must be a Target or the result of morph_metrop')
        if morph is None:
            morph = prev["morph"]
            target = prev["lud"]
            start = prev["morph.final"]                                  # morph.metrop.R:9-10
        else:
            target = prev["base"].with_morph(morph)
            fin = np.asarray(prev["final"], dtype=np.float64)
            start = (morph.transform(fin) if fin.ndim == 1
                     else np.stack([morph.transform(v) for v in fin]))   # morph.metrop.R:14
        outfun = prev["outfun"] if outfun is None else outfun            # morph.metrop.R:17-21
        blen = prev["blen"] if blen is None else blen
        nspac = prev["nspac"] if nspac is None else nspac
        debug = prev["debug"] if debug is None else debug
        scale = prev["scale"] if scale is None else scale
        base = prev["base"]
        prev["lud"], prev["final"] = target, start
        if initial is not None:
            prev["final"] = initial
        res = metrop(prev, nbatch=nbatch, blen=blen, nspac=nspac, scale=scale, outfun=outfun, debug=debug,
                     seed=seed, z=z, u=u, path=abi.PATH_STREAM)
    else:
        if not isinstance(obj, Target):
            raise TypeError('argument "obj" must be a Target or the result of morph_metrop')
        if morph is None:
            morph = Morph()                                              # morph.identity(), morph.metrop.R:39
        if initial is None or nbatch is None:
            raise ValueError('argument "initial"/"nbatch" missing')
        base = obj
        target = obj.with_morph(morph)
        ini = np.asarray(initial, dtype=np.float64)
        start = morph.transform(ini) if ini.ndim == 1 else np.stack([morph.transform(v) for v in ini])
        # nspac is not forwarded on the first call (morph.metrop.R:42-49)
        res = metrop(target, start, nbatch, blen=blen, nspac=None, scale=scale, outfun=outfun, debug=debug,
                     seed=seed, z=z, u=u, path=abi.PATH_STREAM)
    mf = np.asarray(res["final"], dtype=np.float64)                       # .morph.unmorph, morph.metrop.R:55-62
    res["morph"] = morph
    res["morph.final"] = mf
    res["final"] = morph.inverse(mf) if mf.ndim == 1 else np.stack([morph.inverse(v) for v in mf])
    res["outfun"] = outfun
    res["base"] = base
    return res


def temper(obj, initial=None, neighbors=None, nbatch=None, blen=None, nspac=None, scale=None, outfun=None,
           debug=None, parallel=None, seed=None, unif=None, norm=None, ens_offset=0, _iter_offset=0, _path=0,
           devices=None):
    """temper(obj, initial, neighbors, nbatch, blen = 1, nspac = 1, scale = 1, outfun,
              debug = FALSE, parallel = FALSE)   -- R/temper.R:2-75

    initial  parallel: k x p matrix (or nens x k x p); serial: c(i, x) (or nens x (1+p))
    Returns the reference's names: lud, neighbors, nbatch, blen, nspac, scale, outfun, debug,
    parallel, initial.seed, final.seed, time, batch, acceptx, accepti, initial, final [, ibatch].
    outfun   None (identity, temper.c:851-867); "witch.corner", the indicator functional of the reference's
             witch's-hat tests (tests/temp-par-witch.R:52-59, temp-ser-witch.R:51-62); an index or logical vector
             over the identity output (metrop's rules, metrop.c:435-497; parallel: the k x p state in R's
             column-major order) or "moments2" = c(v, v^2); R closures cannot run on the device.  A restart keeps
             the previous outfun when none is given (R/temper.R:16-18).
    debug = TRUE returns the reference's trace
    (which, unif.which, state, log.hastings, unif.hastings, proposal, acceptd, norm, unif.choose[, coproposal]).
    """
    if isinstance(obj, dict):
        prev = obj
        target = prev["lud"]
        initial = prev["final"] if initial is None else initial
        neighbors = prev["neighbors"] if neighbors is None else neighbors
        nbatch = prev["nbatch"] if nbatch is None else nbatch
        blen = prev["blen"] if blen is None else blen      # R/temper.R:6-35
        nspac = prev["nspac"] if nspac is None else nspac
        scale = prev["scale"] if scale is None else scale
        parallel = prev["parallel"] if parallel is None else parallel
        debug = prev["debug"] if debug is None else debug
        outfun = prev["outfun"] if outfun is None else outfun
        seed = prev["final.seed"][0] if seed is None else seed
        _iter_offset = prev["final.seed"][1]
        ens_offset = prev["final.seed"][2]
    else:
        target = obj
    blen = 1 if blen is None else blen
    nspac = 1 if nspac is None else nspac
    scale = 1 if scale is None else scale
    parallel = False if parallel is None else parallel
    debug = False if debug is None else debug
    if not isinstance(target, Target) or not target.is_temper_kind:
        raise TypeError('argument "obj" must be a tempering Target or a previous result')
    init_arr = np.asarray(initial, dtype=np.float64)
    out_index = None
    if outfun is None:
        out_kind = abi.TOUT_IDENTITY
    elif isinstance(outfun, str) and outfun == "witch.corner":
        out_kind = abi.TOUT_WITCH_CORNER
    elif isinstance(outfun, str) and outfun == "moments2":
        out_kind = abi.TOUT_MOMENTS2
    elif callable(outfun):
        raise NotImplementedError("temper: outfun functions cannot run on the device path")
    elif isinstance(outfun, str):
        raise ValueError('argument "outfun" must be function or NULL')          # temper.c:68-69
    else:
        # index / logical vector over the identity output (parallel: the k x p state in R's column-major order,
        # serial: x), with metrop's rules (metrop.c:435-497)
        k_ = np.asarray(neighbors).shape[0]
        nid = k_ * target.d if parallel else target.d
        kind_, mask, _ = abi.resolve_outfun(outfun, nid)
        out_kind, out_index = abi.TOUT_INDEX, mask
    single = init_arr.ndim == (2 if parallel else 1)
    if unif is not None or norm is not None:
        mode = abi.RNG_INJECTED
    else:
        mode = abi.RNG_PHILOX
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
    call = abi.TemperCall(target.d, init_arr, neighbors, nbatch, blen, nspac, scale, parallel, rng_mode=mode,
                          unif=unif, norm=norm, seed=seed or 0, ens_offset=ens_offset,
                          iter_offset=_iter_offset, debug=debug, out_kind=out_kind, path=_path, out_index=out_index)
    t0 = time.perf_counter()
    if devices is None:
        _check(load_library().mcmchip_temper(target.handle, C.addressof(call.args), C.addressof(call.out)))
    else:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        _check(load_library().mcmchip_temper_multi(target.handle, C.addressof(call.args), C.addressof(call.out),
                                                   devs.ctypes.data, devs.size))
    elapsed = time.perf_counter() - t0
    r = call.result()

    def sq(a):
        return a[0] if single else a

    out = {"lud": target, "neighbors": np.asarray(neighbors, dtype=bool), "nbatch": int(nbatch),
           "blen": int(blen), "nspac": int(nspac), "scale": scale, "outfun": outfun, "debug": bool(debug),
           "parallel": bool(parallel), "initial.seed": (seed, _iter_offset, ens_offset),
           "final.seed": (seed, r["iter_offset"], ens_offset), "time": elapsed,
           "batch": sq(r["batch"]), "acceptx": sq(r["acceptx"]), "accepti": sq(r["accepti"]),
           "initial": init_arr, "final": sq(r["final"])}
    if not parallel:
        out["ibatch"] = sq(r["ibatch"])
    if debug:      # slot order of temper.c:282-464
        for key in ("which", "unif.which", "state", "log.hastings", "unif.hastings", "proposal", "acceptd",
                    "norm", "unif.choose") + (("coproposal",) if parallel else ()):
            out[key] = sq(r[key])
    return out


def logistic_models_target(x, y, models, prior_sd=2.0, log_pseudo_prior=None, normalized_prior=True, device=None):
    """the model-selection posterior of vignettes/bfst.Rnw:472-489 (serial tempering over sub-models of a logistic
    regression): `models` is the k x (d - 1) 0/1 matrix of which predictors enter model i (the intercept always does),
    the prior is iid N(0, prior_sd^2) with its normalizing constant (bfst.Rnw:453-459: needed when models of different
    dimension are compared) unless normalized_prior is False (tests/temp-par.R:39-56)"""
    x = np.asarray(x, dtype=np.float64)
    models = np.asarray(models)
    masks = np.column_stack([np.ones(models.shape[0]), models]).astype(np.float64)
    d = x.shape[1]
    const = -d * np.log(prior_sd * np.sqrt(2.0 * np.pi)) if normalized_prior else 0.0
    par = abi.logistic_models_par(masks, log_pseudo_prior, const)
    return Target(abi.T_LOGISTIC_MODELS, d, x=x, y=y, prior_c=1.0 / (2.0 * prior_sd ** 2), par=par, device=device)


def tune_pseudo_prior(x, y, models, initial, neighbors, nbatch, blen=1, nspac=1, scale=1, prior_sd=2.0,
                      log_pseudo_prior=None, max_rounds=50, ratio=2.0, cap=10.0, seed=None):
    """The trial-and-error loop of vignettes/bfst.Rnw:497-559 as a driver: serial tempering over the sub-models, then
        qux <- qux + pmin(log(max(ibar) / ibar), 10);  qux <- qux - min(qux)        (bfst.Rnw:516-517, 546-547)
    with ibar = colMeans(ibatch), continuing from the last state (544), until max(ibar) / min(ibar) < 2 (550).
    `initial` is c(i, beta) or nens x (1 + d) (ibar then averages over the ensembles too).
    As in the vignette the update comes before the test, so "log.pseudo.prior" is one update ahead of the last run;
    "log.pseudo.prior.run" is what `out` was sampled with.
    Returns {"log.pseudo.prior", "log.pseudo.prior.run", "qux.save", "ibar.save", "rounds", "out"}; the log 10
    unnormalized Bayes factors of bfst.Rnw:610 are (log.pseudo.prior.run - log(colMeans(out["ibatch"]))) / log(10)."""
    models = np.asarray(models)
    k = models.shape[0]
    qux = np.zeros(k) if log_pseudo_prior is None else np.asarray(log_pseudo_prior, dtype=np.float64).copy()
    if seed is None:
        seed = int.from_bytes(os.urandom(8), "little")
    qux_save, ibar_save = [], []
    out = None
    for rounds in range(1, int(max_rounds) + 1):
        tg = logistic_models_target(x, y, models, prior_sd=prior_sd, log_pseudo_prior=qux)
        if out is None:
            out = temper(tg, initial, neighbors, nbatch, blen, nspac, scale, parallel=False, seed=seed)
        else:
            prev = dict(out)
            prev["lud"] = tg                       # temper(out, log.pseudo.prior = qux): same stream, new pseudo-prior
            out = temper(prev)
        ib = np.asarray(out["ibatch"])
        ibar = ib.reshape(-1, k).mean(axis=0)
        ibar_save.append(ibar)
        qux_run = qux                              # what this run sampled with: Bayes factors pair it with this ibar
        with np.errstate(divide="ignore"):
            qux = qux + np.minimum(np.log(ibar.max() / ibar), cap)
        qux = qux - qux.min()
        qux_save.append(qux.copy())
        if ibar.min() > 0 and ibar.max() / ibar.min() < ratio:      # after the update, as bfst.Rnw:546-550 has it
            break
    return {"log.pseudo.prior": qux, "log.pseudo.prior.run": qux_run, "qux.save": np.array(qux_save),
            "ibar.save": np.array(ibar_save), "rounds": rounds, "out": out}


def initseq(x, centred=False):
    """initseq(x): R/initseq.R:1-5 -- x numeric, all finite; the series is centred here and the
    centred series goes to the native routine.  A 2-d x (len x ncol) is a batch of series and
    returns a list of results.
    A DeviceSeries (upload_series) or a float64 CUDA tensor of shape (ncol, len) is resident in HBM already and is not
    modified: with centred = True it is taken as the native routine takes its argument (initseq.c:7), otherwise about its
    column means in the same passes (mcmchip_initseq_uncentred: nothing is subtracted in place).  The device checks that it
    is finite."""
    about_mean = False
    if isinstance(x, DeviceSeries) or (hasattr(x, "data_ptr") and hasattr(x, "is_cuda")):
        single = False
        src = x
        about_mean = not centred
    else:
        x = np.asarray(x)
        if not np.issubdtype(x.dtype, np.number):
            raise TypeError("is.numeric(x) is not TRUE")
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        xm = x[:, None] if single else x
        if x.nbytes >= _UPLOAD_MIN_BYTES:
            # a large series: uploaded as it is, taken about its means (and checked for non-finite values) on the device --
            # three passes over a GB-sized array on the host cost ten times the upload
            init()
            src = upload_series(xm)
            about_mean = not centred
        else:
            if not np.all(np.isfinite(x)):
                raise ValueError("all(is.finite(x)) is not TRUE")
            src = xm if centred else xm - xm.mean(axis=0)
    init()
    ptr, length, ncol, dp, keep = _colmajor(src)             # column-major len x ncol, on the host or in HBM
    L = load_library()
    cap = max(min(length // 2, 4096), 1)
    while True:
        n_out = np.zeros(ncol, dtype=np.int64)
        g0, vp, vd, vc = (np.empty(ncol) for _ in range(4))
        Gp, Gd, Gc = (np.empty((ncol, cap)) for _ in range(3))
        if about_mean:
            rc = L.mcmchip_initseq_uncentred(ptr, length, ncol, dp, cap, n_out.ctypes.data, g0.ctypes.data,
                                             Gp.ctypes.data, Gd.ctypes.data, Gc.ctypes.data, vp.ctypes.data,
                                             vd.ctypes.data, vc.ctypes.data, None)
        else:
            rc = L.mcmchip_initseq(ptr, length, ncol, dp, cap, n_out.ctypes.data, g0.ctypes.data,
                                   Gp.ctypes.data, Gd.ctypes.data, Gc.ctypes.data, vp.ctypes.data,
                                   vd.ctypes.data, vc.ctypes.data)
        if rc == abi.ECAP:
            cap = int(n_out.max())
            continue
        if rc == abi.EINVAL and b"must be finite" in L.mcmchip_last_error():
            raise ValueError("all(is.finite(x)) is not TRUE")
        _check(rc)
        break
    res = []
    for c in range(ncol):
        m = int(n_out[c])
        res.append({"gamma0": g0[c], "Gamma.pos": Gp[c, :m].copy(), "Gamma.dec": Gd[c, :m].copy(),
                    "Gamma.con": Gc[c, :m].copy(), "var.pos": vp[c], "var.dec": vd[c], "var.con": vc[c]})
    return res[0] if single else res


class DeviceSeries:
    """A len x ncol batch of series resident in HBM in the diagnostics' column-major layout (mcmchip_series_upload):
    pass it to initseq / lagsums / olbm / olbm_device in place of the array, as often as needed; PCIe is crossed once."""

    def __init__(self, ptr, length, ncol):
        self.ptr, self.length, self.ncol = ptr, int(length), int(ncol)

    def centre(self):
        """subtract the column means in place (R/initseq.R:3, on the device); returns the means"""
        mean = np.empty(self.ncol)
        _check(load_library().mcmchip_series_centre(self.ptr, self.length, self.ncol, mean.ctypes.data))
        return mean

    def free(self):
        if self.ptr:
            ptr, self.ptr = self.ptr, None
            try:
                load_library().mcmchip_series_free(ptr)
            except Exception:           # interpreter shutdown: the library may be gone already
                pass

    __del__ = free


# host arrays at least this large that are not already column-major go to the device as they are and are transposed
# there (numpy transposes a 1 GB matrix at ~0.4-2.5 GB/s; the upload runs at ~25 GB/s and the device transpose at TB/s)
_UPLOAD_MIN_BYTES = 8 << 20


def upload_series(x):
    """numpy len x ncol (or a vector) -> DeviceSeries.  A Fortran-ordered array (R's layout) is uploaded as it is, a
    C-ordered one is uploaded as it is and transposed on the device."""
    init()
    a = np.asarray(x, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    length, ncol = a.shape
    if a.flags.f_contiguous:
        src, row_major = a, 0
    else:
        src, row_major = np.ascontiguousarray(a), 1
    out = C.c_void_p()
    _check(load_library().mcmchip_series_upload(src.ctypes.data, length, ncol, row_major, C.byref(out)))
    return DeviceSeries(out.value, length, ncol)


def _torch_ready(t):
    """The library launches on streams of its own: work torch has queued on ITS current stream for a tensor handed over
    (a copy, a slice made contiguous, a generator's fill) must have finished before the library reads it -- a 3.4 GB
    `.contiguous()` was still copying its last columns when olbm started on the first."""
    import torch
    torch.cuda.current_stream(t.device).synchronize()


def _colmajor(x):
    """(pointer, len, ncol, device_ptrs, keepalive) of a batch of series for the C-ABI's column-major layout x[t + len*c].
    numpy: len x ncol (or a vector); column-major (Fortran-ordered) arrays are used in place, small C-ordered ones are
    transposed on the host, large ones on the device (upload_series).  DeviceSeries and torch CUDA tensors (ncol x len
    contiguous: already column-major from R's point of view) are used in place -- nothing crosses PCIe."""
    if isinstance(x, DeviceSeries):
        return x.ptr, x.length, x.ncol, 1, x
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda"):
        if not x.is_cuda or x.dim() != 2 or not x.is_contiguous() or str(x.dtype) != "torch.float64":
            raise ValueError("device series must be a contiguous float64 CUDA tensor of shape (ncol, len)")
        _torch_ready(x)
        return x.data_ptr(), int(x.shape[1]), int(x.shape[0]), 1, x
    a = np.asarray(x, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    if a.nbytes >= _UPLOAD_MIN_BYTES and a.shape[1] > 1 and not a.flags.f_contiguous:
        ds = upload_series(a)
        return ds.ptr, ds.length, ds.ncol, 1, ds
    xc = np.ascontiguousarray(a.T)
    return xc.ctypes.data, a.shape[0], a.shape[1], 0, xc


def lagsums(x, nlags):
    """raw lag sums sum_t x[t] x[t + l], l < nlags, per column (ncol x nlags): the additive half of initseq, for series
    sharded by time (mcmc_amd.dist.initseq_time_sharded).  x as for `_colmajor`; used as given (the caller centres)."""
    init()
    ptr, length, ncol, dp, keep = _colmajor(x)
    out = np.empty((ncol, int(nlags)))
    _check(load_library().mcmchip_lagsums(ptr, length, ncol, int(nlags), dp, out.ctypes.data))
    return out


def initseq_from_lagsums(sums, length):
    """the stop rule and the three sequences of initseq (initseq.c:24-105) from TOTAL lag sums (ncol x nlags) of a
    series of `length` elements; host only.  Returns a list of result dicts; None for a column whose stopping lag lies
    beyond the lags given."""
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    ncol, nlags = sums.shape
    cap = max(nlags // 2, 1)
    n_out = np.zeros(ncol, dtype=np.int64)
    g0, vp, vd, vc = (np.empty(ncol) for _ in range(4))
    Gp, Gd, Gc = (np.empty((ncol, cap)) for _ in range(3))
    _check(load_library().mcmchip_initseq_from_lagsums(sums.ctypes.data, int(length), ncol, nlags, cap, n_out.ctypes.data,
                                                       g0.ctypes.data, Gp.ctypes.data, Gd.ctypes.data, Gc.ctypes.data,
                                                       vp.ctypes.data, vd.ctypes.data, vc.ctypes.data))
    res = []
    for c in range(ncol):
        m = int(n_out[c])
        res.append(None if m < 0 else
                   {"gamma0": g0[c], "Gamma.pos": Gp[c, :m].copy(), "Gamma.dec": Gd[c, :m].copy(),
                    "Gamma.con": Gc[c, :m].copy(), "var.pos": vp[c], "var.dec": vd[c], "var.con": vc[c]})
    return res


_SUMMARY_VECTORS = ("grand_mean", "between", "within", "mcse_between", "mcse_within", "rhat")


def _summary_struct(nout, want_initseq):
    vec = {k: np.empty(nout) for k in _SUMMARY_VECTORS}
    if want_initseq:
        vec["mcse_initseq"] = np.empty(nout)
    vec["stats"] = np.empty((5, nout))
    o = abi.ChainSummaryOut()
    for k, v in vec.items():
        setattr(o, k, v.ctypes.data)
    return o, vec


def chain_summary(batch, initseq=False, per_chain=False):
    """Cross-chain diagnostics of a batched run (mcmchip_chain_summary; no counterpart in the one-chain reference):
    batch = out["batch"] of metrop, nchain x nbatch x nout (numpy, or a contiguous float64 CUDA tensor: nothing crosses
    PCIe but 5 nout doubles).  Per output functional: "grand_mean", "between" (variance of the chain means), "within"
    (mean variance of a chain's batch means), "mcse_between" = sqrt(between / nchain), "mcse_within" =
    sqrt(within / (nbatch nchain)), "rhat"; initseq=True adds "mcse_initseq" (R/initseq.R on every chain's batch
    means, var.con pooled); per_chain=True adds "chain_mean" and "chain_var" (nchain x nout).  "stats" (5 x nout) are
    the sufficient statistics that chain_summary_combine / mcmc_amd.dist.chain_summary_sharded merge across ranks."""
    init()
    on_device = hasattr(batch, "data_ptr") and hasattr(batch, "is_cuda")
    if on_device:
        if not batch.is_cuda or batch.dim() != 3 or not batch.is_contiguous() or str(batch.dtype) != "torch.float64":
            raise ValueError("a device batch must be a contiguous float64 CUDA tensor of shape (nchain, nbatch, nout)")
        _torch_ready(batch)
        ptr, keep = batch.data_ptr(), batch
    else:
        keep = np.ascontiguousarray(batch, dtype=np.float64)
        if keep.ndim == 2:
            keep = keep[None]                      # one chain: the reference's nbatch x nout matrix
        if keep.ndim != 3:
            raise ValueError("batch must be nchain x nbatch x nout")
        ptr = keep.ctypes.data
    nchain, nbatch, nout = (int(v) for v in keep.shape)
    o, vec = _summary_struct(nout, initseq)
    if per_chain:
        if on_device:
            import torch
            vec["chain_mean"] = torch.empty((nchain, nout), dtype=torch.float64, device=batch.device)
            vec["chain_var"] = torch.empty((nchain, nout), dtype=torch.float64, device=batch.device)
            o.chain_mean, o.chain_var = vec["chain_mean"].data_ptr(), vec["chain_var"].data_ptr()
        else:
            vec["chain_mean"], vec["chain_var"] = np.empty((nchain, nout)), np.empty((nchain, nout))
            o.chain_mean, o.chain_var = vec["chain_mean"].ctypes.data, vec["chain_var"].ctypes.data
    _check(load_library().mcmchip_chain_summary(ptr, nchain, nbatch, nout, int(on_device), C.addressof(o)))
    vec.update(nchain=nchain, nbatch=nbatch)
    return vec


def chain_summary_combine(stats, nbatch):
    """merge the "stats" (5 x nout each) of several shards IN THE ORDER GIVEN and return the summary of all their chains
    (host only: mcmchip_chain_summary_combine)."""
    st = np.ascontiguousarray(stats, dtype=np.float64)
    if st.ndim != 3 or st.shape[1] != 5:
        raise ValueError("stats must be nranks x 5 x nout")
    nranks, _, nout = st.shape
    o, vec = _summary_struct(nout, True)
    _check(load_library().mcmchip_chain_summary_combine(st.ctypes.data, nranks, int(nbatch), nout, C.addressof(o)))
    vec.update(nchain=int(round(vec["stats"][0, 0])), nbatch=int(nbatch))
    return vec


def olbm_device(x, batch_length, mean=None):
    """mcmchip_olbm on a series that is already in HBM: x is a contiguous float64 CUDA tensor of shape (p, n) (column-major
    n x p); mean given -> the no.calc.mean = TRUE contract (olbm.c:37-43).  Returns (var, mean) as numpy arrays."""
    init()
    ptr, n, p, dp, keep = _colmajor(x)
    if int(batch_length) > n:
        raise ValueError("batch.length must be <= nrow(x)")
    nocalc = mean is not None
    m = np.zeros(p) if mean is None else np.ascontiguousarray(mean, dtype=np.float64).copy()
    var = np.empty((p, p))
    _check(load_library().mcmchip_olbm(ptr, n, p, int(batch_length), m.ctypes.data, var.ctypes.data, int(nocalc), dp))
    return var, m


def olbm(x, batch_length, demean=True, mean=None):
    """olbm(x, batch.length, demean = TRUE): R/olbm.R:2-24.  x is n x p (a vector is n x 1).
    Like the reference, demean = FALSE zeroes `mean` but leaves no.calc.mean FALSE, so the
    native routine recomputes the means (R/olbm.R:8-14 with olbm.c:37): both settings demean.
    mean (not an argument of the R front end): the native routine's no.calc.mean = TRUE contract with means
    supplied by the caller (olbm.c:37-43), used by the time-sharded driver mcmc_amd.dist.olbm_sharded."""
    if isinstance(x, DeviceSeries):
        if mean is None and demean:
            demean = False              # the means of a resident series are the device's to compute (same result)
        n, p = x.length, x.ncol
    else:
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x[:, None]
        n, p = x.shape
    batch_length = int(batch_length)
    if batch_length > n:
        raise ValueError("batch.length must be <= nrow(x)")
    if mean is not None:
        mean = np.asarray(mean, dtype=np.float64).copy()
        if mean.shape != (p,):
            raise ValueError("mean must have one entry per column")
        no_calc_mean = True
    elif demean:
        mean = x.mean(axis=0)
        no_calc_mean = True
    else:
        mean = np.zeros(p)
        no_calc_mean = False
    mean = np.ascontiguousarray(mean, dtype=np.float64)
    var = np.empty((p, p))
    init()
    ptr, n, p, dp, keep = _colmajor(x)                      # column-major n x p, on the host or in HBM
    _check(load_library().mcmchip_olbm(ptr, n, p, batch_length, mean.ctypes.data,
                                       var.ctypes.data, int(no_calc_mean), dp))
    return var
