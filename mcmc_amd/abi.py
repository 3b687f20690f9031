"""ctypes mirror of include/mcmchip.h (the C-ABI boundary).

Field order and types must match the header exactly; tests/test_abi.py checks the
struct sizes against the values the compiled library reports.
"""
import ctypes as C

OK, EINVAL, EDOMAIN, EHIP, ECAP, EINTR = 0, 1, 2, 3, 4, 5

T_LOGISTIC, T_LOGISTIC_NAIVE, T_MVNORM, T_NORMMIX = 1, 2, 3, 4
T_SIMPLEX, T_BALL, T_WITCH, T_TEMPERED_NORMMIX = 5, 6, 7, 8
T_STUDENT_T = 9
TOUT_IDENTITY, TOUT_WITCH_CORNER, TOUT_INDEX, TOUT_MOMENTS2 = 0, 1, 2, 3
T_LOGISTIC_MODELS = 10
T_BINOMIAL_LOGIT = 11
TEMPER_KINDS = (T_WITCH, T_TEMPERED_NORMMIX, T_LOGISTIC_MODELS)

SCALE_CONSTANT, SCALE_DIAGONAL, SCALE_FULL, SCALE_LIST = 1, 2, 3, 8
OUT_IDENTITY, OUT_INDEX, OUT_MOMENTS2 = 0, 1, 2
RNG_INJECTED, RNG_PHILOX = 1, 2
PATH_AUTO, PATH_STREAM, PATH_MFMA = 0, 1, 2
PATH_LDS, PATH_RUN, PATH_TMFMA = 1, 2, 3      # temper kernels (mcmchip_temper_args.path)

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class TargetDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("d", C.c_int32),
        ("n", C.c_int64),
        ("x", C.c_void_p),
        ("y", C.c_void_p),
        ("prior_c", C.c_double),
        ("par", C.c_void_p),
        ("npar", C.c_int64),
        ("morph_flags", C.c_int32),
        ("morph_pad_", C.c_int32),
        ("morph_b", C.c_double),
        ("morph_r", C.c_double),
        ("morph_p", C.c_double),
        ("morph_center", C.c_void_p),
    ]


class MetropArgs(C.Structure):
    _fields_ = [
        ("nchain", C.c_int64),
        ("nbatch", C.c_int32),
        ("blen", C.c_int32),
        ("nspac", C.c_int32),
        ("initial", C.c_void_p),
        ("scale_kind", C.c_int32),
        ("scale", C.c_void_p),
        ("out_kind", C.c_int32),
        ("out_index", C.c_void_p),
        ("rng_mode", C.c_int32),
        ("z", C.c_void_p),
        ("u", C.c_void_p),
        ("seed", C.c_uint64),
        ("chain_offset", C.c_uint64),
        ("iter_offset", C.c_uint64),
        ("debug", C.c_int32),
        ("device_ptrs", C.c_int32),
        ("path", C.c_int32),
    ]


class MetropOut(C.Structure):
    _fields_ = [
        ("accept", C.c_void_p),
        ("batch", C.c_void_p),
        ("final", C.c_void_p),
        ("accept_batch", C.c_void_p),
        ("current", C.c_void_p),
        ("proposal", C.c_void_p),
        ("z", C.c_void_p),
        ("log_green", C.c_void_p),
        ("u", C.c_void_p),
        ("debug_accept", C.c_void_p),
        ("iter_offset_out", C.c_uint64),
        ("nout", C.c_int32),
        ("path_used", C.c_int32),
    ]


class TemperArgs(C.Structure):
    _fields_ = [
        ("nens", C.c_int64),
        ("ncomp", C.c_int32),
        ("parallel", C.c_int32),
        ("nbatch", C.c_int32),
        ("blen", C.c_int32),
        ("nspac", C.c_int32),
        ("neighbors", C.c_void_p),
        ("initial", C.c_void_p),
        ("scale_kind", C.c_int32),
        ("scale", C.c_void_p),
        ("rng_mode", C.c_int32),
        ("unif", C.c_void_p),
        ("norm", C.c_void_p),
        ("seed", C.c_uint64),
        ("ens_offset", C.c_uint64),
        ("iter_offset", C.c_uint64),
        ("device_ptrs", C.c_int32),
        ("debug", C.c_int32),
        ("out_kind", C.c_int32),
        ("path", C.c_int32),
        ("out_index", C.c_void_p),
    ]


class TemperOut(C.Structure):
    _fields_ = [
        ("batch", C.c_void_p),
        ("ibatch", C.c_void_p),
        ("acceptx", C.c_void_p),
        ("accepti", C.c_void_p),
        ("final", C.c_void_p),
        ("iter_offset_out", C.c_uint64),
        ("which", C.c_void_p),
        ("unif_which", C.c_void_p),
        ("state", C.c_void_p),
        ("log_hastings", C.c_void_p),
        ("unif_hastings", C.c_void_p),
        ("proposal", C.c_void_p),
        ("acceptd", C.c_void_p),
        ("norm", C.c_void_p),
        ("unif_choose", C.c_void_p),
        ("coproposal", C.c_void_p),
    ]


class ChainSummaryOut(C.Structure):
    """mcmchip_chain_summary_out (include/mcmchip.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("grand_mean", "between", "within", "mcse_between", "mcse_within", "rhat",
                                          "mcse_initseq", "stats", "chain_mean", "chain_var")]


# every symbol include/mcmchip.h declares (tests check the library exports them all)
EXPORTS = [
    "mcmchip_init",
    "mcmchip_last_error",
    "mcmchip_version",
    "mcmchip_target_create",
    "mcmchip_target_destroy",
    "mcmchip_logdens",
    "mcmchip_logdens_noalloc",
    "mcmchip_metrop",
    "mcmchip_temper",
    "mcmchip_metrop_multi",
    "mcmchip_temper_multi",
    "mcmchip_initseq",
    "mcmchip_initseq_uncentred",
    "mcmchip_olbm_comb_plan_check",
    "mcmchip_olbm_lane_plan_check",
    "mcmchip_olbm_few_plan_check",
    "mcmchip_lagsums",
    "mcmchip_initseq_from_lagsums",
    "mcmchip_olbm",
    "mcmchip_chain_summary",
    "mcmchip_chain_summary_combine",
    "mcmchip_series_upload",
    "mcmchip_series_free",
    "mcmchip_series_centre",
    "mcmchip_set_interrupt_poll",
    "mcmchip_set_launch_limit",
    "mcmchip_last_kernel_time",
    "mcmchip_sync",
]


def np_ptr(a):
    """address of a C-contiguous numpy array (or None)"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


# ---------------------------------------------------------------------------
# host-side marshalling shared by the product wrapper (mcmc_amd.api) and by the
# test oracle's wrapper (oracle/__init__.py): numpy arrays -> C structs
# ---------------------------------------------------------------------------
import numpy as np


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class HostTarget:
    """A log-density descriptor held in host memory (numpy) plus its C struct."""

    def __init__(self, kind, d, x=None, y=None, prior_c=0.0, par=None, morph=None):
        self.kind, self.d = int(kind), int(d)
        self.morph = morph
        self.x, self.y, self.par = _f64(x), _f64(y), _f64(par)
        self.prior_c = float(prior_c)
        self.n = 0 if self.x is None else self.x.shape[0]
        if self.x is not None:
            if self.x.ndim != 2 or self.x.shape[1] != self.d:
                raise ValueError("x must be n x d")
            if self.y is None or self.y.shape != (self.n,):
                raise ValueError("y must have length n")
        self.desc = TargetDesc(self.kind, self.d, self.n, np_ptr(self.x), np_ptr(self.y),
                               self.prior_c, np_ptr(self.par),
                               0 if self.par is None else self.par.size)
        if morph is not None and morph.flags():
            if self.kind in TEMPER_KINDS:
                raise ValueError("morph applies to metrop targets only")
            self.morph_center = morph.center_vector(self.d)
            self.desc.morph_flags = morph.flags()
            self.desc.morph_b, self.desc.morph_r, self.desc.morph_p = morph.b, morph.r, morph.p
            self.desc.morph_center = np_ptr(self.morph_center)

    @property
    def is_temper_kind(self):
        return self.kind in TEMPER_KINDS


def logistic_models_par(models, log_pseudo_prior=None, constant=0.0):
    """parameter block of MCMCHIP_T_LOGISTIC_MODELS: `models` is the k x d 0/1 matrix of which coefficients
    (intercept included) enter the linear predictor of component i (tests/temp-par.R:12-13,48-50)"""
    m = np.asarray(models, dtype=np.float64)
    k = m.shape[0]
    lpp = np.zeros(k) if log_pseudo_prior is None else np.asarray(log_pseudo_prior, dtype=np.float64)
    if lpp.shape != (k,):
        raise ValueError("length(log.pseudo.prior) == nrow(models) is not TRUE")
    return np.concatenate([[float(k)], m.ravel(), lpp, [float(constant)]])


def normmix_par(weights, means, sds):
    """parameter block of MCMCHIP_T_NORMMIX for an isotropic finite normal mixture"""
    means = np.atleast_2d(np.asarray(means, dtype=np.float64))
    K, d = means.shape
    w = np.asarray(weights, dtype=np.float64)
    s = np.broadcast_to(np.asarray(sds, dtype=np.float64), (K,))
    rec = np.empty((K, 2 + d))
    rec[:, 0] = np.log(w) - 0.5 * d * np.log(2 * np.pi * s * s)
    rec[:, 1] = 1.0 / (2 * s * s)
    rec[:, 2:] = means
    return np.concatenate([[float(K)], rec.ravel()])


def classify_scale(scale, d, ncomp=None):
    """(kind, flat float64 array) -- metrop.c:333-365 / temper.c:915-953 rules"""
    if isinstance(scale, (list, tuple)) and ncomp is not None:
        if len(scale) != ncomp:
            raise ValueError('argument "scale" must have length k if list')
        kinds, flats = zip(*(classify_scale(s, d) for s in scale))
        if len(set(kinds)) != 1:
            raise ValueError("list scale: all components must have the same form")
        return kinds[0] | SCALE_LIST, np.concatenate(flats)
    s = np.asarray(scale, dtype=np.float64)
    if s.ndim == 2:
        if s.shape != (d, d):
            raise ValueError('dimensions of "scale" matrix not d by d')
        return SCALE_FULL, np.ascontiguousarray(s.T).ravel()  # column-major like R
    s = s.ravel()
    if s.size == d and d != 1:
        return SCALE_DIAGONAL, s.copy()
    if s.size == 1:
        return SCALE_CONSTANT, s.copy()
    if s.size == d:
        return SCALE_DIAGONAL, s.copy()
    raise ValueError('length of "scale" vector not d or 1')


def resolve_outfun(outfun, d):
    """outfun -> (out_kind, mask) following metrop.c:416-503 (logical / +-index vectors)"""
    if outfun is None:
        return OUT_IDENTITY, None, d
    if isinstance(outfun, str):
        if outfun == "moments2":
            return OUT_MOMENTS2, None, 2 * d
        raise ValueError("outfun must be NULL, a function, a numeric vector, or a logical vector")
    a = np.asarray(outfun)
    if a.dtype == bool:
        if a.size != d:
            raise ValueError("is.logical(outfun) & (length(outfun) != length(initial))")
        mask = a.astype(np.int32)
    else:
        f = a.astype(np.float64).ravel()
        if np.any(np.isnan(f)):
            raise ValueError("NA or NaN index for outfun")
        if np.any(~np.isfinite(f)):
            raise ValueError("-Inf or Inf index for outfun")
        if np.any(f == 0):
            raise ValueError("is.numeric(outfun) & any(outfun == 0)")
        if np.any(f != np.trunc(f)):
            raise ValueError("is.numeric(outfun) & any(outfun != as.integer(outfun))")
        if np.any(np.abs(f) > d):
            raise ValueError("is.numeric(outfun) & any(abs(outfun) > length(initial)")
        if np.any(f > 0) and np.any(f < 0):
            raise ValueError("is.numeric(outfun) & any(outfun > 0) & any(outfun < 0)")
        idx = np.abs(f).astype(int) - 1
        if np.all(f > 0):
            mask = np.zeros(d, dtype=np.int32)
            mask[idx] = 1
        else:
            mask = np.ones(d, dtype=np.int32)
            mask[idx] = 0
    return OUT_INDEX, np.ascontiguousarray(mask, dtype=np.int32), int(mask.sum())


class MetropCall:
    """numpy-backed argument/result block for one mcmchip_metrop / orc_metrop call"""

    def __init__(self, d, initial, nbatch, blen, nspac, scale, outfun=None, rng_mode=RNG_PHILOX,
                 z=None, u=None, seed=0, chain_offset=0, iter_offset=0, debug=False, path=PATH_AUTO):
        init = _f64(initial)
        if init.ndim == 1:
            init = init[None, :]
        if init.shape[1] != d:
            raise ValueError("initial must be nchain x d")
        self.d, self.nchain = d, init.shape[0]
        self.nbatch, self.blen, self.nspac = int(nbatch), int(blen), int(nspac)
        self.niter = self.nbatch * self.blen * self.nspac
        self.initial = init
        self.scale_kind, self.scale = classify_scale(scale, d)
        self.out_kind, self.mask, self.nout = resolve_outfun(outfun, d)
        self.z, self.u = _f64(z), _f64(u)
        if rng_mode == RNG_INJECTED:
            if self.z is None or self.z.size != self.nchain * self.niter * d:
                raise ValueError("injected z must be nchain x niter x d")
            if self.u is None or self.u.size != self.nchain * self.niter:
                raise ValueError("injected u must be nchain x niter")
        C_, nb = self.nchain, self.nbatch
        self.accept = np.empty(C_)
        self.batch = np.empty((C_, nb, self.nout))
        self.final = np.empty((C_, d))
        self.accept_batch = np.empty((C_, nb))
        self.debug = bool(debug)
        if self.debug:
            self.current = np.empty((C_, self.niter, d))
            self.proposal = np.empty((C_, self.niter, d))
            self.zout = np.empty((C_, self.niter, d))
            self.log_green = np.empty((C_, self.niter))
            self.uout = np.empty((C_, self.niter))
            self.debug_accept = np.empty((C_, self.niter), dtype=np.int32)
        else:
            self.current = self.proposal = self.zout = self.log_green = self.uout = self.debug_accept = None
        self.args = MetropArgs(C_, self.nbatch, self.blen, self.nspac, np_ptr(self.initial),
                               self.scale_kind, np_ptr(self.scale), self.out_kind, np_ptr(self.mask),
                               int(rng_mode), np_ptr(self.z), np_ptr(self.u), int(seed),
                               int(chain_offset), int(iter_offset), int(self.debug), 0, int(path))
        self.out = MetropOut(np_ptr(self.accept), np_ptr(self.batch), np_ptr(self.final),
                             np_ptr(self.accept_batch), np_ptr(self.current), np_ptr(self.proposal),
                             np_ptr(self.zout), np_ptr(self.log_green), np_ptr(self.uout),
                             np_ptr(self.debug_accept), 0, 0, 0)

    def result(self):
        r = dict(accept=self.accept, batch=self.batch, initial=self.initial, final=self.final,
                 accept_batch=self.accept_batch, iter_offset=int(self.out.iter_offset_out),
                 path_used=int(self.out.path_used), nout=self.nout)
        if self.debug:
            r.update(current=self.current, proposal=self.proposal, z=self.zout,
                     log_green=self.log_green, u=self.uout, debug_accept=self.debug_accept)
        return r


class TemperCall:
    """numpy-backed argument/result block for one mcmchip_temper / orc_temper call"""

    def __init__(self, p, initial, neighbors, nbatch, blen, nspac, scale, parallel, rng_mode=RNG_PHILOX,
                 unif=None, norm=None, seed=0, ens_offset=0, iter_offset=0, nout=None, debug=False, out_kind=0, path=0,
                 out_index=None):
        nb = np.asarray(neighbors)
        if nb.ndim != 2:
            raise ValueError('argument "neighbors" must be matrix')
        if nb.shape[0] != nb.shape[1]:
            raise ValueError('argument "neighbors" must have same row and column dimension')
        k = nb.shape[0]
        self.k, self.p, self.parallel = k, p, bool(parallel)
        # column-major k x k (symmetric, so layout is moot once validated)
        self.neighbors = np.ascontiguousarray(nb.astype(bool).T, dtype=np.int32)
        init = _f64(initial)
        nstate = k * p if self.parallel else p + 1
        if self.parallel:
            # accept (k, p) or (nens, k, p) row-major numpy matrices; store column-major per ensemble
            if init.ndim == 2:
                init = init[None]
            if init.shape[1:] != (k, p):
                raise ValueError('row dim of args "initial" and "neighbors" must be same in parallel case')
            init = np.ascontiguousarray(np.transpose(init, (0, 2, 1))).reshape(-1, nstate)
        else:
            if init.ndim == 1:
                init = init[None]
            if init.shape[1] != nstate:
                raise ValueError('argument "initial" must have length > 1 in serial case')
        self.initial = np.ascontiguousarray(init)
        self.nens = init.shape[0]
        self.nbatch, self.blen, self.nspac = int(nbatch), int(blen), int(nspac)
        self.niter = self.nbatch * self.blen * self.nspac
        self.scale_kind, self.scale = classify_scale(scale, p, ncomp=k)
        self.unif, self.norm = _f64(unif), _f64(norm)
        E = self.nens
        self.out_kind = int(out_kind)
        if nout is None and self.out_kind == TOUT_WITCH_CORNER:
            nout = k if self.parallel else 2 * k
        nid = k * p if self.parallel else p            # length of the identity output
        self.out_index = None
        if self.out_kind == TOUT_INDEX:
            self.out_index = np.ascontiguousarray(out_index, dtype=np.int32)
            if self.out_index.shape != (nid,):
                raise ValueError("out_index must have one entry per element of the identity output")
            nout = int((self.out_index != 0).sum())
        elif self.out_kind == TOUT_MOMENTS2:
            nout = 2 * nid
        self.identity_out = nout is None
        self.nout = nout if nout is not None else nstate - (0 if self.parallel else 1)
        self.batch = np.empty((E, self.nout, self.nbatch))       # per ensemble batch[kb + nbatch*i]
        self.ibatch = None if self.parallel else np.empty((E, k, self.nbatch))
        self.acceptx = np.empty((E, k))
        self.accepti = np.empty((E, k, k))                         # column-major k x k per ensemble
        self.final = np.empty((E, nstate))
        self.debug = bool(debug)
        self.nstate = nstate
        T = self.niter
        dbg = {}
        if self.debug:
            dbg = dict(which=np.empty((E, T), dtype=np.int32), unif_which=np.empty((E, T)),
                       state=np.empty((E, nstate, T)), log_hastings=np.empty((E, T)),
                       unif_hastings=np.empty((E, T)), proposal=np.empty((E, p + 1, T)),
                       acceptd=np.empty((E, T), dtype=np.int32), norm=np.empty((E, p, T)),
                       unif_choose=np.empty((E, 2, T)) if self.parallel else np.empty((E, T)),
                       coproposal=np.empty((E, p + 1, T)) if self.parallel else None)
        self.dbg = dbg
        self.args = TemperArgs(E, k, int(self.parallel), self.nbatch, self.blen, self.nspac,
                               np_ptr(self.neighbors), np_ptr(self.initial), self.scale_kind,
                               np_ptr(self.scale), int(rng_mode), np_ptr(self.unif), np_ptr(self.norm),
                               int(seed), int(ens_offset), int(iter_offset), 0, int(self.debug), self.out_kind, int(path),
                               np_ptr(self.out_index))
        self.out = TemperOut(np_ptr(self.batch), np_ptr(self.ibatch), np_ptr(self.acceptx),
                             np_ptr(self.accepti), np_ptr(self.final), 0,
                             *[np_ptr(dbg.get(n)) for n in ("which", "unif_which", "state", "log_hastings",
                                                            "unif_hastings", "proposal", "acceptd", "norm",
                                                            "unif_choose", "coproposal")])

    def result(self):
        E, k, p = self.nens, self.k, self.p
        r = dict(acceptx=self.acceptx, accepti=np.transpose(self.accepti, (0, 2, 1)),
                 iter_offset=int(self.out.iter_offset_out))
        if self.parallel and self.identity_out:
            # per ensemble nbatch x k x p array, R layout batch[kb + nbatch*(i + k*j)]
            r["batch"] = np.transpose(self.batch.reshape(E, p, k, self.nbatch), (0, 3, 2, 1))
            r["final"] = np.transpose(self.final.reshape(E, p, k), (0, 2, 1))
        else:
            r["batch"] = np.transpose(self.batch, (0, 2, 1))
            r["final"] = self.final.reshape(E, p, k).transpose(0, 2, 1) if self.parallel else self.final
        if not self.parallel:
            r["ibatch"] = np.transpose(self.ibatch, (0, 2, 1))
        if self.debug:
            d = self.dbg
            r["which"] = d["which"] != 0
            r["unif.which"] = d["unif_which"]
            # niter x nstate (serial) or niter x k x p (parallel), like the reference's arrays
            st = np.transpose(d["state"], (0, 2, 1))
            r["state"] = st.reshape(E, self.niter, p, k).transpose(0, 1, 3, 2) if self.parallel else st
            r["log.hastings"] = d["log_hastings"]
            r["unif.hastings"] = d["unif_hastings"]
            r["proposal"] = np.transpose(d["proposal"], (0, 2, 1))
            r["acceptd"] = d["acceptd"] != 0
            r["norm"] = np.transpose(d["norm"], (0, 2, 1))
            r["unif.choose"] = np.transpose(d["unif_choose"], (0, 2, 1)) if self.parallel else d["unif_choose"]
            if self.parallel:
                r["coproposal"] = np.transpose(d["coproposal"], (0, 2, 1))
        return r
