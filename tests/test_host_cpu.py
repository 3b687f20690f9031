"""CPU-only tests (-m "not gpu"): the C-ABI library loads and exports every symbol of
include/mcmchip.h, ctypes struct layouts match the header, the host logic mirrors the
reference's argument rules, the product fails loudly without a GPU, and the N>1 sharding
path works under gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from mcmc_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcmchip.h")


def test_library_exports_every_declared_symbol():
    from mcmc_amd import api
    L = api.load_library()
    declared = re.findall(r"\b(mcmchip_[a-z_]+)\s*\(", open(HEADER).read())
    declared = sorted(set(d for d in declared if not d.endswith("_desc")))
    assert sorted(abi.EXPORTS) == declared
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.mcmchip_version()


def test_struct_layouts_match_header():
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "mcmchip.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu\n", sizeof(mcmchip_target_desc), sizeof(mcmchip_metrop_args),
             sizeof(mcmchip_metrop_out), sizeof(mcmchip_temper_args), sizeof(mcmchip_temper_out));
      printf("%zu %zu %zu %zu\n", offsetof(mcmchip_metrop_args, seed), offsetof(mcmchip_metrop_args, path),
             offsetof(mcmchip_metrop_out, nout), offsetof(mcmchip_temper_args, seed));
      return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [ctypes.sizeof(s) for s in (abi.TargetDesc, abi.MetropArgs, abi.MetropOut, abi.TemperArgs, abi.TemperOut)]
    offs = [abi.MetropArgs.seed.offset, abi.MetropArgs.path.offset, abi.MetropOut.nout.offset, abi.TemperArgs.seed.offset]
    assert [int(v) for v in out] == sizes + offs


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mcmc_amd import api
    with pytest.raises(api.McmcHipError):
        api.Target(abi.T_BALL, 3)
    with pytest.raises(api.McmcHipError):
        api.olbm(np.zeros((10, 2)), 2)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mcmc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "oracle.h" not in text and "liboracle" not in text, f


def test_scale_and_outfun_rules():
    # metrop.c:333-365
    assert abi.classify_scale(0.4, 5)[0] == abi.SCALE_CONSTANT
    assert abi.classify_scale(np.ones(5), 5)[0] == abi.SCALE_DIAGONAL
    k, flat = abi.classify_scale(np.arange(9.0).reshape(3, 3), 3)
    assert k == abi.SCALE_FULL and list(flat[:3]) == [0.0, 3.0, 6.0]       # column-major like R
    with pytest.raises(ValueError, match="not d by d"):
        abi.classify_scale(np.eye(4), 3)
    with pytest.raises(ValueError, match="not d or 1"):
        abi.classify_scale([1.0, 2.0], 3)
    # metrop.c:416-503
    assert abi.resolve_outfun(None, 4)[2] == 4
    assert list(abi.resolve_outfun([2, 3], 4)[1]) == [0, 1, 1, 0]
    assert list(abi.resolve_outfun([-2], 4)[1]) == [1, 0, 1, 1]
    assert list(abi.resolve_outfun([True, False, True, False], 4)[1]) == [1, 0, 1, 0]
    for bad, msg in [([0, 1], "outfun == 0"), ([1.5], "as.integer"), ([7], "abs\\(outfun\\)"),
                     ([1, -2], "outfun > 0\\) & any\\(outfun < 0"), ([np.nan], "NA or NaN"),
                     ([True, False], "length\\(outfun\\) != length\\(initial\\)")]:
        with pytest.raises(ValueError, match=msg):
            abi.resolve_outfun(bad, 4)
    # temper.c:78-87: list scale
    k, flat = abi.classify_scale([0.1, 0.2, 0.3], 2, ncomp=3)
    assert k == (abi.SCALE_CONSTANT | abi.SCALE_LIST) and flat.size == 3


def test_oracle_error_behaviour(oracle_mod):
    # metrop.c:60-65,196-197; temper.c:36-41
    tg = oracle_mod.Target(abi.T_SIMPLEX, 3, par=[0.0])
    with pytest.raises(oracle_mod.OracleError, match="-Inf at initial state"):
        oracle_mod.metrop(tg, np.array([0.5, 0.6, 0.1]), 10, seed=1)
    with pytest.raises(oracle_mod.OracleError, match='"nbatch" must be positive'):
        oracle_mod.metrop(tg, np.array([0.1, 0.1, 0.1]), 0, seed=1)
    w = oracle_mod.Target(abi.T_WITCH, 2, par=[0.2, 0.5, 0.7])
    nb = np.array([[0, 1, 0], [0, 0, 1], [0, 1, 0]], dtype=bool)
    with pytest.raises(oracle_mod.OracleError, match="symmetric"):
        oracle_mod.temper(w, np.zeros((3, 2)), nb, 5, seed=1, parallel=True)


def test_oracle_philox_sharding_and_restart(oracle_mod):
    # the Philox addressing the device uses: independent of sharding, restartable (tests/saveseed.R)
    tg = oracle_mod.Target(abi.T_BALL, 4)
    init = np.zeros((6, 4))
    whole = oracle_mod.metrop(tg, init, 20, 2, scale=0.3, seed=9)
    part = oracle_mod.metrop(tg, init[4:], 20, 2, scale=0.3, seed=9, chain_offset=4)
    assert np.array_equal(whole["batch"][4:], part["batch"])
    a = oracle_mod.metrop(tg, init, 10, 2, scale=0.3, seed=9)
    b = oracle_mod.metrop(tg, a["final"], 10, 2, scale=0.3, seed=9, iter_offset=a["iter_offset"])
    assert np.array_equal(np.concatenate([a["batch"], b["batch"]], axis=1), whole["batch"])


def test_softplus_arithmetic_accuracy():
    # the MFMA path's table-driven M(eta) = |eta|/2 + log1p(exp(-|eta|)) (mcmc_amd/csrc/softplus.h), host build
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_softplus
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(-36, 36, 300000), np.arange(0, 2035) / 56.5, (np.arange(0, 2035) + 0.5) / 56.5,
                        [0.0, -0.0, 1e-300, 36.0, 36.01, 40.0, -745.0, 1e4, 1e9, 1e30]])
    out, ref = check_softplus.run(a)
    assert np.max(np.abs(out - ref) / ref) < 4.5 * 2.0 ** -53      # <= 4 ulp (softplus.h header), mean 0.4
    edge, _ = check_softplus.run(np.array([np.inf, -np.inf, np.nan]))
    assert edge[0] == np.inf and edge[1] == np.inf and np.isnan(edge[2])


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch.distributed as dist
import oracle
from mcmc_amd import abi, dist as mdist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
tg = oracle.Target(abi.T_BALL, 4)
init = np.zeros((7, 4))                       # ragged: 4 + 3 chains
def run_local(block, lo):
    r = oracle.metrop(tg, block, 12, 3, scale=0.3, seed=5, chain_offset=lo)
    return {"batch": r["batch"], "accept": r["accept"], "accept.batch": r["accept_batch"]}
out = mdist.metrop_sharded(run_local, init)
whole = oracle.metrop(tg, init, 12, 3, scale=0.3, seed=5)
assert out["shard"] == ((0, 4) if dist.get_rank() == 0 else (4, 7))
assert np.array_equal(out["batch.all"], whole["batch"])
assert np.array_equal(out["accept.all"], whole["accept"])
assert np.array_equal(out["accept.batch.all"], whole["accept_batch"])

# temper: ensembles sharded, tables gathered
par = 1 - 0.5 ** (1 / 2) * 0.25 ** (np.arange(4) / 2)
wt = oracle.Target(abi.T_WITCH, 2, par=par)
nb = np.abs(np.subtract.outer(np.arange(4), np.arange(4))) == 1
tinit = np.zeros((5, 4, 2))                   # ragged: 3 + 2 ensembles
def run_t(block, lo):
    return oracle.temper(wt, block, nb, 6, 4, 2, 0.3, parallel=True, seed=8, ens_offset=lo)
tout = mdist.temper_sharded(run_t, tinit)
twhole = oracle.temper(wt, tinit, nb, 6, 4, 2, 0.3, parallel=True, seed=8)
for key in ("batch", "acceptx", "accepti"):
    assert np.array_equal(tout[key + ".all"], twhole[key], equal_nan=True), key

# initseq: functionals (columns) sharded, ragged Gamma sequences gathered
rng = np.random.default_rng(3)
cols = []
for rho in (0.2, 0.9, 0.6):
    e = rng.standard_normal(3000)
    x = np.empty(3000); x[0] = e[0]
    for t in range(1, 3000): x[t] = rho * x[t - 1] + e[t]
    cols.append(x - x.mean())
iout = mdist.initseq_sharded(lambda lo, hi: [oracle.initseq(cols[c]) for c in range(lo, hi)], 3)
for c in range(3):
    ref = oracle.initseq(cols[c])
    m = len(ref["Gamma.pos"])
    assert iout["length"][c] == m and iout["var.con"][c] == ref["var.con"] and iout["gamma0"][c] == ref["gamma0"]
    assert np.array_equal(iout["Gamma.con"][c, :m], ref["Gamma.con"]) and np.all(np.isnan(iout["Gamma.con"][c, m:]))

# olbm: time sharded with a halo, two small all-reduces
xs = np.column_stack(cols) + 2.0
n, blen = xs.shape[0], 50
lo, hi, own = mdist.olbm_time_shard(n, blen, dist.get_rank(), 2)
var, mean = mdist.olbm_sharded(lambda xb, l, m: oracle.olbm(xb, l, mean=m)[0], xs[lo:hi], n, blen)
vref, mref = oracle.olbm(xs, blen)
assert np.allclose(mean, mref, rtol=1e-14) and np.max(np.abs(var - vref)) <= 1e-12 * np.max(np.abs(vref))

# initseq: TIME sharded with a halo; per round one all-reduce of raw lag sums (numpy lag sums stand in for the device
# kernel; the stop rule and the sequences come from the library's host-only mcmchip_initseq_from_lagsums)
from mcmc_amd import api
def np_lagsums(x, nlags):
    x = np.asarray(x)
    return np.array([[np.dot(x[: x.shape[0] - l, c], x[l:, c]) if l < x.shape[0] else 0.0 for l in range(nlags)]
                     for c in range(x.shape[1])])
raw = np.column_stack(cols) + 1.5                 # not centred: the driver centres with the global mean
halo = 400
lo, hi, own = mdist.initseq_time_shard(raw.shape[0], halo, dist.get_rank(), 2)
res = mdist.initseq_time_sharded(np_lagsums, api.initseq_from_lagsums, raw[lo:hi], raw.shape[0], halo, first_lags=16)
for c in range(3):
    ref = oracle.initseq(raw[:, c] - raw[:, c].mean())
    assert len(res[c]["Gamma.pos"]) == len(ref["Gamma.pos"]), c
    for key in ("gamma0", "Gamma.pos", "Gamma.dec", "Gamma.con", "var.pos", "var.dec", "var.con"):
        assert np.max(np.abs(np.asarray(res[c][key]) - np.asarray(ref[key]))) <= 1e-10 * np.max(np.abs(ref[key])), (c, key)
# a halo below the default first round of 64 lags: the first round is clamped to the halo (sums for lags beyond it would
# miss the terms whose partner row lies past the rank's block), so 48 rows serve these columns and give the same numbers
lo48, hi48, _ = mdist.initseq_time_shard(raw.shape[0], 48, dist.get_rank(), 2)
res48 = mdist.initseq_time_sharded(np_lagsums, api.initseq_from_lagsums, raw[lo48:hi48], raw.shape[0], 48)
for c in range(3):
    ref = oracle.initseq(raw[:, c] - raw[:, c].mean())
    assert len(ref["Gamma.pos"]) * 2 <= 48, "the test wants stopping lags inside the halo"
    assert len(res48[c]["Gamma.pos"]) == len(ref["Gamma.pos"]), c
    assert abs(res48[c]["var.con"] - ref["var.con"]) <= 1e-10 * abs(ref["var.con"]), c
# ... and a halo that cannot hold the stopping lag is an error, with the default first round as well
lo8, hi8, _ = mdist.initseq_time_shard(raw.shape[0], 8, dist.get_rank(), 2)
for kw in ({"first_lags": 8}, {}):
    try:
        mdist.initseq_time_sharded(np_lagsums, api.initseq_from_lagsums, raw[lo8:hi8], raw.shape[0], 8, **kw)
        raise SystemExit("a halo of 8 rows cannot hold the stopping lag of the rho = 0.9 column")
    except RuntimeError as e:
        assert "beyond the halo" in str(e)
# cross-chain diagnostics: every rank reduces its chains (numpy moments stand in for the device kernel), one all-gather
# of 5 nout doubles per rank, merged in rank order by the library's host-only mcmchip_chain_summary_combine
def np_summary(b):
    b = np.asarray(b)
    m = b.mean(axis=1)
    s2 = ((b - m[:, None, :]) ** 2).sum(axis=1) / (b.shape[1] - 1)
    gm = m.mean(axis=0)
    stats = np.stack([np.full(b.shape[2], float(b.shape[0])), gm, ((m - gm) ** 2).sum(axis=0), s2.sum(axis=0),
                      np.zeros(b.shape[2])])
    return {"stats": stats, "nbatch": b.shape[1]}
rngs = np.random.default_rng(17)
ball = 2.0 + 1e-3 * rngs.standard_normal((11, 8, 3)) + 1e-3 * rngs.standard_normal((11, 1, 3))     # ragged: 6 + 5 chains
lo, hi = mdist.shard_range(11, dist.get_rank(), 2)
summ = mdist.chain_summary_sharded(np_summary, api.chain_summary_combine, ball[lo:hi])
sref = oracle.chain_summary(ball)
assert summ["nchain"] == 11 and summ["local"]["stats"][0, 0] == hi - lo
for key in ("grand_mean", "between", "within", "mcse_between", "mcse_within", "rhat"):
    assert np.all(np.abs(summ[key] - sref[key]) <= 1e-12 * np.abs(sref[key])), key
dist.destroy_process_group()
print("ok")
'''


def test_sharded_metrop_world_size_2_gloo(oracle_mod):
    # N>1 path: shard chains over ranks, Philox keyed by global chain id, one all-gather at the end; likewise
    # temper (ensembles), initseq (columns) and olbm (time + halo, two small all-reduces).
    # (the oracle stands in for the device sampler here: what is under test is the sharding)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    with tempfile.TemporaryDirectory() as td:
        w = os.path.join(td, "worker.py")
        open(w, "w").write(_WORKER)
        procs = [subprocess.Popen([sys.executable, w, ROOT, port, str(r)], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT) for r in range(2)]
        outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ok" in o, o


def test_shard_range():
    from mcmc_amd.dist import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]


def test_plain_c_client_builds():
    """examples/capi_demo.c: the header is valid C99 and a gcc-built client links against libmcmchip.so"""
    import subprocess
    ex = os.path.join(ROOT, "examples")
    subprocess.check_call(["make", "-s", "-B", "-C", ex])
    assert os.access(os.path.join(ex, "capi_demo"), os.X_OK)


def test_olbm_comb_plan_covers_every_window_once():
    """The host-side plan of olbm_comb.hip (teeth one window length apart swept together, split over the workgroups, pieces
    chained afterwards) is pure host code: every window of [0, nwin) in exactly one piece, pieces in time order without
    gaps, step counts consistent and balanced -- over random shapes and the C5 shape.  No GPU call."""
    import ctypes as C
    from mcmc_amd import api
    f = api.load_library().mcmchip_olbm_comb_plan_check
    f.restype = C.c_int
    f.argtypes = [C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    rng = np.random.default_rng(3)
    npieces, lo, hi = C.c_int32(), C.c_int64(), C.c_int64()
    cases = [(99_990_257, 10_000, 5, 256), (17, 16, 5, 256), (1, 1, 1, 1), (33, 16, 8, 3), (4097, 1024, 5, 512), (2_000_001, 1_000_000, 5, 256)]
    for _ in range(300):
        ln = int(rng.choice([1, 2, 15, 16, 17, 31, 100, 1024, 1027, 4096]))
        nwin = int(rng.integers(1, 60)) * ln + int(rng.integers(0, ln))
        cases.append((max(nwin, 1), ln, int(rng.integers(1, 9)), int(rng.choice([1, 2, 7, 256, 1024]))))
    for nwin, ln, teeth, groups in cases:
        assert f(nwin, ln, teeth, groups, C.byref(npieces), C.byref(lo), C.byref(hi)) == 0, (nwin, ln, teeth, groups)
        assert npieces.value >= -(-nwin // ln)
    assert f(99_990_257, 10_000, 5, 256, C.byref(npieces), C.byref(lo), C.byref(hi)) == 0
    # C5: the busiest of the 256 workgroups is within one comb step (5 slabs) of the mean
    assert hi.value <= -(-99_990_257 // (16 * 256)) + 5 and 10_000 <= npieces.value <= 12_000


def test_olbm_lane_plan_covers_every_window_once():
    """The plan of olbm_lane.hip (super-blocks of four teeth, units of the sweep dealt round-robin to the workgroups): every
    window in exactly one piece, pieces in time order without gaps, step counts consistent; the busiest workgroup of the
    n = 1e8, len = 1e4 shape within 10 % of the mean.  No GPU call."""
    import ctypes as C
    from mcmc_amd import api
    f = api.load_library().mcmchip_olbm_lane_plan_check
    f.restype = C.c_int
    f.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    rng = np.random.default_rng(4)
    npieces, lo, hi = C.c_int32(), C.c_int64(), C.c_int64()
    cases = [(99_990_001, 10_000, 2304), (17, 16, 256), (1, 1, 1), (33, 16, 3), (4097, 1024, 512), (2_000_001, 1_000_000, 256),
             (1_600_004, 400_000, 2304), (51_201, 12_800, 512)]
    for _ in range(300):
        ln = int(rng.choice([1, 2, 15, 16, 17, 31, 100, 256, 1024, 1027, 4096, 20_000]))
        nwin = int(rng.integers(1, 60)) * ln + int(rng.integers(0, ln))
        cases.append((max(nwin, 1), ln, int(rng.choice([1, 2, 7, 256, 1024, 2304]))))
    for nwin, ln, groups in cases:
        assert f(nwin, ln, groups, C.byref(npieces), C.byref(lo), C.byref(hi)) == 0, (nwin, ln, groups)
        assert npieces.value >= -(-nwin // ln)
    for groups in (512, 768, 1024, 2304):
        assert f(99_990_001, 10_000, groups, C.byref(npieces), C.byref(lo), C.byref(hi)) == 0
        mean_steps = 99_990_001 / 16 / groups
        assert hi.value <= 1.10 * mean_steps + 8, (groups, hi.value, mean_steps)


def test_olbm_few_plan_covers_every_window_once():
    """The plan of olbm_few_kernel (p <= 8: 2 or 4 consecutive units of olbm_lane.hip's plan side by side in one instruction,
    bundles dealt round-robin): every window in exactly one piece, pieces in time order without gaps.  No GPU call."""
    import ctypes as C
    from mcmc_amd import api
    f = api.load_library().mcmchip_olbm_few_plan_check
    f.restype = C.c_int
    f.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    rng = np.random.default_rng(6)
    npieces, lo, hi = C.c_int32(), C.c_int64(), C.c_int64()
    cases = [(99_990_001, 10_000, 4608, 4), (99_990_001, 10_000, 4608, 2), (17, 16, 256, 4), (1, 1, 1, 2), (33, 16, 3, 4),
             (4097, 1024, 512, 2), (2_000_001, 1_000_000, 256, 4), (1_600_004, 400_000, 2304, 4), (51_201, 12_800, 512, 2)]
    for _ in range(300):
        ln = int(rng.choice([1, 2, 15, 16, 17, 31, 100, 256, 1024, 1027, 4096, 20_000]))
        nwin = int(rng.integers(1, 60)) * ln + int(rng.integers(0, ln))
        cases.append((max(nwin, 1), ln, int(rng.choice([1, 2, 7, 256, 1024, 4608])), int(rng.choice([2, 4]))))
    for nwin, ln, groups, sub in cases:
        assert f(nwin, ln, groups, sub, C.byref(npieces), C.byref(lo), C.byref(hi)) == 0, (nwin, ln, groups, sub)
        assert npieces.value >= -(-nwin // ln)
    for sub in (2, 4):
        assert f(99_990_001, 10_000, 4608, sub, C.byref(npieces), C.byref(lo), C.byref(hi)) == 0
        mean_steps = 99_990_001 / 16 / 4608
        assert hi.value <= 1.15 * mean_steps + 8 * sub, (sub, hi.value, mean_steps)
