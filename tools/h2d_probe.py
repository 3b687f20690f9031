"""How long a host-resident series takes to reach the device through mcmchip_olbm (pageable input, column-major as R
holds it) against the kernel time of the same call; the numpy mirror's transpose is timed apart.
usage (GPU box): python tools/h2d_probe.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401  (before the library: two HIP runtimes)
from mcmc_amd import api

rng = np.random.default_rng(0)
L = api.load_library()
api.init()
for n, p in ((4_000_000, 32), (2_000_000, 256)):
    x = rng.standard_normal((n, p))
    m = x.mean(axis=0)
    t0 = time.perf_counter()
    xc = np.ascontiguousarray(x.T)
    t1 = time.perf_counter()
    print("n=%d p=%d (%.2f GB): numpy transpose %.1f ms" % (n, p, n * p * 8 / 1e9, (t1 - t0) * 1e3), flush=True)
    var = np.empty((p, p))
    for rep in range(3):
        t0 = time.perf_counter()
        rc = L.mcmchip_olbm(xc.ctypes.data, n, p, 1000, m.ctypes.data, var.ctypes.data, 1, 0)
        t1 = time.perf_counter()
        assert rc == 0
        print("   C call %.1f ms (%.1f GB/s), kernel %s" % ((t1 - t0) * 1e3, n * p * 8 / 1e9 / (t1 - t0), api.last_kernel_time()), flush=True)
    # the numpy mirror end to end: C-ordered input, uploaded as it is and transposed on the device
    for rep in range(2):
        t0 = time.perf_counter()
        v = api.olbm(x, 1000, mean=m)
        t1 = time.perf_counter()
        print("   api.olbm(C-ordered numpy) %.1f ms" % ((t1 - t0) * 1e3), flush=True)
    assert np.array_equal(v, var)
    if p == 32:
        # initseq through the mirror: raw upload, centring and the finite check on the device; against the host steps
        # it replaces (finite check, mean, subtraction, transpose), timed here with numpy on the same array
        for rep in range(2):
            t0 = time.perf_counter()
            r = api.initseq(x)
            t1 = time.perf_counter()
            print("   api.initseq(C-ordered numpy) %.1f ms" % ((t1 - t0) * 1e3), flush=True)
        t0 = time.perf_counter()
        ok = np.all(np.isfinite(x)); xcen = x - x.mean(axis=0); xt = np.ascontiguousarray(xcen.T)
        t1 = time.perf_counter()
        print("   the host steps it replaces: %.1f ms" % ((t1 - t0) * 1e3), flush=True)
