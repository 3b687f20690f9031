"""C2 (logit 100 x 5, 4096 chains x 10 000 iterations, 1.6 GB of batch means to pageable host memory) through the C-ABI with
the SAME host arrays call after call: the first call takes the page faults of a fresh destination, the later ones show
what the copy costs once the pages are there.  Piece by piece under the launches (metrop_stream.hip) against the copy behind
the run (MCMCHIP_NO_PIECEWISE_COPY=1).  usage: python tools/c2_host_copy.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mcmc_amd import abi, api  # noqa: E402
from rhelpers import load_logit  # noqa: E402

api.init(0)
L = api.load_library()
X, y = load_logit()
tg = api.logistic_target(X, y)
Cn, nbatch, d = 4096, 10_000, 5
rng = np.random.default_rng(3)
init_h = 0.1 * rng.standard_normal((Cn, d))
scale = np.array([0.4])
shapes = ((Cn,), (Cn, nbatch, d), (Cn, d), (Cn, nbatch))
for mode in ("piecewise", "behind the run"):
    if mode == "piecewise":
        os.environ.pop("MCMCHIP_NO_PIECEWISE_COPY", None)
    else:
        os.environ["MCMCHIP_NO_PIECEWISE_COPY"] = "1"
    outs = [np.empty(s_) for s_ in shapes]                       # fresh: the first call faults the pages in
    args = abi.MetropArgs(Cn, nbatch, 1, 1, init_h.ctypes.data, abi.SCALE_CONSTANT, scale.ctypes.data, abi.OUT_IDENTITY,
                          None, abi.RNG_PHILOX, None, None, 7, 0, 0, 0, 0, abi.PATH_STREAM)
    out = abi.MetropOut(*[a_.ctypes.data for a_ in outs], None, None, None, None, None, None, 0, 0, 0)
    ms = []
    for rep in range(5):
        t0 = time.perf_counter()
        rc = L.mcmchip_metrop(tg.handle, C.addressof(args), C.addressof(out))
        ms.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0, L.mcmchip_last_error()
    kms, nl, _ = api.last_kernel_time()[1], None, None
    print("%-16s call ms: first (fresh destination) %.1f, then %s; checksum %.6f" % (mode, ms[0], " ".join("%.1f" % m for m in ms[1:]), float(outs[1][::97, ::53].sum())), flush=True)
